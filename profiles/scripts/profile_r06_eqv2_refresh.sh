#!/bin/bash
# round 6, after the last EquiformerV2 change (second convolution through the streamed-fragment kernel): the bench line again and
# the EquiformerV2 parts of profile_r06.sh (kernel stats, FETCH_SIZE / WRITE_SIZE passes).  Run on the GPU box through gpurun.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r06b; rm -rf "$o"; mkdir -p "$o"
python3 bench.py > $o/bench.json 2> $o/bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $o/eq_hbm/$c -o p --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline --no-secondary > $o/eq_hbm_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $o/eq_stats -o eq --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 10 --no-cpu-baseline --no-secondary > $o/eq_under_rocprof.log 2>&1
find $o -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections, os, json
base = "gpurun_out/r06b/"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(base + "eq_hbm/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        os.remove(f)
with open(base + "eqv2_hbm_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "FETCH_SIZE_KB_mean", "FETCH_SIZE_KB_max", "WRITE_SIZE_KB_mean", "WRITE_SIZE_KB_max",
                "HBM_GB_per_full_size_launch (2 x FETCH_max + WRITE_max)"])
    for k in sorted(per, key=lambda k: -(sum(per[k]["WRITE_SIZE"]) + 2 * sum(per[k]["FETCH_SIZE"]))):
        if "at::native" in k or "rocprim" in k or "rocclr" in k:
            continue
        row = [k[:120], len(per[k]["WRITE_SIZE"]) or len(per[k]["FETCH_SIZE"])]
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 1), round(max(v), 1)]
        row.append(round((2.0 * max(per[k]["FETCH_SIZE"] or [0.0]) + max(per[k]["WRITE_SIZE"] or [0.0])) * 1024 / 1e9, 3))
        w.writerow(row)
# the SO(2) convolution products: eq_gemm16pw / eq_gemm16p, eq_gemm16_256, and the edge-level launches of adf_gemm_f16x3 (EPI 0)
conv = {k: v for k, v in per.items() if "eq_gemm16p" in k or "eq_gemm16_256_kernel" in k or "adf_gemm_f16x3_kernelILi0ELi3ELi2ELi0ELb1ELi4E" in k}
kern, tot_b, tot_n = {}, 0.0, 0
for k, v in conv.items():
    n = min(len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"]))
    b = (2.0 * sum(v["FETCH_SIZE"]) + sum(v["WRITE_SIZE"])) * 1024.0
    kern[k[:48]] = {"launches": n, "hbm_bytes_per_launch": b / max(n, 1)}
    tot_b += b; tot_n += n
json.dump({"systems": 64, "edges": 256000, "conv_launches": tot_n, "hbm_bytes_per_conv_launch": tot_b / max(tot_n, 1),
           "hbm_bytes_per_conv_launch_and_edge": tot_b / max(tot_n, 1) / 256000, "kernels": kern,
           "note": "round 6, final code. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `bench.py --model eqv2 --systems 64 "
                   "--num-steps 2` (profiles/r06_eqv2_hbm_per_kernel.csv, counter unit KB); 2*FETCH + WRITE per launch (gfx950 correction of "
                   "MI355X_MICROARCH.md), averaged over the launches of the SO(2)-convolution product kernels (eq_gemm16pw_kernel, "
                   "eq_gemm16_256_kernel, adf_gemm_f16x3_kernel<EPI 0, streamed fragments, 8 waves>); includes the cheaper launches of the force blocks"},
          open(base + "eqv2_conv_pmc.json", "w"), indent=1)
for f in glob.glob(base + "**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
ls $o

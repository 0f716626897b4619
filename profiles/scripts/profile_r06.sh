#!/bin/bash
# round-6 profile set (run on the GPU box): the driver's bench line, rocprofv3 kernel stats of the same command,
# MfmaUtil / VALUBusy pass, HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) per kernel of the PaiNN sampler
# (-> r06_pmc_hbm_per_kernel.csv: the node kernels' bytes per launch), of the training step (-> train_step_pmc.json) and
# of the EquiformerV2 convolution products (-> eqv2_conv_pmc.json), training and EquiformerV2 kernel stats
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r06; rm -rf "$o"; mkdir -p "$o"
python3 bench.py > $o/bench.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats -d $o/stats -o r06 --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > $o/under_rocprof.log 2>&1
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/pmc_util -o u --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic-probe --steps 1 --warmup 0 --num-steps 3 > $o/pmc_util.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $o/hbm/$c -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic-probe --steps 1 --warmup 0 --num-steps 3 > $o/hbm_$c.log 2>&1
  rocprofv3 --pmc $c -d $o/train_hbm/$c -o p --output-format csv -- python3 bench.py --mode train --steps 1 --warmup 1 > $o/train_hbm_$c.log 2>&1
  rocprofv3 --pmc $c -d $o/eq_hbm/$c -o p --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline --no-secondary > $o/eq_hbm_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $o/train_stats -o tr --output-format csv -- python3 bench.py --mode train --steps 3 --warmup 1 > $o/train_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/eq_stats -o eq --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 10 --no-cpu-baseline --no-secondary > $o/eq_under_rocprof.log 2>&1
find $o -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections, os, json
base = "gpurun_out/r06/"
def collect(dirs):
    """dirs: list of directories under base; every *counter_collection.csv below them -> {kernel: {counter: [values]}}"""
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(base + d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            os.remove(f)
    return per
per = collect(["pmc_util"])
with open(base + "pmc_util_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "MfmaUtil_mean", "MfmaUtil_mean_of_full_size_launches", "VALUBusy_mean", "VALUBusy_mean_of_full_size_launches"])
    for k in sorted(per, key=lambda k: -len(per[k]["MfmaUtil"])):
        row = [k, len(per[k]["MfmaUtil"])]
        for c in ("MfmaUtil", "VALUBusy"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 3), round(max(v), 3)]
        w.writerow(row)
def hbm_table(sub, out_csv):
    per = collect([sub + "/FETCH_SIZE", sub + "/WRITE_SIZE"])
    tot_f = tot_w = 0.0
    with open(base + out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Launches", "FETCH_SIZE_KB_mean", "FETCH_SIZE_KB_max", "WRITE_SIZE_KB_mean", "WRITE_SIZE_KB_max",
                    "HBM_GB_per_full_size_launch (2 x FETCH_max + WRITE_max)"])
        for k in sorted(per, key=lambda k: -(sum(per[k]["WRITE_SIZE"]) + 2 * sum(per[k]["FETCH_SIZE"]))):
            tot_f += sum(per[k]["FETCH_SIZE"]); tot_w += sum(per[k]["WRITE_SIZE"])
            if "at::native" in k or "rocprim" in k or "rocclr" in k:
                continue
            row = [k[:120], len(per[k]["WRITE_SIZE"]) or len(per[k]["FETCH_SIZE"])]
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                v = per[k][c] or [0.0]
                row += [round(sum(v) / len(v), 1), round(max(v), 1)]
            row.append(round((2.0 * max(per[k]["FETCH_SIZE"] or [0.0]) + max(per[k]["WRITE_SIZE"] or [0.0])) * 1024 / 1e9, 3))
            w.writerow(row)
    return per, tot_f, tot_w
hbm_table("hbm", "pmc_hbm_per_kernel.csv")
_, tot_f, tot_w = hbm_table("train_hbm", "train_hbm_per_kernel.csv")
steps, graphs = 2, 256
json.dump({"hbm_bytes_per_step_and_graph": (2.0 * tot_f + tot_w) * 1024.0 / steps / graphs,
           "fetch_size_kb_total": tot_f, "write_size_kb_total": tot_w, "steps_profiled": steps, "graphs_per_step": graphs,
           "source": "round 6: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --mode train --steps 1 --warmup 1`, "
                     "2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of the two steps, KB -> bytes, per step and graph "
                     "(profiles/scripts/profile_r06.sh)"}, open(base + "train_step_pmc.json", "w"), indent=1)
per, _, _ = hbm_table("eq_hbm", "eqv2_hbm_per_kernel.csv")
conv = {k: v for k, v in per.items() if "eq_gemm16p" in k or "eq_gemm16_256_kernel" in k}   # (eq_gemm16p_kernel and eq_gemm16pw_kernel)
kern, tot_b, tot_n = {}, 0.0, 0
for k, v in conv.items():
    n = min(len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"]))
    b = (2.0 * sum(v["FETCH_SIZE"]) + sum(v["WRITE_SIZE"])) * 1024.0
    kern[k[:44]] = {"launches": n, "hbm_bytes_per_launch": b / max(n, 1)}
    tot_b += b; tot_n += n
try:
    cfg = json.loads(open(base + "eq_hbm_FETCH_SIZE.log").read().strip().splitlines()[-1])["config"]
    edges = int(round(cfg["edges_per_system"] * cfg.get("systems_per_gpu", 64)))
except Exception:
    edges = 256000
json.dump({"systems": 64, "edges": edges, "conv_launches": tot_n, "hbm_bytes_per_conv_launch": tot_b / max(tot_n, 1),
           "hbm_bytes_per_conv_launch_and_edge": tot_b / max(tot_n, 1) / edges, "kernels": kern,
           "note": "round 6. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `bench.py --model eqv2 --systems 64 "
                   "--num-steps 2` (profiles/r06_eqv2_hbm_per_kernel.csv, counter unit KB); 2*FETCH + WRITE per launch (gfx950 correction of "
                   "MI355X_MICROARCH.md), averaged over the launches of the SO(2)-convolution product kernels (eq_gemm16pw_kernel / eq_gemm16p_kernel, "
                   "eq_gemm16_256_kernel); includes the cheaper launches of the force blocks"}, open(base + "eqv2_conv_pmc.json", "w"), indent=1)
for f in glob.glob(base + "**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
du -sh $o; ls $o

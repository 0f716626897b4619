#!/bin/bash
# round-5 profile set (run on the GPU box): the driver's bench line, rocprofv3 kernel stats of the same command,
# MfmaUtil / VALUBusy pass, training and EquiformerV2 kernel stats, SQ counters of the message kernel, HBM counters of the
# training step (-> profiles/train_step_pmc.json), the training kernels per launch shape, SQ / TCP / TCC counters of the fused
# rbf_proj weight-gradient kernel, the MFMA/VALU overlap microbenchmark
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r05; rm -rf "$o"; mkdir -p "$o"
python3 bench.py > $o/bench.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats -d $o/stats -o r05 --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > $o/under_rocprof.log 2>&1
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/pmc_util -o u --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --num-steps 3 > $o/pmc_util.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/train_stats -o tr --output-format csv -- python3 bench.py --mode train --steps 3 --warmup 1 > $o/train_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/eq_stats -o eq --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 10 --no-cpu-baseline --no-secondary > $o/eq_under_rocprof.log 2>&1
bash profiles/scripts/msg_pmc.sh adsorbdiff_amd/libadsorbdiff_hip.so 200 > $o/message_kernel_sq_counters_200_systems.txt 2>&1
rm -rf gpurun_out/msgpmc
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $o/train_hbm/$c -o p --output-format csv -- python3 bench.py --mode train --steps 1 --warmup 1 > $o/train_hbm_$c.log 2>&1
done
rocprofv3 --kernel-trace -d $o/train_trace -o t --output-format csv -- python3 bench.py --mode train --steps 2 --warmup 1 > /dev/null 2>&1
python3 profiles/scripts/trace_by_grid.py $o/train_trace 40 > $o/train_kernels_by_launch_shape.txt; rm -rf $o/train_trace
bash profiles/scripts/rw_pmc.sh > $o/rbf_wgrad_counters.txt 2>&1; rm -rf gpurun_out/rwpmc
(cd profiles/scripts && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize mfma_valu_overlap.hip -o /tmp/overlap 2>/dev/null && /tmp/overlap) > $o/mfma_valu_overlap.txt 2>&1
find $o -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections, os, json
base = "gpurun_out/r05/"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "pmc_util/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    os.remove(f)
with open(base + "pmc_util_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "MfmaUtil_mean", "MfmaUtil_mean_of_full_size_launches", "VALUBusy_mean", "VALUBusy_mean_of_full_size_launches"])
    for k in sorted(per, key=lambda k: -len(per[k]["MfmaUtil"])):
        row = [k, len(per[k]["MfmaUtil"])]
        for c in ("MfmaUtil", "VALUBusy"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 3), round(max(v), 3)]
        w.writerow(row)
# training step: HBM bytes per kernel and per step (one profiled step of 256 graphs after one warm-up step: the
# counters of BOTH steps are collected, so per-step = total / 2)
per = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(base + "train_hbm/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                per[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
        os.remove(f)
tot_f = tot_w = 0.0
with open(base + "train_hbm_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "FETCH_SIZE_KB_mean", "FETCH_SIZE_KB_max", "WRITE_SIZE_KB_mean", "WRITE_SIZE_KB_max"])
    for k in sorted(per, key=lambda k: -(sum(per[k]["WRITE_SIZE"]) + sum(per[k]["FETCH_SIZE"]))):
        tot_f += sum(per[k]["FETCH_SIZE"]); tot_w += sum(per[k]["WRITE_SIZE"])
        if "at::native" in k or "rocprim" in k or "rocclr" in k:
            continue
        row = [k[:120], len(per[k]["WRITE_SIZE"]) or len(per[k]["FETCH_SIZE"])]
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 1), round(max(v), 1)]
        w.writerow(row)
steps, graphs = 2, 256
json.dump({"hbm_bytes_per_step_and_graph": (2.0 * tot_f + tot_w) * 1024.0 / steps / graphs,
           "fetch_size_kb_total": tot_f, "write_size_kb_total": tot_w, "steps_profiled": steps, "graphs_per_step": graphs,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --mode train --steps 1 --warmup 1`, "
                     "2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of the two steps, KB -> bytes, per step and graph "
                     "(profiles/scripts/profile_r05.sh)"}, open(base + "train_step_pmc.json", "w"), indent=1)
for f in glob.glob(base + "**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
du -sh $o; ls $o

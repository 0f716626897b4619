#!/bin/bash
# round-4 profile set (run on the GPU box): the driver's bench line, rocprofv3 kernel stats of the same command,
# MfmaUtil / VALUBusy passes, training and EquiformerV2 kernel stats, SQ counters of the two message kernels
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r04; rm -rf "$o"; mkdir -p "$o"
python3 bench.py > $o/bench.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats -d $o/stats -o r04 --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > $o/under_rocprof.log 2>&1
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/pmc_util -o u --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --num-steps 3 > $o/pmc_util.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/train_stats -o tr --output-format csv -- python3 bench.py --mode train --steps 3 --warmup 1 > $o/train_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/eq_stats -o eq --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 10 --no-cpu-baseline --no-secondary > $o/eq_under_rocprof.log 2>&1
bash profiles/scripts/r04_msg_pmc.sh 200 > $o/message_kernels_sq_counters_200_systems.txt 2>&1
find $o -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/r04/"
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
for f in glob.glob(base + "**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
du -sh $o; ls $o

#!/bin/bash
# round-3 profile set (run on the GPU box): bench lines, rocprofv3 kernel stats of the same commands, PMC passes
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r03; rm -rf $o; mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err
python3 bench.py --mode train > $o/bench_train.json 2> $o/bench_train.err
python3 bench.py --model eqv2 --systems 256 --steps 1 --warmup 0 > $o/bench_eqv2.json 2> $o/bench_eqv2.err
rocprofv3 --kernel-trace --stats -d $o/stats -o r03 --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > $o/under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $o/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --num-steps 3 > $o/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --num-steps 3 > $o/pmc_write.log 2>&1
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/pmc_util -o u --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --num-steps 3 > $o/pmc_util.log 2>&1
# EquiformerV2 (config 4)
rocprofv3 --kernel-trace --stats -d $o/eq_stats -o eq --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 10 --no-cpu-baseline > $o/eq_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $o/eq_pmc_fetch -o f --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline > $o/eq_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/eq_pmc_write -o w --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline > $o/eq_pmc_write.log 2>&1
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/eq_pmc_util -o u --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline > $o/eq_pmc_util.log 2>&1
# training step kernel stats
rocprofv3 --kernel-trace --stats -d $o/train_stats -o tr --output-format csv -- python3 bench.py --mode train --steps 3 --warmup 1 > $o/train_under_rocprof.log 2>&1
find $o -name "*agent_info.csv" -delete
# reduce the per-dispatch counter files to per-kernel summaries (the raw files are too large to merge back)
python3 - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/r03/"
def summarise(tag, counters):
    files = glob.glob(base + tag + "/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        os.remove(f)
    with open(base + tag + "_per_kernel.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Launches"] + [c + s for c in counters for s in ("_mean", "_mean_of_full_size_launches", "_sum")])
        for k in sorted(per, key=lambda k: -sum(per[k][counters[0]])):
            row = [k, len(per[k][counters[0]])]
            for c in counters:
                v = per[k][c]
                full = [x for x in v if x > 0.5 * max(v)] or v
                row += [round(sum(v) / max(len(v), 1), 3), round(sum(full) / max(len(full), 1), 3), round(sum(v), 3)]
            w.writerow(row)
for tag, cs in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_util", ["MfmaUtil", "VALUBusy"]),
                ("eq_pmc_fetch", ["FETCH_SIZE"]), ("eq_pmc_write", ["WRITE_SIZE"]), ("eq_pmc_util", ["MfmaUtil", "VALUBusy"])):
    summarise(tag, cs)
# per-layer message launch durations, then drop the big traces
rows = [r for r in csv.DictReader(open(glob.glob(base + "stats/**/*kernel_trace.csv", recursive=True)[0])) if "message_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows][-300:]
with open(base + "message_launch_ms_by_step_and_layer.csv", "w", newline="") as fh:
    w = csv.writer(fh); w.writerow(["reverse_step"] + [f"layer{l}_ms" for l in range(6)])
    for s in range(50): w.writerow([s] + [round(x, 3) for x in dur[6 * s:6 * s + 6]])
for f in glob.glob(base + "**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
du -sh $o; ls $o

#!/bin/bash
# round 6: MfmaUtil / VALUBusy per kernel of the EquiformerV2 sampler (64 systems, 2 reverse steps) on the last build
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; export TMPDIR=/tmp
o=gpurun_out/r06_eq_pmc; rm -rf "$o"; mkdir -p "$o"
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/util -o u --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 2 --no-cpu-baseline --no-secondary > $o/util.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/r06_eq_pmc/"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "util/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    os.remove(f)
with open(base + "eqv2_pmc_util_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "MfmaUtil_mean", "MfmaUtil_max", "VALUBusy_mean", "VALUBusy_max"])
    for k in sorted(per, key=lambda k: -sum(per[k]["MfmaUtil"]) - sum(per[k]["VALUBusy"])):
        if "at::native" in k or "rocprim" in k or "rocclr" in k: continue
        row = [k[:110], len(per[k]["MfmaUtil"])]
        for c in ("MfmaUtil", "VALUBusy"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 2), round(max(v), 2)]
        w.writerow(row)
PY
find $o -name "*agent_info.csv" -delete; find $o -name "*kernel_trace.csv" -delete
head -14 $o/eqv2_pmc_util_per_kernel.csv | cut -c1-170

#!/bin/bash
# MfmaUtil / VALUBusy per kernel of the training step (run on the GPU box): --pmc pass only, no tracing flags beside it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r04_train_pmc; rm -rf "$o"; mkdir -p "$o"
rocprofv3 --pmc MfmaUtil VALUBusy -d $o/pmc -o u --output-format csv -- python3 bench.py --mode train --steps 2 --warmup 1 > $o/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/r04_train_pmc/"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    os.remove(f)
with open(base + "train_pmc_util_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "MfmaUtil_mean", "MfmaUtil_max", "VALUBusy_mean", "VALUBusy_max"])
    for k in sorted(per, key=lambda k: -len(per[k]["MfmaUtil"])):
        if "at::native" in k or "rocprim" in k or "rocclr" in k:
            continue
        row = [k[:120], len(per[k]["MfmaUtil"])]
        for c in ("MfmaUtil", "VALUBusy"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 3), round(max(v), 3)]
        w.writerow(row)
PY
find $o -name "*agent_info.csv" -delete
cat $o/train_pmc_util_per_kernel.csv | head -30

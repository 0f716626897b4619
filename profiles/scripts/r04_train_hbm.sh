#!/bin/bash
# HBM traffic counters per kernel of the training step (run on the GPU box): FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md: unit KB; on gfx950 FETCH_SIZE counts half the bytes of wide reads - bench.py's probe calibrates the x2)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/r04_train_hbm; rm -rf "$o"; mkdir -p "$o"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $o/$c -o p --output-format csv -- python3 bench.py --mode train --steps 1 --warmup 1 > $o/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
base = "gpurun_out/r04_train_hbm/"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(base + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                per[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
        os.remove(f)
with open(base + "train_hbm_per_kernel.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Launches", "FETCH_SIZE_KB_mean", "FETCH_SIZE_KB_max", "WRITE_SIZE_KB_mean", "WRITE_SIZE_KB_max"])
    keys = [k for k in per if not ("at::native" in k or "rocprim" in k or "rocclr" in k)]
    for k in sorted(keys, key=lambda k: -(sum(per[k]["WRITE_SIZE"]) + sum(per[k]["FETCH_SIZE"]))):
        row = [k[:120], len(per[k]["WRITE_SIZE"]) or len(per[k]["FETCH_SIZE"])]
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = per[k][c] or [0.0]
            row += [round(sum(v) / len(v), 1), round(max(v), 1)]
        w.writerow(row)
PY
find $o -name "*agent_info.csv" -delete
head -12 $o/train_hbm_per_kernel.csv

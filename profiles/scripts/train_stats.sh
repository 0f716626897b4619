#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/trs -o tr --output-format csv -- python3 bench.py --mode train --steps 3 --warmup 1 > /tmp/trs.log 2>&1
mkdir -p gpurun_out/train_stats; cp $(find /tmp/trs -name "*kernel_stats.csv" | head -1) gpurun_out/train_stats/tr_kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/train_stats/tr_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms per step", tot / 1e6 / 4)
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e6:8.3f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% max {float(r["MaxNs"])/1e6:.2f}')
PY

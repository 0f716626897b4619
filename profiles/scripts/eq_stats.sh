#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/eq_stats; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats -d /tmp/eqs -o eq --output-format csv -- python3 bench.py --model eqv2 --systems 64 --steps 1 --warmup 0 --num-steps 10 --no-cpu-baseline > $o/log.txt 2>&1
cp $(find /tmp/eqs -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/eq_stats/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in rows[:22]:
    print(f'{r["Name"][:64]:64s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.1f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% avg {float(r["AverageNs"])/1e3:9.1f} us')
PY

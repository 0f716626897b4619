#!/bin/bash
# per-kernel cost of the per-row lifts: kernel stats of 10 full-row reverse steps with and without them
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/lift_cost; rm -rf $o; mkdir -p $o
for lift in 1 0; do
  ADF_LIFT=$lift rocprofv3 --kernel-trace --stats -d /tmp/lc_$lift -o lc --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --no-incremental --steps 1 --warmup 0 --num-steps 10 > $o/log_$lift.txt 2>&1
  cp $(find /tmp/lc_$lift -name "*kernel_stats.csv" | head -1) $o/kernel_stats_lift$lift.csv
done
python3 - <<'PY'
import csv
def load(f):
    return {r["Name"][:70]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("gpurun_out/lift_cost/kernel_stats_lift1.csv"), load("gpurun_out/lift_cost/kernel_stats_lift0.csv")
tot_a, tot_b = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"total ms: lifts {tot_a:.1f}  none {tot_b:.1f}")
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1])):
    va, vb = a.get(k, (0, 0.0)), b.get(k, (0, 0.0))
    if max(va[1], vb[1]) > 0.004 * tot_a:
        print(f"{k:70s} {va[0]:5d} {va[1]:9.1f} | {vb[0]:5d} {vb[1]:9.1f}  d={va[1]-vb[1]:+8.1f}")
PY

#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf "$R/gpurun_out/p125"; mkdir -p "$R/gpurun_out/p125"
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/p125" -o k --output-format csv -- python3 "$R/bench.py" --systems 125 --no-secondary --no-traffic-probe --no-cpu-baseline --steps 2 --warmup 1 > "$R/gpurun_out/p125/log.txt" 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/p125/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:16]:
    print(f"{r['Name'][:85]:85s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
grep '^{"metric"' "$R/gpurun_out/p125/log.txt" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['gpu_ms_per_pass'])"

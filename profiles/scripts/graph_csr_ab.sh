#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for mode in 0 1; do
mkdir -p "$R/gpurun_out/graph_prof$mode"
ADF_GRAPH_SYS_CSR=$mode rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/graph_prof$mode" -o g --output-format csv -- python3 "$R/bench.py" --no-secondary --no-traffic-probe --no-cpu-baseline --steps 1 --warmup 0 --num-steps 6 > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$R/gpurun_out/graph_prof$mode/**/*kernel_stats.csv", recursive=True)):
    rows=list(csv.DictReader(open(f)))
    print("mode $mode")
    for r in rows:
        if any(k in r['Name'] for k in ('count','fill','sort','topk','scan','validate','inc_compare')):
            print(f"  {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us")
PY
done

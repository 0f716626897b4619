#!/bin/bash
# usage (on the GPU box, through gpurun): ab_env.sh VAR v1 v2 ...  : two interleaved repetitions of the short bench with VAR=v
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $VAR=$v python bench.py --no-secondary --no-traffic-probe --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v rep $rep value', round(d['value'],1), d['gpu_ms_per_pass'], d['sites_sha256_16'])"
done
done

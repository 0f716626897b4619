#!/bin/bash
# usage (on the GPU box): ab_lib.sh name1 name2 ...   (alternative builds linked as scratch/ab/lib_<name>.so, loaded through ADF_LIB_PATH), two interleaved repetitions of the short bench
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "$@"; do
  ADF_LIB_PATH=$PWD/scratch/ab/lib_$v.so python bench.py --no-secondary --no-traffic-probe --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v rep $rep value', round(d['value'],1), d['gpu_ms_per_pass'], d['sites_sha256_16'])"
done
done

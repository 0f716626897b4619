#!/bin/bash
# usage (on the GPU box): train_lib_ab.sh name1 name2 ...  (scratch/ab/lib_<name>.so), two interleaved repetitions of the training bench
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "$@"; do
  ADF_LIB_PATH=$PWD/scratch/ab/lib_$v.so python bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train $v rep $rep', round(d['value'],1), 'graphs/s', round(d['ms_per_step'],2), 'loss', d.get('loss'), 'gnorm', d.get('grad_norm'))"
done
done

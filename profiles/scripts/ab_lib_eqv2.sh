#!/bin/bash
# usage (on the GPU box): eq_lib_ab.sh name1 name2 ...  (scratch/ab/lib_<name>.so): EquiformerV2 bench, two interleaved repetitions
export PYTHONPATH=$PWD
for rep in 1 2; do
for v in "$@"; do
  ADF_LIB_PATH=$PWD/scratch/ab/lib_$v.so python bench.py --model eqv2 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > /tmp/o.json
  python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print("$v rep $rep", round(d["value"],3), d.get("gpu_ms_per_pass"), d.get("sites_sha256_16"))
PY
done
done

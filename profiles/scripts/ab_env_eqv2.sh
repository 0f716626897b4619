#!/bin/bash
# usage (on the GPU box): ab_env_eqv2.sh VAR v1 v2 ...   (EquiformerV2 bench, one pass each, prints value + per-category ms + hash)
export PYTHONPATH=$PWD
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --model eqv2 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > /tmp/o.json
  python - <<PY
import json
d=json.load(open('/tmp/o.json'))
print("$var=$v", round(d["value"],3), d.get("gpu_ms_per_pass"), d.get("sites_sha256_16"))
PY
done

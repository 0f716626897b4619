#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for v in skip1 skip2; do
  for n in 200 1000; do ADF_LIB_PATH=$PWD/scratch/ab/lib_$v.so python profiles/scripts/msg_time.py $n 2>/dev/null | tail -1 | sed "s/^/$n systems /"; done
done
done

#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
o=gpurun_out/r4a; rm -rf $o; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_eqv2.py tests/test_gpu_parity.py -x -q -m gpu -k "config4 or tag_based" -s > $o/tests_new.log 2>&1; echo "new tests rc=$?" > $o/rc.txt
(time timeout 900 python3 bench.py > $o/bench.json 2> $o/bench.err); echo "bench rc=$?" >> $o/rc.txt
tail -c 600 $o/bench.err

#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
o=gpurun_out/r4b; rm -rf $o; mkdir -p $o
timeout 2400 python -m pytest tests -x -q -m gpu > $o/tests.log 2>&1; echo "tests rc=$?" > $o/rc.txt
tail -5 $o/tests.log
(time timeout 900 python3 bench.py > $o/bench.json 2> $o/bench.err); echo "bench rc=$?" >> $o/rc.txt

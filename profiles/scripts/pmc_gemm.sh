#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  raw=/tmp/pg_$RANDOM; rocprofv3 --pmc $set -d $raw -o g --output-format csv -- python3 profiles/scripts/gemm_bench.py > /tmp/pg.log 2>&1
  python3 - $raw <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    if "gemm16" in k:
        print(k, " ".join(f"{n}={max(v):.4g}" for n, v in c.items()))
PY
done

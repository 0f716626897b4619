#!/bin/bash
# SQ / TCP / LDS counters of rw_rbf_wgrad_kernel (csrc/rbf_wgrad.hip) in one training step.  usage (GPU box): rw_pmc.sh [lib]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
[ -n "${1:-}" ] && export ADF_LIB_PATH=$PWD/$1
o=gpurun_out/rwpmc; rm -rf $o; mkdir -p $o
run() { rocprofv3 --pmc "$@" -d $o/$tag -o p --output-format csv -- python3 bench.py --mode train --steps 1 --warmup 0 > $o/$tag.log 2>&1; }
tag=a; run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
tag=b; run SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
tag=c; run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVES
tag=d; run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
tag=e; run TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
python3 - $o <<'PY'
import csv, glob, collections, sys
o = sys.argv[1]
for tag in "abcde":
    per = collections.defaultdict(list)
    for f in glob.glob(f"{o}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rw_rbf_wgrad_kernel<false>" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(per.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
rm -rf $o/a $o/b $o/c $o/d $o/e

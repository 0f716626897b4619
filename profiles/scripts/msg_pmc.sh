#!/bin/bash
# SQ counters of the message kernel for one or more library builds. usage: msg_pmc.sh "libA libB" nsys
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
libs=${1:-"adsorbdiff_amd/libadsorbdiff_hip.so"}; n=${2:-200}
o=gpurun_out/msgpmc; rm -rf $o; mkdir -p $o
i=0
for l in $libs; do
  i=$((i+1))
  export ADF_LIB_PATH=$PWD/$l ADF_MSG_KERNEL=v1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d $o/a$i -o a --output-format csv -- python3 profiles/scripts/msg_time.py $n > $o/a$i.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_IFETCH SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $o/b$i -o b --output-format csv -- python3 profiles/scripts/msg_time.py $n > $o/b$i.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $o/c$i -o c --output-format csv -- python3 profiles/scripts/msg_time.py $n > $o/c$i.log 2>&1
  echo "== $l"
  python3 - $o $i <<'PY'
import csv, glob, collections, sys
o, i = sys.argv[1], sys.argv[2]
for tag in ("a", "b", "c"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{o}/{tag}{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "message_kernel" in r["Kernel_Name"]:
                per[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in per.items():
        print(k)
        for c, v in sorted(d.items()):
            full = [x for x in v if x > 0.5 * max(v)] or v
            print(f"   {c:28s} n={len(v):3d} mean_full={sum(full)/len(full):.4g}")
PY
done

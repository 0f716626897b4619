#!/bin/bash
# SQ counters of the two message kernels on the comparison harness (profiles/scripts/r04_msg_cmp.py)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
o=gpurun_out/r4pmc; rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d $o/a -o a --output-format csv -- python3 profiles/scripts/r04_msg_cmp.py ${1:-200} > $o/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_IFETCH SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $o/b -o b --output-format csv -- python3 profiles/scripts/r04_msg_cmp.py ${1:-200} > $o/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r4pmc/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "message" in r["Kernel_Name"]:
                per[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in per.items():
        print(k)
        for c, v in sorted(d.items()):
            full = [x for x in v if x > 0.5 * max(v)] or v
            print(f"   {c:28s} n={len(v):3d} mean_full={sum(full)/len(full):.4g}")
PY

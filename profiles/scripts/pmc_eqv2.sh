#!/bin/bash
# per-kernel memory-unit counters of one EquiformerV2 forward (64 systems); raw output stays in /tmp, summaries come back
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"; export TMPDIR=/tmp
o=gpurun_out/pmc_eqv2; mkdir -p $o
i=0
for set in "MemUnitStalled WriteUnitStalled" "L2CacheHit VALUBusy" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); raw=/tmp/pmc_raw_$i; rm -rf $raw
  rocprofv3 --pmc $set -d $raw -o run --output-format csv -- python3 profiles/scripts/eqv2_time.py 64 > $o/set$i.log 2>&1
  echo "set $i rc=$?"; tail -n 3 $o/set$i.log
  python3 - $raw $o/set${i}_summary.txt "$set" <<'PY'
import csv, glob, collections, sys
raw, out, cs = sys.argv[1], sys.argv[2], sys.argv[3]
files = glob.glob(raw + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("files", len(files), "kernels", len(agg))
with open(out, "w") as w:
    w.write("# rocprofv3 --pmc " + cs + " -- python3 profiles/scripts/eqv2_time.py 64 (per-launch means, n = launches)\n")
    for k, c in sorted(agg.items()):
        line = k + " " + " ".join(f"{n}={sum(v)/len(v):.4g}(n={len(v)})" for n, v in c.items())
        w.write(line + "\n")
        if any(t in k for t in ("rotate", "s2act", "gemm16p", "gemm16_256", "from_grid", "to_grid", "alpha")):
            print(line)
PY
done

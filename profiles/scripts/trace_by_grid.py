"""Group a rocprofv3 --kernel-trace csv by (kernel, grid): launches, mean / total duration.  Usage: trace_by_grid.py <dir> [top]"""
import collections, csv, glob, sys

rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print(f"total kernel time {tot / 1e6:.1f} ms")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{sum(v) / 1e6:9.2f} ms {100 * sum(v) / tot:5.1f}%  n={len(v):5d} avg={sum(v) / len(v) / 1e3:9.1f} us  grid={k[1]:>9} wg={k[2]:>4}  {k[0]}")

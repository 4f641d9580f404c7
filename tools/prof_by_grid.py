#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup) so GEMM shapes can be told apart."""
import csv, re, sys
from collections import defaultdict
src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # drop the first `skip` fraction (percent) of dispatches (warm-up)
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * skip // 100:]
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r.get("Grid_Size_Y", 1) or 1), int(r.get("LDS_Block_Size", 0) or 0))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[key][0] += 1
    agg[key][1] += d
tot = sum(v[1] for v in agg.values())
with open(dst, "w") as f:
    f.write(f"# {title}\n\ntotal kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches (first {skip}% dropped as warm-up)\n\n")
    f.write("| kernel | workgroups x | grid y | LDS B | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
    for (name, gx, gy, lds), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        f.write(f"| `{name}` | {gx} | {gy} | {lds} | {n} | {t/1e6:.2f} | {t/n/1e3:.1f} | {100*t/tot:.2f} |\n")
print("wrote", dst)

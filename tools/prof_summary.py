#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) into a committed summary table."""
import csv, re, sys
src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    f.write(f"# {title}\n\n")
    f.write("command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...` (see title)\n\n")
    f.write(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in rows[:30]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(.*", "", name)[:90]
        f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print("wrote", dst)

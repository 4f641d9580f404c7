#!/usr/bin/env python3
"""Mean per-launch value of every collected counter for kernels matching a substring
(rocprofv3 --pmc output dirs given as arguments after the substring)."""
import csv, glob, sys, collections
sub = sys.argv[1]
acc = collections.defaultdict(list)
for d in sys.argv[2:]:
    for path in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if sub in r["Kernel_Name"]:
                acc[(r["Counter_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
for (c, g), v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"grid {g:9d}  {c:34s} n={len(v):3d}  mean {sum(v)/len(v):16.1f}")

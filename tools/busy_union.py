#!/usr/bin/env python3
"""GPU busy time from a rocprofv3 kernel trace (CSV): over the last `frac` of the trace, the union of all kernel intervals,
the sum of kernel durations and the wall span — how much of the step the two lanes leave idle, and how much they overlap.
Usage: tools/busy_union.py <kernel_trace.csv> [first=-3500 last=-500]  (kernel indices in start order: a window inside the timed steps)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else -3500
last = int(sys.argv[3]) if len(sys.argv) > 3 else -500
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)[first:last]
span = iv[-1][1] - iv[0][0]
tot = sum(e - s for s, e in iv)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"{len(iv)} kernels over {span / 1e6:.2f} ms: busy (union) {union / 1e6:.2f} ms = {100.0 * union / span:.1f} %, "
      f"sum of durations {tot / 1e6:.2f} ms = {tot / span:.2f} x the span (overlap of the lanes), idle {100.0 * (span - union) / span:.1f} %")

#!/usr/bin/env python3
"""Average number of library kernels in flight, from a rocprofv3 --kernel-trace CSV: the sum of the kernels'
durations over the length of the union of their intervals (1.0 = never two at once).
    python3 tools/trace_overlap.py <kernel_trace.csv> [name substring = ky::]"""
import csv
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else "ky::"
iv = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
busy = sum(b - a for a, b in iv)
union, cur_a, cur_b = 0, None, None
for a, b in iv:
    if cur_b is None or a > cur_b:
        if cur_b is not None:
            union += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
if cur_b is not None:
    union += cur_b - cur_a
print("%d kernels, %.2f ms of kernel time in %.2f ms with at least one running: %.2f in flight on average" % (len(iv), busy / 1e6, union / 1e6, busy / max(union, 1)))

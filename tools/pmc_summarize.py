#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel name:
   python3 tools/pmc_summarize.py gpurun_out/pmc_*/ > profiles/rNN_pmc_summary.txt"""
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg):
    if not k.startswith("ky::"):
        continue
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s total %16.0f  launches %4d  per-launch %16.1f" % (c, agg[k][c], cnt[k][c], agg[k][c] / cnt[k][c]))

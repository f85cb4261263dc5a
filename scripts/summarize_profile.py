#!/usr/bin/env python3
"""Condenses a scripts/profile.sh output directory into a text summary (kernel stats + per-kernel
PMC means) suitable for committing under profiles/."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 12:
            print("%-90s calls=%5s avg_ns=%12s total_ns=%14s pct=%s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
print()
print("== per-dispatch trace_bvh durations from kernel_trace.csv (ns), by grid size ==")
for f in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trace_bvh" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][-48:], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(d.items()):
        print("%-50s grid=%9s n=%4d mean=%10.0f min=%10d max=%10d" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)))
print()
print("== PMC (one pass per counter set), mean per dispatch ==")
summary = {}
for f in sorted(glob.glob(out + "/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "trace_bvh" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-48:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        for c, v in cs.items():
            print("%-50s grid=%9s %-30s n=%4d mean=%.5g" % (k[0], k[1], c, len(v), sum(v) / len(v)))
            summary["%s|%s|%s" % (k[0], k[1], c)] = sum(v) / len(v)
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1)

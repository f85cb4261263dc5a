#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into text for profiles/.

  summarize_rocprof.py trace <dir>   per kernel x grid size: dispatch count, mean / min / max duration (ns) from
                                     *_kernel_trace.csv (what `--kernel-trace --stats` aggregates, split by launch shape)
  summarize_rocprof.py pmc <dir>...  per kernel x grid size x counter: mean per dispatch from *_counter_collection.csv
  summarize_rocprof.py json <pmc.txt>...  the `pmc` text of several workloads -> one {"kernel|grid|counter": mean} object
                                     (profiles/*_pmc_summary.json, which bench.py reads for roofline.binding)
Kernel names are shortened to the function name."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("ntr::", "").replace("(anonymous namespace)::", "")
    return name[-70:]


def trace(d):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[(short(r["Kernel_Name"]), r["Grid_Size_X"], r.get("Workgroup_Size_X", "?"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in rows.values()) or 1
    print("%-72s %10s %5s %6s %12s %12s %12s %6s" % ("kernel", "grid", "wg", "calls", "mean_ns", "min_ns", "max_ns", "pct"))
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print("%-72s %10s %5s %6d %12.0f %12d %12d %6.2f" % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))


def pmc(dirs):
    """One rocprofv3 --pmc pass per directory.  Besides every counter's per-dispatch mean, each pass contributes what its OWN launches
    cost: `<counter>@ns` (mean duration of the same kernel x grid in this pass's kernel trace) and, when the pass also collected
    GRBM_GUI_ACTIVE, `<counter>@cycles` (GRBM_GUI_ACTIVE / 8 XCDs) -- so that a share is a pass's counts over the same pass's cycles."""
    for d in dirs:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                agg[(short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = collections.defaultdict(list)
        for f in sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                dur[(short(r["Kernel_Name"]), r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, cs in sorted(agg.items()):
            gui = cs.get("GRBM_GUI_ACTIVE")
            for c, v in sorted(cs.items()):
                print("%-72s grid=%10s %-32s n=%4d mean=%.6g" % (k[0], k[1], c, len(v), sum(v) / len(v)))
                if c == "GRBM_GUI_ACTIVE":
                    continue
                if k in dur:
                    print("%-72s grid=%10s %-32s n=%4d mean=%.6g" % (k[0], k[1], c + "@ns", len(dur[k]), sum(dur[k]) / len(dur[k])))
                if gui:
                    print("%-72s grid=%10s %-32s n=%4d mean=%.6g" % (k[0], k[1], c + "@cycles", len(gui), sum(gui) / len(gui) / 8.0))
            if gui and k in dur:
                print("%-72s grid=%10s %-32s n=%4d mean=%.6g" % (k[0], k[1], "GRBM_GUI_ACTIVE@ns", len(dur[k]), sum(dur[k]) / len(dur[k])))


def to_json(files):
    out = {}
    pat = re.compile(r"^(.*?)\s+grid=\s*(\d+)\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)$")
    for f in files:
        for line in open(f):
            m = pat.match(line.rstrip())
            if m and not m.group(1).startswith(("__amd", "l<", "void at::")):
                out["%s|%s|%s" % (m.group(1).strip(), m.group(2), m.group(3))] = float(m.group(4))
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "json":
        to_json(sys.argv[2:])
    elif sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2:])

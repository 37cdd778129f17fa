#!/usr/bin/env python3
"""Mean per-dispatch PMC counters per kernel from a rocprofv3 --output-format csv run.
usage: pmc_summary.py <dir> [name-substring ...]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
keys = sys.argv[2:] or ["mf::"]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not files:
    raise SystemExit(f"no counter_collection.csv under {d}")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    if any(s in k for s in keys):
        print(k, {n: round(sum(v) / len(v), 1) for n, v in sorted(c.items())}, "dispatches", len(next(iter(c.values()))))

# kernel durations from the --kernel-trace csv of the same run
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(dur.items()):
        if any(s in k for s in keys):
            print("  duration_us", k, round(sum(v) / len(v) / 1e3, 1), "n", len(v))

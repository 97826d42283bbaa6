#!/usr/bin/env python3
"""Median per-kernel PMC values from a rocprofv3 counter_collection CSV directory, as ratios to SQ_WAVE_CYCLES when present.
Usage: pmc_view.py <dir> [name-filter ...]"""
import csv, glob, statistics, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        vals[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2:]
for k, cs in sorted(vals.items()):
    if flt and not any(t in k[0] for t in flt): continue
    wc = statistics.median(cs["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in cs else None
    print(f"{k[0]} grid {k[1]} (x{len(next(iter(cs.values())))})")
    for c, v in sorted(cs.items()):
        m = statistics.median(v)
        print(f"    {c:28s} {m:16.0f}" + (f"   {m / wc:6.3f} of wave cycles" if wc and c != "SQ_WAVE_CYCLES" else ""))

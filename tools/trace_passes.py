#!/usr/bin/env python3
"""Summarise a WM_TRACE_EVENTS=1 stderr log: per pass, encoder start/end and decode start/end (ms), and how many decode
chains were active on average.  Usage: trace_passes.py <stderr log>"""
import re, sys, collections
ev = []
for l in open(sys.argv[1]):
    m = re.match(r"\[wm-trace\]\s+([\d.]+) ms\s+state (\S+) (.*)", l)
    if m: ev.append((float(m.group(1)), m.group(2), m.group(3).strip()))
ev.sort()
ids = {}
cur = {}
passes = []
for t, s, what in ev:
    sid = ids.setdefault(s, len(ids))
    if what == "encoder start": cur[sid] = {"slot": sid, "enc0": t}
    elif what == "encoder end": cur[sid]["enc1"] = t
    elif what.endswith("decode start"): cur[sid]["dec0"] = t
    elif what.endswith("prefill end"): cur[sid]["pre"] = t
    elif what.endswith("decode end"):
        cur[sid]["dec1"] = t
        passes.append(cur.pop(sid))
for p in passes:
    print(f"slot {p['slot']}: enc {p['enc0']:8.2f}-{p['enc1']:8.2f} ({p['enc1']-p['enc0']:5.2f})  decode {p['dec0']:8.2f}-{p['dec1']:8.2f} ({p['dec1']-p['dec0']:6.2f}; prefill {p['pre']-p['dec0']:4.2f})")
ends = [p["dec1"] for p in passes]
if len(ends) > 4:
    k = len(ends) // 3
    print(f"steady-state period (last {len(ends)-k} passes): {(ends[-1]-ends[k-1])/(len(ends)-k):.2f} ms per pass")

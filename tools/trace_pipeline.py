#!/usr/bin/env python3
"""Parse a rocprofv3 kernel-trace CSV of the pipelined bench: per stream (Queue_Id/Stream), when do encoder kernels run and
how fast do decode steps (argmax_step launches) tick while they do?  Usage: trace_pipeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
t0 = rows[0]["s"]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
enc = [r for r in rows if "flash_attn_enc" in r["Kernel_Name"] or "gemm_nt_lds" in r["Kernel_Name"] or "layernorm_rows" in r["Kernel_Name"]]
steps = collections.defaultdict(list)
for r in rows:
    if "argmax_step" in r["Kernel_Name"]: steps[r[qkey]].append(r["e"])
# encoder busy intervals (merge gaps < 20 us)
iv = []
for r in enc:
    if iv and r["s"] - iv[-1][1] < 20000: iv[-1][1] = max(iv[-1][1], r["e"])
    else: iv.append([r["s"], r["e"]])
iv = [x for x in iv if x[1] - x[0] > 500000]
print("encoder-active intervals (ms from start):", [(round((a - t0) / 1e6, 2), round((b - t0) / 1e6, 2)) for a, b in iv][-8:])
def in_enc(t): return any(a <= t <= b for a, b in iv)
for q, ts in steps.items():
    d_in, d_out = [], []
    for a, b in zip(ts, ts[1:]):
        if b - a > 2e6: continue
        (d_in if in_enc(b) and in_enc(a) else d_out).append((b - a) / 1e3)
    if d_in and d_out:
        print(f"queue {q}: {len(ts)} steps; step period while an encoder runs: {sum(d_in)/len(d_in):.1f} us ({len(d_in)}), otherwise {sum(d_out)/len(d_out):.1f} us ({len(d_out)})")
tot = (rows[-1]["e"] - t0) / 1e6
print(f"trace span {tot:.1f} ms; encoder-active total {sum(b - a for a, b in iv)/1e6:.1f} ms")

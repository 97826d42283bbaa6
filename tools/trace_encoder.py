#!/usr/bin/env python3
"""Parse a rocprofv3 kernel-trace CSV and print the per-launch durations of the LAST encoder chunk (the launches between
the last two mel_transpose_pad dispatches ... end).  Usage: trace_encoder.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mel_transpose_pad" in r["Kernel_Name"]]
lo = idx[-1]
names = ["mel_transpose", "conv1", "conv2"] + [f"L{l}.{n}" for l in range(4) for n in ("ln1", "qkv", "attn", "o", "ln2", "fc1", "fc2")] + ["ln_post", "cross_kv"]
tot = 0
for k, r in enumerate(rows[lo:lo + len(names)]):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    print(f"{names[k]:14s} {dur:9.1f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}  {r['Kernel_Name'][:60]}")
span = (int(rows[lo + len(names) - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
print(f"sum {tot:.1f} us, span {span:.1f} us")

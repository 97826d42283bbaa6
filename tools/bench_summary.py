#!/usr/bin/env python3
"""Reads bench.py JSON lines from the files named on the command line (or stdin) and prints a compact summary (developer tool)."""
import fileinput, json, sys
for line in fileinput.input():
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r, ds, e = d["roofline"], d["decode_step"], d["encoder"]
    print(f"{d['config']['name']}: RTF {d['value']:.0f}  {d['ms_per_step']:.2f} ms/pass  tok/s {d['tokens_per_sec']:.0f} | cross-attn {r['us_per_launch']} us "
          f"{r['achieved']} GB/s ({r['frac']:.3f}) | step(1 lane) {ds['us']} us {ds['GBps']} GB/s | enc {e['ms']} ms {e['TFLOPs']} TF"
          + (f" | step x4 {d['decode_step_4_in_flight']['us_per_step_of_each_chain']} us {d['decode_step_4_in_flight']['aggregate_GBps']} GB/s ({d['decode_step_4_in_flight']['frac_of_hbm_peak']:.3f})" if "decode_step_4_in_flight" in d else "")
          + (f" | unpipelined {d['unpipelined']['ms_per_step']} ms" if d.get("unpipelined") else "")
          + (f" | cpu RTF {d['cpu_baseline']['value']} ({d['cpu_baseline']['cores']} thr)" if "cpu_baseline" in d else ""))

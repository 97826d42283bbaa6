#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes of `python tools/enc_bench.py 2` (--pmc SQ_VALU_MFMA_BUSY_CYCLES and
--pmc GRBM_GUI_ACTIVE, each with --kernel-trace only) into MFMA utilisation per encoder kernel:
busy cycles summed over the chip's 1024 SIMDs / (1024 x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
is the sum over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back').  Usage: pmc_mfma.py <busy_dir> <active_dir> <out.json>"""
import csv, glob, json, statistics, sys

def collect(d, counter):
    vals = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return vals

busy, act = collect(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES"), collect(sys.argv[2], "GRBM_GUI_ACTIVE")
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / --pmc GRBM_GUI_ACTIVE (separate passes, --kernel-trace only) of "
               "`python tools/enc_bench.py 2` (tiny, 64 clips, bf16).  utilisation = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).  "
               "Reduced by tools/pmc_mfma.py.", "kernels": {}}
for key in sorted(busy):
    name, grid = key
    if key not in act or not any(t in name for t in ("gemm_nt", "flash_attn")): continue
    b, a = statistics.median(busy[key]), statistics.median(act[key])
    cyc = a / 8.0
    out["kernels"][f"{name[:70]} grid {grid}"] = {"launches": len(busy[key]), "mfma_busy_cycles": b, "kernel_cycles": round(cyc),
                                                  "mfma_utilisation": round(b / (1024.0 * cyc), 4)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items(): print(f"{v['mfma_utilisation']:.3f}  {v['kernel_cycles']:>8} cyc  x{v['launches']:<4} {k}")

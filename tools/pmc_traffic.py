#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only) of
`python bench.py --steps 1 --warmup 0 --no-cpu-baseline` into profiles/<name>.json: HBM bytes per launch of the
cross-attention and logits kernels, with the gfx950 correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE reports half of a
wide coalesced 16 B/lane streaming read; WRITE_SIZE is exact; both in KB).
Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <algorithmic_bytes_cross_attn> [workload]
workload tiny_b64_bf16enc_f32dec (round 3's headline): the fp32-KV instantiation of the cross-attention and the split-fp32 logits kernel."""
import csv, glob, json, statistics, sys

def collect(d, counter):
    vals = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return vals

fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --steps 1 "
               "--warmup 0 --no-cpu-baseline` (tiny_b64_bf16). Units: counter value = KB.  gfx950 correction (MI355X_MICROARCH.md "
               "§HBM): FETCH_SIZE reports 1/2 of a wide coalesced 16 B/lane streaming read -> fetch_bytes = 2 * FETCH_SIZE * 1024; "
               "WRITE_SIZE is exact.  Reduced by tools/pmc_traffic.py.", "kernels": {}}
workload = sys.argv[5] if len(sys.argv) > 5 else "tiny_b64_bf16"
out["note"] = out["note"].replace("(tiny_b64_bf16)", f"({workload})")
f32 = workload == "tiny_b64_bf16enc_f32dec"
alg = {"cross": float(sys.argv[4]), "logits": 51865 * 384 * (4 if f32 else 2) + 64 * 384 * 4 + 64 * 250 * 8}  # embedding once + x + (value, index) partials
kernels = ((("cross", ("attn_decode_kernel<float, 16, false, true, 4, 1>", "attn_decode_kernelIfLi16ELb0ELb1ELi4ELi1E"), "attn_decode_kernel<float,NT> (cross-attention, fp32 KV, one layer, B=64)"),
            ("logits", ("dec_logits_split_kernel",), "dec_logits_split_kernel<3,4> (fp32 embedding, B=64)")) if f32 else
           (("cross", ("attn_decode_kernelIDF16bLi8ELb1ELb1ELi4ELi1E",), "attn_decode_kernel<bf16,NT> (cross-attention, one layer, B=64)"),  # the step kernel (NQ = 1), not the 4-position prefill variant
            ("logits", ("dec_logits_kernelIDF16b",), "dec_logits_kernel<bf16> (B=64)")))
for key, pats, label in kernels:
    fk = [k for k in fetch if any(pat in k for pat in pats)]
    if not fk: continue
    fv, wv = fetch[fk[0]], write.get(fk[0], [0.0])
    fb, wb = 2 * statistics.median(fv) * 1024, statistics.median(wv) * 1024
    out["kernels"][label] = {"FETCH_SIZE_KB_median": statistics.median(fv), "WRITE_SIZE_KB_median": statistics.median(wv),
                             "launches_sampled": len(fv), "hbm_fetch_bytes_corrected": fb, "hbm_write_bytes": wb,
                             "traffic_bytes": fb + wb, "algorithmic_bytes": alg[key],
                             "traffic_over_algorithmic": round((fb + wb) / alg[key], 4)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))

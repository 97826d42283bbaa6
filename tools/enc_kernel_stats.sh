#!/bin/bash
# Per-kernel averages of the encoder alone (tools/enc_bench.py) under rocprofv3 --kernel-trace --stats.  Run on the GPU box.
# Prints one line per (kernel, grid): calls, average µs.  Extra environment (WM_USE_DEV_LIB=1 WM_GEMM_NO_FULLROW=1 ...) is inherited.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/encstats_$$; rm -rf $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/enc_bench.py 10 > /dev/null 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"][:86], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/tot*100:5.1f} %  x{len(v):<4} {sum(v)/len(v):8.1f} us  grid {k[1]:>8}  {k[0]}")
PY

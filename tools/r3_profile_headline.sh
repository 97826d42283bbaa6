#!/bin/bash
# rocprofv3 per-kernel stats of the round-3 headline (BASELINE config 3 literal: bf16 encoder, fp32 decoder + KV), every launch alone.
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-a}
O=gpurun_out/r3_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -- python3 bench.py --workload tiny_b64_bf16enc_f32dec --no-pipeline --no-cpu-baseline --no-x4 --no-extras --steps 4 --warmup 1 > $O/bench_solo.json 2> $O/bench_solo.err
cp $(ls $O/solo/*/*kernel_stats.csv | head -1) $O/kernel_stats_solo.csv
cat $O/bench_solo.json
python3 - $O/kernel_stats_solo.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s}%")
PY

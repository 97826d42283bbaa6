#!/bin/bash
# HBM traffic of the headline's dominant kernel (cross-attention at fp32 KV) and of the split-fp32 logits kernel: separate
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only), reduced with the gfx950 correction by tools/pmc_traffic.py.
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3_pmc; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-x4 --no-extras --no-pipeline"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- $CMD > $O/fetch.json 2> $O/fetch.err
echo "[$(date +%T)] fetch pass done"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- $CMD > $O/write.json 2> $O/write.err
echo "[$(date +%T)] write pass done"
ALG=$(python3 -c "import json; print(json.load(open('$O/fetch.json'))['roofline']['bytes_per_launch'])")
python3 tools/pmc_traffic.py $O/fetch $O/write $O/r3_pmc_traffic.json $ALG tiny_b64_bf16enc_f32dec

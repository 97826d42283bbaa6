#!/bin/bash
# A/B of developer switches on the headline workload, PIPELINED (coalesced, 8 submits in flight): ms per 64-clip pass.
# usage: tools/r3_ab_pipe.sh "VAR=1" "OTHER=2" ...   (each argument = one arm)
cd $GRAFT_REPO_ROOT
export WM_USE_DEV_LIB=1
for arm in "$@"; do
  for rep in 1 2; do
    env $arm python bench.py --workload ${WM_AB_WORKLOAD:-tiny_b64_bf16enc_f32dec} --no-extras --no-cpu-baseline --no-x4 --steps 16 --warmup 8 ${WM_AB_ARGS:-} 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('arm [$arm] rep $rep: ms/pass', d['ms_per_step'], 'value', d['value'], 'alone', d['unpipelined']['ms_per_step'])"
  done
done

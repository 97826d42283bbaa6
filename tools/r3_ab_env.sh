#!/bin/bash
# A/B of developer switches on the headline workload, passes strictly one after another (unpipelined ms per pass + decode step).
# usage: tools/r3_ab_env.sh "VAR=1 VAR2=2" "OTHER=1" ...   (each argument = one arm; "" = defaults)
cd $GRAFT_REPO_ROOT
export WM_USE_DEV_LIB=1
for arm in "$@"; do
  for rep in 1 2; do
    env $arm python bench.py --workload ${WM_AB_WORKLOAD:-tiny_b64_bf16enc_f32dec} --no-pipeline --no-extras --no-cpu-baseline --no-x4 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('arm [$arm] rep $rep: ms/pass', d['ms_per_step'], 'step', d['decode_step']['us'], 'in-pass', d['decode_step'].get('in_pass_avg_us'), 'enc', d['encoder']['ms'])"
  done
done

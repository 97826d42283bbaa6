#!/bin/bash
export WM_USE_DEV_LIB=1
run() { WM_DEV_LIB_PATH=$PWD/whisper.mojo_amd/csrc/libwm_$1.so python bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', 'ms/pass', d['ms_per_step'], 'alone', d['unpipelined']['ms_per_step'], 'step', d['decode_step']['us'], 'x4', d['decode_step_4_in_flight']['us_per_step_of_each_chain'])"; }
for i in 1 2 3; do run A; run B; done

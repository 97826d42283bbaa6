#!/bin/bash
export WM_USE_DEV_LIB=1
run() { python bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', 'ms/pass', d['ms_per_step'], 'alone', d['unpipelined']['ms_per_step'], 'xattn us', d['roofline']['us_per_launch'], 'step', d['decode_step']['us'], 'x4', d['decode_step_4_in_flight']['us_per_step_of_each_chain'])"; }
run kpw3
WM_LIN_KPW=6 run kpw6
WM_LIN_KPW=4 run kpw4

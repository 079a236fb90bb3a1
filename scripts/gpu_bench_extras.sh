#!/bin/bash
HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 48 --warmup 16 --no-cpu-baseline --config ${1:-cfg3} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=d['two_chains_per_gpu']
print('value', round(d['value'],1), 'near_true', round(d['near_true_state']['steps_per_s'],1), 'two_chains', round(t['steps_per_s_aggregate'],1), t['persistent_solves'], t['slots_per_xcd'], 'straight', round(d['straight_line']['steps_per_s'],1), 'cold', round(d['cold_start']['steps_per_s'],1))
"

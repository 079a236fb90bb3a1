#!/bin/bash
# round 5 (third session): same-box A/B of two builds (build_ab/<name>.so via HMCMT_LIB_PATH), headline protocol, alternating
mkdir -p gpurun_out
A=${A:-base}; B=${B:-mom}; REPS=${REPS:-3}; CFGS=${CFGS:-cfg3}
for rep in $(seq $REPS); do
for lib in $A $B; do
for cfg in $CFGS; do
  HMCMT_LIB_PATH=$PWD/build_ab/$lib.so HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 96 --warmup 16 --config $cfg --no-cpu-baseline --no-extras 2>gpurun_out/r5c_${lib}_${cfg}.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $cfg rep $rep', round(d['value'],1), 'steps/s  check', d['check']['true_res_max_at_last_model'], 'iters', d['chain']['iters_fwd_max_last_step_mean'], d['chain']['iters_adj_max_last_step_mean'])"
done
done
done

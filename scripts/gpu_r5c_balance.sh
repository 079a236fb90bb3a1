#!/bin/bash
# round 5 (third session): the queues of the persistent kernel balanced from the previous solve's iteration counts (cfg5), on / off
mkdir -p gpurun_out
REPS=${REPS:-3}; CFGS=${CFGS:-cfg5}
for rep in $(seq $REPS); do
for bal in 0 1; do
for cfg in $CFGS; do
  HMCMT_PERSIST_BALANCE=$bal HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 96 --warmup 16 --config $cfg --no-cpu-baseline --no-extras 2>gpurun_out/r5c_bal${bal}_${cfg}.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('balance $bal $cfg rep $rep', round(d['value'],1), 'steps/s  check', d['check']['true_res_max_at_last_model'], 'iters', d['chain']['iters_fwd_max_last_step_mean'], d['chain']['iters_adj_max_last_step_mean'])"
done
done
done

#!/bin/bash
# round 5 (third session): knobs of the iteration count re-measured on the final build (headline protocol, same box)
mkdir -p gpurun_out
run() { # label, cfg, env...
  local label=$1 cfg=$2; shift 2
  env "$@" HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 96 --warmup 16 --config $cfg --no-cpu-baseline --no-extras 2>gpurun_out/r5c_knobs.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label $cfg', round(d['value'],1), 'steps/s iters', round(d['chain']['iters_fwd_max_last_step_mean'],1), round(d['chain']['iters_adj_max_last_step_mean'],1), 'check', d['check']['true_res_max_at_last_model'])"
}
for rep in 1 2; do
for cfg in cfg3 cfg5; do
  run "default" $cfg A=1
  for np in 2 3 4 6; do run "extrap_points=$np" $cfg HMCMT_EXTRAP_POINTS=$np; done
  for w in 0.7 0.9; do run "jacobi_w=$w" $cfg HMCMT_JACOBI_W=$w; done
done
done

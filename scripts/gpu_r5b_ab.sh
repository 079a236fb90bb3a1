#!/bin/bash
# round 5 (second session): generic vs width-specialised persistent kernel -- correctness, phase stamps, short bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_persist.py -x -q 2>&1 | tail -3
for wk in 0 1; do
for cfg in cfg3 cfg5; do echo "== widthK $wk $cfg"; HMCMT_PERSIST_WIDTHK=$wk timeout 300 python -m scripts.gpu_persist_stamps $cfg 2 2>&1 | tail -3; done
done
if [ "$1" != "nobench" ]; then
for wk in 0 1; do
for cfg in cfg3 cfg5; do
  HMCMT_PERSIST_WIDTHK=$wk HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 48 --warmup 16 --config $cfg --no-cpu-baseline --no-extras 2>gpurun_out/r5b_bench_${cfg}_$wk.err | tee gpurun_out/r5b_bench_${cfg}_$wk.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('widthK $wk $cfg', round(d['value'],1), 'steps/s  check', d['check']['true_res_max_at_last_model'], 'iters', d['chain']['iters_fwd_max_last_step_mean'], d['chain']['iters_adj_max_last_step_mean'])"
done
done
fi

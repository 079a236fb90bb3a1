#!/bin/bash
# round 5 quick loop: correctness of the persistent kernel's shapes, phase stamps, the two bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_persist.py -x -q 2>&1 | tail -3
for cfg in cfg3 cfg5; do timeout 300 python -m scripts.gpu_persist_stamps $cfg 2 2>&1 | tail -2; done
if [ "$1" != "nobench" ]; then
for cfg in cfg3 cfg5; do
  HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 48 --warmup 16 --config $cfg --no-cpu-baseline --no-extras 2>gpurun_out/r5_bench_$cfg.err | tee gpurun_out/r5_bench_$cfg.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value'],1), 'steps/s  check', d['check']['true_res_max_at_last_model'], 'iters', d['chain']['iters_fwd_max_last_step_mean'], d['chain']['iters_adj_max_last_step_mean'])"
done
fi

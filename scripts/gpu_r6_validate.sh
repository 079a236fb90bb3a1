#!/bin/bash
# round 6: the whole GPU suite, then the headline with the solves' start inside the persistent kernel (HMCMT_PS_START) against launches of their own
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r6_tests_all.log; tail -8 gpurun_out/r6_tests_all.log
for ps in 1 0 1 0; do
  HMCMT_PS_START=$ps HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ps_start $ps headline', d['value'], d['check']['true_res_max_at_last_model'])"
done
timeout 300 python scripts/gpu_ticks_chain.py rough 6 > gpurun_out/r6_ticks_rough.log 2>&1; tail -22 gpurun_out/r6_ticks_rough.log

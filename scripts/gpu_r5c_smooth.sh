#!/bin/bash
# round 5 (third session): weight of the older costs in the balance of the persistent kernel's queues (cfg5)
mkdir -p gpurun_out
for rep in 1 2 3; do
for sm in 0 0.5 0.75; do
  HMCMT_PERSIST_BALANCE_SMOOTH=$sm HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 96 --warmup 16 --config cfg5 --no-cpu-baseline --no-extras 2>gpurun_out/r5c_sm.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('smooth $sm rep $rep', round(d['value'],1), 'steps/s')"
done
done
for sm in 0 0.5 0.75; do HMCMT_PERSIST_BALANCE_SMOOTH=$sm timeout 500 python -m scripts.gpu_iters_by_system cfg5 48 2>&1 | grep "^kind" | sed "s/^/smooth $sm /"; done

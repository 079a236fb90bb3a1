#!/bin/bash
# round 6: quick check of a build of the four-strip persistent kernel (kernels_persist4.h): the persist tests, the phase stamps and the headline, strips 4 against 2
# usage: gpu_r6_quick.sh [notests]
mkdir -p gpurun_out
if [ "$1" != "notests" ]; then
timeout 1200 python -m pytest tests/test_gpu_persist.py -x -q 2>&1 | tail -15 > gpurun_out/r6_persist_tests.log; tail -5 gpurun_out/r6_persist_tests.log
fi
for st in 4 2; do
  for sw in 1 2; do echo "strips $st sweeps $sw"; HMCMT_PERSIST_STRIPS=$st timeout 200 python -m scripts.gpu_persist_stamps cfg3 $sw 2>&1 | grep -A2 "HMCMT_STAMPS persist"; done
done
for st in 4 2 4 2; do
  HMCMT_PERSIST_STRIPS=$st HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strips $st headline', d['value'], d['check']['true_res_max_at_last_model'])"
done

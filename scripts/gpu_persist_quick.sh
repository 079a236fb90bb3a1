#!/bin/bash
# round 4: quick check of a build of the persistent kernel: correctness on cfg2 / cfg3, phase stamps, headline
timeout 300 python -m scripts.gpu_persist_check cfg2 cfg3 > gpurun_out/persist_q.log 2>&1; grep "persist 1" gpurun_out/persist_q.log | tail -4
for sw in 1 2; do timeout 200 python -m scripts.gpu_persist_stamps cfg3 $sw 2>&1 | tail -1; done
HMCMT_PERSIST=1 HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['check']['true_res_max_at_last_model'])"

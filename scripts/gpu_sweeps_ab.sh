#!/bin/bash
# headline chain with the smoother fixed at one / two sweeps per side and chosen per solve (default)
for sw in 1 2 0; do
  HMCMT_SWEEPS=$sw HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HMCMT_SWEEPS=$sw headline', d['value'], d['check']['true_res_max_at_last_model'], d.get('iterations'))"
done

#!/bin/bash
# round 6: the layer-blocked boundary-field kernel (k_bc_blocked) -- tests, then rocprofv3 averages of the boundary kernels at the stress
# size per "columns,threads,up layers,down layers" (0: the two-kernel form)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python -m pytest tests/test_gpu_parity_full.py -q -m gpu -k "boundary_fields or cfg5" 2>&1 | tail -3
for cw in ${BCB_SET:-0 default 64,1024,15,8 32,512,14,8}; do
  echo "== HMCMT_BC_BLOCKED=$cw"
  d=gpurun_out/prof_bcb_${cw//,/_}
  rm -rf $d
  if [ $cw = default ]; then unset HMCMT_BC_BLOCKED; else export HMCMT_BC_BLOCKED=$cw; fi
  HMCMT_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 24 --warmup 8 --config cfg5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps/s', d['value'])"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  grep -E "k_bc_|k_coef_all" $f | awk -F, '{print $1, "calls", $2, "avg ns", $4, "min", $7, "max", $8}'
done

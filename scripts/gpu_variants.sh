#!/bin/bash
# A/B of builds of the library (HMCMT_LIB_PATH): phase stamps of the persistent kernel, parity check, headline
# usage: bash scripts/gpu_variants.sh lib1.so lib2.so ...   ("default" = the in-tree build)
for lib in "$@"; do
  if [ "$lib" = default ]; then unset HMCMT_LIB_PATH; else export HMCMT_LIB_PATH=$PWD/$lib; fi
  echo "== $lib"
  timeout 200 python -m scripts.gpu_persist_stamps cfg3 2 2>&1 | tail -2
  timeout 300 python -m scripts.gpu_persist_check cfg3 2>&1 | grep "sweeps 2 persist 1"
  HMCMT_PERSIST=1 HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['check']['true_res_max_at_last_model'])"
done

#!/bin/bash
# round 5 (second session): full validation of the width-specialised / pipelined persistent kernels
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5b_fullsuite2.log 2>&1
bash scripts/gpu_r5b_stamps.sh > gpurun_out/r5b_st10.log 2>&1
for cfg in cfg3 cfg5; do HMCMT_BENCH_NOPROF=1 timeout 600 python bench.py --steps 96 --warmup 16 --config $cfg --no-cpu-baseline --no-extras 2>gpurun_out/r5b_bench_$cfg.err > gpurun_out/r5b_bench_$cfg.json; done
for wk in 0 1; do echo "== widthK $wk"; HMCMT_PERSIST_WIDTHK=$wk timeout 600 python scripts/gpu_run_example.py dprism3d 600 100 2>&1 | tail -4; done > gpurun_out/r5b_example.log 2>&1

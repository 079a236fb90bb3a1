#!/bin/bash
# round 5, final numbers (second session: width-specialised / pipelined persistent kernels): phase stamps (one / two sweeps,
# generic and specialised), profiles of both configs (bench line, rocprofv3 --stats, PMC passes), SQ counters of the kernel
mkdir -p gpurun_out
for wk in 1 0; do for cfg in cfg3 cfg5; do for sw in 1 2; do echo "== stamps $cfg sweeps $sw widthK $wk"; HMCMT_PERSIST_WIDTHK=$wk timeout 300 python -m scripts.gpu_persist_stamps $cfg $sw 2>&1 | tail -3; done; done; done > gpurun_out/r05_stamps.log 2>&1
bash scripts/gpu_profile_all.sh r05_prof3 cfg3 > gpurun_out/r05_prof3.log 2>&1
bash scripts/gpu_profile_all.sh r05_prof5 cfg5 > gpurun_out/r05_prof5.log 2>&1
cat gpurun_out/r05_stamps.log

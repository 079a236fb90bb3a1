#!/bin/bash
# round 5, final numbers: phase stamps (one / two sweeps), profiles of both configs (twice: the second bench.py run reads the first run's PMC bytes)
mkdir -p gpurun_out
for cfg in cfg3 cfg5; do for sw in 1 2; do echo "== stamps $cfg sweeps $sw"; timeout 300 python -m scripts.gpu_persist_stamps $cfg $sw 2>&1 | tail -3; done; done > gpurun_out/r05_stamps.log 2>&1
bash scripts/gpu_profile_all.sh r05_prof3 cfg3 > gpurun_out/r05_prof3.log 2>&1
bash scripts/gpu_profile_all.sh r05_prof5 cfg5 > gpurun_out/r05_prof5.log 2>&1
cat gpurun_out/r05_stamps.log

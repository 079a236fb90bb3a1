#!/bin/bash
# round 5 bring-up: the persistent kernel after the argument-block change (CS = 1) and with column parts (CS = 2)
mkdir -p gpurun_out
echo "== CS=1 regression: persist tests"; timeout 900 python -m pytest tests/test_gpu_persist.py -x -q 2>&1 | tail -5
echo "== CS=2 forced on cfg2 / cfg3"; HMCMT_PERSIST_CS=2 timeout 600 python -m scripts.gpu_persist_check cfg2 cfg3 > gpurun_out/r5_cs2_small.log 2>&1; grep -v "^ *$" gpurun_out/r5_cs2_small.log | tail -40
echo "== cfg5 (CS=2 by shape)"; timeout 900 python -m scripts.gpu_persist_check cfg5 > gpurun_out/r5_cs2_cfg5.log 2>&1; tail -30 gpurun_out/r5_cs2_cfg5.log

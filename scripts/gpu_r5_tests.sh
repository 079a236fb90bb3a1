#!/bin/bash
# the whole GPU suite, the way the driver runs it
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu ${1:+-k "$1"} > gpurun_out/r5_gputests.log 2>&1; grep -n "Error\|assert\|^E " gpurun_out/r5_gputests.log | head -30; tail -8 gpurun_out/r5_gputests.log

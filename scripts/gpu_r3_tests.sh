#!/bin/bash
# the GPU test suite, all failures listed (arguments are passed to pytest)
O=gpurun_out/r3; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -s --durations=12 "$@" ) > $O/tests.log 2>&1
grep -n "^\[\|passed\|failed\|FAILED\|Error" $O/tests.log | head -60

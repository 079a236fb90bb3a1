#!/bin/bash
# round 3, first GPU call: the GPU test suite (new parity tests included), the bench at both step counts, and the
# 16-mode-slab variant of k_fdm_fwd (HMCMT_FWD_NTW=1) beside the default
O=gpurun_out/r3; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q -s --ignore=tests/test_gpu_posterior.py --durations=12 ) > $O/tests.log 2>&1
tail -25 $O/tests.log
timeout 600 python bench.py --steps 96 > $O/b96.json 2> $O/b96.err; python scripts/bench_brief.py default96 < $O/b96.json
timeout 600 python bench.py --steps 20 --no-extras --no-cpu-baseline > $O/b20.json 2> $O/b20.err; python scripts/bench_brief.py default20 < $O/b20.json
HMCMT_FWD_NTW=1 timeout 600 python bench.py --steps 96 --no-cpu-baseline > $O/b96_ntw1.json 2> $O/b96_ntw1.err; python scripts/bench_brief.py ntw1 < $O/b96_ntw1.json

#!/bin/bash
# A/B of library builds: bench.py (96 steps, no CPU leg) with the default library and with each build_ab/*.so given
# (arguments: names of files in build_ab/ without .so; environment passes through)
O=gpurun_out/r3; mkdir -p $O
python -c "from hmcmt2d_amd import lib; lib.build_library()" 
timeout 600 python bench.py --steps 96 --no-cpu-baseline > $O/ab_base.json 2> $O/ab_base.err; python scripts/bench_brief.py base < $O/ab_base.json
for v in "$@"; do
  HMCMT_LIB_PATH=$PWD/build_ab/$v.so timeout 600 python bench.py --steps 96 --no-cpu-baseline > $O/ab_$v.json 2> $O/ab_$v.err; python scripts/bench_brief.py $v < $O/ab_$v.json || tail -5 $O/ab_$v.err
done

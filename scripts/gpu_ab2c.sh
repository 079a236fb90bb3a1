#!/bin/bash
# as gpu_ab2.sh, with a config: gpu_ab2c.sh cfg5 "lib|ENV" ...
O=gpurun_out/r3; mkdir -p $O
CFG=$1; shift
i=0
for a in "$@"; do
  i=$((i+1)); lib=${a%%|*}; e=${a#*|}
  if [ "$lib" != "-" ]; then e="$e HMCMT_LIB_PATH=$PWD/build_ab/$lib.so"; fi
  env $e timeout 900 python bench.py --steps 96 --no-cpu-baseline --config $CFG > $O/ab2c_$i.json 2> $O/ab2c_$i.err; python scripts/bench_brief.py "[$a]" < $O/ab2c_$i.json || tail -3 $O/ab2c_$i.err
done

#!/bin/bash
# A/B of (library build, environment) pairs: each argument "lib|ENV..." (lib = name in build_ab/ without .so, or "-" for the default build)
O=gpurun_out/r3; mkdir -p $O
i=0
for a in "$@"; do
  i=$((i+1)); lib=${a%%|*}; e=${a#*|}
  if [ "$lib" != "-" ]; then e="$e HMCMT_LIB_PATH=$PWD/build_ab/$lib.so"; fi
  env $e timeout 900 python bench.py --steps 96 --no-cpu-baseline > $O/ab2_$i.json 2> $O/ab2_$i.err; python scripts/bench_brief.py "[$a]" < $O/ab2_$i.json || tail -3 $O/ab2_$i.err
done

#!/bin/bash
# A/B of environment settings: bench.py (96 steps, no CPU leg) once per argument, each argument a quoted env string ("" = default)
# usage: gpu_env_ab.sh [-c cfg] "" "HMCMT_XMAP=0" ...
O=gpurun_out/r3; mkdir -p $O
CFG=cfg3; if [ "$1" = "-c" ]; then CFG=$2; shift 2; fi
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 900 python bench.py --steps 96 --no-cpu-baseline --config $CFG > $O/env_$i.json 2> $O/env_$i.err; python scripts/bench_brief.py "[$e]" < $O/env_$i.json || tail -5 $O/env_$i.err
done

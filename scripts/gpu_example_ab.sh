#!/bin/bash
# A/B of environment settings on the reference's dprism3d example (96x49 cells, 22 systems): evaluations per second of a
# 600-sample chain; each argument a quoted env string
for e in "$@"; do
  r=$(env $e timeout 600 python scripts/gpu_run_example.py dprism3d 600 100 2>&1 | grep "samples in")
  echo "[$e] $r"
done

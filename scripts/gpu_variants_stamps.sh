#!/bin/bash
# phase stamps only, for builds whose numerics are deliberately broken (timing probes)
for lib in "$@"; do
  if [ "$lib" = default ]; then unset HMCMT_LIB_PATH; else export HMCMT_LIB_PATH=$PWD/$lib; fi
  echo "== $lib"
  timeout 200 python -m scripts.gpu_persist_stamps cfg3 2 2>&1 | tail -3
done

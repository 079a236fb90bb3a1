#!/bin/bash
# generic vs width-specialised persistent kernel: phase stamps only
for wk in ${WKS:-0 1}; do
for cfg in ${CFGS:-cfg3 cfg5}; do echo "== widthK $wk $cfg"; HMCMT_PERSIST_WIDTHK=$wk timeout 300 python -m scripts.gpu_persist_stamps $cfg 2 2>&1 | tail -3; done
done

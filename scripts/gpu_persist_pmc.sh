#!/bin/bash
# SQ counters of the persistent kernel on three cold evaluations of cfg3 (two sweeps): what the waves spend their cycles on
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/persist_pmc
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  (cd $R && HMCMT_PERSIST=1 HMCMT_SWEEPS=2 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 -m scripts.gpu_persist_pmc_child > $O/p$i.log 2>&1)
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_cocg_persist' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:36s} per launch {v / n:16.0f}   ({n} launches)")
PY
done

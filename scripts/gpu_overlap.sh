#!/bin/bash
# which kernels overlap in time with k_pivot / k_extrap / k_sens_profile (kernel trace of a few evaluations)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ovl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -- python3 $R/scripts/gpu_profile_run.py cfg3 6 > /tmp/ovl.log 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/ovl/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2] == 'k_sigma']
a, b = starts[-2], starts[-1]
t0 = ks[a][0]
for s, e, n, q in ks[a:b]:
    if (s - t0) / 1e3 < 420: print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  q{q}  {n}")
PY

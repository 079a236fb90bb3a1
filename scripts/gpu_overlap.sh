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
import os
lo, hi = float(os.environ.get("OVL_FROM", "0")), float(os.environ.get("OVL_TO", "420"))
tend = ks[b][0]
print("evaluation wall", (tend - t0) / 1e3, "us")
for s, e, n, q in ks[a:b + 3]:
    t = (s - t0) / 1e3
    if hi > 0 and lo <= t < hi or hi < 0 and t > (tend - t0) / 1e3 + hi: print(f"{t:8.1f} .. {(e - t0) / 1e3:8.1f} us  q{q}  {n}")
PY

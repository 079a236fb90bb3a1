#!/bin/bash
# rocprofv3 kernel stats of scripts/gpu_chain_traj.py <state> (environment passes through): the iteration kernels' average durations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/scripts/gpu_chain_traj.py ${1:-true} > /tmp/ks.log 2>&1
tail -2 /tmp/ks.log | cut -c1-100
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/ks/*/*kernel_stats.csv')[0])))
for r in rows[:9]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"{n:60s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:7.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms  {float(r['Percentage']):5.1f} %")
PY

#!/bin/bash
# per-kernel averages of N gradient evaluations (rocprofv3 --stats); usage: gpu_kstat.sh [cfg] [n] [grep pattern]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat -- python3 $R/scripts/gpu_profile_run.py ${1:-cfg3} ${2:-12} > /tmp/kstat.log 2>&1
python3 - "$3" <<'PY'
import csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r in csv.DictReader(open(glob.glob('/tmp/kstat/*/*kernel_stats.csv')[0])):
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if pat in n and float(r['TotalDurationNs']) > 2e4:
        print(f"{n:40s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY

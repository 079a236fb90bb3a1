#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc -- python3 $R/scripts/gpu_spin_calib.py 2>&1 | grep calibrated
python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob('/tmp/sc/*/*kernel_stats.csv')[0])):
    if 'k_spin' in r['Name'] or 'k_null' in r['Name']:
        print(r['Name'], 'calls', r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3, 'max', float(r['MaxNs'])/1e3)
PY
cd $R && python3 scripts/gpu_spin_calib.py

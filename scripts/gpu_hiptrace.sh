#!/bin/bash
# host-side API cost of the bench loop: rocprofv3 --hip-trace --kernel-trace --stats (no counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hiptrace
HMCMT_BENCH_NOPROF=1 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/hiptrace -- python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-sampler > /tmp/hiptrace.log 2>&1
tail -1 /tmp/hiptrace.log | cut -c80-170
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/hiptrace/*/*hip_api_stats.csv')
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f"{r['Name'][:40]:40s} calls {int(r['Calls']):7d} avg {float(r['AverageNs'])/1e3:8.2f} us total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {r['Percentage']}%")
PY

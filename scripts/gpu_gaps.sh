#!/bin/bash
# idle time of the main queue inside one steady-state evaluation (kernel trace): every gap > GAP_MIN us between
# consecutive kernels of the busiest queue, with the kernels on either side
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaps
# usage: gpu_gaps.sh [cfg]        host-API evaluations on distinct models (scripts/gpu_profile_run.py)
#        gpu_gaps.sh bench        the timed loop of bench.py (device API, leapfrog-like trajectories)
if [ "$1" = bench ]; then
    HMCMT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline > /tmp/gaps.log 2>&1
else
    rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python3 $R/scripts/gpu_profile_run.py ${1:-cfg3} 8 > /tmp/gaps.log 2>&1
fi
python3 - <<'PY'
import csv, glob, os, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/gaps/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2] == 'k_sigma']
gmin = float(os.environ.get("GAP_MIN", "6"))
for ev in ((12, 13) if os.environ.get('GAPS_BENCH') else (-3, -2)):
    a, b = starts[ev], starts[ev + 1]
    seg = ks[a:b + 1]
    mainq = collections.Counter(k[3] for k in seg).most_common(1)[0][0]
    m = [k for k in seg if k[3] == mainq]
    t0 = m[0][0]
    wall = (m[-1][0] - t0) / 1e3
    busy = sum(e - s for s, e, _, _ in m[:-1]) / 1e3
    print(f"evaluation wall {wall:.1f} us, main-queue kernels {busy:.1f} us ({len(m)-1} launches), idle {wall-busy:.1f} us")
    small = 0.0
    for (s0, e0, n0, _), (s1, e1, n1, _) in zip(m[:-1], m[1:]):
        g = (s1 - e0) / 1e3
        if g > gmin: print(f"   {(e0 - t0)/1e3:8.1f} us  gap {g:6.1f} us  {n0} -> {n1}")
        else: small += g
    print(f"   sum of the gaps <= {gmin} us: {small:.1f} us")
PY

#!/bin/bash
# Where a leapfrog step of a REAL chain goes (kernel trace of scripts/gpu_chain_traj.py): for the steady-state
# evaluations, wall time per evaluation, main-queue busy time by kernel, idle time and the gaps > GAP_MIN us.
# usage: gpu_step_breakdown.sh [true|rough]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sb
rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -- python3 $R/scripts/gpu_chain_traj.py ${1:-true} > /tmp/sb.log 2>&1
tail -3 /tmp/sb.log
python3 - <<'PY'
import csv, glob, os, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/sb/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2].startswith('k_sigma')]
gmin = float(os.environ.get("GAP_MIN", "8"))
n = len(starts)
evs = range(n - 30, n - 6)            # steady-state evaluations of the last trajectories
tot = collections.Counter(); cnt = collections.Counter(); wall = busy = 0.0; gaps = collections.Counter(); side = collections.Counter()
for ev in evs:
    a, b = starts[ev], starts[ev + 1]
    seg = ks[a:b]
    mainq = collections.Counter(k[3] for k in seg).most_common(1)[0][0]
    m = [k for k in seg if k[3] == mainq]
    wall += (ks[b][0] - m[0][0]) / 1e3
    for s, e, nm, q in seg:
        if q == mainq: tot[nm] += (e - s) / 1e3; cnt[nm] += 1; busy += (e - s) / 1e3
        else: side[nm] += (e - s) / 1e3
    for (s0, e0, n0, _), (s1, e1, n1, _) in zip(m[:-1], m[1:] ):
        g = (s1 - e0) / 1e3
        if g > gmin: gaps[f"{n0} -> {n1}"] += g
ne = len(list(evs))
print(f"per evaluation (mean of {ne}): wall {wall/ne:.1f} us, main-queue kernels {busy/ne:.1f} us, idle {(wall-busy)/ne:.1f} us")
for nm, t in tot.most_common(): print(f"   main {nm:28s} {t/ne:8.1f} us/eval  {cnt[nm]/ne:6.1f} launches  avg {t/cnt[nm]:6.1f} us")
for nm, t in side.most_common(8): print(f"   side {nm:28s} {t/ne:8.1f} us/eval")
print(f"gaps > {gmin} us (us per evaluation):")
for nm, t in gaps.most_common(12): print(f"   {t/ne:7.1f}  {nm}")
PY
python3 - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/sb/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2].startswith('k_sigma')]
ev = len(starts) - 12
a, b = starts[ev], starts[ev + 1]
seg = ks[a:b]
mainq = collections.Counter(k[3] for k in seg).most_common(1)[0][0]
m = [k for k in seg if k[3] == mainq]
t0 = m[0][0]; nit = 0
print("one evaluation: gaps > 5 us on the main queue (time since k_sigma, iteration count so far)")
for (s0, e0, n0, _), (s1, e1, n1, _) in zip(m[:-1], m[1:]):
    if n0 == 'k_spmv_fused': nit += 1
    g = (s1 - e0) / 1e3
    if g > 5: print(f"   {(e0 - t0)/1e3:8.1f} us  gap {g:6.1f} us  after {nit:3d} spmv launches  {n0} -> {n1}")
PY
python3 - <<'PY'
# timeline of everything that is not one of the four iteration kernels, all queues, of the same evaluation
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/sb/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2].startswith('k_sigma')]
ev = len(starts) - 12
a, b = starts[ev], starts[ev + 1]
seg = ks[a - 6:b + 1]
mainq = collections.Counter(k[3] for k in seg).most_common(1)[0][0]
t0 = ks[a][0]
ITER = ('k_spmv_fused', 'k_update_fused', 'k_fdm_fwd', 'k_back_post')
print("timeline (us since k_sigma): start  end  dur  queue  kernel   [iteration kernels: first and last of each run only]")
run = []
def flush():
    global run
    if run:
        print(f"   {(run[0][0]-t0)/1e3:8.1f} {(run[-1][1]-t0)/1e3:8.1f}   ...   main   {len(run)} iteration kernels")
        run = []
for s, e, nm, q in seg:
    if nm in ITER: run.append((s, e)); continue
    if q == mainq: flush()
    print(f"   {(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} {(e-s)/1e3:6.1f}   {'main' if q == mainq else 'q'+q:5s}  {nm}")
flush()
PY

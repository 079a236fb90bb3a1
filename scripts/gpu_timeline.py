#!/usr/bin/env python3
"""All queues' kernels of one evaluation from a scripts/gpu_gaps.sh trace (/tmp/gaps): start, end, queue, name.
usage (on the GPU box, after gpu_gaps.sh): python3 scripts/gpu_timeline.py [evaluation index, default 13] [from us] [to us]"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob('/tmp/gaps/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2] == 'k_sigma']
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 13
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -200.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
t0 = ks[starts[ev]][0]
for s, e, n, q in ks:
    t = (s - t0) / 1e3
    if lo <= t < hi:
        print(f"{t:8.1f} .. {(e - t0) / 1e3:8.1f} us  q{q}  {n}")

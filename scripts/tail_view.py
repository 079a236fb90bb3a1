import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/gaps/*/*kernel_trace.csv')[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0], r['Queue_Id']) for r in rows)
starts = [i for i, k in enumerate(ks) if k[2] == 'k_sigma']
a, b = starts[12], starts[13]
t0 = ks[a][0]
tend = ks[b][0]
for s, e, n, q in ks[a:b+8]:
    t = (s - t0) / 1e3
    if t > (tend - t0)/1e3 - 260: print(f"{t:8.1f} .. {(e - t0) / 1e3:8.1f} us  q{q}  {n}")

#!/bin/bash
# rocprofv3 kernel stats of a short bench run: gpu_kstat_quick.sh <cfg> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${2:-kstat}
mkdir -p $O
HMCMT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 24 --warmup 8 --config ${1:-cfg5} --no-cpu-baseline --no-extras > $O/stats.log 2>&1
f=$(ls -t $O/stats/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
print("total kernel ms", tot/1e6)
PY
tail -2 $O/stats.log | cut -c1-300

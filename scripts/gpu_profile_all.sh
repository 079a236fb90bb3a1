#!/bin/bash
# bench line + rocprofv3 kernel stats + PMC passes of the same command (run on the GPU box through gpurun)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${1:-prof}
CFG=${2:-cfg3}
mkdir -p $O
BENCH="$R/bench.py --steps 48 --warmup 8 --config $CFG --no-cpu-baseline --no-extras"
(cd $R && python3 bench.py --config $CFG > $O/bench.json 2> $O/bench.err)
HMCMT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $BENCH > $O/stats.log 2>&1
HMCMT_BENCH_NOPROF=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $BENCH > $O/pmc_fetch.log 2>&1
HMCMT_BENCH_NOPROF=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $BENCH > $O/pmc_write.log 2>&1
cd $R && python3 scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $CFG > $O/pmc_summary.txt 2>&1
# keep the merge small: drop the per-dispatch traces of the PMC passes, keep the stats csvs
find $O/pmc_fetch $O/pmc_write -name '*kernel_trace.csv' -delete
ls -la $O $O/stats/* | head -40
tail -3 $O/bench.err; cat $O/bench.json | cut -c1-600

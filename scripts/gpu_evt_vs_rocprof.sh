#!/bin/bash
# HIP-event averages of bench.py against rocprofv3 --stats averages of the SAME command (96 steps, no extras)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3; mkdir -p $O
bash $R/scripts/gpu_spin_calib.sh
cd $R && python3 bench.py --steps 96 --no-cpu-baseline --no-extras > $O/evt.json 2> $O/evt.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/er
HMCMT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/er -- python3 $R/bench.py --steps 96 --no-cpu-baseline --no-extras > $O/evt_rocprof.json 2> $O/evt_rocprof.err
cp /tmp/er/*/*kernel_stats.csv $O/evt_kernel_stats.csv
cd $R && python3 - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/r3/evt.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1))
for r in [d["roofline"]] + d["roofline_other"]:
    print("  events  %-20s avg %6.2f us  n %d  overhead subtracted %.2f" % (r["kernel"][:20], r["avg_launch_us"], r["launches_timed"], r["event_bracket_overhead_us_subtracted"]))
for r in list(csv.DictReader(open("gpurun_out/r3/evt_kernel_stats.csv")))[:6]:
    print("  rocprof %-40s calls %6d avg %6.2f us" % (r["Name"].replace("(anonymous namespace)::", "")[:40], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY

#!/bin/bash
# round 6, final build: bench + rocprofv3 stats + PMC passes (cfg3, cfg5), three driver-style runs, the untraced step timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r6g; O=gpurun_out/r6g
bash scripts/gpu_profile_all.sh r6g/prof_cfg3 cfg3 > $O/prof_cfg3.log 2>&1; tail -1 $O/prof_cfg3.log | cut -c1-200
bash scripts/gpu_profile_all.sh r6g/prof_cfg5 cfg5 > $O/prof_cfg5.log 2>&1; tail -1 $O/prof_cfg5.log | cut -c1-200
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/driver_style_$i.json; python -c "import json; d=json.load(open('$O/driver_style_$i.json')); print('driver-style', d['value'], 'near', d['near_true_state']['steps_per_s'], 'two', d['two_chains_per_gpu']['steps_per_s_aggregate'], d['two_chains_per_gpu']['ms_per_step_by_chain'], 'median traj ms', d['median_ms_per_step_by_trajectory'], 'cpu', d['cpu_baseline']['value'])"; done | tee $O/driver_style_runs.log
for st in rough true; do timeout 300 python scripts/gpu_ticks_chain.py $st 6 > $O/ticks_$st.log 2>&1; done; tail -17 $O/ticks_rough.log | head -15

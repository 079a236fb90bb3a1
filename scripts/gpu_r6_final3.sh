#!/bin/bash
# round 6, last build (k_bc_blocked, writers-only k_coef_all, k_sens_fused, the "grid resident" signal): GPU suite + smoke, bench + rocprofv3 stats +
# PMC passes (cfg3, cfg5), three driver-style runs, the untraced step timelines, the shape grid on the production path, the soak
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r6h; O=gpurun_out/r6h
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/tests.log; tail -2 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash scripts/gpu_profile_all.sh r6h/prof_cfg3 cfg3 > $O/prof_cfg3.log 2>&1; tail -1 $O/prof_cfg3.log | cut -c1-200
bash scripts/gpu_profile_all.sh r6h/prof_cfg5 cfg5 > $O/prof_cfg5.log 2>&1; tail -1 $O/prof_cfg5.log | cut -c1-200
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/driver_style_$i.json; python -c "import json; d=json.load(open('$O/driver_style_$i.json')); print('driver-style', d['value'], 'near', d['near_true_state']['steps_per_s'], 'two', d['two_chains_per_gpu']['steps_per_s_aggregate'], d['two_chains_per_gpu']['ms_per_step_by_chain'], 'median traj ms', d['median_ms_per_step_by_trajectory'], 'cpu', d['cpu_baseline']['value'])"; done | tee $O/driver_style_runs.log
for st in rough true; do timeout 300 python scripts/gpu_ticks_chain.py $st 6 > $O/ticks_$st.log 2>&1; done; tail -17 $O/ticks_rough.log | head -15
timeout 300 python scripts/gpu_ticks_chain.py rough 4 cfg5 > $O/ticks_rough_cfg5.log 2>&1; tail -17 $O/ticks_rough_cfg5.log | head -15
timeout 900 python -m scripts.gpu_shape_fuzz 20 guard > $O/shape_fuzz_guard.log 2>&1; tail -1 $O/shape_fuzz_guard.log
timeout 1500 python -m scripts.gpu_persist_soak 3000 3 > $O/soak_full.log 2>&1; grep -c "failed 0" $O/soak_full.log

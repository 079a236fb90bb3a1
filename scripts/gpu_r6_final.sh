#!/bin/bash
# round 6: the round's measurements in one call -- GPU suite, bench + rocprofv3 stats + PMC passes (cfg3, cfg5), driver-style runs, SQ counters of both
# persistent kernels, the untraced step timeline, phase stamps, strips A/B, soak of the four-strip kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r6f; O=gpurun_out/r6f
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/tests.log; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash scripts/gpu_profile_all.sh r6f/prof_cfg3 cfg3 > $O/prof_cfg3.log 2>&1; tail -2 $O/prof_cfg3.log | cut -c1-300
bash scripts/gpu_profile_all.sh r6f/prof_cfg5 cfg5 > $O/prof_cfg5.log 2>&1; tail -2 $O/prof_cfg5.log | cut -c1-300
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/driver_style_$i.json; python -c "import json; d=json.load(open('$O/driver_style_$i.json')); print('driver-style', d['value'], 'near', d['near_true_state']['steps_per_s'], 'two', d['two_chains_per_gpu']['steps_per_s_aggregate'], d['two_chains_per_gpu']['ms_per_step_by_chain'], 'median traj ms', d['median_ms_per_step_by_trajectory'])"; done | tee $O/driver_style_runs.log
bash scripts/gpu_persist_pmc.sh > $O/sq_strips2.log 2>&1; HMCMT_PERSIST_STRIPS=4 bash scripts/gpu_persist_pmc.sh > $O/sq_strips4.log 2>&1; tail -4 $O/sq_strips2.log
for st in rough true; do timeout 300 python scripts/gpu_ticks_chain.py $st 6 > $O/ticks_$st.log 2>&1; done; tail -20 $O/ticks_rough.log
for stp in 2 4; do for sw in 1 2; do echo "strips $stp sweeps $sw"; HMCMT_PERSIST_STRIPS=$stp timeout 200 python -m scripts.gpu_persist_stamps cfg3 $sw 2>&1 | grep -A2 "HMCMT_STAMPS persist"; done; done > $O/stamps_cfg3.log 2>&1
for sw in 1 2; do echo "cfg5 sweeps $sw"; timeout 300 python -m scripts.gpu_persist_stamps cfg5 $sw 2>&1 | grep -A2 "HMCMT_STAMPS persist"; done > $O/stamps_cfg5.log 2>&1
for st in 4 2 4 2 4 2; do HMCMT_PERSIST_STRIPS=$st HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strips $st headline', d['value'], d['check']['true_res_max_at_last_model'])"; done > $O/strips_ab.log 2>&1; cat $O/strips_ab.log
HMCMT_PERSIST_STRIPS=4 timeout 900 python -m scripts.gpu_persist_soak 1000 3 > $O/soak_strips4.log 2>&1; grep -c "failed 0" $O/soak_strips4.log
timeout 1500 python -m scripts.gpu_persist_soak 3000 3 > $O/soak_full.log 2>&1; grep -c "failed 0" $O/soak_full.log

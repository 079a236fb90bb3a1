#!/bin/bash
# round 4: the persistent solve kernel against the launch-per-phase loop on bench.py's headline chain (+ its phase stamps)
mkdir -p gpurun_out
for p in 0 1; do
  HMCMT_PERSIST=$p HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 48 --warmup 16 --no-cpu-baseline ${EXTRA:---no-extras} > gpurun_out/ab_p$p.json 2> gpurun_out/ab_p$p.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_p$p.json").read().strip().splitlines()[-1])
    print("persist=$p", d["value"], "steps/s", d["ms_per_step"], "ms/step", {k: d.get(k) for k in ("near_true_state", "straight_line", "cold_start") if isinstance(d.get(k), dict) and 0} )
    for k in ("near_true_state", "straight_line", "cold_start", "two_chains_per_gpu"):
        if isinstance(d.get(k), dict): print("   ", k, {kk: vv for kk, vv in d[k].items() if "steps_per_s" in kk or kk == "ms_per_step"})
    print("    check", d.get("check"))
except Exception as e:
    print("persist=$p failed", e); print(open("gpurun_out/ab_p$p.err").read()[-2000:])
PY
done
HMCMT_PERSIST=1 HMCMT_STAMPS=persist HMCMT_BENCH_NOPROF=1 timeout 300 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras > gpurun_out/ab_st.json 2> gpurun_out/ab_st.err
grep "HMCMT_STAMPS" gpurun_out/ab_st.err | tail -3

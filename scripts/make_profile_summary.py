#!/usr/bin/env python3
"""Copy the judged artefacts of a scripts/gpu_profile_all.sh run into profiles/ and write the round summary.

    python scripts/make_profile_summary.py gpurun_out/<run dir> [round tag, default r02] [config, default cfg3]
"""
import csv
import glob
import json
import os
import shutil
import sys

O = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
stats_csv = os.path.join(P, f"{tag}_bench_{cfg}_kernel_stats.csv")
shutil.copy(max(glob.glob(os.path.join(O, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime), stats_csv)   # (the newest, if the run dir was reused)
shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, f"{tag}_bench_{cfg}.json"))
# PMC bytes: merge this config's entry into the tracked file bench.py reads
run_pm = json.load(open(os.path.join(O, "pmc_traffic.json")))
tracked = os.path.join(P, "pmc_traffic.json")
allc = json.load(open(tracked)) if os.path.exists(tracked) else {}
if "per_launch_bytes" in allc:
    allc = {"cfg3": allc}
allc[cfg] = run_pm[cfg]
json.dump(allc, open(tracked, "w"), indent=1)
rows = list(csv.DictReader(open(stats_csv)))
# launch-weighted rocprofv3 average duration per bench.py category (the cross-check of bench.py's HIP-event averages)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import CATS  # noqa: E402
rp = {}
for c, names in CATS.items():
    tns = sum(float(r["TotalDurationNs"]) for r in rows if any(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").startswith(nm) for nm in names))
    ncl = sum(int(r["Calls"]) for r in rows if any(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").startswith(nm) for nm in names))
    rp[c] = tns / ncl / 1e3 if ncl else None
allc[cfg]["rocprof_avg_us"] = rp
allc[cfg]["rocprof_source"] = f"rocprofv3 --kernel-trace --stats of bench.py --steps 48 --warmup 8 --config {cfg} (profiles/{tag}_bench_{cfg}_kernel_stats.csv), launch-weighted over the kernels of a category"
json.dump(allc, open(tracked, "w"), indent=1)
d = json.load(open(os.path.join(P, f"{tag}_bench_{cfg}.json")))
pm = allc[cfg]
out = [f"# Round {tag} — rocprofv3 summary of `bench.py --config {cfg}` (1 chain, 1 MI355X)\n",
       "Commands (on the GPU box, `scripts/gpu_profile_all.sh`; this file: `scripts/make_profile_summary.py`):\n",
       f"```\npython3 bench.py --config {cfg}                     -> {tag}_bench_{cfg}.json (HIP-event sampling on, CPU leg on, extras on)\n"
       f"HMCMT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 48 --warmup 8 --config {cfg} --no-cpu-baseline --no-extras\n"
       f"                                                    -> {tag}_bench_{cfg}_kernel_stats.csv\n"
       "HMCMT_BENCH_NOPROF=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace ... (same command)  \\\n"
       "HMCMT_BENCH_NOPROF=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace ... (same command)  /  -> scripts/pmc_summary.py -> pmc_traffic.json\n"
       f"                                                                                      (entry \"{cfg}\", read by bench.py)\n```\n"]
cb = d.get("cpu_baseline") or {}
nt = d.get("near_true_state") or {}
sl = d.get("straight_line") or {}
cs = d.get("cold_start") or {}
ch = d.get("chain") or {}
out.append(f"Bench line: **{d['value']:.1f} leapfrog steps/s** ({d['ms_per_step']:.3f} ms/step) on real trajectories of the chain started at "
           f"the rough state (accepted {ch.get('accepted')} / rejected {ch.get('rejected')} in the timed region, iterations of the last step of a "
           f"trajectory {ch.get('iters_fwd_max_last_step_mean', float('nan')):.0f} forward / {ch.get('iters_adj_max_last_step_mean', float('nan')):.0f} adjoint); "
           f"chain started at the true model {nt.get('steps_per_s', float('nan')):.0f} steps/s "
           f"({nt.get('iters_fwd_max_last_step_mean', float('nan')):.0f}/{nt.get('iters_adj_max_last_step_mean', float('nan')):.0f} iterations); idealised straight-line "
           f"trajectories (round 1's headline) {sl.get('steps_per_s', float('nan')):.0f}; cold starts {cs.get('steps_per_s', float('nan')):.0f}; "
           f"CPU baseline (oracle, {cb.get('cores')} cores) {cb.get('value', float('nan')):.3f} steps/s.\n")
ri, rs = d["roofline_iteration"], d["roofline_step"]
out.append(f"Roofline (HIP events, every launch of every n-th evaluation of the timed region (n: `sampled_every` in the JSON); numerators scaled by the device-counted number "
           f"of active systems): dominant kernel `{d['roofline']['kernel'].split(' ')[0]}` {d['roofline']['achieved']:.0f} GB/s algorithmic = "
           f"**{d['roofline']['frac']:.3f}** of the 8 TB/s HBM peak at {d['roofline']['active_systems_per_launch']:.1f} active systems per " +
           ("iteration (one launch = one whole solve, " + f"{d['roofline'].get('avg_launch_us', 0):.0f} us); one COCG iteration inside it" if 'us_per_iteration' in d['roofline'] else "launch; one COCG iteration") +
           f" ({ri['kernels']} launch{'es' if ri['kernels'] != 1 else ''}) {ri['us']:.1f} us = {ri['frac']:.3f}; whole step (iteration bytes / wall time) {rs['frac']:.3f}.\n")
nev = None
for r in rows:
    if "k_sigma" in r["Name"]:
        nev = int(r["Calls"])
nev = nev or 1
out.append(f"The rocprofv3 run covers {nev} evaluations (chain set-up, 8 warm-up steps, 48 timed steps, the verify evaluation).\n")
out.append("| kernel | calls/eval | avg us, all launches (rocprofv3 --stats) | us/eval | % | HIP-event avg us in bench.py (same population: all launches, sampled evaluations of the timed region) | active systems / launch | PMC bytes/launch (MB) | algorithmic bytes/launch (MB) |")
out.append("|---|---|---|---|---|---|---|---|---|")
ev = {}
for r in [d["roofline"]] + d["roofline_other"]:
    ev[r["kernel"].split(" ")[0].split("<")[0]] = r
tot = sum(float(r["TotalDurationNs"]) for r in rows) / nev / 1e3
for r in rows:
    t = float(r["TotalDurationNs"]) / nev / 1e3
    if t < 3:
        continue
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    e = ev.get(nm.split("<")[0])
    pk = pm["kernels"].get(nm)
    cells = [f"`{nm}`", f"{int(r['Calls']) / nev:.1f}", f"{float(r['AverageNs']) / 1e3:.1f}", f"{t:.1f}", f"{100 * t / tot:.1f}",
             f"{e['avg_launch_us']:.1f}" if e else "", f"{e['active_systems_per_launch']:.1f}" if e else "",
             f"{pk['bytes_per_launch'] / 1e6:.1f}" if pk else "", f"{e['bytes_per_launch'] / 1e6:.1f}" if e else ""]
    out.append("| " + " | ".join(cells) + " |")
out.append(f"\nSum of kernel time per evaluation: {tot:.0f} us (k_sens_profile, k_bcsens_pre, k_pivot and the extrapolation kernels run on the side "
           "streams beside the solves).\n")
out.append("PMC bytes are `(2*FETCH_SIZE + WRITE_SIZE)*1024` averaged over ALL launches of the run (late iterations with most systems "
           "converged included). FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads; 2- and 8-byte-per-lane "
           "accesses are uncalibrated, so treat the complex64 / bf16 rows as +-2x on the read side. At cfg3 the ~230 MB working set sits in "
           "the 256 MB Infinity Cache; the stencil kernels' load phase runs at ITS bandwidth (profiles/r03_launch_shapes.md).\n")
open(os.path.join(P, f"{tag}_bench_{cfg}_summary.md"), "w").write("\n".join(out))
print("\n".join(out)[:4000])

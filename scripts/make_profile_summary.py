#!/usr/bin/env python3
"""Copy the judged artefacts of a scripts/gpu_profile_all.sh run into profiles/ and write the round summary.

    python scripts/make_profile_summary.py gpurun_out/<run dir> [round tag, default r01]
"""
import csv
import glob
import json
import os
import shutil
import sys

O = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
shutil.copy(glob.glob(os.path.join(O, "stats", "*", "*kernel_stats.csv"))[0], os.path.join(P, f"{tag}_bench_cfg3_kernel_stats.csv"))
shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, f"{tag}_bench_cfg3.json"))
shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(P, f"{tag}_pmc_traffic.json"))
shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_bench_cfg3_kernel_stats.csv"))))
d = json.load(open(os.path.join(P, f"{tag}_bench_cfg3.json")))
pm = json.load(open(os.path.join(P, "pmc_traffic.json")))
ne = 56
out = [f"# Round {tag} — rocprofv3 summary of `bench.py` at cfg3 (200x100 cells, 16 freq, 1 chain, 1 MI355X)\n",
       "Commands (on the GPU box, `scripts/gpu_profile_all.sh`; this file: `scripts/make_profile_summary.py`):\n",
       "```\npython3 bench.py                                    -> %s_bench_cfg3.json (HIP-event sampling on, CPU leg on)\n"
       "HMCMT_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-sampler\n"
       "                                                    -> %s_bench_cfg3_kernel_stats.csv\n"
       "HMCMT_BENCH_NOPROF=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace ... (same command)  \\\n"
       "HMCMT_BENCH_NOPROF=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace ... (same command)  /  -> scripts/pmc_summary.py -> %s_pmc_traffic.json\n"
       "                                                                                      (= pmc_traffic.json, read by bench.py)\n```\n" % (tag, tag, tag)]
cb = d.get("cpu_baseline") or {}
ss = d.get("structured_state") or {}
out.append(f"Bench line: **{d['value']:.1f} steps/s** ({d['ms_per_step']:.3f} ms/step), iterations fwd/adj max "
           f"{d['config']['iters_fwd_max']}/{d['config']['iters_adj_max']}; samples/s (reference cost structure) "
           f"{(d.get('samples') or {}).get('samples_per_s', float('nan')):.1f}; structured-model state "
           f"{ss.get('steps_per_s', float('nan')):.0f} steps/s (iterations {ss.get('iters_fwd_max')}/{ss.get('iters_adj_max')}); "
           f"CPU baseline (oracle, {cb.get('cores')} cores) {cb.get('value', float('nan')):.3f} steps/s.\n")
out.append("The rocprofv3 run covers 56 evaluations (8 warm-up + 48 timed; the first warm-up evaluations are cold starts "
           "with more iterations, so calls/eval is above the steady state).\n")
out.append("`avg µs, working launches`: the convergence polls wait on an event behind `k_spmv_fused` with the rest of the iteration "
           "already queued, so every solve ends with three launches that find all systems inactive and exit at once (~3 µs), and "
           "late iterations run with part of the systems converged; the column averages the launches of the kernel trace that last "
           "longer than 40 % of the kernel's median — the population bench.py's HIP events sample (it drops the launches the host "
           "knows to be empty).\n")
out.append("| kernel | calls/eval | avg µs, all launches (rocprofv3 --stats) | avg µs, working launches (kernel trace) | µs/eval | % | HIP-event avg µs in bench.py | PMC bytes/launch (MB) | algorithmic bytes/launch (MB) |")
out.append("|---|---|---|---|---|---|---|---|---|")
import statistics
durs = {}
tr = glob.glob(os.path.join(O, "stats", "*", "*kernel_trace.csv"))
if tr:
    for r in csv.DictReader(open(tr[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        durs.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def working_avg(n):
    v = durs.get(n)
    if not v:
        return ""
    med = statistics.median(v)
    w = [x for x in v if x > 0.4 * med]
    return f"{sum(w) / len(w):.1f}"
ev = {}
for r in [d["roofline"]] + d["roofline_other"]:
    ev[r["kernel"].split(" ")[0].split("<")[0]] = (r["avg_launch_us"], r["bytes_per_launch"])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / ne / 1e3
for r in rows:
    t = float(r["TotalDurationNs"]) / ne / 1e3
    if t < 3:
        continue
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    e = ev.get(nm.split("<")[0])
    pk = pm["kernels"].get(nm)
    cells = [f"`{nm}`", f"{int(r['Calls']) / ne:.1f}", f"{float(r['AverageNs']) / 1e3:.1f}", working_avg(nm), f"{t:.1f}", f"{100 * t / tot:.1f}",
             f"{e[0]:.1f}" if e else "", f"{pk['bytes_per_launch'] / 1e6:.1f}" if pk else "", f"{e[1] / 1e6:.1f}" if e else ""]
    out.append("| " + " | ".join(cells) + " |")
out.append(f"\nSum of kernel time per evaluation: {tot:.0f} µs (k_sens_profile and the extrapolation kernels run on the side stream "
           "under the forward solve).\n")
out.append("PMC bytes are `(2*FETCH_SIZE + WRITE_SIZE)*1024` averaged over ALL launches of the run, including the late iterations in "
           "which most systems have converged and their workgroups exit at once — hence below the algorithmic bytes of a full launch. "
           "FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads; 2- and 8-byte-per-lane accesses are "
           "uncalibrated, so treat the complex64 / bf16 rows as ±2x on the read side. Only `k_fdm_fwd` moves more than its algorithmic "
           "bytes (each of a system's 7 slab workgroups reads that system's rows; they share one XCD's L2). The launches are "
           "latency-bound: the ~160 MB working set of one solve sits in the 256 MB Infinity Cache.\n")
open(os.path.join(P, f"{tag}_bench_cfg3_summary.md"), "w").write("\n".join(out))
print("\n".join(out)[:3600])

#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes of bench.py into profiles/pmc_traffic.json.

    python scripts/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [out.json] [config]

The output file is keyed by bench.py's --config (cfg3, cfg5, ...): an existing file keeps its other configs.

Each pass is `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`
(separate runs: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  Units and corrections follow
/opt/skills/guides/MI355X_MICROARCH.md 'HBM': both counters are in KiB; FETCH_SIZE tallies 128-B requests at
64 B for wide coalesced reads, so it is doubled.  Bytes are averaged per launch of each kernel and then per
bench.py category (launch-weighted), the granularity of bench.py's HIP-event timing.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CATS = {"fdm_transform": ("k_back_post", "k_transform_lp"), "tridiagonal": ("k_fdm_fwd", "k_thomas32"),
        "spmv": ("k_spmv_fused",), "post_smoother": ("k_post",), "vector_ops": ("k_update_fused",),
        "persist": ("k_cocg_persist",)}          # round 4: the whole solve in one launch (kernels_persist.h)


def per_kernel(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                name = name.split("(")[0].strip()
                tot[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        kernels[k] = {"fetch_kib_raw": f, "write_kib": w, "launches": max(nf, nw),
                      "bytes_per_launch": (2.0 * f + w) * 1024.0}
    cats = {}
    for c, names in CATS.items():
        b = n = 0.0
        for k, v in kernels.items():
            if any(k.startswith(nm) for nm in names):
                b += v["bytes_per_launch"] * v["launches"]
                n += v["launches"]
        cats[c] = b / n if n else None
    config = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
    allc = {}
    if os.path.exists(out):
        try:
            allc = json.load(open(out))
            if "per_launch_bytes" in allc:          # round-1 layout (not keyed by config): it was cfg3
                allc = {"cfg3": allc}
        except ValueError:
            allc = {}
    allc[config] = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --config %s; "
                              "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, averaged over all launches of the run" % config,
                    "per_launch_bytes": cats, "kernels": {k: v for k, v in kernels.items() if k.startswith("k_")}}
    with open(out, "w") as f:
        json.dump(allc, f, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:16]:
        print(f"{k:40s} n={v['launches']:6d} fetch(raw KiB)={v['fetch_kib_raw']:10.1f} write(KiB)={v['write_kib']:10.1f} "
              f"bytes/launch={v['bytes_per_launch'] / 1e6:8.2f} MB")
    print(json.dumps(cats))


if __name__ == "__main__":
    main()

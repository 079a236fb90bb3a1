import json
d=json.loads(open("gpurun_out/r3/driver_style.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","scaling")})
print("roofline", {k:d["roofline"].get(k) for k in ("bound","achieved","peak","unit","frac","traffic","avg_launch_us","rocprofv3_avg_launch_us","evaluations_sampled","frac_canonical_csr")})
print("cpu_baseline", d["cpu_baseline"])

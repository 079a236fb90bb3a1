"""What an initialised RCCL process group does to torch's host-side waits (round 6: bench.py --gpus N > 1 read 15 % below N = 1 per GPU with the same
kernel times).  Times stream.synchronize() behind a small kernel and a 32-byte device-to-host read, before and after init_process_group.
    python scripts/probe/pg_sync_cost.py"""
import os, time, torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 16, device=dev, dtype=torch.float64)
h = torch.zeros(4, dtype=torch.float64, device=dev)

def measure(tag):
    torch.cuda.synchronize()
    for name, fn in (("kernel + stream.synchronize", lambda: (x.add_(1.0), torch.cuda.current_stream().synchronize())),
                     ("kernel + 32-byte .tolist()", lambda: (x.add_(1.0), h.tolist())),
                     ("kernel + event.query spin", None)):
        t0 = time.perf_counter()
        for _ in range(300):
            if fn is not None:
                fn()
            else:
                x.add_(1.0); e = torch.cuda.Event(); e.record()
                while not e.query():
                    pass
        print(f"{tag:12s} {name:30s} {(time.perf_counter() - t0) / 300 * 1e6:8.1f} us")

measure("no group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
measure("rccl group")
t = torch.ones(8, device=dev); dist.all_reduce(t)
measure("after a collective")
dist.destroy_process_group()
measure("destroyed")

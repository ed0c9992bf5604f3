"""Fixed cost of the exchange's RCCL calls on a one-GPU box: a process group of ONE rank under backend "nccl", each collective
timed alone (HIP events, after warm-up) at the bucket sizes of the workloads.  With one rank no byte crosses a link, so what is
left is RCCL's per-call launch / staging cost -- the floor every step of the keyframe-parallel loop pays per collective."""
import json
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29671")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


out = {}
flag = torch.zeros(1, dtype=torch.int32, device=dev)
out["all_reduce_4B_us"] = timed(lambda: dist.all_reduce(flag))
for name, n in (("c4_11MB", 14 * 200_000), ("c2_28MB", 14 * 500_000), ("c5_85MB", 71 * 300_000), ("3m_168MB", 14 * 3_000_000)):
    x = torch.zeros(n, dtype=torch.float32, device=dev)
    y = torch.zeros(n, dtype=torch.float32, device=dev)
    out[name] = {
        "bytes": 4 * n,
        "all_reduce_us": timed(lambda: dist.all_reduce(x)),
        "reduce_scatter_us": timed(lambda: dist.reduce_scatter_tensor(y, x)),
        "all_gather_us": timed(lambda: dist.all_gather_into_tensor(y, x)),
        "device_copy_us": timed(lambda: y.copy_(x)),
    }
print(json.dumps(out, indent=1))
dist.destroy_process_group()

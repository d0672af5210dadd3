"""ONE-rank RCCL process group on this GPU: does torch.distributed's all_gather_into_tensor run, and on which stream?  (measurement tool:
`bench.py --emulate-collation rccl` relies on it.)  Prints the steps it passed so that a crash is located."""
import faulthandler
import os
import sys
import time

faulthandler.enable()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print("init ...", flush=True)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("initialised; rccl", torch.cuda.nccl.version(), flush=True)
x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize()
print("gather on the current stream ok:", bool(torch.equal(out, x)), flush=True)
s = torch.cuda.Stream(device=dev)
big = torch.empty(1 << 28, device=dev)          # 1 GiB fill: ~0.4 ms of work in front of the gather on the side stream
for mode in ("side stream",):
    out.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        big.fill_(1.0)
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(out, x)
        t1 = time.perf_counter()
        ev = torch.cuda.Event()
        ev.record()
    ev.synchronize()
    print(f"gather on a {mode}: ok {bool(torch.equal(out, x))}, host time of the call {1e3 * (t1 - t0):.3f} ms", flush=True)
# timing: 200 gathers of a 6.3 MB block back to back on the side stream
blk = torch.empty(8 * 16 * 3 * 64 * 64, device=dev)
o2 = torch.empty_like(blk)
with torch.cuda.stream(s):
    for _ in range(10):
        dist.all_gather_into_tensor(o2, blk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        dist.all_gather_into_tensor(o2, blk)
    e1.record()
e1.synchronize()
print(f"one-rank gather of {blk.numel() * 4 / 1e6:.1f} MB: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us each", flush=True)
dist.destroy_process_group()
print("done", flush=True)

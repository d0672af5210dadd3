"""Experiment (round 5): does running the two HALVES of a batch as two concurrent decoder forwards on two streams (two handles, two
workspaces) beat one forward over the whole batch?  Idea: the forward is a strictly dependent chain of matrix-core-bound convs and
HBM-bound operand writers; two independent chains could fill each other's idle resource.  Prints ms per whole batch.
    python tools/experiments/half_batch_streams.py [config] [batch] [offset_kernels]"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "image2video-synthesis-using-cinns_amd")); sys.path.insert(0, REPO)
import i2v_synth as synth
from stage1_VAE.modules.decoder import Generator
import bench
torch.set_grad_enabled(False)
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "bair64"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["batch"]
dsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.decoder_state_dict(seed=7, channel_factor=cfg["nf"]).items()}
def make():
    g = Generator({"channel_factor": cfg["nf"], "z_dim": 64, "upsample_s": cfg["ups"], "upsample_t": cfg["upt"], "spectral_norm": True})
    g.load_state_dict(dsd)
    return g.cuda().eval()
g0, g1, g2 = make(), make(), make()
x0, z, _ = synth.bench_inputs(B, cfg["img"], 64)
x0, z = x0.cuda(), z.cuda()
h = B // 2
xa, za, xb, zb = x0[:h].contiguous(), z[:h].contiguous(), x0[h:].contiguous(), z[h:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def whole(n):
    for _ in range(n): o = g0(x0, z)
    return o
def halves(n):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(n):
        with torch.cuda.stream(s1): oa = g1(xa, za)
        with torch.cuda.stream(s2): ob = g2(xb, zb)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return oa, ob
ref = whole(2); oa, ob = halves(2); torch.cuda.synchronize()
assert torch.equal(ref[:h], oa) and torch.equal(ref[h:], ob)
for name, fn in (("whole batch, one forward", whole), ("two half-batch forwards on two streams", halves), ("whole batch, one forward", whole),
                 ("two half-batch forwards on two streams", halves)):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(10); torch.cuda.synchronize()
    print(f"{cfg['name']} B={B}: {name}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per batch", flush=True)

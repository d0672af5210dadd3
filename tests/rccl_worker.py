"""Worker of test_two_process_rccl_collation: rank r of 2 on GPU r, RCCL backend.  Shards a small batch, runs cINN
inverse + decoder on its shard, collates with the overlapped all-gather, and compares with the full batch computed locally."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "image2video-synthesis-using-cinns_amd")):
    sys.path.insert(0, p)
import i2v_dist  # noqa: E402
import i2v_synth as synth  # noqa: E402
from stage1_VAE.modules.decoder import Generator  # noqa: E402
from stage2_cINN.modules.flow_blocks import ConditionalFlow  # noqa: E402

rank, world = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
torch.set_grad_enabled(False)
T = lambda sd: {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}  # noqa: E731
flow = ConditionalFlow(64, 64, 512, 2, 20, conditioning_option="None")
flow.load_state_dict(T(synth.flow_state_dict(seed=7, embedding_dim=64)))
gen = Generator({"channel_factor": 8, "z_dim": 64, "upsample_s": [2, 1], "upsample_t": [2, 1], "spectral_norm": True})
gen.load_state_dict(T(synth.decoder_state_dict(seed=5, channel_factor=8)))
flow, gen = flow.to(dev).eval(), gen.to(dev).eval()
total = 6
x0, residual, embed = synth.bench_inputs(total, 64, 64)


def model(x, r, e):
    return gen(x.contiguous(), flow(r.contiguous(), e.contiguous(), reverse=True).view(x.size(0), -1))


full = model(x0.to(dev), residual.to(dev), embed.to(dev))
lo, hi = i2v_dist.shard_bounds(total, world, rank)
col = i2v_dist.OverlappedCollator(total)
for _ in range(3):   # a short stream of steps: the gather of step k overlaps step k + 1
    col.submit(model(x0[lo:hi].to(dev), residual[lo:hi].to(dev), embed[lo:hi].to(dev)))
out = col.result()
torch.cuda.synchronize()
assert out.shape == full.shape and torch.equal(out, full), (rank, float((out - full).abs().max()))
# the bench's N > 1 form: the gathers issued on the stream the cINN prefetch of the NEXT step runs on (three streams per rank)
import i2v_pipeline  # noqa: E402
pf = i2v_pipeline.LatentPrefetcher(lambda r, e: flow(r, e, reverse=True), device=dev)
col3 = i2v_dist.OverlappedCollator(total, stream=pf.stream)
xs, rs, es = x0[lo:hi].to(dev).contiguous(), residual[lo:hi].to(dev).contiguous(), embed[lo:hi].to(dev).contiguous()
tk = pf.submit(rs, es)
for k in range(4):
    z = pf.get(tk)
    if k + 1 < 4:
        tk = pf.submit(rs, es)
    col3.submit(gen(xs, z.view(xs.size(0), -1)))
out3 = col3.result()
torch.cuda.synchronize()
assert torch.equal(out3, full), (rank, "collation on the prefetch stream")
out2 = i2v_dist.synthesize_sharded(model, x0.to(dev), residual.to(dev), embed.to(dev))
assert torch.equal(out2, full)
dist.barrier()
dist.destroy_process_group()
print("ok", rank, dist.is_nccl_available())

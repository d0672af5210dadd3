#!/usr/bin/env python
"""Benchmark of the hot path: cINN inverse + stage-1 decoder, synthesized frames/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one Model.synthesize-equivalent call on synthetic inputs already resident in HBM: cINN inverse on the
rank's shard of the globally drawn residual/embedding, one decoder pass (16 frames per sample), and for N > 1 one RCCL
all-gather collating the [B/N,16,3,H,W] blocks.  Weak scaling: the per-GPU batch is fixed (BASELINE configs[1]:
BAIR 64x64, seq_len 16, batch 64 per MI355X), the global batch grows with N.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     dominant kernel = the 3x3x3 implicit-GEMM Conv3d on fp32 MFMA: achieved = algorithmic FLOPs of all its
               launches / their summed duration, measured with HIP events on the launch stream inside the timed region
  roofline_cinn  the coupling-block pass against the HBM roofline: algorithmic bytes (parameters + I/O) / pass time
  cpu_baseline the CPU oracle ("port" of the reference op sequence, spectral norm folded once) timed on the host cores
               on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "image2video-synthesis-using-cinns_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1]: BAIR 64x64, seq_len 16, batch 64 on one MI355X (full cINN stack + decoder)
    "bair64": dict(nf=64, emb=64, img=64, ups=[2, 1], upt=[2, 1], batch=64, name="BAIR 64x64x16 nf=64 E=64"),
    # BASELINE.json configs[2]: Landscape 128x128, seq_len 16, batch 32
    "land128": dict(nf=32, emb=128, img=128, ups=[2, 2], upt=[2, 1], batch=32, name="Landscape 128x128x16 nf=32 E=128"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (the split-fp16 path issues 3 MFMA FLOPs per FLOP)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="bair64", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--vid-length", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import i2v_dist
    import i2v_native
    import i2v_synth as synth
    from stage1_VAE.modules.decoder import Generator
    from stage2_cINN.modules.flow_blocks import ConditionalFlow

    cfg = CONFIGS[args.config]
    # the host driver only supports dmabuf IPC: without this RCCL's buffer exchange fails (hipIpcGetMemHandle); it is
    # exported on the GPU boxes already -- keep it for any environment this is launched from (read at HSA start-up)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched through torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.set_grad_enabled(False)

    per_gpu = args.batch or cfg["batch"]
    total = per_gpu * world
    # weights: deterministic synthetic (no checkpoints reachable), replicated on every rank
    fsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.flow_state_dict(seed=7, embedding_dim=cfg["emb"]).items()}
    dsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.decoder_state_dict(seed=7, channel_factor=cfg["nf"]).items()}
    flow = ConditionalFlow(64, cfg["emb"], 512, 2, 20, conditioning_option="None")
    flow.load_state_dict(fsd)
    gen = Generator({"channel_factor": cfg["nf"], "z_dim": 64, "upsample_s": cfg["ups"], "upsample_t": cfg["upt"],
                     "spectral_norm": True})
    gen.load_state_dict(dsd)
    flow, gen = flow.to(dev).eval(), gen.to(dev).eval()

    # inputs drawn for the GLOBAL batch (CPU generators, fixed seeds), then sliced per rank and made resident
    x0, residual, embed = synth.bench_inputs(total, cfg["img"], cfg["emb"])
    lo, hi = i2v_dist.shard_bounds(total, world, rank)
    x0_d, res_d, emb_d = x0[lo:hi].to(dev), residual[lo:hi].to(dev), embed[lo:hi].to(dev)

    def step():
        z = flow(res_d, emb_d, reverse=True).view(hi - lo, -1)
        seq = gen(x0_d, z)
        while seq.shape[1] < args.vid_length:
            seq = torch.cat((seq, gen(seq[:, -1].contiguous(), z)), dim=1)
        return i2v_dist.collate(seq, total)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    gen.native().set_profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = gen.native().get_profile()
    gen.native().set_profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    frames_per_step = out.shape[0] * out.shape[1]
    assert out.shape[0] == total

    # cINN pass latency (device-timed, median of 50 after 5 warm-ups), rank 0 only
    cinn = {}
    if rank == 0:
        for direction in ("inv", "fwd"):
            fn = (lambda: flow(res_d, emb_d, reverse=True)) if direction == "inv" else (lambda: flow(res_d, emb_d))
            for _ in range(5):
                fn()
            ts = []
            for _ in range(50):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            cinn[direction + "_us"] = float(np.median(ts))
    if rank == 0:
        nb = hi - lo
        cinn_bytes = flow.native().param_bytes + 4 * nb * (64 + cfg["emb"] + 64)
        result = {
            "metric": "synthesized frames/sec (BxT): cINN inverse + VAE decoder",
            "value": frames_per_step * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if gen.mma == 0 else "f32 (split-fp16 MFMA: fp16 hi/lo operand pairs, 3 MFMAs per product, fp32 accumulate)",
            "data": "synthetic (seeded start frames / latents / embeddings, deterministic synthetic weights)",
            "config": {"workload": f"{cfg['name']}, batch {per_gpu}/GPU, vid_length {args.vid_length}: "
                                   "20-block cINN inverse + decoder pass(es)" + (" + RCCL all-gather" if world > 1 else ""),
                       "global_batch": total, "frames_per_step": frames_per_step, "parallelism": f"batch-shard x{world}"},
            "roofline": roofline(prof, dt, gen.mma, args.config == "bair64" and per_gpu == 64 and args.vid_length == 16),
            "roofline_cinn": {
                "kernel": "cINN inverse pass (flow_pre_kernel, then the flow_hidden_kernel / flow_tail_kernel chain)",
                "bound": "hbm", "bytes_per_pass": cinn_bytes,
                "achieved": cinn_bytes / (cinn["inv_us"] * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": cinn_bytes / (cinn["inv_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS,
                "inv_latency_us": cinn["inv_us"], "fwd_latency_us": cinn["fwd_us"], "batch": nb,
                "measured_hbm_bytes_per_pass": cinn_measured_bytes(args.config == "bair64" and per_gpu == 64),
            },
        }
        if gen.mma == 1 and result["roofline"]:
            # the data-sheet peak assumes 2.4 GHz; with live operands the matrix cores sustain less (power management).
            # An MFMA-only loop of the conv kernel's shape, measured here on this box, gives the sustained rate.
            sustained = i2v_native.probe_mfma_f16(dev)
            r = result["roofline"]
            r["sustained_mfma"] = {
                "what": "MFMA-only loop (12 v_mfma_f32_32x32x16_f16 per k-step on 4 accumulators, 2 waves/SIMD, live "
                        "pseudo-random register operands, no memory traffic), measured on this GPU after the timed steps",
                "peak_live_operands": sustained, "unit": "TFLOP/s (fp16 MFMA FLOPs executed)",
                "frac_of_data_sheet_peak": sustained / PEAK_F16_MFMA_TFLOPS,
                "conv_kernel_issue_frac_of_sustained": r["mfma_issue_frac"] * PEAK_F16_MFMA_TFLOPS / sustained,
            }
        result["embedder"] = embedder_latency(cfg, x0_d)
        result["encoder"] = encoder_latency(cfg, x0_d)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, fsd, dsd)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cinn_measured_bytes(default_workload):
    """HBM bytes one cINN pass moves, from the PMC counters of the flow kernels (profiles/r01_l_pmc_hbm_traffic.json:
    FETCH_SIZE / WRITE_SIZE of every flow_* launch divided by the number of passes in that run = flow_pre_kernel launches).
    Valid for the default workload only (the activations' share grows with the batch)."""
    if not default_workload:
        return None
    try:
        with open(os.path.join(REPO, "profiles", "r01_l_pmc_hbm_traffic.json")) as f:
            k = json.load(f)["kernels"]
        flow = {n: v for n, v in k.items() if "flow_" in n}
        passes = next(v["launches"] for n, v in flow.items() if "flow_pre_kernel" in n)
        return sum(v["read_bytes"] + v["write_bytes"] for v in flow.values()) / passes
    except (OSError, KeyError, ValueError, StopIteration):
        return None


def embedder_latency(cfg, x0_d):
    """Row N1 (conditioning ResNet-50, in FRONT of the benchmarked path -- the step consumes a ready embedding):
    device time of ResnetEncoder.encode(x_0).mode() on the same start frames, reported separately."""
    import i2v_synth as synth
    from stage2_cINN.AE.modules.AE import ResnetEncoder
    enc = ResnetEncoder({"z_dim": cfg["emb"], "deterministic": False, "in_size": cfg["img"], "encoder_type": "resnet50", "norm": "in"})
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.embedder_state_dict(seed=7, z_dim=cfg["emb"]).items()})
    enc = enc.to(x0_d.device).eval()
    for _ in range(2):
        enc.encode(x0_d).mode()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        enc.encode(x0_d).mode()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return {"what": "ResNet-50 conditioning embedder (InstanceNorm variant), same batch; not part of `value`",
            "ms_per_batch": float(np.median(ts)), "batch": int(x0_d.shape[0])}


def encoder_latency(cfg, x0_d):
    """Row N3 (motion encoder, 3D ResNet-18 in front of the cINN's forward direction in `Model.transfer`): device time of
    Encoder.forward on a synthetic 16-frame clip per sample, reported separately like the embedder."""
    import i2v_synth as synth
    from stage1_VAE.modules.resnet3D import Encoder
    bair = cfg["img"] == 64
    geo = dict(channels=[64, 128, 256, 512, 512], stride_s=[1, 2, 2, 2]) if bair else \
        dict(channels=[64, 128, 128, 256, 512], stride_s=[2, 2, 2, 2])  # stage1_VAE/configs/{bair,landscape}_config.yaml
    enc = Encoder({"res_type_encoder": "resnet18", "use_max_pool": False, "z_dim": 64, "stride_t": [1, 2, 2, 2], **geo})
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.encoder3d_state_dict(seed=7, z_dim=64, **geo).items()})
    enc = enc.to(x0_d.device).eval()
    B = int(x0_d.shape[0])
    g = torch.Generator().manual_seed(97)
    clip = (2 * torch.rand(B, 3, 16, cfg["img"], cfg["img"], generator=g) - 1).to(x0_d.device)
    for _ in range(2):
        enc(clip)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        enc(clip)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return {"what": "motion encoder (3D ResNet-18, GroupNorm) on 16-frame clips, same batch; not part of `value`",
            "ms_per_batch": float(np.median(ts)), "batch": B}


def roofline(prof, dt, mma, default_workload=False):
    """Dominant kernel = the 3x3x3 Conv3d implicit GEMM.  achieved = ALGORITHMIC FLOPs (2*M*N*K per launch, summed) /
    summed launch duration (HIP events on the launch stream, inside the timed region).  In split-fp16 mode every
    algorithmic FLOP costs three fp16 MFMA FLOPs, so the fraction of the dense fp16 peak that the matrix cores are
    actually issuing is reported separately as mfma_issue_frac (3 MFMA FLOPs per executed product; conv_0 behind a x2
    temporal up-sampling executes 18 of the 27 algorithmic taps)."""
    if prof["conv3_ms"] <= 0:
        return None
    ach = prof["conv3_flops"] / (prof["conv3_ms"] * 1e-3) / 1e12
    if mma == 1:
        kernel = "conv_mfma_f16x3_kernel (3x3x3 Conv3d implicit GEMM, split-fp16: 3x v_mfma_f32_32x32x16_f16 per product)"
        peak = PEAK_F16_MFMA_TFLOPS
    else:
        kernel = "conv_mfma_f32_kernel (3x3x3 Conv3d implicit GEMM, v_mfma_f32_32x32x2_f32)"
        peak = PEAK_FP32_MFMA_TFLOPS
    # HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    # rocprofv3 --pmc passes and corrected as MI355X_MICROARCH.md prescribes; profiles/r01_l_pmc_hbm_traffic.json).
    # bench.py cannot read PMCs itself: the figure is valid for the default workload (bair64, batch 64, mma = 1) only.
    traffic = None
    if mma == 1 and default_workload:
        try:
            with open(os.path.join(REPO, "profiles", "r01_l_pmc_hbm_traffic.json")) as f:
                traffic = json.load(f)["dominant_kernel"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
    return {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": traffic, "mfma_issue_frac": prof["conv3_mfma_flops"] / (prof["conv3_ms"] * 1e-3) / 1e12 / peak,
            "launches": prof["conv3_launches"],
            "avg_launch_ms": prof["conv3_ms"] / max(prof["conv3_launches"], 1), "time_share": prof["conv3_ms"] * 1e-3 / dt}


def cpu_baseline(cfg, fsd, dsd):
    """The CPU oracle (torch-CPU port of the reference op sequence; W/sigma folded once, i.e. the "folded" variant of
    SURVEY §8d -- not inflated by the reference's per-call renormalisation) on a bounded sample of the same workload."""
    from oracle import decoder_ref, model_ref
    import i2v_synth as synth
    # one thread per physical core of one socket is what torch-CPU conv3d scales to; 256 SMT threads ran 4x slower
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    nb = 8  # 2 timed calls of batch 8 = ~10 s of CPU work on a 64-core EPYC
    x0, residual, embed = synth.bench_inputs(nb, cfg["img"], cfg["emb"])
    folded = decoder_ref.fold_spectral_norm(dsd)
    model_ref.synthesize(fsd, folded, x0[:1], residual[:1], embed[:1], 16, cfg["ups"], cfg["upt"], faithful=False)  # warm-up
    t0 = time.perf_counter()
    ncall = 2
    for _ in range(ncall):
        seq = model_ref.synthesize(fsd, folded, x0, residual, embed, 16, cfg["ups"], cfg["upt"], faithful=False)
    dt = (time.perf_counter() - t0) / ncall
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": seq.shape[0] * seq.shape[1] / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"mean of 2 warm calls (after a batch-1 warm-up), batch {nb} x 16 frames, same geometry and weights (oracle/model_ref.synthesize, folded)",
            "cpu": cpu, "seconds": dt}


if __name__ == "__main__":
    main()

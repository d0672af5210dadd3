#!/usr/bin/env python
"""Benchmark of the hot path: cINN inverse + stage-1 decoder, synthesized frames/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (plain shell: re-executes itself under torch.distributed.run,
                                                            one process per GPU, 127.0.0.1 rendezvous on a free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config dtdb128 --scaling strong]
    python bench.py --gpus 2 --dry                          (CPU / gloo: launch path, sharding, collation and the JSON line
                                                            with a stand-in step -- no kernels, `value` is meaningless)

A "step" = one Model.synthesize-equivalent call on synthetic inputs already resident in HBM: cINN inverse on the
rank's shard of the globally drawn residual/embedding, the decoder pass(es) (16 frames per sample and pass; vid_length 32 =
two dependent passes), and for N > 1 one RCCL all-gather collating the [B/N,T,3,H,W] blocks, issued on a side stream so
that it overlaps the next step (the last one is waited for inside the timed region).

Configurations (BASELINE.json `configs`): bair64 = configs[1] (the default, the metric's workload), land128 = configs[2],
dtdb128 = configs[3] (global batch 256), iper128_t32 = configs[4] (global batch 128, vid_length 32; SURVEY §8a: run on the
128x128 decoder geometry nf = 32 with E = 128).  --scaling weak (default): the config's batch PER GPU; --scaling strong: the
config's batch is the GLOBAL batch, sharded over the ranks (the fixed-global-batch jobs of configs[3], configs[4]).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     dominant kernels = the 3x3x3 Conv3d launches (Winograd / direct split-fp16 implicit GEMM): achieved =
               algorithmic FLOPs of all their launches / their summed duration, measured with HIP events on the launch
               stream inside the timed region; `per_layer` carries the same per layer
  roofline_cinn  the coupling-block pass against the HBM roofline: algorithmic bytes (parameters + I/O) / pass time
  cpu_baseline the CPU oracle (torch-CPU restatement of the reference op sequence) timed on the host cores on
               BASELINE configs[0] (B = 4): median of 3 calls after 1 warm-up, `faithful` (per-call spectral
               renormalisation, materialised SPADE maps -- what the reference does) and `folded` (sigma folded once)
  output_check finiteness, range and checksums of the last timed output.
"""
import argparse
import glob
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
    "bair64": dict(nf=64, emb=64, img=64, ups=[2, 1], upt=[2, 1], batch=64, vid=16, name="BAIR 64x64x16 nf=64 E=64"),
    # configs[2]: Landscape 128x128, seq_len 16, batch 32
    "land128": dict(nf=32, emb=128, img=128, ups=[2, 2], upt=[2, 1], batch=32, vid=16, name="Landscape 128x128x16 nf=32 E=128"),
    # configs[3]: DTDB textures 128x128, seq_len 16, batch 256 sharded over the node (use --scaling strong)
    "dtdb128": dict(nf=32, emb=128, img=128, ups=[2, 2], upt=[2, 1], batch=256, vid=16, name="DTDB 128x128x16 nf=32 E=128"),
    # configs[4]: "iPER 128x128, seq_len 32, batch 128" -- not a reference config (its iPER is 64x64); per SURVEY §8a the
    # 128x128 decoder geometry (nf = 32, upsample_s [2,2]) with E = 128, two dependent decoder passes
    "iper128_t32": dict(nf=32, emb=128, img=128, ups=[2, 2], upt=[2, 1], batch=128, vid=32,
                        name="iPER-like 128x128x32 (128x128 geometry nf=32, E=128, 2 decoder passes)"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="bair64", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's batch per GPU; strong: the config's batch is the global batch")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak) / global batch (strong); default: the config's")
    ap.add_argument("--vid-length", type=int, default=0, help="default: the config's (16, or 32 for iper128_t32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the post-timing measurements (cINN latency loop, MFMA probe, embedder / encoder latency): use "
                         "under rocprofv3 so that the kernel trace holds the timed steps only")
    ap.add_argument("--per-layer", type=str, help="write the per-layer table of the 3x3x3 conv launches (CSV) here")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1],
                    help="1 (default): the cINN pass of step k+1 runs on a side stream underneath the decoder of step k "
                         "(i2v_pipeline.LatentPrefetcher; every step still has its own pass, the first one is exposed); 0: serial")
    ap.add_argument("--dry", action="store_true",
                    help="CPU / gloo rehearsal of the launch path (self-launch, sharding, collation, JSON line); the step is a "
                         "stand-in without kernels and the line says so")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.dry:
        return dry_run(args)

    import i2v_dist
    import i2v_native
    import i2v_synth as synth
    from stage1_VAE.modules.decoder import Generator
    from stage2_cINN.modules.flow_blocks import ConditionalFlow

    cfg = CONFIGS[args.config]
    vid_length = args.vid_length or cfg["vid"]
    # the host driver only supports dmabuf IPC: without this RCCL's buffer exchange fails (hipIpcGetMemHandle); it is
    # exported on the GPU boxes already -- keep it for any environment this is launched from (read at HSA start-up)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a {world}-process job (WORLD_SIZE): the two must agree")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.set_grad_enabled(False)

    if args.scaling == "weak":
        per_gpu = args.batch or cfg["batch"]
        total = per_gpu * world
    else:
        total = args.batch or cfg["batch"]
        if total < world:
            raise SystemExit(f"--scaling strong: global batch {total} < {world} ranks")
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
    collator = i2v_dist.OverlappedCollator(total)

    import i2v_pipeline
    prefetch = i2v_pipeline.LatentPrefetcher(lambda r, e: flow(r, e, reverse=True), device=dev, enabled=bool(args.pipeline))

    last_z = {}

    def decode(z):
        z = z.view(hi - lo, -1)
        last_z["z"] = z
        seq = gen(x0_d, z)
        while seq.shape[1] < vid_length:
            seq = torch.cat((seq, gen(seq[:, -1].contiguous(), z)), dim=1)
        collator.submit(seq)   # N > 1: all-gather on a side stream, overlapping the next step; N = 1: keeps the tensor
        return seq

    def run_steps(n):
        """n steps = n cINN inverse passes + n decoder runs.  Pipelined like the batching loop of generate_samples.py: the
        pass of step k+1 is enqueued (side stream) before the decoder of step k; the first pass is exposed."""
        if n <= 0:
            return
        ticket = prefetch.submit(res_d, emb_d)
        for k in range(n):
            z = prefetch.get(ticket)
            if k + 1 < n:
                ticket = prefetch.submit(res_d, emb_d)
            decode(z)

    def single_call_ms():
        """One serial call: cINN pass, then the decoder (the latency of ONE Model.synthesize, nothing overlapped)."""
        ts = []
        for _ in range(3):
            barrier()
            t = time.perf_counter()
            decode(flow(res_d, emb_d, reverse=True))
            collator.result()
            barrier()
            ts.append((time.perf_counter() - t) * 1e3)
        return float(np.median(ts))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    if args.warmup:
        collator.result()
    barrier()
    gen.native().set_profile(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    out = collator.result()    # the current stream waits for the last gather; the barrier below covers it
    barrier()
    dt = time.perf_counter() - t0
    prof = gen.native().get_profile()
    layers = gen.native().get_layer_profile()
    gen.native().set_profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)   # the ONLY all-reduce: the timing scalar
        dt = float(tmax.item())
    frames_per_step = out.shape[0] * out.shape[1]
    assert out.shape[0] == total, (out.shape, total)
    # the timed output must be a valid result, not just a fast one
    finite = bool(torch.isfinite(out).all())
    amax = float(out.abs().max())
    if not finite or amax > 1.0:
        raise SystemExit(f"bench.py: invalid output of the timed steps (finite={finite}, max|y|={amax}): tanh frames must lie in [-1, 1]")
    # the latent the last timed step decoded (computed on the side stream, underneath the previous decoder) must be the one a
    # serial pass gives, bit for bit
    z_serial = flow(res_d, emb_d, reverse=True).view(hi - lo, -1)
    z_ok = bool(torch.equal(z_serial, last_z["z"]))
    if not z_ok:
        raise SystemExit("bench.py: the pipelined cINN pass of the last timed step differs from a serial pass on the same inputs")
    flags = gen.native().status()
    if flags:
        raise SystemExit(f"bench.py: the decoder reported status flags {flags} (fp16 range of the split-fp16 operands exceeded)")
    od = out.double()
    output_check = {"finite": finite, "max_abs": amax, "sum": float(od.sum()), "sum_sq": float((od * od).sum()),
                    "mean_abs": float(od.abs().mean()), "shape": list(out.shape),
                    "pipelined_latent_equals_serial": z_ok}

    nb = hi - lo
    single_ms = None if args.no_extras else single_call_ms()   # (every rank: the collation inside is a collective)
    # cINN pass latency (device-timed, median of 100 after 10 warm-ups: SURVEY §8d), rank 0 only
    cinn = {}
    if rank == 0 and not args.no_extras:
        for direction in ("inv", "fwd"):
            fn = (lambda: flow(res_d, emb_d, reverse=True)) if direction == "inv" else (lambda: flow(res_d, emb_d))
            for _ in range(10):
                fn()
            ts = []
            for _ in range(100):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            cinn[direction + "_us"] = float(np.median(ts))
            cinn[direction + "_us_min"] = float(np.min(ts))
    if rank == 0:
        default_workload = args.config == "bair64" and nb == 64 and vid_length == 16
        result = {
            "metric": "synthesized frames/sec (BxT): cINN inverse + VAE decoder",
            "value": frames_per_step * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if gen.mma == 0 else "f32 (split-fp16 MFMA: fp16 hi/lo operand pairs, 3 MFMAs per product, fp32 accumulate)",
            "data": "synthetic (seeded start frames / latents / embeddings, deterministic synthetic weights)",
            "config": {"workload": f"{cfg['name']}, batch {nb}/GPU (global {total}), vid_length {vid_length}: "
                                   "20-block cINN inverse + decoder pass(es)" + (" + RCCL all-gather (overlapped)" if world > 1 else ""),
                       "global_batch": total, "per_gpu_batch": nb, "frames_per_step": frames_per_step,
                       "parallelism": f"batch-shard x{world}"},
            "pipeline": {"cinn_of_next_step_under_decoder": bool(args.pipeline),
                         "single_call_ms": single_ms,
                         "note": "value counts `steps` cINN passes + `steps` decoder runs inside the timed region; with pipelining "
                                 "the pass of step k+1 overlaps the decoder of step k (first pass exposed); single_call_ms = one "
                                 "serial call (median of 3)"},
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "rccl_version": list(torch.cuda.nccl.version()) if world > 1 else None,
            "output_check": output_check,
            "roofline": roofline(prof, dt, gen.mma, default_workload, layers, args.steps),
        }
        if cinn:
            cinn_bytes = flow.native().param_bytes + 4 * nb * (64 + cfg["emb"] + 64)
            measured, msrc = cinn_measured_bytes(default_workload)
            result["roofline_cinn"] = {
                "kernel": "cINN inverse pass (flow_pre_tile_kernel, then the flow_hid_tile_kernel / flow_tail_tile_kernel chain: "
                          "v_mfma_f32_16x16x4_f32 tiles)",
                "bound": "hbm", "bytes_per_pass": cinn_bytes,
                "achieved": cinn_bytes / (cinn["inv_us"] * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": cinn_bytes / (cinn["inv_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS,
                "inv_latency_us": cinn["inv_us"], "fwd_latency_us": cinn["fwd_us"],
                "inv_latency_us_min": cinn["inv_us_min"], "fwd_latency_us_min": cinn["fwd_us_min"],
                "latency_method": "HIP events, median of 100 passes after 10 warm-ups", "batch": nb,
                "measured_hbm_bytes_per_pass": measured, "measured_hbm_bytes_source": msrc,
            }
        if gen.mma == 1 and result["roofline"] and not args.no_extras:
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
        if not args.no_extras:
            result["embedder"] = embedder_latency(cfg, x0_d)
            result["encoder"] = encoder_latency(cfg, x0_d)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        if args.per_layer:
            write_per_layer(args.per_layer, layers, args.steps, gen.mma)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE): re-execute under torch.distributed.run, one process per
    GPU, 127.0.0.1 rendezvous on a free port; the children's stdout (rank 0's JSON line) passes straight through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Rehearsal of everything around the kernels on CPU tensors over gloo: rank/world from the launcher's environment,
    global draw + contiguous shards, a stand-in step (the shard's start frames repeated over T), collation, the max-over-
    ranks timing and rank 0's single JSON line.  No native code is touched; `value` carries no meaning."""
    import i2v_dist
    import i2v_synth as synth
    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a {world}-process job (WORLD_SIZE): the two must agree")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    total = (args.batch or 8) * (world if args.scaling == "weak" else 1)
    x0, residual, embed = synth.bench_inputs(total, cfg["img"], cfg["emb"])
    lo, hi = i2v_dist.shard_bounds(total, world, rank)
    collator = i2v_dist.OverlappedCollator(total)

    def step():
        seq = x0[lo:hi, None].expand(-1, 16, -1, -1, -1).contiguous()
        collator.submit(seq)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    out = collator.result()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ok = bool(torch.equal(out, x0[:, None].expand(-1, 16, -1, -1, -1)))
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (CPU / gloo stand-in step, no kernels): launch path only", "value": out.shape[0] * 16 * args.steps / dt,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
                          "vs_baseline": None, "dtype": "none", "data": "synthetic", "dry": True,
                          "config": {"workload": "dry run", "global_batch": total, "per_gpu_batch": hi - lo},
                          "ranks_seen": dist.get_world_size() if world > 1 else 1, "collation_ok": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def _latest_traffic_file():
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic.json")))
    return files[-1] if files else None


def cinn_measured_bytes(default_workload):
    """HBM bytes one cINN pass moves, from the PMC counters of the flow kernels (tools/pmc_hbm_traffic.py: FETCH_SIZE /
    WRITE_SIZE of every flow_* launch divided by the number of passes in that run = flow_pre_kernel launches).  bench.py
    cannot read PMCs itself: this is a STATIC figure from the newest committed summary (returned with its file name),
    valid for the default workload only (the activations' share grows with the batch)."""
    if not default_workload:
        return None, None
    path = _latest_traffic_file()
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]
        flow = {n: v for n, v in k.items() if "flow_" in n}
        passes = next(v["launches"] for n, v in flow.items() if "flow_pre" in n)
        return sum(v["read_bytes"] + v["write_bytes"] for v in flow.values()) / passes, "static: profiles/" + os.path.basename(path)
    except (OSError, KeyError, ValueError, StopIteration, TypeError):
        return None, None


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
    B = min(int(x0_d.shape[0]), 64)
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
    return {"what": "motion encoder (3D ResNet-18, GroupNorm) on 16-frame clips; not part of `value`",
            "ms_per_batch": float(np.median(ts)), "batch": B}


def roofline(prof, dt, mma, default_workload=False, layers=None, steps=1):
    """Dominant kernels = the 3x3x3 Conv3d launches.  achieved = ALGORITHMIC FLOPs (2*M*N*K of the reference's conv per
    launch, summed) / summed launch duration (HIP events on the launch stream, inside the timed region).  Every
    algorithmic FLOP costs three fp16 MFMA FLOPs in split-fp16 mode; the Winograd kernel executes 2/3 of the products,
    conv_0 behind a x2 temporal up-sampling 18 of the 27 taps: the fraction of the dense fp16 peak that the matrix cores
    actually issue is reported separately as mfma_issue_frac."""
    if prof["conv3_ms"] <= 0:
        return None
    ach = prof["conv3_flops"] / (prof["conv3_ms"] * 1e-3) / 1e12
    if mma == 1:
        kernel = ("conv_wino_f16x3_kernel (3x3x3 Conv3d, Winograd F(2,3) along W on split-fp16 operands: 4 GEMMs per output pair, "
                  "3x v_mfma_f32_32x32x16_f16 per product) for g_1..g_4; conv_mfma_f16x3_kernel (direct split-fp16 implicit GEMM) "
                  "for head_0, g_0 and shapes the Winograd tiling does not cover")
        peak = PEAK_F16_MFMA_TFLOPS
    else:
        kernel = "conv_mfma_f32_kernel (3x3x3 Conv3d implicit GEMM, v_mfma_f32_32x32x2_f32)"
        peak = PEAK_FP32_MFMA_TFLOPS
    # HBM bytes per launch of the dominant kernels from the PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    # rocprofv3 --pmc passes and corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot read PMCs itself: STATIC
    # figure from the newest committed summary, valid for the default workload (bair64, batch 64, mma = 1) only.
    traffic, tsrc = None, None
    if mma == 1 and default_workload:
        path = _latest_traffic_file()
        try:
            with open(path) as f:
                traffic = json.load(f)["dominant_kernel"]["hbm_bytes_per_launch"]
            tsrc = "static: profiles/" + os.path.basename(path)
        except (OSError, KeyError, ValueError, TypeError):
            traffic = None
    r = {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
         "traffic": traffic, "traffic_source": tsrc,
         "mfma_issue_frac": prof["conv3_mfma_flops"] / (prof["conv3_ms"] * 1e-3) / 1e12 / peak,
         "launches": prof["conv3_launches"],
         "avg_launch_ms": prof["conv3_ms"] / max(prof["conv3_launches"], 1), "time_share": prof["conv3_ms"] * 1e-3 / dt}
    if layers:
        r["per_layer"] = [{"layer": L["layer"], "kernel": L["kernel"], "ms_per_launch": L["ms"] / L["launches"],
                           "launches_per_step": L["launches"] / max(steps, 1),
                           "tflops_algorithmic": L["flops"] / (L["ms"] * 1e-3) / 1e12,
                           "tflops_mfma_issued": L["mfma_flops"] / (L["ms"] * 1e-3) / 1e12} for L in layers]
    return r


def write_per_layer(path, layers, steps, mma):
    """profiles/rNN_conv16_per_layer.csv: one row per 3x3x3 conv layer from the HIP-event pairs of the timed steps."""
    with open(path, "w") as f:
        f.write("layer,kernel,launches_per_step,ms_per_launch,algorithmic_gflop_per_launch,tflops_algorithmic,tflops_mfma_issued,"
                "frac_of_peak_algorithmic\n")
        peak = PEAK_F16_MFMA_TFLOPS if mma == 1 else PEAK_FP32_MFMA_TFLOPS
        for L in layers:
            ms = L["ms"] / L["launches"]
            ta = L["flops"] / (L["ms"] * 1e-3) / 1e12
            f.write(f"{L['layer']},{L['kernel']},{L['launches'] / max(steps, 1):g},{ms:.4f},{L['flops'] / L['launches'] / 1e9:.2f},"
                    f"{ta:.1f},{L['mfma_flops'] / (L['ms'] * 1e-3) / 1e12:.1f},{ta / peak:.4f}\n")


def cpu_baseline():
    """BASELINE.md §3 / SURVEY §8d: the CPU oracle (torch-CPU restatement of the reference op sequence, pinned against the
    reference's own modules) on BASELINE configs[0] -- BAIR 64x64, seq_len 16, batch 4, one cINN inverse + decoder pass --
    on the host cores: 1 warm-up + 3 timed calls, median.  Two variants: `faithful` (per-call W/sigma renormalisation and
    materialised gamma/beta maps, what the reference executes) and `folded` (sigma folded once, so the comparison is not
    inflated by the reference's waste); `value` is the faithful figure."""
    from oracle import decoder_ref, flow_ref, model_ref
    import i2v_synth as synth
    c1 = CONFIGS["bair64"]
    # one thread per physical core of one socket is what torch-CPU conv3d scales to; 256 SMT threads ran 4x slower
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    nb = 4
    fsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.flow_state_dict(seed=7, embedding_dim=c1["emb"]).items()}
    dsd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.decoder_state_dict(seed=7, channel_factor=c1["nf"]).items()}
    x0, residual, embed = synth.bench_inputs(nb, c1["img"], c1["emb"])
    folded = decoder_ref.fold_spectral_norm(dsd)
    out = {}
    for variant, sd, faithful in (("faithful", dsd, True), ("folded", folded, False)):
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            seq = model_ref.synthesize(fsd, sd, x0, residual, embed, 16, c1["ups"], c1["upt"], faithful=faithful)
            if it:
                ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        out[variant] = {"frames_per_s": seq.shape[0] * seq.shape[1] / med, "seconds_per_call": med}
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        flow_ref.flow_reverse(fsd, residual, embed)
        if it:
            ts.append(time.perf_counter() - t0)
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": out["faithful"]["frames_per_s"], "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "BASELINE configs[0]: BAIR 64x64x16, batch 4, cINN inverse + decoder; median of 3 calls after 1 warm-up "
                      "(oracle/model_ref.synthesize)",
            "faithful": out["faithful"], "folded": out["folded"], "cinn_inverse_ms": float(np.median(ts)) * 1e3, "cpu": cpu}


if __name__ == "__main__":
    sys.exit(main())

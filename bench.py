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
  roofline     THE dominant kernel alone (the one with the largest summed duration among the 3x3x3 Conv3d launches of the
               timed steps; on every shipped config conv_wino4_f16x3_kernel, the Winograd F(4,3) split-fp16 conv): achieved =
               algorithmic FLOPs of its launches / their summed duration, measured with HIP events on the launch stream
               inside the timed region; `per_layer` carries every 3x3x3 layer
  roofline_all_conv3  the same aggregate over ALL 3x3x3 conv launches (three different kernels)
  single_call  frames/s and ms of ONE serial Model.synthesize-equivalent call; `value` is the pipelined
               stream rate (`value_is`)
  steps_check  a device-side checksum of EVERY timed step's output against a serial reference call (bit-identical)
  exact_fp32   the same step on the exact-fp32 MFMA kernels (mma = 0), 3 steps, against the 157.3 TFLOP/s fp32 peak
  sustained    >= 10 s of back-to-back steps after the timed region (steady-state clock)
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
    ap.add_argument("--sustain", type=float, default=10.0, help="seconds of back-to-back steps after the timed region (0: skip)")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32 (mma = 0) leg")
    ap.add_argument("--live-traffic", dest="live_traffic", action="store_true", default=None,
                    help="after the timing, measure the HBM bytes per launch of the dominant kernel and per cINN pass NOW (two child runs of this "
                         "workload under `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace`, tools/pmc_hbm_traffic.py; ~1-2 min) instead of "
                         "quoting the newest committed profiles/r*_pmc_hbm_traffic.json.  DEFAULT for a full N = 1 run (no --no-extras); if "
                         "rocprofv3 is missing or the child passes exceed I2V_PMC_TIMEOUT (default 240 s) the line falls back to the static figure "
                         "and says so")
    ap.add_argument("--no-live-traffic", dest="live_traffic", action="store_false")
    ap.add_argument("--small-batch", type=int, default=8,
                    help="second first-class workload of the default line: the per-GPU share of the BAIR B = 64 job on 8 GPUs (0: skip)")
    ap.add_argument("--per-layer", type=str, help="write the per-layer table of the 3x3x3 conv launches (CSV) here")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1],
                    help="1 (default): the cINN pass of step k+1 runs on a side stream underneath the decoder of step k "
                         "(i2v_pipeline.LatentPrefetcher; every step still has its own pass, the first one is exposed); 0: serial")
    ap.add_argument("--collation-stream", default="auto", choices=["auto", "own", "prefetch"],
                    help="N > 1 (or --emulate-collation): which stream the all-gather of a step is issued on -- a stream of the collator's own "
                         "(a FOURTH stream next to main, cINN prefetch and the decoder's side stream: HIP's four hardware queues then cost the "
                         "own-side-stream form 7 % / 46 % per step at B = 64 / 8) or the cINN prefetch stream (the gather of step k and the "
                         "pass of step k + 2 serialise there with a step of slack; three streams).  auto = prefetch")
    ap.add_argument("--side-stream", default="auto", choices=["auto", "shared", "own"],
                    help="own: the decoder handle runs its side work (SPADE branches, learned shortcuts) on a stream of its own next to the "
                         "cINN prefetch stream (best on ONE GPU: 3 streams); shared: on the SAME stream as the cINN prefetch, so that a job "
                         "with a collation stream still has three side streams at most -- HIP multiplexes streams onto four hardware queues, "
                         "and with the collation stream as the FOURTH the `own` form loses 7 % at B = 64 and 46 % at B = 8 "
                         "(profiles/r06_c_stream_configurations.txt); auto (default): own for N = 1, shared for N > 1 / --emulate-collation")
    ap.add_argument("--emulate-collation", nargs="?", const="copy", default=None, choices=["copy", "rccl"],
                    help="N = 1 measurement: run the collation of an N > 1 rank on one GPU -- the same stream, events and double buffers; "
                         "copy (default): a device copy stands in for the RCCL all-gather; rccl: a ONE-rank RCCL process group is brought up "
                         "in this process and every step's all-gather goes through torch.distributed / RCCL itself (whatever streams and "
                         "events ProcessGroupNCCL adds are then part of the measurement)")
    ap.add_argument("--lean", action="store_true",
                    help="timed steps, checksums, single_call and the per-layer roofline only (no sustained / exact-fp32 / cINN / probe / "
                         "embedder legs): what the default run's `config_128` child leg uses")
    ap.add_argument("--no-config-128", action="store_true", help="skip the Landscape 128x128 child leg of the default line")
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
    # multi: this process takes every N > 1 code path.  I2V_BENCH_FORCE_MULTI=1 (measurement / test) does so with ONE rank -- a one-rank RCCL
    # group on this GPU -- so that the lines only a multi-GPU job reaches (collectives around the timing, the rank lists, the line's N > 1
    # keys) run on a one-GPU box too
    forced = world == 1 and os.environ.get("I2V_BENCH_FORCE_MULTI") == "1"
    multi = world > 1 or forced
    if forced:
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        args.emulate_collation = "rccl"
    if multi:
        # Every N runs the same three streams (main, the cINN prefetch stream, the decoder handle's side stream); for N > 1 the prefetch
        # stream also carries each step's all-gather (--collation-stream auto).  Round 5 switched the in-call side stream
        # off here and thereby projected scaling from a configuration the N > 1 job did not run.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with c_stdout_to_stderr():   # (RCCL's start-up banner goes to stderr)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # the job must really be N ranks over RCCL on N distinct GPUs -- not N replicas that never met
            if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl" or not torch.cuda.nccl.version():
                raise SystemExit(f"bench.py: expected {args.gpus} ranks over RCCL, got world {dist.get_world_size()} / backend {dist.get_backend()}")
            rank_devices = [None] * world    # (recorded in the line; RCCL itself refuses two ranks on one GPU)
            try:
                uuid = str(torch.cuda.get_device_properties(dev).uuid)
            except Exception:
                uuid = None
            dist.all_gather_object(rank_devices, (rank, torch.cuda.current_device(), uuid))
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)            # the first device collective brings the communicator up (inside the redirection)
            torch.cuda.synchronize()
    else:
        rank_devices = [(0, torch.cuda.current_device(), None)]
        if args.emulate_collation == "rccl":
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            with c_stdout_to_stderr():
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                warm = torch.zeros(1, device=dev)
                dist.all_reduce(warm)
                torch.cuda.synchronize()
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
    import i2v_pipeline
    prefetch = i2v_pipeline.LatentPrefetcher(lambda r, e: flow(r, e, reverse=True), device=dev, enabled=bool(args.pipeline))
    coll_mode = "prefetch" if args.collation_stream == "auto" else args.collation_stream
    coll_on_prefetch = coll_mode == "prefetch" and prefetch.enabled and (multi or args.emulate_collation)
    collator = i2v_dist.OverlappedCollator(total, emulate=args.emulate_collation if (not multi or forced) else None, stream=prefetch.stream if coll_on_prefetch else None)
    # the decoder's side work: its handle's own stream, unless the collation has a stream of its own too (then the fourth stream must go)
    side_mode = args.side_stream if args.side_stream != "auto" else ("shared" if ((multi or args.emulate_collation) and not coll_on_prefetch) else "own")
    shared = side_mode == "shared" and prefetch.enabled
    if shared:
        gen.share_side_stream(prefetch.stream)

    last_z = {}
    step_sums = []   # one device-side checksum per step (see checksum())

    def checksum(t):
        """Exact, order-independent checksum of a float tensor: the int64 sum of its bit patterns (one reduction pass over the
        output, ~15 us per 50 MB; it runs inside the timed region so that EVERY timed step is checked, not only the last)."""
        # (viewed as int64 PAIRS of bit patterns: one reduction kernel; an int32 view summed into int64 costs an extra cast pass over the
        #  output -- 80 us per 100 MB step in the round-5 kernel trace of iper128_t32)
        return t.view(torch.int64).sum()

    def decode(z, g=None):
        g = g or gen
        z = z.view(hi - lo, -1)
        last_z["z"] = z
        seq = g.decode_sequence(x0_d, z, vid_length)   # get_model.py:68-73, decoded in place into one [B, vid_length, 3, H, W] buffer
        step_sums.append(checksum(seq))
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
            # (shared side stream: the pass of step k + 1 is enqueued FIRST, the decoder's side work of step k behind it -- the handle
            #  runs the two tiny first SPADE levels inline then, so that the main chain does not wait for the pass: i2v_dec.hip fork_spade)
            if k + 1 < n:
                ticket = prefetch.submit(res_d, emb_d)
            decode(z)

    def single_call_ms(g=None, n=3):
        """One serial call: cINN pass, then the decoder (the latency of ONE Model.synthesize, nothing overlapped)."""
        ts = []
        for _ in range(n):
            barrier()
            t = time.perf_counter()
            if args.pipeline:   # like get_model.Model.synthesize: the cINN pass on the side stream, the SPADE branches meanwhile
                tk = prefetch.submit(res_d, emb_d)
                (g or gen).prepare(x0_d)
                decode(prefetch.get(tk), g)
            else:
                decode(flow(res_d, emb_d, reverse=True), g)
            collator.result()
            barrier()
            ts.append((time.perf_counter() - t) * 1e3)
        return float(np.median(ts))

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    if args.warmup:
        collator.result()
    barrier()
    del step_sums[:]
    gen.native().set_profile(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    out = collator.result()    # the current stream waits for the last gather; the barrier below covers it
    barrier()
    dt = time.perf_counter() - t0
    prof = gen.native().get_profile()
    layers = gen.native().get_layer_profile()
    gen.native().set_profile(False)
    timed_sums = list(step_sums)
    dt_rank = dt
    rank_ms = [dt / max(args.steps, 1) * 1e3]
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)   # the timing scalar (max over ranks)
        dt = float(tmax.item())
        every = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, torch.tensor([dt_rank], dtype=torch.float64, device=dev))   # (reporting only)
        rank_ms = [float(v) / max(args.steps, 1) * 1e3 for v in every.tolist()]
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
    if flags & 1:
        raise SystemExit(f"bench.py: the decoder reported status flags {flags} (fp16 range of the split-fp16 operands exceeded)")
    # EVERY timed step must have produced the frames a serial reference call produces, bit for bit (tanh maps garbage into
    # (-1, 1), so a range check alone would pass a wrong step): one device-side checksum per step, compared here
    ref_seq = decode(z_serial)
    collator.result()
    ref_sum = int(checksum(ref_seq).item())
    sums = [int(v.item()) for v in timed_sums]
    bad_steps = [k for k, v in enumerate(sums) if v != ref_sum]
    if len(sums) != args.steps or bad_steps or not torch.equal(ref_seq, out[lo:hi]):
        raise SystemExit(f"bench.py: timed steps {bad_steps} of {len(sums)} (expected {args.steps}) differ from a serial reference call "
                         f"on the same inputs (checksums {sums} vs {ref_sum})")
    steps_check = {"steps_checked": len(sums), "all_bit_identical_to_serial_reference": True, "checksum": ref_sum,
                   "method": "int64 sum of the float bit patterns (taken in pairs) of each timed step's [B/N,T,3,H,W] output, computed on the device "
                             "inside the timed region; compared with one serial cINN + decoder call after the timing"}
    od = out.double()
    output_check = {"finite": finite, "max_abs": amax, "sum": float(od.sum()), "sum_sq": float((od * od).sum()),
                    "mean_abs": float(od.abs().mean()), "shape": list(out.shape),
                    "pipelined_latent_equals_serial": z_ok, "status_flags": flags}
    del ref_seq, od

    nb = hi - lo
    single_ms = None if args.no_extras else single_call_ms()   # (every rank: the collation inside is a collective)
    if args.lean:
        args.sustain, args.no_exact, args.small_batch, args.no_cpu_baseline, args.no_config_128 = 0.0, True, 0, True, True
        if args.live_traffic is None:
            args.live_traffic = False
    # steady state: >= 10 s of back-to-back steps (a fresh box clocks higher for the first seconds than under sustained load)
    sustained = None
    if not args.no_extras and args.sustain > 0:
        n_c = max(int(args.sustain / 5 / max(dt / args.steps, 1e-4)) + 1, 1)   # steps per chunk: a fifth of the span
        chunks, done_s = [], 0
        barrier()
        ts0 = time.perf_counter()
        # six chunks = 1.2x the span at the timed rate: the SAME count on every rank (dt is the max over ranks), so that the
        # collectives inside stay matched; a single process may add chunks until the span is really covered
        while len(chunks) < 6 or (not multi and time.perf_counter() - ts0 < args.sustain + 0.05):
            tc = time.perf_counter()
            run_steps(n_c)
            collator.result()
            barrier()
            chunks.append((time.perf_counter() - tc) / n_c * 1e3)
            done_s += n_c
        tot_s = time.perf_counter() - ts0
        sustained = {"seconds": tot_s, "steps": done_s, "ms_per_step": tot_s / done_s * 1e3,
                     "ms_per_step_by_chunk": chunks, "frames_per_s": frames_per_step * done_s / tot_s,
                     "note": "back-to-back pipelined steps after the timed region, same code path; chunks in time order"}
        bad = [k for k, v in enumerate(step_sums[-done_s:]) if int(v.item()) != ref_sum] if done_s <= 4096 else []
        if bad:
            raise SystemExit(f"bench.py: sustained-load steps {bad[:8]} differ from the serial reference")
    # second first-class workload of the default line (before the exact-fp32 leg creates another handle with its own side stream)
    small = None
    if rank == 0 and not multi and not args.no_extras and args.small_batch > 0 and args.config == "bair64" and nb == 64 and vid_length == 16:
        small = small_batch_leg(flow, gen, x0_d, res_d, emb_d, vid_length, args.small_batch,
                                {"single_call": {"ms": single_ms}, "ms_per_step": dt / args.steps * 1e3}, prefetch)
    # the un-emulated number: the same step on the exact-fp32 MFMA kernels (mma = 0), N = 1 only
    exact = None
    if not args.no_extras and not multi and gen.mma != 0 and not args.no_exact:
        gen0 = Generator({"channel_factor": cfg["nf"], "z_dim": 64, "upsample_s": cfg["ups"], "upsample_t": cfg["upt"],
                          "spectral_norm": True, "mma": 0})
        gen0.load_state_dict(dsd)
        gen0 = gen0.to(dev).eval()
        if shared:
            gen0.share_side_stream(prefetch.stream)
        single_call_ms(gen0, 1)                        # warm-up
        gen0.native().set_profile(True)
        ms0 = single_call_ms(gen0, 3)
        p0 = gen0.native().get_profile()
        gen0.native().set_profile(False)
        ach0 = p0["conv3_flops"] / (p0["conv3_ms"] * 1e-3) / 1e12 if p0["conv3_ms"] > 0 else None
        iss0 = p0["conv3_mfma_flops"] / (p0["conv3_ms"] * 1e-3) / 1e12 if p0["conv3_ms"] > 0 else None
        exact = {"what": "the same step with every conv on the fp32 matrix cores (mma = 0: v_mfma_f32_32x32x2_f32, no fp16 value anywhere); "
                         "since round 5 the 3x3x3 convs from the 8x8 level on run Winograd F(4,3) along W (csrc/i2v_wino32.hip: half the "
                         "MFMA work; I2V_DEC_WINO32=0 = the 27-tap kernel); serial calls, median of 3 after 1 warm-up",
                 "ms_per_step": ms0, "frames_per_s": frames_per_step / (ms0 * 1e-3),
                 "conv3_tflops": ach0, "conv3_tflops_mfma_issued": iss0, "peak": PEAK_FP32_MFMA_TFLOPS,
                 "frac": None if iss0 is None else iss0 / PEAK_FP32_MFMA_TFLOPS,
                 "frac_algorithmic": None if ach0 is None else ach0 / PEAK_FP32_MFMA_TFLOPS,
                 "frac_is": "matrix-core FLOPs actually issued / fp32 MFMA peak (the 3x3x3 launches incl. Winograd's output transform); "
                            "frac_algorithmic counts the reference conv's 2*M*N*K"}
        del gen0
    # cINN pass latency (device-timed, median of 100 after 10 warm-ups: SURVEY §8d), rank 0 only
    cinn = {}
    if rank == 0 and not args.no_extras and not args.lean:
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
            "dtype": "f32" if gen.mma == 0 else "f32 (split-fp16 MFMA: fp16 hi/lo operand pairs, 3 MFMAs per product, fp32 accumulate)" +
                     ("; auto mode: per-layer fallback to exact fp32 behind the range guard, switched so far: %s" % gen.native().fallback_layers() if gen.mma == 2 else ""),
            "data": "synthetic (seeded start frames / latents / embeddings, deterministic synthetic weights)",
            "config": {"workload": f"{cfg['name']}, batch {nb}/GPU (global {total}), vid_length {vid_length}: "
                                   "20-block cINN inverse + decoder pass(es)" + (" + RCCL all-gather (overlapped)" if multi else ""),
                       "global_batch": total, "per_gpu_batch": nb, "frames_per_step": frames_per_step,
                       "parallelism": f"batch-shard x{world}"},
            "value_is": ("THE METRIC (BASELINE.json: synthesized frames/sec, whole-job throughput over `steps` steps): " +
                         ("pipelined stream rate -- `steps` cINN passes + `steps` decoder runs, the pass of step k+1 enqueued under the "
                          "decoder of step k" if args.pipeline else "serial steps") +
                         ".  `single_call` is SURVEY §8d(i)'s frames/s of ONE call (latency figure, not the metric); both are printed"),
            "single_call": None if single_ms is None else {"ms": single_ms, "frames_per_s": frames_per_step / (single_ms * 1e-3),
                                                           "what": "ONE Model.synthesize-equivalent call, median of 3: the cINN pass on a side stream while the "
                                                                   "decoder's SPADE branches (start frame only) are computed, then the rest of the decoder"},
            "steps_check": steps_check,
            "sustained": sustained,
            "exact_fp32": exact,
            "rank_ms_per_step": rank_ms,
            "rank_devices": [list(t) for t in rank_devices],
            "pipeline": {"cinn_of_next_step_under_decoder": bool(args.pipeline),
                         "single_call_ms": single_ms,
                         "note": "value counts `steps` cINN passes + `steps` decoder runs inside the timed region; with pipelining "
                                 "the pass of step k+1 overlaps the decoder of step k (first pass exposed); single_call_ms = one "
                                 "serial call (median of 3)"},
            "ranks_seen": dist.get_world_size() if multi else 1,
            "rccl_version": list(torch.cuda.nccl.version()) if multi else None,
            "output_check": output_check,
        }
        result.update(roofline(prof, dt, gen.mma, default_workload, layers, args.steps))
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
                **cinn_latency_floor(flow, nb, cinn["inv_us"]),
                "measured_hbm_bytes_per_pass": measured, "measured_hbm_bytes_source": msrc,
            }
        if gen.mma != 0 and result.get("roofline") and not args.no_extras and not multi and not args.lean:
            result["roofline"]["undisturbed"] = undisturbed_roofline(cfg, dsd, dev, x0_d, last_z["z"], vid_length, result["roofline"])
        if gen.mma != 0 and result.get("roofline") and not args.no_extras and not args.lean:
            # the data-sheet peak assumes 2.4 GHz; with live operands the matrix cores sustain less (power management).
            # An MFMA-only loop of the conv kernel's shape, measured here on this box, gives the sustained rate.
            sustained = i2v_native.probe_mfma_f16(dev)
            r = result["roofline"]
            result["roofline_all_conv3"]["sustained_mfma"] = {
                "what": "MFMA-only loop (12 v_mfma_f32_32x32x16_f16 per k-step on 4 accumulators, 2 waves/SIMD, live "
                        "pseudo-random register operands, no memory traffic), measured on this GPU after the timed steps",
                "peak_live_operands": sustained, "unit": "TFLOP/s (fp16 MFMA FLOPs executed)",
                "frac_of_data_sheet_peak": sustained / PEAK_F16_MFMA_TFLOPS,
                "dominant_kernel_issue_frac_of_sustained": r["mfma_issue_frac"] * PEAK_F16_MFMA_TFLOPS / sustained,
            }
        if not args.no_extras and not args.lean:
            result["embedder"] = embedder_latency(cfg, x0_d)
            result["encoder"] = encoder_latency(cfg, x0_d)
        if not multi and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        result["streams"] = {"side_stream": side_mode if prefetch.enabled else "none",
                             "what": ("main + ONE side stream: the cINN prefetch and the decoder handle's side work (SPADE branches, learned "
                                      "shortcuts) share it" if shared else "main + the cINN prefetch stream + the decoder handle's own side stream") +
                                     ("; + the collation stream and RCCL's" if multi else ""),
                             "dec_overlap_env": os.environ.get("I2V_DEC_OVERLAP"), "collation_stream_emulated": (args.emulate_collation or False) if not multi else False,
                             "collation_stream": ("the cINN prefetch stream" if coll_on_prefetch else "its own") if (multi or args.emulate_collation) else None,
                             "rule": "three streams at every N: main + the decoder handle's side stream + the cINN prefetch stream, which for N > 1 also "
                                     "carries the all-gathers (--collation-stream auto); `small_batch` is measured in the one-GPU and in the N > 1 form and "
                                     "projects from the second"}
        if small is not None:
            result["small_batch"] = small
        if not multi and default_workload and not args.no_extras and not args.no_config_128:
            result["config_128"] = config_128_leg(args)
        live = args.live_traffic if args.live_traffic is not None else (not args.no_extras)
        if live and not multi:
            live_traffic(result, args)
        if args.per_layer:
            write_per_layer(args.per_layer, layers, args.steps, gen.mma)
        validate_line(result, full=not multi and gen.mma != 0 and not (args.no_extras or args.no_cpu_baseline or args.no_exact or args.sustain < 10))
        line = json.dumps(result)
    if multi:
        dist.barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the ONE line, and the last thing on stdout: RCCL prints its version banner through C stdio (block-buffered on a pipe, flushed
        # at exit -- behind a line printed earlier), so the group is torn down and C stdio flushed first
        flush_c_stdio()
        print(line, flush=True)


class c_stdout_to_stderr:
    """RCCL prints a five-line version banner through C stdio when its first communicator comes up.  While the process group is brought
    up, file descriptor 1 points at stderr (and C stdio is flushed before it is restored), so that stdout carries the ONE JSON line only."""

    def __enter__(self):
        flush_c_stdio()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        flush_c_stdio()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def flush_c_stdio():
    """RCCL (and the HIP runtime) write to the C-level stdout, which is block-buffered on a pipe and would otherwise be flushed at
    exit -- BEHIND the one JSON line the caller parses (seen with a one-rank RCCL group: five banner lines after the line)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


CONTRACT_KEYS = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
                 "higher_is_better": bool, "scaling": str, "vs_baseline": type(None), "dtype": str, "data": str, "config": dict}
LINE_KEYS_N1 = ("roofline", "roofline_all_conv3", "cpu_baseline", "single_call", "steps_check", "sustained", "exact_fp32", "value_is",
                "rank_ms_per_step", "roofline_cinn", "output_check")


def validate_line(r, full=True):
    """Schema of the JSON line (the driver's contract + what the round-3 review asked the line to carry).  `full`: a default
    N = 1 run with all extras; otherwise only the contract keys and the always-present extras are required."""
    for k, t in CONTRACT_KEYS.items():
        if k not in r or not isinstance(r[k], t):
            raise ValueError(f"bench line: key {k!r} missing or not {t.__name__}: {r.get(k)!r}")
    if "workload" not in r["config"]:
        raise ValueError("bench line: config.workload missing")
    for k in ("roofline", "roofline_all_conv3", "steps_check", "value_is", "rank_ms_per_step", "output_check") + (LINE_KEYS_N1 if full else ()):
        if r.get(k) is None:
            raise ValueError(f"bench line: key {k!r} missing")
    ro = r["roofline"]
    for k in ("kernel", "kernel_name", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        if k not in ro:
            raise ValueError(f"bench line: roofline.{k} missing")
    if ro["bound"] not in ("hbm", "mfma") or abs(ro["frac"] - ro["achieved"] / ro["peak"]) > 1e-9:
        raise ValueError("bench line: roofline.bound / frac inconsistent")
    if " for g_1" in ro["kernel"] or ";" in ro["kernel_name"]:
        raise ValueError("bench line: roofline must name ONE kernel")
    sc = r["steps_check"]
    if sc["steps_checked"] != r["steps"] or sc["all_bit_identical_to_serial_reference"] is not True:
        raise ValueError("bench line: steps_check does not cover every timed step")
    if len(r["rank_ms_per_step"]) != r["n_gpus"]:
        raise ValueError("bench line: rank_ms_per_step must have one entry per rank")
    if full:
        cb = r["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            if k not in cb:
                raise ValueError(f"bench line: cpu_baseline.{k} missing")
        if r["sustained"]["seconds"] < 10.0:
            raise ValueError("bench line: sustained.seconds < 10")
        for k in ("ms_per_step", "frames_per_s", "frac"):
            if k not in r["exact_fp32"]:
                raise ValueError(f"bench line: exact_fp32.{k} missing")
    return True


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
        flush_c_stdio()
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


def undisturbed_roofline(cfg, dsd, dev, x0_d, z, vid_length, ro):
    """Since round 5 a forward runs its SPADE branches and learned shortcuts on a side stream underneath the main chain, so the HIP-event
    durations of the dominant kernel's launches in the timed region include whatever shared the chip with them (the step gets shorter,
    the individual launches longer).  This leg times the SAME launches with everything inline on one stream (a second handle created
    with I2V_DEC_OVERLAP=0, 3 profiled decoder runs after 1 warm-up): the kernel's own rate, next to `roofline.frac` of the timed region."""
    from stage1_VAE.modules.decoder import Generator
    old = os.environ.get("I2V_DEC_OVERLAP")
    os.environ["I2V_DEC_OVERLAP"] = "0"
    try:
        g = Generator({"channel_factor": cfg["nf"], "z_dim": 64, "upsample_s": cfg["ups"], "upsample_t": cfg["upt"], "spectral_norm": True})
        g.load_state_dict(dsd)
        g = g.to(dev).eval()
        g.decode_sequence(x0_d, z, vid_length)
        torch.cuda.synchronize()
        g.native().set_profile(True)
        for _ in range(3):
            g.decode_sequence(x0_d, z, vid_length)
        torch.cuda.synchronize()
        layers = g.native().get_layer_profile()
        g.native().set_profile(False)
    finally:
        if old is None:
            os.environ.pop("I2V_DEC_OVERLAP", None)
        else:
            os.environ["I2V_DEC_OVERLAP"] = old
    dom = ro["kernel_name"].replace("_kernel", "")
    ms = sum(L["ms"] for L in layers if L["kernel"] == dom)
    fl = sum(L["flops"] for L in layers if L["kernel"] == dom)
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else None
    del g
    return {"what": "the same kernel's launches with the SPADE branches and shortcuts INLINE on the launch stream (second handle, "
                    "I2V_DEC_OVERLAP=0; 3 decoder runs): nothing else on the chip while they run",
            "achieved": ach, "frac": None if ach is None else ach / ro["peak"], "ms_per_decoder_run": ms / 3}


def small_batch_leg(flow, gen, x0_d, res_d, emb_d, vid_length, nb, result, pf):
    """The per-GPU share of the default job on 8 GPUs (BASELINE north_star: >= 6x at 8 GPUs for BAIR 64x64x16): the same step at
    batch `nb` on THIS GPU -- one serial call (median of 5) and the pipelined stream rate (30 steps) -- in TWO stream configurations:
    `n1` = what a one-GPU job runs, and `multi_gpu` = what every rank of an N > 1 job runs (the same three streams, the all-gather of
    every step issued on the cINN prefetch stream -- here a device copy of the rank's block standing in for the RCCL kernel).  The strong-scaling figure is PROJECTED from the second: T(64 on one GPU) / T(nb in the N > 1 configuration),
    with an explicit all-gather term.  A projection from one GPU, labelled so; no multi-GPU claim."""
    import i2v_dist
    x, r, e = x0_d[:nb].contiguous(), res_d[:nb].contiguous(), emb_d[:nb].contiguous()
    # (pf: the run's own LatentPrefetcher -- a second high-priority stream would be one stream too many: HIP multiplexes streams
    #  onto four hardware queues, and streams that share a queue serialise)
    frames = nb * 16 * max(1, -(-vid_length // 16))
    was_shared = getattr(gen, "_shared_side", None)

    def measure(shared, collator):
        gen.share_side_stream(pf.stream if shared else None)

        def one_call():
            tk = pf.submit(r, e)
            gen.prepare(x)
            seq = gen.decode_sequence(x, pf.get(tk).view(nb, -1), vid_length)
            if collator is not None:
                collator.submit(seq)
                collator.result()
            return seq

        def stream(n):
            tk = pf.submit(r, e)
            for k in range(n):
                z = pf.get(tk)
                if k + 1 < n:
                    tk = pf.submit(r, e)
                seq = gen.decode_sequence(x, z.view(nb, -1), vid_length)
                if collator is not None:
                    collator.submit(seq)
            if collator is not None:
                collator.result()

        for _ in range(2):
            one_call()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            one_call()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        single = float(np.median(ts))
        stream(3)
        torch.cuda.synchronize()
        t = time.perf_counter()
        stream(30)
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t) / 30 * 1e3
        return {"single_call_ms": single, "single_call_frames_per_s": frames / (single * 1e-3),
                "pipelined_ms_per_step": piped, "pipelined_frames_per_s": frames / (piped * 1e-3)}

    try:
        n1 = measure(False, None)
        multi = measure(False, i2v_dist.OverlappedCollator(nb, emulate="rccl" if dist.is_initialized() else "copy", stream=pf.stream))   # the N > 1 form: gathers on the cINN stream
    finally:
        gen.share_side_stream(was_shared)
    big_single = (result.get("single_call") or {}).get("ms")
    out = {"what": f"the same step at batch {nb} = the per-GPU share of the B = 64 job on {64 // nb} GPUs, measured on this one GPU in the "
                   "stream configuration of a one-GPU job (`n1`) and in the one every rank of an N > 1 job runs (`multi_gpu`: the same streams, "
                   "every step's all-gather -- here a device copy of the block -- issued on the cINN prefetch stream)",
           "batch": nb, "n1": n1, "multi_gpu": multi,
           "projected_strong_scaling": projected_scaling(64 // nb, nb, frames // nb, big_single, multi["single_call_ms"], result["ms_per_step"],
                                                         multi["pipelined_ms_per_step"])}
    out.update({k: multi[k] for k in multi})   # (top-level figures = the N > 1 configuration, the one the projection uses)
    return out


XGMI_LINK_GBS = 153.0   # MI355X_MICROARCH.md: one xGMI link, per direction (7 links per GPU, point to point)


def projected_scaling(n, nb, frames_per_sample, big_single_ms, single_ms, big_piped_ms, piped_ms):
    """T(64) / T(64 / n) from one-GPU runs in the stream configuration EVERY N runs (--side-stream), with an explicit term for the
    one collective of the path: the all-gather of the [B/N, T, 3, 64, 64] fp32 blocks as a ring over ONE xGMI link per hop
    ((n - 1) hops of one rank's block) plus one launch.  A single call waits for it; the pipelined stream overlaps it with the next
    step (i2v_dist.OverlappedCollator), so there it is reported next to the ratio, not inside it."""
    block = nb * frames_per_sample * 3 * 64 * 64 * 4
    ag_ms = (n - 1) * block / (XGMI_LINK_GBS * 1e9) * 1e3 + 0.02
    return {"gpus": n,
            "single_call": None if not big_single_ms else big_single_ms / (single_ms + ag_ms),
            "single_call_before_collation": None if not big_single_ms else big_single_ms / single_ms,
            "pipelined": big_piped_ms / piped_ms,
            "all_gather_ms_model": ag_ms, "all_gather_bytes_per_rank": block,
            "all_gather_model": f"ring: (N - 1) x one rank's block over one {XGMI_LINK_GBS:.0f} GB/s xGMI link + 20 us launch; exposed in a "
                                "single call, overlapped with the next step in the pipelined stream (where it has to stay under the step time)",
            "note": "PROJECTION from one-GPU runs: T(B = 64, one-GPU job) / T(B = %d, measured in the stream configuration every rank of the "
                    "N > 1 job runs); not a multi-GPU measurement" % nb}


def config_128_leg(args):
    """Second workload of the default line (round-5 review: three of five BASELINE configs had never been timed by the driver):
    BASELINE configs[2], Landscape 128x128x16 nf = 32 E = 128, batch 32 -- run as a CHILD `bench.py --config land128 --lean` (its own
    process: its own handles and streams, this process idles meanwhile), 10 timed steps after 3 warm-ups, every step checksummed
    against a serial call, single call, dominant-kernel roofline and the g_4 rows of the per-layer table."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "land128", "--steps", "10", "--warmup", "3", "--lean",
           "--side-stream", args.side_stream, "--pipeline", str(args.pipeline)]
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=float(os.environ.get("I2V_CONFIG128_TIMEOUT", "240")))
        if p.returncode != 0:
            raise RuntimeError(f"rc {p.returncode}: {p.stderr[-300:]}")
        r = json.loads(p.stdout.strip().splitlines()[-1])
    except (OSError, subprocess.SubprocessError, ValueError, RuntimeError, IndexError) as e:
        return {"error": repr(e)[:400], "seconds": time.perf_counter() - t0}
    ro = r["roofline"] or {}
    return {"workload": r["config"]["workload"], "command": "bench.py " + " ".join(cmd[2:]),
            "ms_per_step": r["ms_per_step"], "frames_per_s": r["value"], "steps": r["steps"], "warmup": r["warmup"],
            "single_call": r.get("single_call"), "steps_check": r["steps_check"],
            "roofline": {k: ro.get(k) for k in ("kernel_name", "bound", "achieved", "peak", "unit", "frac", "mfma_issue_frac", "ms_per_step",
                                                "avg_launch_ms", "launches", "time_share")},
            "roofline_all_conv3_frac": (r.get("roofline_all_conv3") or {}).get("frac"),
            "per_layer_g4": [L for L in (r.get("roofline_all_conv3") or {}).get("per_layer", []) if L["layer"].startswith("g_4.")],
            "per_layer": (r.get("roofline_all_conv3") or {}).get("per_layer"),
            "output_check": {k: r["output_check"].get(k) for k in ("finite", "max_abs", "shape", "status_flags")},
            "seconds": time.perf_counter() - t0}



def live_traffic(result, args):
    """--live-traffic: run tools/pmc_hbm_traffic.py (separate FETCH_SIZE / WRITE_SIZE passes of one step of THIS workload under
    rocprofv3, the guide's gfx950 correction) in a child process group while this process idles, and replace the static figures.
    On any failure (no rocprofv3, time limit) the static figures stay, labelled, with the reason next to them."""
    import shutil
    import signal
    import subprocess
    import tempfile
    r = result.get("roofline")
    if not r:
        return
    if not shutil.which("rocprofv3"):
        r["traffic_live_error"] = "rocprofv3 not on PATH"
        return
    out = os.environ.get("I2V_PMC_OUT") or tempfile.mkdtemp(prefix="i2v_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))   # (I2V_PMC_OUT: keep the summary)
    cmd = [sys.executable, os.path.join(REPO, "tools", "pmc_hbm_traffic.py"), out, "--config", args.config, "--scaling", args.scaling]
    if args.batch:
        cmd += ["--batch", str(args.batch)]
    limit = float(os.environ.get("I2V_PMC_TIMEOUT", "240"))
    t0 = time.perf_counter()
    try:
        proc = subprocess.Popen(cmd, cwd=REPO, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = proc.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)   # the whole group: pmc_hbm_traffic.py -> rocprofv3 -> bench.py
            proc.wait()
            raise
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd)
        with open(os.path.join(out, "hbm_traffic.json")) as f:
            k = json.load(f)["kernels"]
    except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
        r["traffic_live_error"] = repr(e)[:200]
        if r.get("traffic_source"):
            r["traffic_source"] += f"  [live measurement failed after {time.perf_counter() - t0:.0f} s]"
        return
    r["traffic_live_seconds"] = time.perf_counter() - t0
    r = result["roofline"]
    name = next((n for n in k if r["kernel_name"] in n and "3x3x3" in n), None)
    if name:
        r["traffic"] = k[name]["hbm_bytes_per_launch"]
        r["traffic_read_write"] = [k[name]["read_bytes"] / k[name]["launches"], k[name]["write_bytes"] / k[name]["launches"]]
        r["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two child passes of one step of this workload, "
                               f"tools/pmc_hbm_traffic.py), mean over the {k[name]['launches']} launches of {name}")
    flow = {n: v for n, v in k.items() if "flow_" in n}
    passes = next((v["launches"] for n, v in flow.items() if "flow_pre" in n), 0)
    if passes and "roofline_cinn" in result:
        result["roofline_cinn"]["measured_hbm_bytes_per_pass"] = sum(v["read_bytes"] + v["write_bytes"] for v in flow.values()) / passes
        result["roofline_cinn"]["measured_hbm_bytes_source"] = "measured by this run (same child passes)"


BOUNDARY_US = 1.45      # MI355X_MICROARCH.md: dependent kernel boundary, same stream (eager == hipGraph), between trivial 256-WG kernels
L2_ROUND_TRIP_US = 0.11  # MI355X_MICROARCH.md: global_load L2-hit latency ~180-225 cycles (200 cycles at the 1.8 GHz these small launches clock at)


def cinn_latency_floor(flow, nb, inv_us):
    """What the 160-deep dependent exchange of one pass cannot go below AS A CHAIN OF LAUNCHES: every launch pays the dependent
    kernel boundary and, inside, at least one dependent trip to L2 for the activations the previous launch left there plus one for
    its weight fragments (issued together: one trip), and the HBM time of the parameters it streams.  `frac_of_floor` = floor /
    measured: how much of the pass is explained by the launch structure, next to `frac` (bytes / time against 8 TB/s), which a
    latency chain cannot approach."""
    launches = 82 if nb <= 64 else 122   # folded chain while a workgroup holds one sample tile (csrc/i2v_flow_tile.hip)
    stream_us = flow.native().param_bytes / (PEAK_HBM_GBS * 1e9) * 1e6
    floor = launches * (BOUNDARY_US + L2_ROUND_TRIP_US) + stream_us
    return {"launches_per_pass": launches, "latency_floor_us": floor, "frac_of_floor": floor / inv_us,
            "latency_floor_model": f"{launches} launches x ({BOUNDARY_US} us dependent boundary + {L2_ROUND_TRIP_US} us L2 round trip) + "
                                   f"{stream_us:.1f} us to stream the parameters once at {PEAK_HBM_GBS:.0f} GB/s (MI355X_MICROARCH.md figures)"}


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


KERNEL_TEXT = {
    "conv_wino4_f16x3": "conv_wino4_f16x3_kernel (3x3x3 Conv3d, Winograd F(4,3) along W on split-fp16 operands: 6 transformed planes per 4 "
                        "outputs in two passes over the K loop, 3x v_mfma_f32_32x32x16_f16 per product; i2v_conv16w4.hip)",
    "conv_wino4g_f16x3": "conv_wino4g_f16x3_kernel (the same F(4,3) conv with the operand generated in the kernel: 8 MFMA waves + 4 producer waves "
                         "that form lrelu(norm(x)), B^T d and the fp16 hi / lo split from the conv's fp32 input; i2v_conv16w4g.hip)",
    "conv_wino_f16x3": "conv_wino_f16x3_kernel (3x3x3 Conv3d, Winograd F(2,3) along W on split-fp16 operands: 4 planes per output pair; "
                       "i2v_conv16w.hip)",
    "conv_mfma_f16x3": "conv_mfma_f16x3_kernel (3x3x3 Conv3d, direct split-fp16 implicit GEMM; i2v_conv16.hip)",
    "conv_mfma_f32": "conv_mfma_f32_kernel (3x3x3 Conv3d implicit GEMM, v_mfma_f32_32x32x2_f32; i2v_conv.hip)",
}


def roofline(prof, dt, mma, default_workload=False, layers=None, steps=1):
    """`roofline`: THE dominant kernel = the kernel with the largest summed duration among the 3x3x3 Conv3d launches of the
    timed steps.  achieved = ALGORITHMIC FLOPs (2*M*N*K of the reference's conv per launch) of ITS launches / their summed
    duration (HIP events on the launch stream, inside the timed region).  Every algorithmic FLOP costs three fp16 MFMA FLOPs
    in split-fp16 mode; F(4,3) executes 1/2 of the products, F(2,3) 2/3, conv_0 behind a x2 temporal up-sampling 18 of the 27
    taps: what the matrix cores actually issue is mfma_issue_frac.  `roofline_all_conv3`: the same over all 3x3x3 launches."""
    if prof["conv3_ms"] <= 0 or not layers:
        return {"roofline": None, "roofline_all_conv3": None}
    peak = PEAK_F16_MFMA_TFLOPS if mma != 0 else PEAK_FP32_MFMA_TFLOPS   # (mma 2 = auto runs the split-fp16 kernels unless the range guard switched a layer)
    by_kernel = {}
    for L in layers:
        k = by_kernel.setdefault(L["kernel"], {"ms": 0.0, "flops": 0.0, "mfma_flops": 0.0, "launches": 0, "layers": []})
        k["ms"] += L["ms"]; k["flops"] += L["flops"]; k["mfma_flops"] += L["mfma_flops"]; k["launches"] += L["launches"]
        k["layers"].append(L["layer"])
    dom = max(by_kernel, key=lambda n: by_kernel[n]["ms"])
    d = by_kernel[dom]
    # HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE collected in separate
    # rocprofv3 --pmc passes and corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot read PMCs itself: STATIC
    # figure from the newest committed summary (tools/pmc_hbm_traffic.py), valid for the default workload only.
    traffic, tsrc = None, None
    if mma != 0 and default_workload:
        path = _latest_traffic_file()
        try:
            with open(path) as f:
                kern = json.load(f)["kernels"]
            name = next(n for n in kern if dom + "_kernel" in n and "3x3x3" in n)
            traffic = kern[name]["hbm_bytes_per_launch"]
            tsrc = f"static (not measured by this run): profiles/{os.path.basename(path)} [{name}]: all launches of that kernel in one step"
        except (OSError, KeyError, ValueError, TypeError, StopIteration):
            traffic = None
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    r = {"kernel": KERNEL_TEXT.get(dom, dom), "kernel_name": dom + "_kernel", "layers": d["layers"], "bound": "mfma",
         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": tsrc,
         "mfma_issue_frac": d["mfma_flops"] / (d["ms"] * 1e-3) / 1e12 / peak, "launches": d["launches"],
         "avg_launch_ms": d["ms"] / max(d["launches"], 1), "ms_per_step": d["ms"] / max(steps, 1),
         "time_share": d["ms"] * 1e-3 / dt,
         "flops_per_launch_algorithmic": d["flops"] / max(d["launches"], 1)}
    ach_all = prof["conv3_flops"] / (prof["conv3_ms"] * 1e-3) / 1e12
    allc = {"kernels": {n: {"ms_per_step": k["ms"] / max(steps, 1), "tflops_algorithmic": k["flops"] / (k["ms"] * 1e-3) / 1e12,
                            "layers": k["layers"]} for n, k in by_kernel.items()},
            "achieved": ach_all, "peak": peak, "unit": "TFLOP/s", "frac": ach_all / peak,
            "mfma_issue_frac": prof["conv3_mfma_flops"] / (prof["conv3_ms"] * 1e-3) / 1e12 / peak,
            "launches": prof["conv3_launches"], "time_share": prof["conv3_ms"] * 1e-3 / dt,
            "per_layer": [{"layer": L["layer"], "kernel": L["kernel"], "ms_per_launch": L["ms"] / L["launches"],
                           "launches_per_step": L["launches"] / max(steps, 1),
                           "tflops_algorithmic": L["flops"] / (L["ms"] * 1e-3) / 1e12,
                           "tflops_mfma_issued": L["mfma_flops"] / (L["ms"] * 1e-3) / 1e12} for L in layers]}
    return {"roofline": r, "roofline_all_conv3": allc}


def write_per_layer(path, layers, steps, mma):
    """profiles/rNN_conv16_per_layer.csv: one row per 3x3x3 conv layer from the HIP-event pairs of the timed steps."""
    with open(path, "w") as f:
        f.write("layer,kernel,launches_per_step,ms_per_launch,algorithmic_gflop_per_launch,tflops_algorithmic,tflops_mfma_issued,"
                "frac_of_peak_algorithmic\n")
        peak = PEAK_F16_MFMA_TFLOPS if mma != 0 else PEAK_FP32_MFMA_TFLOPS
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

"""CPU-side checks: the C-ABI library loads and exports every symbol of include/i2v_hip.h, the class-surface mirror
keeps the reference's state_dict layout, host logic (sharding/collation over gloo, config reader, GIF tiling), and
the product path refuses to run without a GPU (no silent fallback)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import i2v_synth as synth
from conftest import PKG, REPO


def test_library_exports_every_declared_symbol():
    import ctypes
    import i2v_native
    if not os.path.exists(i2v_native.LIB_PATH):
        i2v_native.build()
    lib = ctypes.CDLL(i2v_native.LIB_PATH)
    header = open(os.path.join(REPO, "include", "i2v_hip.h")).read()
    declared = set(re.findall(r"\b(i2v_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/i2v_hip.h but not exported"
    assert declared == set(i2v_native.SYMBOLS), declared ^ set(i2v_native.SYMBOLS)
    assert i2v_native.lib().i2v_version() >= 1


def test_production_library_has_no_launch_path_environment_switches():
    """Round-5 review item: an environment variable silently changing the production kernel is not a drop-in property.  The F(4,3)
    kernel's structure switches (I2V_W4_*) and conv_img's frames-per-workgroup switch exist in the measurement build only
    (`make measure`, -DI2V_MEASURE); the production library neither contains their names nor the persistent-kernel instantiations,
    and every `getenv` left in csrc/ sits in a handle-creation / weight-packing function or inside `#ifdef I2V_MEASURE`."""
    import i2v_native
    if not os.path.exists(i2v_native.LIB_PATH):
        i2v_native.build()
    blob = open(i2v_native.LIB_PATH, "rb").read()
    for name in (b"I2V_W4_PIPE", b"I2V_W4_BN", b"I2V_W4_ORDER", b"I2V_W4_NTH", b"I2V_W4_SKEW", b"I2V_W4_TRACE", b"I2V_W4_LOADER", b"I2V_CONVIMG_TCH", b"I2V_C16_"):
        assert name not in blob, name
    assert b"conv_wino4_f16x3_kernelILi9ELi64ELi0ELi512E" in blob
    for pipe in (1, 2):
        assert b"conv_wino4_f16x3_kernelILi9ELi64ELi%dELi512E" % pipe not in blob
    if os.path.exists(i2v_native.MEASURE_LIB_PATH):
        mblob = open(i2v_native.MEASURE_LIB_PATH, "rb").read()
        assert b"I2V_W4_PIPE" in mblob and b"conv_wino4_f16x3_kernelILi9ELi64ELi1ELi512E" in mblob
    # source level: getenv only at creation / pack time (functions named below) or under I2V_MEASURE
    allowed = {"i2v_dec.hip": ("i2v_dec_create", "i2v_gblock_create"), "i2v_flow.hip": ("i2v_flow_create",),
               "i2v_flow_tile.hip": ("env_int",), "i2v_conv16w4.hip": ("w4_switches",), "i2v_convimg.hip": ("conv_img_mfma_forward",),
               "i2v_conv16.hip": ("c16_switch",)}     # (under -DC16_TUNE only: the stand-alone tools/conv16_bench build)
    import glob
    for f in glob.glob(os.path.join(PKG, "csrc", "*.hip")):
        text = open(f).read()
        if "getenv" not in text:
            continue
        assert os.path.basename(f) in allowed, f
    w4 = open(os.path.join(PKG, "csrc", "i2v_conv16w4.hip")).read()
    body = w4[w4.index("static W4Switches w4_switches()"):]
    body = body[:body.index("\n}\n")]
    assert w4.count("getenv(") == body.count("getenv(") and body.index("#ifdef I2V_MEASURE") < body.index("getenv(") < body.index("#endif")
    c16 = open(os.path.join(PKG, "csrc", "i2v_conv16.hip")).read()
    assert c16.count("getenv(") == 1 and c16.index("#ifdef C16_TUNE") < c16.index("getenv(") < c16.index("#else", c16.index("#ifdef C16_TUNE"))
    ci = open(os.path.join(PKG, "csrc", "i2v_convimg.hip")).read()
    assert ci.count("getenv(") == 1 and ci.index("#ifdef I2V_MEASURE") < ci.index("getenv(")
    ft = open(os.path.join(PKG, "csrc", "i2v_flow_tile.hip")).read()
    assert len(re.findall(r'env_int\("', ft)) == 2 and all(m.start() > ft.index("int flow_tile_pack(FlowTilePack& p") for m in re.finditer(r'env_int\("', ft))


def test_state_dict_layout_matches_reference_keys():
    from stage1_VAE.modules.decoder import Generator
    from stage2_cINN.modules.flow_blocks import ConditionalFlow
    for emb, ctrl in ((64, False), (94, True)):
        ref = synth.flow_state_dict(seed=1, n_flows=4, embedding_dim=emb, control=ctrl)
        flow = ConditionalFlow(64, emb, 512, 2, 4, conditioning_option="None", control=ctrl)
        mine = flow.state_dict()
        assert set(mine) == set(ref)
        for k, v in ref.items():
            assert tuple(mine[k].shape) == tuple(np.asarray(v).shape), k
        flow.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()})  # strict
    ref = synth.decoder_state_dict(seed=1, channel_factor=8)
    gen = Generator({"channel_factor": 8, "z_dim": 64, "upsample_s": [2, 1], "upsample_t": [2, 1], "spectral_norm": True})
    mine = gen.state_dict()
    assert set(mine) == set(ref)
    for k, v in ref.items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()})
    # without spectral norm the convs carry plain .weight
    gen2 = Generator({"channel_factor": 8, "z_dim": 64, "upsample_s": [2, 2], "upsample_t": [2, 1], "spectral_norm": False})
    assert set(gen2.state_dict()) == set(synth.decoder_state_dict(seed=1, channel_factor=8, spectral_norm=False))


def test_embedder_state_dict_layout():
    from stage2_cINN.AE.modules.AE import ResnetEncoder
    for norm in ("in", "bn"):
        ref = synth.embedder_state_dict(seed=1, z_dim=128, norm=norm)
        enc = ResnetEncoder({"z_dim": 128, "deterministic": False, "in_size": 128, "encoder_type": "resnet50", "norm": norm})
        mine = enc.state_dict()
        assert set(mine) == set(ref), set(mine) ^ set(ref)
        for k, v in ref.items():
            assert tuple(mine[k].shape) == tuple(np.asarray(v).shape), k
    # torchvision resnet50: 25,557,032 parameters - fc (2,049,000) - BatchNorm (53,120) = 23,454,912 conv weights
    assert sum(v.size for v in synth.embedder_state_dict(z_dim=64, norm="in").values()) == 23454912 + 2048 * 128 + 128


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of computing on the CPU."""
    import i2v_native
    from stage2_cINN.modules.flow_blocks import ConditionalFlow, InvLeakyRelu
    flow = ConditionalFlow(64, 64, 512, 2, 2, conditioning_option="None")
    for b in flow.sub_layers:
        b.norm_layer.initialized.fill_(1)
    with pytest.raises(i2v_native.I2VError):
        flow(torch.zeros(2, 64), torch.zeros(2, 64), reverse=True)
    with pytest.raises(i2v_native.I2VError):
        InvLeakyRelu()(torch.zeros(2, 64))
    with pytest.raises(NotImplementedError):
        ConditionalFlow(64, 64, 512, 2, 2, conditioning_option="parallel")


def test_child_load_invalidates_ancestor_handles():
    """A native handle packs the parameters of the whole sub-tree: loading a state_dict into a CHILD must drop the handles
    of every NativeBacked ancestor too (the next call rebuilds them), and so must moving the module."""
    from stage1_VAE.modules.decoder import Generator
    from stage2_cINN.modules.flow_blocks import ConditionalFlow
    gen = Generator({"channel_factor": 8, "z_dim": 64, "upsample_s": [2, 1], "upsample_t": [2, 1], "spectral_norm": True})
    marker = object()
    for m in (gen, gen.g_0, gen.g_0.norm_0):
        object.__setattr__(m, "_native", marker)
    gen.g_0.norm_0.load_state_dict(gen.g_0.norm_0.state_dict())
    assert gen._native is None and gen.g_0._native is None and gen.g_0.norm_0._native is None
    object.__setattr__(gen, "_native", marker)
    object.__setattr__(gen.g_1, "_native", marker)
    gen.g_1.load_state_dict(gen.g_1.state_dict())
    assert gen._native is None and gen.g_1._native is None
    flow = ConditionalFlow(64, 64, 512, 2, 2, conditioning_option="None")
    object.__setattr__(flow, "_native", marker)
    object.__setattr__(flow, "_init_checked", True)
    flow.sub_layers[1].coupling.s[0].load_state_dict(flow.sub_layers[1].coupling.s[0].state_dict())
    assert flow._native is None and flow._init_checked is False   # Q1: `initialized` is looked at again on the next forward
    assert flow.module_device().type == "cpu"


def test_shard_bounds():
    import i2v_dist
    for total in (1, 7, 8, 64, 65):
        for ws in (1, 2, 3, 8):
            spans = [i2v_dist.shard_bounds(total, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {pkg!r})
import i2v_dist
rank, ws, total = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[4]
dist.init_process_group('gloo', rank=rank, world_size=ws)
g = torch.Generator().manual_seed(3)
x0 = torch.rand(total, 3, 4, 4, generator=g); res = torch.randn(total, 8, generator=g); emb = torch.randn(total, 5, generator=g)
def fake_model(x, r, e):   # per-sample function standing in for cINN inverse + decoder
    return (x.mean(dim=(1, 2, 3))[:, None] + r.sum(1, keepdim=True) * e.sum(1, keepdim=True)).reshape(-1, 1, 1, 1, 1).expand(-1, 2, 3, 4, 4).contiguous()
out = i2v_dist.synthesize_sharded(fake_model, x0, res, emb)
ref = fake_model(x0, res, emb)
assert out.shape == ref.shape and torch.equal(out, ref), (rank, out.shape)
# embed=None is a legal input of the path (Model computes the embedding itself)
out_n = i2v_dist.synthesize_sharded(lambda x, r, e: fake_model(x, r, torch.ones(x.shape[0], 5)), x0, res, None)
assert torch.equal(out_n, fake_model(x0, res, torch.ones(total, 5)))
# the overlapped collator degrades to the plain all-gather for CPU tensors / ragged shards
lo, hi = i2v_dist.shard_bounds(total, ws, rank)
col = i2v_dist.OverlappedCollator(total)
col.submit(fake_model(x0[lo:hi], res[lo:hi], emb[lo:hi]))   # (a collective: every rank takes part, also with 0 rows)
assert torch.equal(col.result(), ref)
dist.destroy_process_group()
print('ok', rank)
"""


@pytest.mark.parametrize("total", [8, 7, 1])
def test_sharded_collation_gloo_world2(total, tmp_path):
    """N > 1 path on CPU: two gloo ranks shard the batch, run a per-sample stand-in model, all-gather; the result must
    equal the single-process result (even and ragged shards, and total < world size: one rank holds an EMPTY shard)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(pkg=PKG))
    port = str(29500 + (os.getpid() + total) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(total), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_config_reader(tmp_path):
    import i2v_config
    p = tmp_path / "c.yaml"
    p.write_text("Flow:\n  n_flows: 20\nTraining:\n  bs: 5\nDecoder:\n  upsample_s: [2, 1]\n")
    c = i2v_config.load(str(p))
    assert c.Flow["n_flows"] == 20 and c.Flow.n_flows == 20
    assert c.Training["control"] is None  # missing key -> None, as omegaconf 2.0.5 (get_model.py:42)
    assert c.Decoder["upsample_s"] == [2, 1]


def test_convert_seq2gif():
    from utils import auxiliaries as aux
    seq = torch.linspace(-1.2, 1.0, 2 * 3 * 3 * 4 * 5).reshape(2, 3, 3, 4, 5)
    gif = aux.convert_seq2gif(seq.clone())
    assert gif.shape == (3, 4, 10, 3) and gif.max() == 255 and gif.min() == 0
    ref = ((seq + 1) / 2).clamp(0, 1).permute(0, 1, 3, 4, 2).numpy()
    ref = np.concatenate((ref[0], ref[1]), axis=2)
    assert np.allclose(gif, 255 * ref / ref.max())


def test_embed_pos_matches_oracle():
    from oracle import flow_ref
    from stage2_cINN.modules.INN import SupervisedTransformer
    st = SupervisedTransformer(flow_in_channels=64, flow_mid_channels=512, flow_hidden_depth=2, n_flows=1,
                               flow_conditioning_option="None", flow_embedding_channels=64, control=True, dic=None)
    pos = torch.tensor([[0.05, 0.5, 1.0], [0.31, 0.999, 0.1001]])
    assert torch.equal(st.embed_pos(pos), flow_ref.embed_pos(pos))
    assert st.flow.cond_channels == 94


def test_bench_evidence_files_parse():
    """bench.py takes `roofline.traffic` and `roofline_cinn.measured_hbm_bytes_per_pass` from the newest committed PMC
    summary and labels them with their source: the file must exist and carry the fields bench.py reads (a silent None in
    the bench line is easy to miss)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    per_pass, src = bench.cinn_measured_bytes(True)
    assert per_pass is not None and 189e6 < per_pass < 2e9  # at least the parameters, not absurdly more
    assert src.startswith("static: profiles/") and os.path.exists(os.path.join(REPO, src.split(": ", 1)[1]))
    prof = {"conv3_ms": 10.0, "conv3_flops": 4e12, "conv3_mfma_flops": 1e13, "conv3_launches": 12}
    layers = [{"layer": "g_3.conv_0", "kernel": "conv_wino4_f16x3", "launches": 2, "ms": 5.0, "flops": 2e12, "mfma_flops": 4e12},
              {"layer": "g_3.conv_1", "kernel": "conv_wino4_f16x3", "launches": 2, "ms": 3.0, "flops": 1e12, "mfma_flops": 2e12},
              {"layer": "g_1.conv_1", "kernel": "conv_wino_f16x3", "launches": 2, "ms": 2.0, "flops": 1e12, "mfma_flops": 4e12}]
    both = bench.roofline(prof, 0.06, 1, default_workload=True, layers=layers, steps=2)
    r, ra = both["roofline"], both["roofline_all_conv3"]
    # `roofline` is the dominant kernel ALONE: the one with the largest summed duration, here F(4,3) with 8 of the 10 ms
    assert r["kernel_name"] == "conv_wino4_f16x3_kernel" and "F(4,3)" in r["kernel"] and r["layers"] == ["g_3.conv_0", "g_3.conv_1"]
    assert abs(r["achieved"] - 3e12 / 8e-3 / 1e12) < 1e-9 and r["launches"] == 4 and abs(r["avg_launch_ms"] - 2.0) < 1e-12
    assert r["traffic"] is not None and r["traffic"] > 1e9 and r["traffic_source"].startswith("static (not measured by this run): profiles/")
    assert "conv_wino4_f16x3_kernel" in r["traffic_source"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["bound"] == "mfma" and r["peak"] == 2500.0
    assert abs(ra["achieved"] - 400.0) < 1e-9 and set(ra["kernels"]) == {"conv_wino4_f16x3", "conv_wino_f16x3"}
    assert ra["per_layer"][0]["launches_per_step"] == 1 and abs(ra["per_layer"][0]["tflops_algorithmic"] - 400.0) < 1e-9
    assert bench.cinn_measured_bytes(False) == (None, None)
    # every BASELINE configuration is a bench workload
    assert {"bair64", "land128", "dtdb128", "iper128_t32"} <= set(bench.CONFIGS)
    assert bench.CONFIGS["dtdb128"]["batch"] == 256 and bench.CONFIGS["iper128_t32"]["vid"] == 32


def test_bench_line_schema():
    """bench.validate_line is what bench.py runs on its own line before printing it: the contract keys, ONE dominant kernel in
    `roofline`, a checksum for every timed step, single_call / sustained / exact_fp32 next to `value`.  Checked here on a
    hand-made line and on the newest committed round-4 line (profiles/r04_*_bench_bair64.json) when there is one."""
    import copy
    import glob
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    line = {"metric": "m", "value": 1.0, "unit": "frames/s", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 1.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w"}, "value_is": "pipelined", "rank_ms_per_step": [1.0], "output_check": {"finite": True},
            "roofline": {"kernel": "conv_wino4_f16x3_kernel (...)", "kernel_name": "conv_wino4_f16x3_kernel", "bound": "mfma",
                         "achieved": 1000.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.4, "traffic": None},
            "roofline_all_conv3": {"achieved": 900.0}, "roofline_cinn": {"frac": 0.05},
            "steps_check": {"steps_checked": 2, "all_bit_identical_to_serial_reference": True},
            "single_call": {"ms": 1.0, "frames_per_s": 1.0}, "sustained": {"seconds": 10.5, "ms_per_step": 1.0},
            "exact_fp32": {"ms_per_step": 5.0, "frames_per_s": 0.2, "frac": 0.7},
            "cpu_baseline": {"value": 1.0, "unit": "frames/s", "cores": 64, "kind": "port", "sample": "s"}}
    assert bench.validate_line(line)
    for breaker in (lambda d: d.pop("single_call"), lambda d: d["steps_check"].update(steps_checked=1),
                    lambda d: d["roofline"].update(frac=0.5), lambda d: d["sustained"].update(seconds=3.0),
                    lambda d: d.update(rank_ms_per_step=[1.0, 1.0]), lambda d: d.pop("value"),
                    lambda d: d["roofline"].update(kernel="a (...) for g_1..g_4; b")):
        bad = copy.deepcopy(line)
        breaker(bad)
        with pytest.raises(ValueError):
            bench.validate_line(bad)
    slim = copy.deepcopy(line)
    for k in ("cpu_baseline", "sustained", "exact_fp32", "single_call"):
        slim[k] = None
    assert bench.validate_line(slim, full=False)
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r04_*_bench_bair64.json")))[-1:]:
        with open(path) as f:
            assert bench.validate_line(json.loads(f.read().strip().splitlines()[-1]))


def test_winograd_kernel_keeps_its_hand_counted_waits_valid(tmp_path):
    """The Winograd conv kernel issues its loads as inline asm (LDS-DMA for the V brick, plain loads into the weight ring)
    and counts every `s_waitcnt vmcnt(n)` by hand.  That arithmetic only holds while the compiler adds no VMEM operation of
    its own inside the tap loop -- a register spill would (scratch accesses count in vmcnt).  Cross-compile the file and
    check, per instantiation: no scratch, no spills, and inside the loop exactly the loads the macros issue."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(PKG, "csrc", "i2v_conv16w.hip")
    asm = tmp_path / "w.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(PKG, "csrc"), "-S", "--cuda-device-only", src,
                    "-o", str(asm)], check=True, capture_output=True, timeout=600)
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN3i2v22conv_wino_f16x3_kernelILi(\d)ELi(\d+)EEEvNS_8WinoArgsE):[^\n]*\n(.*?)\.end_amdhsa_kernel", text,
                         flags=re.S | re.M)
    assert len(kernels) == 6, [k[0] for k in kernels]
    for name, nt, bn, whole in kernels:
        nt = int(nt)
        assert "scratch_" not in whole and "buffer_store" not in whole, name
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", whole), name
        # the tap loop = the one backward branch whose body holds MFMAs; one body = a pair of chunks = 2 NT taps
        loops = [mm for mm in re.finditer(r"^(\.LBB\d+_\d+):[^\n]*\n((?:(?!^\.LBB).)*?)s_cbranch_\w+ \1\n", whole, flags=re.S | re.M)
                 if "v_mfma" in mm.group(2)]
        assert len(loops) == 1, (name, len(loops))
        loop = loops[0].group(2)
        taps = 2 * nt
        wm_wn = 4 if bn == "64" else 2
        assert loop.count("v_mfma_f32_32x32x16_f16") == taps * 3 * wm_wn, name
        assert loop.count("global_load_lds_dwordx4") == 2 * 8, name                      # one V brick (8 pieces) per chunk
        assert len(re.findall(r"global_load_dwordx4", loop)) == taps * 2, name           # hi + lo weight fragment per tap
        assert loop.count("s_barrier") == 2, name                                        # one barrier per chunk
        # every MFMA block is preceded by a counted wait, none of them a full drain
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)]
        assert len(waits) >= taps and min(waits) >= 2, (name, waits)
    # replay every compiled tap loop against a model of the in-order VMEM queue (tools/check_asm_waits.py): no instruction
    # may touch a register a load can still be writing, no barrier may be crossed with an LDS-DMA load in flight ...
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_asm_waits as caw
    checked = 0
    for name, loops in caw.kernel_loops(text, "conv_wino_f16x3_kernel"):
        assert len(loops) == 1
        assert caw.check_loop(loops[0]) == [], name
        # ... and the counts are tight: one more outstanding operation at either kind of wait is a detected hazard
        b_wait, bar_wait = max(waits_of(loops[0])), min(waits_of(loops[0]))
        # (3-tap kernels: the chunk barrier's wait, every third tap, already covers the weight ring -- only it is tight)
        for w in ((b_wait, bar_wait) if "ILi3E" not in name else (bar_wait,)):
            mutated = re.sub(r"s_waitcnt vmcnt\(%d\)" % w, "s_waitcnt vmcnt(%d)" % (w + 1), loops[0])
            assert caw.check_loop(mutated) != [], (name, w)
        checked += 1
    assert checked == 6
    # every loop is entered with nothing in flight (the replay and the hand-written counts both assume it), and no asm load
    # reads an SGPR that a VALU instruction wrote within the five wait states the compiler would have inserted for its own loads
    assert caw.check_loop_entries(text, "conv_wino_f16x3_kernel") == []
    assert caw.check_scalar_operands(text, "conv_wino_f16x3_kernel") == []


@pytest.mark.parametrize("measure", [False, True])
def test_f43_kernel_keeps_its_hand_counted_waits_valid(tmp_path, measure):
    """The same static checks for the Winograd F(4,3) kernel (csrc/i2v_conv16w4.hip): every instantiation has TWO tap loops
    (pass A: planes 0..3, pass B: planes 4, 5), no scratch, exactly the loads the macros issue, and the replay of each
    compiled loop against the in-order VMEM queue finds no hazard while a wait relaxed by one does.  Both builds: the production
    library (7 one-workgroup-per-brick instantiations, no `getenv` in the object) and the measurement build (-DI2V_MEASURE: + the
    ten persistent-kernel instantiations behind I2V_W4_PIPE)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(PKG, "csrc", "i2v_conv16w4.hip")
    asm = tmp_path / "w4.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(PKG, "csrc"), "-S", "--cuda-device-only", src,
                    "-o", str(asm)] + (["-DI2V_MEASURE"] if measure else []), check=True, capture_output=True, timeout=900)
    text = asm.read_text()
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_asm_waits as caw
    kernels = re.findall(r"^(_ZN3i2v23conv_wino4_f16x3_kernelILi(\d)ELi(\d+)ELi(\d)ELi(\d+)EEEvNS_6W4ArgsE):[^\n]*\n(.*?)\.end_amdhsa_kernel", text,
                         flags=re.S | re.M)
    # <9,64> <6,64> <3,64> (SPADE's 2-D convs) <9,32> <6,32>, each as round 3's one-workgroup-per-brick kernel (PIPE = 0), as the
    # software-pipelined persistent kernel (PIPE = 1) and as its "lite" form (PIPE = 2); round 5: <9,32> <6,32> as 256-thread
    # workgroups of 64 tiles (two per CU), whose half-requests are 5 + 4 (pass A) and 3 + 2 (pass B) load instructions
    assert len(kernels) == (17 if measure else 7), [k[0] for k in kernels]
    assert measure or all(k[3] == "0" for k in kernels)   # production: PIPE = 0 only
    assert sorted(k[0] for k in kernels if k[4] == "256") == sorted(
        "_ZN3i2v23conv_wino4_f16x3_kernelILi%dELi32ELi0ELi256EEEvNS_6W4ArgsE" % n for n in (9, 6))
    for name, nt, bn, pipe, nth, whole in kernels:
        nt = int(nt)
        assert "scratch_" not in whole and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", whole), name
        loops = [mm.group(2) for mm in re.finditer(r"^(\.LBB\d+_\d+):[^\n]*\n((?:(?!^\.LBB).)*?)s_cbranch_\w+ \1\n", whole, flags=re.S | re.M)
                 if "v_mfma" in mm.group(2)]
        assert len(loops) == 2, (name, len(loops))
        # PIPE: pass B's half-requests carry two extra loads (the next brick's first V brick)
        # V load instructions per chunk (both half-requests)
        for loop, wm, vh2 in zip(loops, (4, 2) if bn == "64" else (2, 1), (9, 5) if nth == "256" else ((8, 8) if pipe == "1" else (8, 4))):
            taps = 2 * nt
            assert loop.count("v_mfma_f32_32x32x16_f16") == taps * 3 * wm, name
            # two chunks per loop iteration; the one-brick kernels (PIPE = 0) request V through a buffer descriptor since round 5
            lds_dma = loop.count("global_load_lds_dwordx4") + len(re.findall(r"buffer_load_dwordx4 [^\n]* lds", loop))
            assert lds_dma == 2 * vh2 and (loop.count("global_load_lds_dwordx4") == 0) == (pipe == "0"), name
            assert len(re.findall(r"global_load_dwordx4", loop)) == taps * 2, name
            assert loop.count("s_barrier") == 2, name
            assert caw.check_loop(loop) == [], name
            b_wait, bar_wait = max(waits_of(loop)), min(waits_of(loop))
            # (3-tap kernels: the chunk barrier's wait, every third tap, already covers the weight ring -- only it is tight)
            for w in ((b_wait, bar_wait) if nt != 3 else (bar_wait,)):
                mutated = re.sub(r"s_waitcnt vmcnt\(%d\)" % w, "s_waitcnt vmcnt(%d)" % (w + 1), loop)
                assert caw.check_loop(mutated) != [], (name, w)
    assert caw.check_loop_entries(text, "conv_wino4_f16x3_kernel") == []
    assert caw.check_scalar_operands(text, "conv_wino4_f16x3_kernel") == []


def test_generating_f43_kernel_static_checks(tmp_path):
    """csrc/i2v_conv16w4g.hip (round 6): the 12-wave workgroup whose four producer waves generate the operand.  Its MFMA role is the
    tap loop of the 512-thread 32-channel kernel WITHOUT V requests: both compiled loops must hold exactly the weight loads (no LDS-DMA),
    the hand-counted waits must replay clean and tight, the kernel must fit 768 threads (<= 168 VGPRs, no scratch), and the producer
    role must not contain packed fp32 arithmetic (-fno-slp-vectorize: MI355X_MICROARCH.md prices v_pk_* above the scalar pair next to
    MFMAs)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(PKG, "csrc", "i2v_conv16w4g.hip")
    asm = tmp_path / "w4g.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I" + os.path.join(PKG, "csrc"), "-S",
                    "--cuda-device-only", src, "-o", str(asm)], check=True, capture_output=True, timeout=900)
    text = asm.read_text()
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_asm_waits as caw
    kernels = re.findall(r"^(_ZN3i2v24conv_wino4g_f16x3_kernelILi9ELi(\d+)ELi([012])EEEvNS_6W4ArgsENS_9W4GenArgsE):[^\n]*\n(.*?)\.end_amdhsa_kernel", text,
                         flags=re.S | re.M)
    # (channels, mode): 0 / 1 generate the operand (ADAIN / SPADE form), 2 = the LOADER form (four extra waves issue the V requests)
    assert sorted((k[1], k[2]) for k in kernels) == [("32", "0"), ("32", "1"), ("32", "2"), ("64", "0"), ("64", "1")], [k[0] for k in kernels]
    for name, cin, spade, whole in kernels:
        assert "scratch_" not in whole and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", whole), name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", whole).group(1)) <= 168, name      # 12 waves per CU = 3 per SIMD
        assert not re.search(r"\bv_pk_(fma|mul|add)_f32", whole), name
        loops = [mm.group(2) for mm in re.finditer(r"^(\.LBB\d+_\d+):[^\n]*\n((?:(?!^\.LBB).)*?)s_cbranch_\w+ \1\n", whole, flags=re.S | re.M)
                 if "v_mfma" in mm.group(2)]
        assert len(loops) == 2, (name, len(loops))
        for loop, wm in zip(loops, (2, 1)):
            assert loop.count("v_mfma_f32_32x32x16_f16") == 18 * 3 * wm, name
            assert loop.count("global_load_lds_dwordx4") == 0 and not re.findall(r"buffer_load_dwordx4 [^\n]* lds", loop), name   # (no V request in a tap loop)
            assert len(re.findall(r"global_load_dwordx4", loop)) == 18 * 2 and loop.count("s_barrier") == 2, name
            assert caw.check_loop(loop) == [], name
            b_wait = max(waits_of(loop))
            mutated = re.sub(r"s_waitcnt vmcnt\(%d\)" % b_wait, "s_waitcnt vmcnt(%d)" % (b_wait + 1), loop)
            assert caw.check_loop(mutated) != [], (name, b_wait)
        if spade == "2":   # the loader role: 16 + 8 LDS-DMA requests per chunk, each behind its own m0, all waited for before the barrier
            dma = re.findall(r"buffer_load_dwordx4 [^\n]* lds", whole)
            assert len(dma) >= 24 and whole.count("s_waitcnt vmcnt(0)") >= 4, (name, len(dma))
    assert caw.check_loop_entries(text, "conv_wino4g_f16x3_kernel") == []
    assert caw.check_scalar_operands(text, "conv_wino4g_f16x3_kernel") == []


def test_f43_virtual_workgroup_map_is_a_bijection(tmp_path):
    """The F(4,3) kernel turns a (virtual) workgroup index into (sample, brick, channel tile, frame parity) -- per XCD, in one of
    three orders, also when the persistent variants loop over it.  A wrong map would skip or double bricks silently for the
    geometries no parity test happens to cover, so the map is compiled for the HOST (-DW4_DECODE_SELFTEST) and enumerated over a
    sweep of geometries: it must hit every work item exactly once."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = tmp_path / "w4_decode_selftest"
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-DW4_DECODE_SELFTEST", "-I" + os.path.join(PKG, "csrc"),
                    "-I" + os.path.join(REPO, "include"), os.path.join(PKG, "csrc", "i2v_conv16w4.hip"), os.path.join(PKG, "csrc", "i2v_common.hip"),
                    "-o", str(exe)], check=True, capture_output=True, timeout=900)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and " 0 bad" in out.stdout, out.stdout + out.stderr


def waits_of(loop):
    return [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)]


def test_bench_self_launches_multi_gpu_jobs_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE must re-execute itself under torch.distributed.run (the driver's
    scaling runs call it exactly like that); --dry rehearses the launch path, sharding, collation and the JSON line on
    CPU over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry", "--steps", "2", "--warmup", "1",
                        "--batch", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints exactly ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["collation_ok"] and line["dry"]
    assert line["config"]["global_batch"] == 6 and line["config"]["per_gpu_batch"] == 3
    # inside a launcher-made job a mismatching --gpus is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--dry"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr


def test_static_asm_checker_flags_the_hazards_it_exists_for():
    """tools/check_asm_waits.py on hand-made snippets: a loop entered with a load in flight (the round-2/3 prologues), an asm
    load behind a VALU-written SGPR, an MFMA that reads a register a load may still write -- and the clean versions of each."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_asm_waits as caw

    def kernel(prologue, loop):
        return "k1: ; @k1\n" + prologue + ".LBB0_1: ; loop\n" + loop + "\ts_cbranch_scc1 .LBB0_1\n\t.end_amdhsa_kernel\n"

    mfma = "\tv_mfma_f32_32x32x16_f16 v[10:25], v[1:4], v[5:8], v[10:25]\n"
    load = "\tglobal_load_dwordx4 v[1:4], v0, s[0:1]\n"
    # loop entry
    late = kernel("\ts_waitcnt vmcnt(0)\n" + load, mfma)
    good = kernel(load + "\ts_waitcnt vmcnt(0)\n", mfma)
    assert caw.check_loop_entries(late, "k1") and "after the last vmcnt(0)" in caw.check_loop_entries(late, "k1")[0]
    assert caw.check_loop_entries(kernel(load, mfma), "k1")
    assert caw.check_loop_entries(good, "k1") == []
    # VALU-written SGPR in front of an asm load: five wait states
    nops = "\ts_nop 0\n"
    assert caw.check_scalar_operands(kernel("\tv_readfirstlane_b32 s1, v9\n" + nops * 2 + load, mfma), "k1")
    assert caw.check_scalar_operands(kernel("\tv_readfirstlane_b32 s1, v9\n" + nops * 5 + load, mfma), "k1") == []
    assert caw.check_scalar_operands(kernel("\tv_readfirstlane_b32 s7, v9\n" + load, mfma), "k1") == []
    # in-loop: the weights of the next iteration are requested, the MFMA must not read them before vmcnt says so
    (name, loops), = caw.kernel_loops(kernel("", load + mfma), "k1")
    assert caw.check_loop(loops[0]) != []
    (name, loops), = caw.kernel_loops(kernel("", load + "\ts_waitcnt vmcnt(0)\n" + mfma), "k1")
    assert caw.check_loop(loops[0]) == []
    (name, loops), = caw.kernel_loops(kernel("", "\tglobal_load_lds_dwordx4 v[30:31], off\n\ts_barrier\n" + mfma), "k1")
    assert any("LDS-DMA" in b for b in caw.check_loop(loops[0]))


def test_every_environment_switch_is_documented():
    """Every I2V_* environment variable the native code or the Python host reads has a row in INTEGRATION.md."""
    import glob
    names = set()
    for f in glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")):
        names.update(re.findall(r'(?:getenv|env_int)\("(I2V_[A-Z0-9_]+)"', open(f).read()))
    for f in glob.glob(os.path.join(PKG, "*.py")) + [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")]:
        names.update(re.findall(r'environ(?:\.get\(|\[)"(I2V_[A-Z0-9_]+)"', open(f).read()))
    assert len(names) >= 8, names
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing

"""Pin the CPU oracle (oracle/*.py) against golden vectors produced by the reference's own
modules (tests/golden/make_golden.py).  CPU-only; part of the `-m "not gpu"` suite."""
import numpy as np
import pytest
import torch

import i2v_synth as synth
from conftest import load_golden, rel_l2
from oracle import decoder_ref, encoder_ref, flow_ref, model_ref

torch.set_grad_enabled(False)
TOL = 2e-5  # oracle vs reference: same ATen ops, differences are blocking/rounding only


def T(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def t(a):
    return torch.from_numpy(a)


def test_flow_units():
    g, meta = load_golden("flow_units")
    sd = T(synth.flow_state_dict(**meta["synth"]))
    p = "sub_layers.0.norm_layer."
    h, ld = flow_ref.actnorm_forward(sd, p, t(g["an_x"]))
    assert rel_l2(h, g["an_fwd"]) < 1e-6 and np.allclose(ld, g["an_logdet"], rtol=1e-6)
    assert rel_l2(flow_ref.actnorm_reverse(sd, p, t(g["an_x"])), g["an_rev"]) < 1e-6
    # Q1 data-dependent init
    loc, scale = flow_ref.actnorm_data_init(t(g["an0_x"]))
    assert rel_l2(loc, g["an0_loc"]) < 1e-6 and rel_l2(scale, g["an0_scale"]) < 1e-6
    # Q2: InvLeakyRelu logdet is reported as 0
    assert float(g["act_logdet"]) == 0.0
    assert np.array_equal(flow_ref.inv_lrelu_forward(t(g["act_x"])).numpy(), g["act_fwd"])
    assert rel_l2(flow_ref.inv_lrelu_reverse(t(g["act_x"])), g["act_rev"]) < 1e-7
    x = t(g["sh_x"])
    assert np.array_equal(x[:, sd["sub_layers.0.shuffle.forward_shuffle_idx"]].numpy(), g["sh_fwd"])
    assert np.array_equal(x[:, sd["sub_layers.0.shuffle.backward_shuffle_idx"]].numpy(), g["sh_rev"])
    assert rel_l2(flow_ref.mlp(sd, "sub_layers.0.coupling.s.0.", t(g["mlp_x"])), g["mlp_y"]) < TOL
    y, ld = flow_ref.coupling_forward(sd, "sub_layers.1.coupling.", t(g["cpl_x"]), t(g["cpl_e"]))
    assert rel_l2(y, g["cpl_fwd"]) < TOL and np.allclose(ld, g["cpl_logdet"], atol=1e-5)
    r = flow_ref.coupling_reverse(sd, "sub_layers.1.coupling.", t(g["cpl_x"]), t(g["cpl_e"]))
    assert rel_l2(r, g["cpl_rev"]) < TOL
    sdc = T(synth.flow_state_dict(**meta["synth_cond"]))
    y, ld = flow_ref.coupling_forward(sdc, "sub_layers.1.coupling.", t(g["cpl_x"]), t(g["cplc_e"]), mode="cond")
    assert rel_l2(y, g["cplc_fwd"]) < TOL and np.allclose(ld, g["cplc_logdet"], atol=1e-5)
    r = flow_ref.coupling_reverse(sdc, "sub_layers.1.coupling.", t(g["cpl_x"]), t(g["cplc_e"]), mode="cond")
    assert rel_l2(r, g["cplc_rev"]) < TOL
    y, ld = flow_ref.block_forward(sd, "sub_layers.1.", t(g["cpl_x"]), t(g["cpl_e"]))
    assert rel_l2(y, g["blk_fwd"]) < TOL and np.allclose(ld, g["blk_logdet"], atol=1e-5)
    assert rel_l2(flow_ref.block_reverse(sd, "sub_layers.1.", t(g["cpl_x"]), t(g["cpl_e"])), g["blk_rev"]) < TOL


@pytest.mark.parametrize("name", ["flow_full_e64", "flow_full_e128", "flow_full_ctrl"])
def test_flow_full(name):
    g, meta = load_golden(name)
    a = meta["synth"]
    sd = T(synth.flow_state_dict(**a))
    zt, ld = flow_ref.flow_forward(sd, t(g["x"]), t(g["e"]), control=a["control"])
    assert list(zt.shape) == list(g["fwd_shape"])
    assert rel_l2(zt.reshape(8, 64), g["fwd"]) < TOL
    assert np.allclose(ld, g["logdet"], rtol=1e-5, atol=1e-4)
    z = flow_ref.flow_reverse(sd, t(g["x"]), t(g["e"]), control=a["control"])
    assert list(z.shape) == list(g["rev_shape"])
    assert rel_l2(z.reshape(8, 64), g["rev"]) < TOL
    # property: inverse(forward(x)) == x
    rt = flow_ref.flow_reverse(sd, zt.reshape(8, 64), t(g["e"]), control=a["control"]).reshape(8, 64)
    assert float((rt - t(g["x"])).abs().max()) < 1e-4


def test_decoder_units():
    g, meta = load_golden("dec_units")
    sd = T(synth.decoder_state_dict(**meta["synth"]))
    x, img, z = t(g["u_x"]), t(g["u_img"]), t(g["u_z"])
    assert rel_l2(decoder_ref.spade(sd, "g_3.norm_0.", x, img), g["spade"]) < TOL
    assert rel_l2(decoder_ref.spade(sd, "g_3.norm_0.", x, img, faithful=False), g["spade"]) < TOL
    assert rel_l2(decoder_ref.adain(sd, "g_3.norm_1.", x[:, :16].contiguous(), z), g["adain"]) < TOL
    assert rel_l2(decoder_ref.norm3d(sd, "g_3.norm_s.", x), g["norm3d"]) < TOL
    xb = t(g["b_x"])
    assert meta["sigma_g1_conv0"] < 0  # signed-sigma quirk D6 is exercised
    assert rel_l2(decoder_ref.generator_block(sd, "g_1", xb, z, img), g["block_g1"]) < TOL
    assert rel_l2(decoder_ref.generator_block(sd, "g_0", xb, z, img), g["block_g0"]) < TOL
    folded = decoder_ref.fold_spectral_norm(sd)
    assert rel_l2(decoder_ref.generator_block(folded, "g_1", xb, z, img, faithful=False), g["block_g1"]) < TOL


def test_decoder_nf8_bair():
    g, meta = load_golden("dec_nf8_bair")
    sd = T(synth.decoder_state_dict(**meta["synth"]))
    out, pre = decoder_ref.generator(sd, t(g["img"]), t(g["z"]), meta["upsample_s"], meta["upsample_t"],
                                     return_pre_tanh=True)
    assert out.shape == (2, 16, 3, 64, 64) and out.is_contiguous()
    assert rel_l2(pre, g["pre_tanh"]) < TOL and rel_l2(out, g["out"]) < TOL
    folded = decoder_ref.fold_spectral_norm(sd)
    out2 = decoder_ref.generator(folded, t(g["img"]), t(g["z"]), faithful=False)
    assert rel_l2(out2, g["out"]) < TOL
    # batch-permutation equivariance (the shardability property)
    perm = torch.tensor([1, 0])
    out3 = decoder_ref.generator(folded, t(g["img"])[perm], t(g["z"])[perm], faithful=False)
    assert rel_l2(out3, g["out"][perm.numpy()]) < TOL


def test_decoder_nf8_128():
    g, meta = load_golden("dec_nf8_128")
    sd = decoder_ref.fold_spectral_norm(T(synth.decoder_state_dict(**meta["synth"])))
    out = decoder_ref.generator(sd, t(g["img"]), t(g["z"]), meta["upsample_s"], meta["upsample_t"], faithful=False)
    assert out.shape == (1, 16, 3, 128, 128)
    assert rel_l2(out[..., ::2, ::2], g["out_s2"]) < TOL


def test_decoder_full_width_bair():
    g, meta = load_golden("dec_nf64_bair")
    sd = decoder_ref.fold_spectral_norm(T(synth.decoder_state_dict(**meta["synth"])))
    out = decoder_ref.generator(sd, t(g["img"]), t(g["z"]), meta["upsample_s"], meta["upsample_t"], faithful=False)
    assert rel_l2(out[..., ::2, ::2], g["out_s2"]) < TOL


def test_model_forward_semantics():
    g, meta = load_golden("model_nf8")
    fsd = T(synth.flow_state_dict(**meta["synth_flow"]))
    dsd = decoder_ref.fold_spectral_norm(T(synth.decoder_state_dict(**meta["synth_dec"])))
    kw = dict(upsample_s=meta["upsample_s"], upsample_t=meta["upsample_t"], faithful=False)
    y32 = model_ref.model_forward(fsd, dsd, t(g["x1"]), t(g["r1"]), t(g["e1"]), vid_length=32, **kw)
    assert y32.shape == (1, 32, 3, 64, 64) and rel_l2(y32, g["y32"]) < 5e-5
    # Q3: the final slice is over the batch dimension
    yq3 = model_ref.model_forward(fsd, dsd, t(g["x3"]), t(g["r3"]), t(g["e3"]), vid_length=2, **kw)
    assert list(yq3.shape) == list(g["yq3_shape"]) == [2, 16, 3, 64, 64]
    assert rel_l2(yq3[:, ::4], g["yq3_t4"]) < 5e-5
    y20 = model_ref.model_forward(fsd, dsd, t(g["x1"]), t(g["r1"]), t(g["e1"]), vid_length=20, **kw)
    assert list(y20.shape) == list(g["y20_shape"]) == [1, 32, 3, 64, 64]


def test_model_128_t32():
    """cfg5 geometry fixture (128x128, nf = 32, E = 128, two decoder passes): the oracle against the reference's output."""
    g, meta = load_golden("model_nf32_128_t32")
    fsd = T(synth.flow_state_dict(**meta["synth_flow"]))
    dsd = decoder_ref.fold_spectral_norm(T(synth.decoder_state_dict(**meta["synth_dec"])))
    seq = model_ref.synthesize(fsd, dsd, t(g["x0"])[:1], t(g["r"])[:1], t(g["e"])[:1], 32, meta["upsample_s"], meta["upsample_t"],
                               faithful=False)
    assert seq.shape == (1, 32, 3, 128, 128)
    assert rel_l2(seq[..., ::4, ::4], g["out_s4"][:1]) < TOL


def golden_clip(meta, g):
    """Regenerates the encoder fixture's input clip from its seed and checks it against the stored head / checksum."""
    x = 2 * torch.rand(*meta["x_shape"], generator=torch.Generator().manual_seed(meta["x_seed"])) - 1
    assert np.array_equal(x.reshape(-1)[:16].numpy(), g["x_head"]) and abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-6
    return x


@pytest.mark.parametrize("name", ["enc3d_bair", "enc3d_land"])
def test_motion_encoder(name):
    g, meta = load_golden(name)
    a = meta["synth"]
    sd = T(synth.encoder3d_state_dict(**a))
    mu, logvar = encoder_ref.encoder(sd, golden_clip(meta, g), a["channels"], a["stride_s"], meta["stride_t"])
    assert rel_l2(mu, g["mu"]) < TOL and rel_l2(logvar, g["logvar"]) < TOL

// Motion encoder of the transfer path (row N3): Encoder.forward (reference stage1_VAE/modules/resnet3D.py:138-219).
//
// 3D ResNet-18: Conv3d(3, c0, (3,7,7), stride 2, pad (1,3,3)) -> GroupNorm(16) -> ReLU -> 4 layers of 2 BasicBlocks
// (3x3x3 convs, strides (stride_t, stride_s, stride_s) on the first conv and on the 3x3x3 down-sample conv of each layer,
// GroupNorm(16, affine) everywhere) -> squeeze(T = 1) -> conv_mu / conv_var = Conv2d(c4, z, 4) on the 4x4 map.
// Model.transfer (get_model.py:87) keeps the MEAN mu; logvar is returned as well.
//
// The 3-channel stem is a direct fp32 convolution on the vector ALU with its whole [3*7*7*3][c0] weight set resident in
// LDS (a 7x7 stride-2 halo brick of 16-channel rows would not fit the MFMA kernel's LDS tile).  The 3x3x3 convs run on the
// split-fp16 matrix-core kernel (i2v_conv16.hip, fp32-class accuracy): the stride-1 ones directly, the strided conv1 /
// down-sample convs as stride-1 2-tap convs on a space-to-depth copy of their input (enc_s2d_hl16_kernel).  GroupNorm +
// ReLU (+ residual) is one elementwise pass over channels-last activations driven by per-(b,c) (A,B) pairs from fp64
// statistics; it emits fp32 (residuals) and / or the split-fp16 operand format (next conv).  conv_mu | conv_var: one GEMM.
#include <algorithm>
#include <memory>

#include "i2v_conv.h"

namespace i2v {

// x [B][3][Tin][Hin][Win] (the reference's NCDHW clip) -> out channels-last [B][To][Ho][Wo][C0]; stride 2, pad (1,3,3).
// Weights in LDS as [tap][cin][C0]; one thread = one output position x 16 output channels.
__global__ __launch_bounds__(256) void enc_stem_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                       float* __restrict__ out, int B, int Tin, int Hin, int Win, int To, int Ho,
                                                       int Wo, int C0) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [147*3][C0]
    const int nw = 147 * 3 * C0;
    for (int i = threadIdx.x; i < nw; i += 256) wl[i] = wp[i];
    __syncthreads();
    const int ngrp = C0 / 16;
    const long npos = (long)B * To * Ho * Wo;
    const long total = (npos + 63) / 64 * 64 * ngrp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int pos_in_blk = (int)(i % 64);  // 64 consecutive positions x ngrp groups: lanes of a wave share the group
        long blk = i / 64;
        const int g = (int)(blk % ngrp);
        const long p = (blk / ngrp) * 64 + pos_in_blk;
        if (p >= npos) continue;
        long q = p;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); q /= Ho;
        const int to = (int)(q % To);
        const int b = (int)(q / To);
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int dt = 0; dt < 3; ++dt) {
            const int t = 2 * to + dt - 1;
            if ((unsigned)t >= (unsigned)Tin) continue;
            for (int dh = 0; dh < 7; ++dh) {
                const int h = 2 * ho + dh - 3;
                if ((unsigned)h >= (unsigned)Hin) continue;
                for (int dw = 0; dw < 7; ++dw) {
                    const int w = 2 * wo + dw - 3;
                    if ((unsigned)w >= (unsigned)Win) continue;
                    const int tap = (dt * 7 + dh) * 7 + dw;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = x[((((long)b * 3 + c) * Tin + t) * Hin + h) * Win + w];
                        const float4* wr = reinterpret_cast<const float4*>(wl + ((tap * 3 + c) * C0 + 16 * g));
#pragma unroll
                        for (int j4 = 0; j4 < 4; ++j4) {
                            const float4 w4 = wr[j4];
                            acc[4 * j4] = fmaf(v, w4.x, acc[4 * j4]);
                            acc[4 * j4 + 1] = fmaf(v, w4.y, acc[4 * j4 + 1]);
                            acc[4 * j4 + 2] = fmaf(v, w4.z, acc[4 * j4 + 2]);
                            acc[4 * j4 + 3] = fmaf(v, w4.w, acc[4 * j4 + 3]);
                        }
                    }
                }
            }
        }
        float4* o = reinterpret_cast<float4*>(out + p * C0 + 16 * g);
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) o[j4] = make_float4(acc[4 * j4], acc[4 * j4 + 1], acc[4 * j4 + 2], acc[4 * j4 + 3]);
    }
}

// Space-to-depth for the strided 3x3x3 convs: a stride-2 conv with pad 1 reads x[2o + d - 1], d = 0..2, i.e. the ODD
// phase at shifts (-1, 0) and the EVEN phase at shift 0.  With X'[o][phase p][c] = x[2o + p][c] it becomes a STRIDE-1 conv
// with a 2-tap kernel (offsets -1, 0; the even phase's -1 tap is zero) over P x C channels -- which runs on the split-fp16
// matrix-core kernel (27 of its 8 x 8 = 64 tap-phase products are non-zero; still 4x faster than the strided fp32 path).
// in: fp32 channels-last [B][T][H][W][C]; out: hl16 [B][T/st][H/2][W/2][P*C], P = st*4, phase index (pt*2 + ph)*2 + pw.
__global__ __launch_bounds__(256) void enc_s2d_hl16_kernel(const float* __restrict__ x, char* __restrict__ out, int B, int T, int H,
                                                           int W, int C, int st) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    const int C8 = C >> 3, To = T / st, Ho = H / 2, Wo = W / 2, P = st * 4;
    const long total = (long)B * To * Ho * Wo * P * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long q = i / C8;
        const int p = (int)(q % P); q /= P;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); q /= Ho;
        const int to = (int)(q % To);
        const int b = (int)(q / To);
        const int pw = p & 1, ph = (p >> 1) & 1, pt = p >> 2;
        const float* src = x + ((((long)b * T + to * st + pt) * H + 2 * ho + ph) * W + 2 * wo + pw) * C + 8 * c8;
        const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
        const float r[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        half8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 hh = (_Float16)r[j];
            hi[j] = hh;
            lo[j] = (_Float16)(r[j] - (float)hh);
        }
        char* o = out + i * 32;  // flat (b, to, ho, wo, p, c8) order == channels-last with channel index p*C + 8*c8
        *reinterpret_cast<half8_t*>(o) = hi;
        *reinterpret_cast<half8_t*>(o + 16) = lo;
    }
}

// eps*std + mu with std = exp(0.5 logvar)  (Encoder.reparameterize, resnet3D.py:199-203); ml = [B][2z] = (mu | logvar)
__global__ void reparam_kernel(const float* __restrict__ ml, const float* __restrict__ eps, float* __restrict__ sample,
                               float* __restrict__ mu, float* __restrict__ logvar, int B, int z) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * z; i += gridDim.x * blockDim.x) {
        const int b = i / z, j = i % z;
        const float m = ml[(long)b * 2 * z + j], lv = ml[(long)b * 2 * z + z + j];
        mu[i] = m;
        logvar[i] = lv;
        if (sample) sample[i] = fmaf(eps[i], expf(0.5f * lv), m);
    }
}

}  // namespace i2v

using namespace i2v;

namespace {

struct GN { DevBuf w, b; int C = 0; };

struct EncBlock {
    ConvWeights c1, c2, down;
    // split-fp16 packing of the stride-1 3x3x3 convs (i2v_conv16.hip: fp16 matrix cores at fp32-class accuracy); the strided
    // conv1 / downsample convs of a stage's first block stay on the exact-fp32 kernel
    Conv16Weights c1_16, c2_16, down_16;  // down_16: a channel-changing block without stride (its 3x3x3 projection conv)
    bool c1_is16 = false;
    // strided conv1 / downsample conv (spatial stride 2, temporal stride st) as stride-1 convs on the space-to-depth input
    // (enc_s2d_hl16_kernel); the *_t1 variants serve a single-frame input, where the temporal stride is the identity
    Conv16Weights c1_s2d, down_s2d, c1_s2d_t1, down_s2d_t1;
    bool use_s2d = false;
    GN n1, n2, nd;
    int planes = 0, ss = 1, st = 1;
    bool has_down = false;
};

// weights [N][C][3][3][3] of a stride-(st,2,2) conv -> [N][P*C][kt][2][2] of the equivalent stride-1 conv on the
// space-to-depth input (see enc_s2d_hl16_kernel): per strided dim, tap shift -1 <- (odd phase, w[0]); shift 0 <- (even
// phase, w[1]) and (odd phase, w[2]).  A dim with stride 1 keeps its three taps.
int pack_s2d(const float* w, int N, int C, int st, Conv16Weights& out) {
    const int P = st * 4, kt = st == 2 ? 2 : 3;
    std::vector<float> w2((size_t)N * P * C * kt * 4, 0.f);
    auto src_tap = [](int phase, int shift) { return phase == 0 ? (shift == 1 ? 1 : -1) : (shift == 0 ? 0 : 2); };  // shift idx 0: -1, 1: 0
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int pt = 0; pt < st; ++pt)
                for (int ph = 0; ph < 2; ++ph)
                    for (int pw = 0; pw < 2; ++pw)
                        for (int it = 0; it < kt; ++it)
                            for (int ih = 0; ih < 2; ++ih)
                                for (int iw = 0; iw < 2; ++iw) {
                                    const int dt = st == 2 ? src_tap(pt, it) : it, dh = src_tap(ph, ih), dw = src_tap(pw, iw);
                                    if (dt < 0 || dh < 0 || dw < 0) continue;
                                    const int p = (pt * 2 + ph) * 2 + pw;
                                    w2[(((size_t)n * P * C + (size_t)p * C + c) * kt + it) * 4 + ih * 2 + iw] =
                                        w[((size_t)n * C + c) * 27 + (dt * 3 + dh) * 3 + dw];
                                }
    return out.pack(w2.data(), nullptr, N, P * C, kt, 2, 2, 1.0);
}

int load_gn(const StateDict& sd, const std::string& name, int C, GN& g) {
    const float* w = sd.f32(name + ".weight", C);
    const float* b = sd.f32(name + ".bias", C);
    if (!w || !b) return I2V_E_MISSING;
    g.C = C;
    int rc = g.w.upload(w, (size_t)C * 4);
    if (rc) return rc;
    return g.b.upload(b, (size_t)C * 4);
}

}  // namespace

struct i2v_encoder3d {
    i2v_encoder3d_cfg cfg;
    int device = 0;
    bool loaded = false;
    DevBuf stem_w;
    GN nstem;
    std::vector<EncBlock> blocks;
    ConvWeights head;  // conv_mu | conv_var as one Linear(16*c4 -> 2z) on the channels-last 4x4 map
};

namespace {

struct EncWs { size_t buf[4], h16[3], sums, coef, ml, total; };  // h16: activations in the split-fp16 operand format

EncWs enc_ws(const i2v_encoder3d* e, int B, int T, int H, int W) {
    const int To = (T + 2 - 3) / 2 + 1, Ho = H / 2, Wo = W / 2;
    size_t mx = (size_t)To * Ho * Wo * e->cfg.channels[0];
    int t = To, h = Ho, w = Wo;
    for (int l = 0; l < 4; ++l) {
        t = (t + e->cfg.stride_t[l] - 1) / e->cfg.stride_t[l];
        h /= e->cfg.stride_s[l]; w /= e->cfg.stride_s[l];
        mx = std::max(mx, (size_t)t * h * w * e->cfg.channels[l + 1]);
    }
    EncWs L;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = align_up(o + floats * 4, 256); return r; };
    for (auto& b : L.buf) b = take((size_t)B * mx);
    for (auto& b : L.h16) b = take((size_t)B * mx);
    L.sums = take((size_t)B * 1024 * 4);
    L.coef = take((size_t)B * 1024 * 2);
    L.ml = take((size_t)B * 2 * e->cfg.z_dim);
    L.total = o;
    return L;
}

// y = act(GroupNorm(16, affine)(x) (+ res))
// (out and / or out16: fp32 for residual connections and strided convs, split-fp16 for the next stride-1 conv)
int gn_act(const GN& g, const float* x, const float* res, float* out, void* out16, int B, long P, bool relu, double* sums,
           float* coef, hipStream_t st) {
    int rc;
    if ((rc = stats_forward(x, sums, B, P, g.C, st))) return rc;
    if ((rc = coef_forward(sums, coef, B, g.C, 16, (double)P, st, g.w.as<float>(), g.b.as<float>()))) return rc;
    return norm_act_forward(x, coef, g.C, res, out, B, P, g.C, relu, st, out16);
}

}  // namespace

extern "C" {

int i2v_encoder3d_create(const i2v_encoder3d_cfg* cfg, i2v_encoder3d** out) {
    I2V_REQUIRE(cfg && out, I2V_E_INVALID, "i2v_encoder3d_create: null argument");
    for (int i = 0; i < 5; ++i)
        I2V_REQUIRE(cfg->channels[i] >= 16 && cfg->channels[i] % 16 == 0 && cfg->channels[i] <= 1024, I2V_E_INVALID,
                    "i2v_encoder3d_create: channels must be multiples of 16 in [16, 1024]");
    for (int i = 0; i < 4; ++i)
        I2V_REQUIRE((cfg->stride_s[i] == 1 || cfg->stride_s[i] == 2) && (cfg->stride_t[i] == 1 || cfg->stride_t[i] == 2), I2V_E_INVALID,
                    "i2v_encoder3d_create: strides must be 1 or 2");
    I2V_REQUIRE(cfg->z_dim > 0 && !cfg->use_max_pool, I2V_E_INVALID,
                "i2v_encoder3d_create: z_dim > 0 required; use_max_pool is false in every shipped config and is not supported");
    I2V_REQUIRE(147 * 3 * cfg->channels[0] * 4 <= 150 * 1024, I2V_E_INVALID, "i2v_encoder3d_create: stem too wide for LDS");
    int ndev = 0;
    I2V_HIP_CHECK(hipGetDeviceCount(&ndev));
    I2V_REQUIRE(ndev > 0, I2V_E_HIP, "i2v_encoder3d_create: no HIP device");
    auto e = std::make_unique<i2v_encoder3d>();
    e->cfg = *cfg;
    I2V_HIP_CHECK(hipGetDevice(&e->device));
    { const char* zp = nullptr; if (int rcz = zero_page(&zp)) return rcz; }  // allocated here, not inside a forward
    *out = e.release();
    return I2V_OK;
}

void i2v_encoder3d_destroy(i2v_encoder3d* e) { delete e; }

int i2v_encoder3d_load(i2v_encoder3d* e, const i2v_tensor* tensors, int32_t n_tensors) {
    if (e) I2V_REQUIRE_DEVICE(e->device, "i2v_encoder3d_load");
    I2V_REQUIRE(e && tensors && n_tensors > 0, I2V_E_INVALID, "i2v_encoder3d_load: null argument");
    StateDict sd(tensors, n_tensors);
    const int* ch = e->cfg.channels;
    int rc;
    {   // conv1.weight [c0][3][3][7][7] -> [tap][cin][c0]
        const float* w = sd.f32("conv1.weight", (int64_t)ch[0] * 3 * 147);
        if (!w) return I2V_E_MISSING;
        std::vector<float> p((size_t)147 * 3 * ch[0]);
        for (int n = 0; n < ch[0]; ++n)
            for (int c = 0; c < 3; ++c)
                for (int tap = 0; tap < 147; ++tap) p[((size_t)tap * 3 + c) * ch[0] + n] = w[((size_t)n * 3 + c) * 147 + tap];
        if ((rc = e->stem_w.upload(p.data(), p.size() * 4))) return rc;
        if ((rc = load_gn(sd, "norm1", ch[0], e->nstem))) return rc;
    }
    e->blocks.clear();
    e->blocks.reserve(8);
    int inplanes = ch[0];
    for (int L = 0; L < 4; ++L)
        for (int i = 0; i < 2; ++i) {  // resnet18: BasicBlock x [2,2,2,2]
            e->blocks.emplace_back();
            EncBlock& b = e->blocks.back();
            const int planes = ch[L + 1];
            b.planes = planes;
            b.ss = i == 0 ? e->cfg.stride_s[L] : 1;
            b.st = i == 0 ? e->cfg.stride_t[L] : 1;
            b.has_down = i == 0 && (e->cfg.stride_s[L] != 1 || inplanes != planes);  // resnet3D.py:180
            const std::string p = "layer." + std::to_string(L) + "." + std::to_string(i) + ".";
            const float* w1 = sd.f32(p + "conv1.weight", (int64_t)planes * inplanes * 27);
            const float* w2 = sd.f32(p + "conv2.weight", (int64_t)planes * planes * 27);
            if (!w1 || !w2) return I2V_E_MISSING;
            b.c1_is16 = b.ss == 1 && b.st == 1;
            b.use_s2d = b.ss == 2 && inplanes % 8 == 0;
            if (b.c1_is16) { if ((rc = b.c1_16.pack(w1, nullptr, planes, inplanes, 3, 3, 3, 1.0))) return rc; }
            else if (b.use_s2d) {
                if ((rc = pack_s2d(w1, planes, inplanes, b.st, b.c1_s2d))) return rc;
                if (b.st == 2 && (rc = pack_s2d(w1, planes, inplanes, 1, b.c1_s2d_t1))) return rc;
            } else if ((rc = b.c1.pack(w1, nullptr, planes, inplanes, 3, 3, 3, 1.0))) return rc;
            if ((rc = b.c2_16.pack(w2, nullptr, planes, planes, 3, 3, 3, 1.0))) return rc;
            if ((rc = load_gn(sd, p + "bn1", planes, b.n1))) return rc;
            if ((rc = load_gn(sd, p + "bn2", planes, b.n2))) return rc;
            if (b.has_down) {
                const float* wd = sd.f32(p + "downsample.0.weight", (int64_t)planes * inplanes * 27);
                if (!wd) return I2V_E_MISSING;
                if (b.c1_is16) {
                    if ((rc = b.down_16.pack(wd, nullptr, planes, inplanes, 3, 3, 3, 1.0))) return rc;
                } else if (b.use_s2d) {
                    if ((rc = pack_s2d(wd, planes, inplanes, b.st, b.down_s2d))) return rc;
                    if (b.st == 2 && (rc = pack_s2d(wd, planes, inplanes, 1, b.down_s2d_t1))) return rc;
                } else if ((rc = b.down.pack(wd, nullptr, planes, inplanes, 3, 3, 3, 1.0))) return rc;
                if ((rc = load_gn(sd, p + "downsample.1", planes, b.nd))) return rc;
            }
            inplanes = planes;
        }
    {   // conv_mu / conv_var: Conv2d(c4, z, 4, 1, 0) on [B, c4, 4, 4] == Linear over (h, w, c) of the channels-last map
        const int c4 = ch[4], z = e->cfg.z_dim;
        const float* wm = sd.f32("conv_mu.weight", (int64_t)z * c4 * 16);
        const float* bm = sd.f32("conv_mu.bias", z);
        const float* wv = sd.f32("conv_var.weight", (int64_t)z * c4 * 16);
        const float* bv = sd.f32("conv_var.bias", z);
        if (!wm || !bm || !wv || !bv) return I2V_E_MISSING;
        std::vector<float> w((size_t)2 * z * 16 * c4), bias((size_t)2 * z);
        for (int part = 0; part < 2; ++part)
            for (int n = 0; n < z; ++n) {
                const float* src = (part ? wv : wm) + (size_t)n * c4 * 16;
                for (int c = 0; c < c4; ++c)
                    for (int hw = 0; hw < 16; ++hw) w[((size_t)(part * z + n) * 16 + hw) * c4 + c] = src[(size_t)c * 16 + hw];
                bias[part * z + n] = (part ? bv : bm)[n];
            }
        if ((rc = e->head.pack(w.data(), bias.data(), 2 * z, 16 * c4, 1, 1, 1, 1.0))) return rc;
    }
    e->loaded = true;
    return I2V_OK;
}

size_t i2v_encoder3d_workspace_bytes(const i2v_encoder3d* e, int32_t batch, int32_t t, int32_t h, int32_t w) {
    if (!e || batch <= 0 || t <= 0 || h <= 0 || w <= 0) return 0;
    return enc_ws(e, batch, t, h, w).total;
}

int i2v_encoder3d_forward(i2v_encoder3d* e, const float* x, int32_t t, int32_t h, int32_t w, const float* eps, float* sample,
                          float* mu, float* logvar, void* workspace, size_t workspace_bytes, int32_t batch, void* stream) {
    if (e) I2V_REQUIRE_DEVICE(e->device, "i2v_encoder3d_forward");
    I2V_REQUIRE(e && e->loaded, I2V_E_STATE, "i2v_encoder3d_forward: weights not loaded");
    I2V_REQUIRE(x && mu && logvar && workspace && batch > 0 && t >= 1, I2V_E_INVALID, "i2v_encoder3d_forward: bad argument");
    I2V_REQUIRE(!sample || eps, I2V_E_INVALID, "i2v_encoder3d_forward: a sample needs eps");
    I2V_REQUIRE(h >= 64 && w >= 64 && (h & (h - 1)) == 0 && (w & (w - 1)) == 0, I2V_E_INVALID,
                "i2v_encoder3d_forward: frame size %dx%d must be a power of two >= 64", h, w);
    const int B = batch;
    const EncWs L = enc_ws(e, B, t, h, w);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_encoder3d_forward: workspace %zu < required %zu", workspace_bytes,
                L.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    float *b0 = F(L.buf[0]), *b1 = F(L.buf[1]), *b2 = F(L.buf[2]), *b3 = F(L.buf[3]);
    double* sums = reinterpret_cast<double*>(ws + L.sums);
    float* coef = F(L.coef);
    int rc;
    int T = (t + 2 - 3) / 2 + 1, H = h / 2, W = w / 2, C = e->cfg.channels[0];
    I2V_REQUIRE((T & (T - 1)) == 0, I2V_E_INVALID, "i2v_encoder3d_forward: %d input frames give %d stem frames (need a power of two)", t, T);
    {
        static bool attr[I2V_MAX_DEV] = {};
        if (int rc2 = ensure_dynamic_lds(reinterpret_cast<const void*>(enc_stem_kernel), 160 * 1024, attr)) return rc2;
        hipLaunchKernelGGL(enc_stem_kernel, dim3(1024), dim3(256), (size_t)147 * 3 * C * 4, st, x, e->stem_w.as<float>(), b0, B, t, h, w, T, H,
                           W, C);
        I2V_HIP_CHECK(hipGetLastError());
    }
    void *x16 = ws + L.h16[0], *y16 = ws + L.h16[1], *t16 = ws + L.h16[2];
    if ((rc = gn_act(e->nstem, b0, nullptr, b1, x16, B, (long)T * H * W, true, sums, coef, st))) return rc;
    float *xcur = b1, *y = b0, *t1 = b2, *t2 = b3;
    for (EncBlock& b : e->blocks) {
        I2V_REQUIRE(T % b.st == 0 || T == 1, I2V_E_INVALID, "i2v_encoder3d_forward: odd temporal extent %d", T);
        const int To = b.st == 2 ? (T + 1) / 2 : T, Ho = H / b.ss, Wo = W / b.ss;
        const long Po = (long)To * Ho * Wo;
        // with T == 1 a temporal stride of 2 is the identity on the single frame (pad 1, kernel 3): use stride 1 there
        const int st_eff = (T == 1) ? 1 : b.st;
        // out = relu(bn1(conv1(x)))   (only conv2 reads it: split-fp16 copy only)
        if (b.use_s2d) {  // strided convs of this block read the space-to-depth copy of its input, built once in y16
                          // (free until this block's output is written there by the last gn_act)
            const long tot = (long)B * To * Ho * Wo * (st_eff * 4) * (C / 8);
            hipLaunchKernelGGL(enc_s2d_hl16_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 16384)), dim3(256), 0, st, xcur,
                               static_cast<char*>(y16), B, T, H, W, C, st_eff);
            I2V_HIP_CHECK(hipGetLastError());
        }
        if (b.c1_is16) rc = conv16_forward(b.c1_16, x16, t1, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st);
        else if (b.use_s2d) rc = conv16_forward(st_eff == b.st ? b.c1_s2d : b.c1_s2d_t1, y16, t1, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st);
        else rc = conv_forward(b.c1, xcur, C, t1, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st, nullptr, b.ss, st_eff);
        if (rc) return rc;
        if ((rc = gn_act(b.n1, t1, nullptr, nullptr, t16, B, Po, true, sums, coef, st))) return rc;
        // out = bn2(conv2(out))
        if ((rc = conv16_forward(b.c2_16, t16, t1, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st))) return rc;
        const float* residual = xcur;
        if (b.has_down) {  // 3x3x3 strided conv + GroupNorm (resnet3D.py:181-189)
            if (b.c1_is16) rc = conv16_forward(b.down_16, x16, t2, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st);
            else if (b.use_s2d) rc = conv16_forward(st_eff == b.st ? b.down_s2d : b.down_s2d_t1, y16, t2, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st);
            else rc = conv_forward(b.down, xcur, C, t2, nullptr, 1, 1, B, To, Ho, Wo, EPI_NONE, st, nullptr, b.ss, st_eff);
            if (rc) return rc;
            if ((rc = gn_act(b.nd, t2, nullptr, y, nullptr, B, Po, false, sums, coef, st))) return rc;
            // y now holds the residual; the block output goes to t2 (and, split, to y16) below
            if ((rc = gn_act(b.n2, t1, y, t2, y16, B, Po, true, sums, coef, st))) return rc;
            std::swap(xcur, t2);
        } else {
            if ((rc = gn_act(b.n2, t1, residual, y, y16, B, Po, true, sums, coef, st))) return rc;
            std::swap(xcur, y);
        }
        std::swap(x16, y16);
        T = To; H = Ho; W = Wo; C = b.planes;
    }
    I2V_REQUIRE(T == 1 && H == 4 && W == 4, I2V_E_INVALID,
                "i2v_encoder3d_forward: the feature map is [%d,%d,%d], conv_mu/conv_var need [1,4,4]", T, H, W);
    if ((rc = conv_forward(e->head, xcur, 16 * C, F(L.ml), nullptr, 1, 1, B, 1, 1, 1, EPI_NONE, st))) return rc;
    hipLaunchKernelGGL(reparam_kernel, dim3((B * e->cfg.z_dim + 255) / 256), dim3(256), 0, st, F(L.ml), eps, sample, mu, logvar, B,
                       e->cfg.z_dim);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // extern "C"

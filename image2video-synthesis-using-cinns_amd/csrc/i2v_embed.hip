// Conditioning embedder: ResnetEncoder.encode(x).mode()  (reference: stage2_cINN/AE/modules/AE.py:91-166,
// distributions.py:6-42; row N1 of the coverage contract -- the step immediately in front of the cINN).
//
// torchvision.models.resnet50 (0.8.1 layout: Bottleneck [3,4,6,3], stride on the 3x3 conv, all convs bias-free) with
// norm_layer = InstanceNorm2d (affine-less, per-sample statistics) or BatchNorm2d (eval: running statistics), AdaptiveAvgPool,
// and `fc` replaced by Conv2d(2048, 2E, 1); .mode() of the diagonal Gaussian = the first E channels (the mean).
// The [-1,1] start frame goes in as is (the ImageNet transform at AE.py:111-114 is never applied).
//
// ~0.34 GFLOP per 64x64 sample: < 0.1 % of the path.  Every conv runs on the exact-fp32 MFMA implicit-GEMM kernel
// (i2v_conv.hip, here with spatial stride and the 7x7 stem), activations channels-last; norm + ReLU (+ residual) is one
// elementwise pass driven by per-(b,c) (A,B) pairs (InstanceNorm: from fused fp64 statistics; BatchNorm: constants folded
// at load).
#include <algorithm>
#include <memory>

#include "i2v_conv.h"

namespace i2v {

// out = act(x * A + B (+ res)); coef index = b * cstride + c (cstride = C for per-sample norms, 0 for BatchNorm constants).
// out16 (optional, C % 8 == 0): the same values in the split-fp16 operand format of i2v_conv16.hip (per 8 channels: 8 x fp16
// hi | 8 x fp16 lo); out may then be null when only the next convolution reads the result.
__global__ __launch_bounds__(256) void norm_act_kernel(const float* __restrict__ x, const float2* __restrict__ coef, long cstride,
                                                       const float* __restrict__ res, float* __restrict__ out,
                                                       char* __restrict__ out16, long per, int C, int relu) {
    const int C4 = C >> 2;
    const int b = blockIdx.y;
    const float2* cp = coef + (long)b * cstride;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const long off = (long)b * per * 4 + i * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + off);
        const float4 ab0 = *reinterpret_cast<const float4*>(cp + 4 * c4), ab1 = *reinterpret_cast<const float4*>(cp + 4 * c4 + 2);
        float4 r = make_float4(fmaf(v.x, ab0.x, ab0.y), fmaf(v.y, ab0.z, ab0.w), fmaf(v.z, ab1.x, ab1.y), fmaf(v.w, ab1.z, ab1.w));
        if (res) {
            const float4 q = *reinterpret_cast<const float4*>(res + off);
            r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
        }
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        if (out) *reinterpret_cast<float4*>(out + off) = r;
        if (out16) {
            typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
            const float rr[4] = {r.x, r.y, r.z, r.w};
            half4_t hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const _Float16 hh = (_Float16)rr[j];
                hi[j] = hh;
                lo[j] = (_Float16)(rr[j] - (float)hh);
            }
            // element index (b, pos, c4): the 8-channel group c4 / 2 occupies 32 bytes, this thread owns half of each 16-byte part
            char* o = out16 + ((long)b * per + (i - c4)) * 16 + (c4 >> 1) * 32 + (c4 & 1) * 8;
            *reinterpret_cast<half4_t*>(o) = hi;
            *reinterpret_cast<half4_t*>(o + 16) = lo;
        }
    }
}

// MaxPool2d(kernel 3, stride 2, padding 1) on channels-last [B][H][W][C] -> [B][H/2][W/2][C]
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W,
                                                         int C) {
    const int C4 = C >> 2, Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long p = i / C4;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dh = -1; dh <= 1; ++dh)
            for (int dw = -1; dw <= 1; ++dw) {
                const int h = 2 * ho + dh, w = 2 * wo + dw;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * C + 4 * c4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        *reinterpret_cast<float4*>(out + i * 4) = m;
    }
}

// AdaptiveAvgPool2d((1,1)) from the fused statistics: mean[b][c] = sum / P
__global__ void mean_from_sums_kernel(const double* __restrict__ sums, float* __restrict__ out, long n, double inv_count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (float)(sums[2 * i] * inv_count);
}

int norm_act_forward(const float* x, const float* coef, long cstride, const float* res, float* out, int B, long P, int C, bool relu,
                     hipStream_t st, void* out_hl16) {
    I2V_REQUIRE(out || out_hl16, I2V_E_INVALID, "norm_act: no output");
    I2V_REQUIRE(!out_hl16 || C % 8 == 0, I2V_E_INVALID, "norm_act: the split-fp16 output needs C %% 8 == 0 (C = %d)", C);
    const long per = P * (C / 4);
    hipLaunchKernelGGL(norm_act_kernel, dim3((unsigned)std::min<long>((per + 255) / 256, 4096), B), dim3(256), 0, st, x,
                       reinterpret_cast<const float2*>(coef), cstride, res, out, static_cast<char*>(out_hl16), per, C, relu ? 1 : 0);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace i2v

using namespace i2v;

namespace {

struct NormP {  // BatchNorm constants folded to (A,B) per channel; empty for InstanceNorm
    DevBuf ab;
};

struct Bneck {
    ConvWeights c1, c2, c3, down;
    Conv16Weights c2_16;  // the stride-1 3x3 convs run on the split-fp16 matrix-core kernel (i2v_conv16.hip)
    bool c2_is16 = false;
    NormP n1, n2, n3, nd;
    int width = 0, stride = 1;
    bool has_down = false;
};

}  // namespace

struct i2v_embedder {
    int E = 0, bn = 0;
    int device = 0;
    bool loaded = false;
    ConvWeights stem, fc;
    NormP nstem;
    std::vector<Bneck> blocks;
    // One handle = (normally) one workspace: forwards on a handle are serialised.  A call that arrives on another stream than the
    // previous one (LatentPrefetcher's side stream next to the main stream) first waits for the previous forward (event recorded
    // behind every forward), like i2v_flow: a direct call on the main stream cannot race a ticket outstanding on the side stream.
    StreamOrder order;   // (capture-aware: see i2v_common.h)
};

namespace {

int load_bn(const StateDict& sd, const std::string& name, int C, NormP& out) {
    const float* w = sd.f32(name + ".weight", C);
    const float* b = sd.f32(name + ".bias", C);
    const float* m = sd.f32(name + ".running_mean", C);
    const float* v = sd.f32(name + ".running_var", C);
    if (!w || !b || !m || !v) return I2V_E_MISSING;
    std::vector<float> ab((size_t)C * 2);
    for (int c = 0; c < C; ++c) {  // F.batch_norm eval: (x - mean) / sqrt(var + eps) * w + b, eps = 1e-5
        const double a = (double)w[c] / std::sqrt((double)v[c] + 1e-5);
        ab[2 * c] = (float)a;
        ab[2 * c + 1] = (float)((double)b[c] - (double)m[c] * a);
    }
    return out.ab.upload(ab.data(), ab.size() * 4);
}

struct EmbWs { size_t img, buf[5], sums, coef, pooled, total; size_t bufsz; };

EmbWs emb_ws(int B, int H, int W) {
    EmbWs L;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = align_up(o + floats * 4, 256); return r; };
    L.bufsz = (size_t)B * H * W * 16;  // = B*(H/2)*(W/2)*64 = B*(H/4)*(W/4)*256: the largest activation
    L.img = take((size_t)B * H * W * 16);
    for (auto& b : L.buf) b = take(L.bufsz);
    L.sums = take((size_t)B * 2048 * 4);
    L.coef = take((size_t)B * 2048 * 2);
    L.pooled = take((size_t)B * 2048);
    L.total = o;
    return L;
}

// y = act(norm(x) (+ res)) for a channels-last [B][P][C] tensor
int norm_act(const i2v_embedder* e, const NormP& np, const float* x, const float* res, float* out, int B, long P, int C, bool relu,
             double* sums, float* coef, hipStream_t st, bool as_hl16 = false) {
    const float2* cp;
    long cstride;
    int rc;
    if (e->bn) {
        cp = np.ab.as<float2>();
        cstride = 0;
    } else {  // InstanceNorm2d(affine=False, track_running_stats=False): statistics of this sample and channel
        if ((rc = stats_forward(x, sums, B, P, C, st))) return rc;
        if ((rc = coef_forward(sums, coef, B, C, C, (double)P, st))) return rc;
        cp = reinterpret_cast<const float2*>(coef);
        cstride = C;
    }
    // as_hl16: `out` receives the split-fp16 operand format (same bytes per element) for a following conv16_forward
    return norm_act_forward(x, reinterpret_cast<const float*>(cp), cstride, res, as_hl16 ? nullptr : out, B, P, C, relu, st,
                            as_hl16 ? out : nullptr);
}

}  // namespace

extern "C" {

int i2v_embedder_create(int32_t z_dim, int32_t use_batchnorm, i2v_embedder** out) {
    I2V_REQUIRE(out && z_dim > 0, I2V_E_INVALID, "i2v_embedder_create: bad argument");
    int ndev = 0;
    I2V_HIP_CHECK(hipGetDeviceCount(&ndev));
    I2V_REQUIRE(ndev > 0, I2V_E_HIP, "i2v_embedder_create: no HIP device");
    auto e = std::make_unique<i2v_embedder>();
    e->E = z_dim;
    e->bn = use_batchnorm ? 1 : 0;
    I2V_HIP_CHECK(hipGetDevice(&e->device));
    { const char* zp = nullptr; if (int rcz = zero_page(&zp)) return rcz; }  // allocated here, not inside a forward
    *out = e.release();
    return I2V_OK;
}

void i2v_embedder_destroy(i2v_embedder* e) { delete e; }

int i2v_embedder_load(i2v_embedder* e, const i2v_tensor* tensors, int32_t n_tensors) {
    if (e) I2V_REQUIRE_DEVICE(e->device, "i2v_embedder_load");
    I2V_REQUIRE(e && tensors && n_tensors > 0, I2V_E_INVALID, "i2v_embedder_load: null argument");
    StateDict sd(tensors, n_tensors);
    int rc;
    const float* w = sd.f32("model.conv1.weight", 64 * 3 * 49);
    if (!w) return I2V_E_MISSING;
    if ((rc = e->stem.pack(w, nullptr, 64, 3, 1, 7, 7, 1.0))) return rc;
    if (e->bn && (rc = load_bn(sd, "model.bn1", 64, e->nstem))) return rc;
    e->blocks.clear();
    e->blocks.reserve(16);
    const int nblk[4] = {3, 4, 6, 3}, width[4] = {64, 128, 256, 512};
    int inplanes = 64;
    for (int L = 0; L < 4; ++L)
        for (int i = 0; i < nblk[L]; ++i) {
            e->blocks.emplace_back();
            Bneck& b = e->blocks.back();
            b.width = width[L];
            b.stride = (i == 0 && L > 0) ? 2 : 1;
            b.has_down = i == 0;
            const std::string p = "model.layer" + std::to_string(L + 1) + "." + std::to_string(i) + ".";
            const int wd = width[L], outp = 4 * wd;
            const float* w1 = sd.f32(p + "conv1.weight", (int64_t)wd * inplanes);
            const float* w2 = sd.f32(p + "conv2.weight", (int64_t)wd * wd * 9);
            const float* w3 = sd.f32(p + "conv3.weight", (int64_t)outp * wd);
            if (!w1 || !w2 || !w3) return I2V_E_MISSING;
            if ((rc = b.c1.pack(w1, nullptr, wd, inplanes, 1, 1, 1, 1.0))) return rc;
            b.c2_is16 = b.stride == 1;
            if (b.c2_is16) { if ((rc = b.c2_16.pack(w2, nullptr, wd, wd, 1, 3, 3, 1.0))) return rc; }
            else if ((rc = b.c2.pack(w2, nullptr, wd, wd, 1, 3, 3, 1.0))) return rc;
            if ((rc = b.c3.pack(w3, nullptr, outp, wd, 1, 1, 1, 1.0))) return rc;
            if (e->bn) {
                if ((rc = load_bn(sd, p + "bn1", wd, b.n1))) return rc;
                if ((rc = load_bn(sd, p + "bn2", wd, b.n2))) return rc;
                if ((rc = load_bn(sd, p + "bn3", outp, b.n3))) return rc;
            }
            if (b.has_down) {
                const float* wdn = sd.f32(p + "downsample.0.weight", (int64_t)outp * inplanes);
                if (!wdn) return I2V_E_MISSING;
                if ((rc = b.down.pack(wdn, nullptr, outp, inplanes, 1, 1, 1, 1.0))) return rc;
                if (e->bn && (rc = load_bn(sd, p + "downsample.1", outp, b.nd))) return rc;
            }
            inplanes = outp;
        }
    // fc = Conv2d(2048, 2E, kernel = 1) (AE.py:121-124); .mode() keeps the mean = output channels [0, E)
    const float* fw = sd.f32("model.fc.sub_layers.0.weight", (int64_t)2 * e->E * 2048);
    const float* fb = sd.f32("model.fc.sub_layers.0.bias", (int64_t)2 * e->E);
    if (!fw || !fb) return I2V_E_MISSING;
    if ((rc = e->fc.pack(fw, fb, e->E, 2048, 1, 1, 1, 1.0))) return rc;
    e->loaded = true;
    return I2V_OK;
}

size_t i2v_embedder_workspace_bytes(const i2v_embedder* e, int32_t batch, int32_t h, int32_t w) {
    if (!e || batch <= 0 || h <= 0 || w <= 0) return 0;
    return emb_ws(batch, h, w).total;
}

int i2v_embedder_forward(i2v_embedder* e, const float* img, int32_t h, int32_t w, float* embed, void* workspace,
                         size_t workspace_bytes, int32_t batch, void* stream) {
    if (e) I2V_REQUIRE_DEVICE(e->device, "i2v_embedder_forward");
    I2V_REQUIRE(e && e->loaded, I2V_E_STATE, "i2v_embedder_forward: weights not loaded");
    I2V_REQUIRE(img && embed && workspace && batch > 0, I2V_E_INVALID, "i2v_embedder_forward: null argument");
    I2V_REQUIRE(h >= 64 && w >= 64 && (h & (h - 1)) == 0 && (w & (w - 1)) == 0, I2V_E_INVALID,
                "i2v_embedder_forward: image size %dx%d must be a power of two >= 64", h, w);
    const int B = batch;
    const EmbWs L = emb_ws(B, h, w);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_embedder_forward: workspace %zu < required %zu", workspace_bytes,
                L.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rco = e->order.entry(st)) return rco;
    StreamOrderMark mark{&e->order, st};   // records the end of this forward on its stream (also on the error paths)
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    float* buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = F(L.buf[i]);
    double* sums = reinterpret_cast<double*>(ws + L.sums);
    float* coef = F(L.coef);
    int rc;
    // stem: conv1 7x7 s2 -> norm -> ReLU -> MaxPool 3x3 s2
    if ((rc = resize_forward(img, F(L.img), B, h, w, h, w, st))) return rc;  // NCHW -> channels-last, 3 -> 16 zero-padded
    int H = h / 2, W = w / 2;
    if ((rc = conv_forward(e->stem, F(L.img), 16, buf[0], nullptr, 1, 1, B, 1, H, W, EPI_NONE, st, nullptr, 2))) return rc;
    if ((rc = norm_act(e, e->nstem, buf[0], nullptr, buf[1], B, (long)H * W, 64, true, sums, coef, st))) return rc;
    {
        const long tot = (long)B * (H / 2) * (W / 2) * 16;
        hipLaunchKernelGGL(maxpool3s2_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 65536)), dim3(256), 0, st, buf[1], buf[0],
                           B, H, W, 64);
        I2V_HIP_CHECK(hipGetLastError());
    }
    H /= 2; W /= 2;
    float *x = buf[0], *y = buf[1], *t1 = buf[2], *t2 = buf[3], *idn = buf[4];
    int C = 64;
    for (Bneck& b : e->blocks) {
        const int wd = b.width, outp = 4 * wd, Ho = H / b.stride, Wo = W / b.stride;
        const long P = (long)H * W, Po = (long)Ho * Wo;
        // conv1 1x1 -> norm -> ReLU
        if ((rc = conv_forward(b.c1, x, C, t1, nullptr, 1, 1, B, 1, H, W, EPI_NONE, st))) return rc;
        if ((rc = norm_act(e, b.n1, t1, nullptr, t2, B, P, wd, true, sums, coef, st, b.c2_is16))) return rc;
        // conv2 3x3 (stride on THIS conv, torchvision >= 0.3) -> norm -> ReLU
        if (b.c2_is16) rc = conv16_forward(b.c2_16, t2, t1, nullptr, 1, 1, B, 1, Ho, Wo, EPI_NONE, st);
        else rc = conv_forward(b.c2, t2, wd, t1, nullptr, 1, 1, B, 1, Ho, Wo, EPI_NONE, st, nullptr, b.stride);
        if (rc) return rc;
        if ((rc = norm_act(e, b.n2, t1, nullptr, t2, B, Po, wd, true, sums, coef, st))) return rc;
        // conv3 1x1 -> norm
        if ((rc = conv_forward(b.c3, t2, wd, t1, nullptr, 1, 1, B, 1, Ho, Wo, EPI_NONE, st))) return rc;
        const float* identity = x;
        if (b.has_down) {  // downsample = Conv2d 1x1 stride s (bias-free) -> norm
            if ((rc = conv_forward(b.down, x, C, t2, nullptr, 1, 1, B, 1, Ho, Wo, EPI_NONE, st, nullptr, b.stride))) return rc;
            if ((rc = norm_act(e, b.nd, t2, nullptr, idn, B, Po, outp, false, sums, coef, st))) return rc;
            identity = idn;
        }
        // out = ReLU(norm(conv3) + identity)
        if ((rc = norm_act(e, b.n3, t1, identity, y, B, Po, outp, true, sums, coef, st))) return rc;
        std::swap(x, y);
        H = Ho; W = Wo; C = outp;
    }
    // AdaptiveAvgPool2d((1,1)) then fc (first E output channels = the mean of the posterior)
    if ((rc = stats_forward(x, sums, B, (long)H * W, C, st))) return rc;
    hipLaunchKernelGGL(mean_from_sums_kernel, dim3((unsigned)(((long)B * C + 255) / 256)), dim3(256), 0, st, sums, F(L.pooled), (long)B * C,
                       1.0 / ((double)H * W));
    I2V_HIP_CHECK(hipGetLastError());
    return conv_forward(e->fc, F(L.pooled), C, embed, nullptr, 1, 1, B, 1, 1, 1, EPI_NONE, st);
}

}  // extern "C"

// Implicit-GEMM Conv3d / Conv2d / Linear on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen/cuDNN convolutions reached by the reference through nn.Conv3d (decoder.py:14-18,81),
// nn.Conv2d (normalization_layer.py:13-15) and nn.Linear (decoder.py:72, normalization_layer.py:44).
//
// GEMM view: Out[M = output positions (b,t,h,w)][N = Cout] = sum_{tap, c} In[pos + tap][c] * W[tap][c][n].
// Activations are channels-last ([B][T][H][W][C]) so the K dimension (channels of one tap) is contiguous.
//
// One 256-thread workgroup (4 wavefronts, one per SIMD) computes a brick of 128 output positions
// (TB x TT x TH x TW, chosen per layer so the halo stays small) times BN output channels:
//   * per 16-channel K chunk the input halo brick is staged ONCE into LDS (rows of 16 floats padded to 20 so the
//     ds_read_b128 fragment reads of 16 consecutive positions fall on distinct bank groups) and reused by all
//     27 taps and all BN output channels;
//   * the [BN][16] weight slab of each tap is double-buffered in LDS (next tap's global loads are issued before
//     the current tap's MFMAs, written after them: one barrier per tap);
//   * each wavefront owns WM x WN accumulator tiles of 32x32 (16 VGPRs each).  MFMA k-slot (s, half) of group g
//     maps to channel 8g + 4*half + s, so one ds_read_b128 per operand feeds four MFMAs;
//   * taps that can only see zero padding for the whole brick (T == 1 layers) are skipped;
//   * epilogue: + bias, + residual read through the nearest-upsample index map (decoder.py:102-114 folded into
//     the index math), optional leaky_relu(0.2), or tanh + [B][T][3][H][W] store for conv_img.
#include <cstdlib>

#include "i2v_conv.h"

namespace i2v {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(ConvArgs a) {
    constexpr int BN = 32 * WN * WAVES_N;
    constexpr int LS = CONV_LDS_STRIDE;
    static_assert(32 * WM * WAVES_M == CONV_BM, "tile");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int half = lane >> 5, l31 = lane & 31;

    const int pt = a.KT / 2, ph = a.KH / 2, pw = a.KW / 2;
    const int sS = a.sS, sT = a.sT;  // strides: the halo brick covers (TH-1)*sS + KH input rows
    const int HT = (a.TT - 1) * sT + a.KT, HH = (a.TH - 1) * sS + a.KH, HW = (a.TW - 1) * sS + a.KW;
    const int NPOS = a.TB * HT * HH * HW;
    const int ntaps = a.KT * a.KH * a.KW;
    const int Hin = a.H * sS, Win = a.W * sS, Tin = a.T * sT;

    float* in_lds = smem;
    float* w_lds = smem + NPOS * LS;
    int* rowpos = reinterpret_cast<int*>(w_lds + 2 * BN * LS);
    int* rowres = rowpos + CONV_BM;
    int* taplist = rowres + CONV_BM;  // [0] = count, [1..] = valid taps (up to 7x7 = 49)

    const int nNt = a.CoutPad / BN;
    const int ntile = blockIdx.x % nNt;
    int brick = blockIdx.x / nNt;
    const int bw = brick % a.nbW; brick /= a.nbW;
    const int bh = brick % a.nbH; brick /= a.nbH;
    const int bt = brick % a.nbT; brick /= a.nbT;
    const int bb = brick;
    const int b0 = bb * a.TB, t0 = bt * a.TT, h0 = bh * a.TH, w0 = bw * a.TW;
    const int n0 = ntile * BN;

    if (tid < CONV_BM) {
        int m = tid;
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        const int b = b0 + m, t = t0 + it, h = h0 + ih, w = w0 + iw;
        const bool ok = b < a.B && m < a.TB;  // (a brick may be only partly filled when LDS limits TB)
        rowpos[tid] = ok ? ((b * a.T + t) * a.H + h) * a.W + w : -1;
        rowres[tid] = ok ? ((b * (a.T / a.rt) + t / a.rt) * (a.H / a.rs) + h / a.rs) * (a.W / a.rs) + w / a.rs : 0;
    }
    if (tid == 0) {
        int cnt = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int dt = tap / (a.KH * a.KW);
            const int lo = t0 * sT + dt - pt, hi = lo + (a.TT - 1) * sT;
            if (hi < 0 || lo >= Tin) continue;  // the whole brick reads zero padding for this tap
            taplist[1 + cnt++] = tap;
        }
        taplist[0] = cnt;
    }

    // LDS float offset (tap (0,0,0), channel 4*half) of this lane's A rows
    int aoff[WM];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        int m = wave_m * (32 * WM) + 32 * wm + l31;
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        aoff[wm] = ((((m < a.TB ? m : 0) * HT + it * sT) * HH + ih * sS) * HW + iw * sS) * LS + 4 * half;
    }
    int boff[WN];
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) boff[wn] = (wave_n * (32 * WN) + 32 * wn + l31) * LS + 4 * half;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    constexpr int WF4 = BN * 4;                  // float4 per weight slab
    constexpr int WLD = (WF4 + 255) / 256;       // float4 per thread
    const long slab = (long)a.CoutPad * CONV_KC;  // floats per (tap, chunk)
    __syncthreads();
    const int ntv = taplist[0];

    // Input staging is software-pipelined over the K chunks: the first NSLOT 16-byte pieces of a thread (all of them for
    // 1x1x1 convs and for the usual 3x3x3 bricks) and the first tap's weight slab of chunk ch+1 are REQUESTED while chunk
    // ch is being multiplied and written to LDS at the next chunk boundary, so a chunk no longer starts with an exposed
    // HBM round trip (it used to: the 1x1(x1) convs -- the decoder's learned shortcuts, two thirds of the ResNet-50
    // embedder -- spent most of their time there).  Piece geometry is decoded once, not per chunk.  All loads are
    // unconditional with clamped addresses (a conditional load forces the compiler to wait for it at once).
    constexpr int NSLOT = 6;
    const int q4 = (tid & 3) * 4;  // channel offset of this thread's pieces inside a chunk (256 % 4 == 0)
    long sl_off[NSLOT]; // element offset of (row, channel 0) in the input tensor, -1: padding / not a piece
    int sl_cb[2];       // coef path: b * CinAct of the first two pieces (1x1x1 bricks have <= 2 per thread)
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
        const int idx = tid + u * 256;
        int p = idx >> 2;
        const int iw = p % HW; p /= HW;
        const int ih = p % HH; p /= HH;
        const int it = p % HT; p /= HT;
        const int b = b0 + p, t = t0 * sT + it - pt, h = h0 * sS + ih - ph, w = w0 * sS + iw - pw;
        const bool ok = idx < NPOS * 4 && b < a.B && (unsigned)t < (unsigned)Tin && (unsigned)h < (unsigned)Hin &&
                        (unsigned)w < (unsigned)Win;
        sl_off[u] = ok ? ((((long)b * Tin + t) * Hin + h) * Win + w) * a.CinAct : -1;
        if (u < 2) sl_cb[u] = ok ? b * a.CinAct : 0;
    }
    float4 pin[NSLOT], pc00, pc01, pc10, pc11, pwr0, pwr1;  // (named, not arrays: arrays defined under a branch go to scratch)
    static_assert(WLD <= 2, "weight pieces per thread");
#define CONV_REQUEST(chn_)                                                                                           \
    {                                                                                                                \
        const int c_ = (chn_) * CONV_KC + q4;                                                                        \
        const bool cok_ = c_ < a.CinAct;                                                                             \
        _Pragma("unroll") for (int u = 0; u < NSLOT; ++u) {                                                          \
            const bool ok_ = sl_off[u] >= 0 && cok_;                                                                 \
            const float4 v_ = *reinterpret_cast<const float4*>(a.in + (ok_ ? sl_off[u] + c_ : 0));                  \
            pin[u] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
        }                                                                                                            \
        {   /* coefficient pairs of the first two pieces (a.coef == null: re-reads the input base, unused) */        \
            const float* cf_ = a.coef ? a.coef : a.in;                                                               \
            const long o0_ = (a.coef && sl_off[0] >= 0 && cok_) ? ((long)sl_cb[0] + c_) * 2 : 0;                     \
            const long o1_ = (a.coef && sl_off[1] >= 0 && cok_) ? ((long)sl_cb[1] + c_) * 2 : 0;                     \
            pc00 = *reinterpret_cast<const float4*>(cf_ + o0_);                                                      \
            pc01 = *reinterpret_cast<const float4*>(cf_ + o0_ + 4);                                                  \
            pc10 = *reinterpret_cast<const float4*>(cf_ + o1_);                                                      \
            pc11 = *reinterpret_cast<const float4*>(cf_ + o1_ + 4);                                                  \
        }                                                                                                            \
        const float* wsrc_ = a.wp + ((long)taplist[1] * a.nchunk + (chn_)) * slab + (long)n0 * CONV_KC;              \
        pwr0 = *reinterpret_cast<const float4*>(wsrc_ + (tid < WF4 ? tid : 0) * 4);                                  \
        pwr1 = *reinterpret_cast<const float4*>(wsrc_ + (WLD > 1 ? tid + 256 : 0) * 4);                              \
    }
    pwr0 = pwr1 = pc00 = pc01 = pc10 = pc11 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) pin[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ntv > 0) CONV_REQUEST(0)  // (taplist[1] is only defined when a tap survives)

    for (int ch = 0; ch < a.nchunk; ++ch) {
        __syncthreads();
        // ---- the input halo brick for channels [16 ch, 16 ch + 16): prefetched pieces, then the rest synchronously
        const int c0 = ch * CONV_KC;
        if (ntv > 0) {
#pragma unroll
            for (int u = 0; u < NSLOT; ++u) {
                const int idx = tid + u * 256;
                float4 v = pin[u];
                if (a.coef && sl_off[u] >= 0 && c0 + q4 < a.CinAct) {
                    // normalisation folded into the load: norm(x)*g + beta == x*A + B per (sample, channel)
                    float4 ab0, ab1;
                    if (u == 0) { ab0 = pc00; ab1 = pc01; }
                    else if (u == 1) { ab0 = pc10; ab1 = pc11; }
                    else {
                        const int b = (int)(sl_off[u] / ((long)a.CinAct * Win * Hin * Tin));
                        ab0 = *reinterpret_cast<const float4*>(a.coef + ((long)b * a.CinAct + c0 + q4) * 2);
                        ab1 = *reinterpret_cast<const float4*>(a.coef + ((long)b * a.CinAct + c0 + q4) * 2 + 4);
                    }
                    v.x = fmaf(v.x, ab0.x, ab0.y); v.y = fmaf(v.y, ab0.z, ab0.w);
                    v.z = fmaf(v.z, ab1.x, ab1.y); v.w = fmaf(v.w, ab1.z, ab1.w);
                }
                if (idx < NPOS * 4) *reinterpret_cast<float4*>(in_lds + (idx >> 2) * LS + q4) = v;
            }
            if (tid < WF4) *reinterpret_cast<float4*>(w_lds + (tid >> 2) * LS + 4 * (tid & 3)) = pwr0;
            if (WLD > 1) *reinterpret_cast<float4*>(w_lds + ((tid + 256) >> 2) * LS + 4 * (tid & 3)) = pwr1;
        }
        for (int idx = tid + NSLOT * 256; idx < NPOS * 4; idx += 256) {  // bricks with a large halo only
            const int q = idx & 3;
            int p = idx >> 2;
            const int iw = p % HW; p /= HW;
            const int ih = p % HH; p /= HH;
            const int it = p % HT; p /= HT;
            const int b = b0 + p, t = t0 * sT + it - pt, h = h0 * sS + ih - ph, w = w0 * sS + iw - pw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = c0 + 4 * q;
            if (b < a.B && (unsigned)t < (unsigned)Tin && (unsigned)h < (unsigned)Hin && (unsigned)w < (unsigned)Win &&
                c < a.CinAct) {
                v = *reinterpret_cast<const float4*>(a.in + ((((long)b * Tin + t) * Hin + h) * Win + w) * a.CinAct + c);
                if (a.coef) {
                    const float4 ab0 = *reinterpret_cast<const float4*>(a.coef + ((long)b * a.CinAct + c) * 2);
                    const float4 ab1 = *reinterpret_cast<const float4*>(a.coef + ((long)b * a.CinAct + c) * 2 + 4);
                    v.x = fmaf(v.x, ab0.x, ab0.y); v.y = fmaf(v.y, ab0.z, ab0.w);
                    v.z = fmaf(v.z, ab1.x, ab1.y); v.w = fmaf(v.w, ab1.z, ab1.w);
                }
            }
            *reinterpret_cast<float4*>(in_lds + (idx >> 2) * LS + 4 * q) = v;
        }
        if (ntv > 0) { const int chn = ch + 1 < a.nchunk ? ch + 1 : ch; CONV_REQUEST(chn) }
        __syncthreads();
        // one tap of one chunk: 2 groups x 4 k-slots, one ds_read_b128 per operand feeds four MFMAs
#define CONV_TAP(tapoff_, wb_)                                                                                       \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                  \
        float4 av[WM], bv[WN];                                                                                       \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm)                                                            \
            av[wm] = *reinterpret_cast<const float4*>(in_lds + aoff[wm] + (tapoff_) + 8 * g);                        \
        _Pragma("unroll") for (int wn = 0; wn < WN; ++wn)                                                            \
            bv[wn] = *reinterpret_cast<const float4*>((wb_) + boff[wn] + 8 * g);                                     \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                              \
            _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) {                                                      \
                const float as = s == 0 ? av[wm].x : s == 1 ? av[wm].y : s == 2 ? av[wm].z : av[wm].w;               \
                _Pragma("unroll") for (int wn = 0; wn < WN; ++wn) {                                                  \
                    const float bs = s == 0 ? bv[wn].x : s == 1 ? bv[wn].y : s == 2 ? bv[wn].z : bv[wn].w;           \
                    acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs, acc[wm][wn], 0, 0, 0);                \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    }
        if (ntv == 1) {
            // 1x1(x1) convs and single surviving taps: the slab is already in buffer 0 and nothing follows it -- no tap
            // prefetch, no trailing barrier (the next chunk's leading barrier orders the LDS reuse); the next chunk's
            // requests issued above overlap these MFMAs
            const int tap = taplist[1];
            const int dw = tap % a.KW, dh = (tap / a.KW) % a.KH, dt = tap / (a.KW * a.KH);
            CONV_TAP(((dt * HH + dh) * HW + dw) * LS, w_lds)
            continue;
        }
        for (int ti = 0; ti < ntv; ++ti) {
            const int tap = taplist[1 + ti];
            // next tap's weight slab: requested before this tap's MFMAs, parked in the other LDS buffer after them
            // (unconditional -- after the last tap the slab is simply re-read -- so that it stays in registers)
            float4 wreg[WLD];
            {
                const int tnext = ti + 1 < ntv ? taplist[2 + ti] : tap;
                const float* src = a.wp + ((long)tnext * a.nchunk + ch) * slab + (long)n0 * CONV_KC;
#pragma unroll
                for (int u = 0; u < WLD; ++u) {
                    const int f = tid + u * 256;
                    wreg[u] = *reinterpret_cast<const float4*>(src + (f < WF4 ? f : 0) * 4);
                }
            }
            const int dw = tap % a.KW, dh = (tap / a.KW) % a.KH, dt = tap / (a.KW * a.KH);
            const int tapoff = ((dt * HH + dh) * HW + dw) * LS;
            const float* wb = w_lds + (ti & 1) * (BN * LS);
            CONV_TAP(tapoff, wb)
            {
                float* wd = w_lds + ((ti + 1) & 1) * (BN * LS);
#pragma unroll
                for (int u = 0; u < WLD; ++u) {
                    const int f = tid + u * 256;
                    if (f < WF4) *reinterpret_cast<float4*>(wd + (f >> 2) * LS + 4 * (f & 3)) = wreg[u];
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int HWo = a.H * a.W;
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int n = n0 + wave_n * (32 * WN) + 32 * wn + l31;
        if (n >= a.Cout) continue;
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int p = rowpos[m];
                if (p < 0) continue;
                float v = acc[wm][wn][r] + bias;
                if (a.res) v += a.res[(long)rowres[m] * a.Cout + n];
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                if (a.epi & EPI_FRAMES) {
                    const int bt_ = p / HWo, hw = p - bt_ * HWo;
                    const int b_ = bt_ / a.T, t_ = bt_ - b_ * a.T;
                    a.out[(long)b_ * a.frames_bstride + ((long)t_ * a.Cout + n) * HWo + hw] = tanhf(v);
                } else if (a.epi & EPI_HL16) {
                    const _Float16 hi = (_Float16)v;
                    char* o = reinterpret_cast<char*>(a.out) + (long)p * a.Cout * 4 + (n >> 3) * 32 + (n & 7) * 2;
                    *reinterpret_cast<_Float16*>(o) = hi;
                    *reinterpret_cast<_Float16*>(o + 16) = (_Float16)(v - (float)hi);
                } else {
                    a.out[(long)p * a.Cout + n] = v;
                }
            }
        }
    }
}

int ConvWeights::pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, int kh, int kw, double scale) {
    Cin = cin; Cout = cout; KT = kt; KH = kh; KW = kw;
    CoutPad = (cout + 31) / 32 * 32;
    if (CoutPad > 64 && CoutPad % 128) CoutPad = (CoutPad + 127) / 128 * 128;
    nchunk = (cin + CONV_KC - 1) / CONV_KC;
    const int ntaps = kt * kh * kw;
    std::vector<float> p((size_t)ntaps * nchunk * CoutPad * CONV_KC, 0.f);
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < ntaps; ++tap) {
                const double v = (double)w_src[((size_t)n * cin + c) * ntaps + tap] * scale;
                p[(((size_t)tap * nchunk + c / CONV_KC) * CoutPad + n) * CONV_KC + c % CONV_KC] = (float)v;
            }
    int rc = w.upload(p.data(), p.size() * 4);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

namespace {

template <int WAVES_M, int WAVES_N, int WM, int WN>
int launch(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    auto kern = conv_mfma_f32_kernel<WAVES_M, WAVES_N, WM, WN>;
    static bool attr_set[I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set)) return rc;
    constexpr int BN = 32 * WN * WAVES_N;
    const long nblk = (long)a.nbB * a.nbT * a.nbH * a.nbW * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 31), I2V_E_INVALID, "conv: grid of %ld workgroups", nblk);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds_bytes, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace

int conv_forward(const ConvWeights& wts, const float* in, int cin_act, float* out, const float* res, int rt, int rs,
                 int B, int T, int H, int W, int epi, hipStream_t st, const float* coef, int stride, int stride_t, long frames_bstride) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "conv: weights not packed");
    I2V_REQUIRE(cin_act % 4 == 0 && cin_act >= wts.Cin, I2V_E_INVALID, "conv: activation channels %d (weights %d)",
                cin_act, wts.Cin);
    I2V_REQUIRE(!coef || (wts.KT == 1 && wts.KH == 1 && wts.KW == 1), I2V_E_INVALID,
                "conv: the on-load affine is only valid without padding (1x1x1 kernels)");
    I2V_REQUIRE(stride == 1 || stride == 2, I2V_E_INVALID, "conv: stride %d", stride);
    I2V_REQUIRE(wts.KT * wts.KH * wts.KW <= 150, I2V_E_INVALID, "conv: kernel too large");
    if (pointwise_supported(wts, res, rt, rs, epi, stride, stride_t))
        return pointwise_forward(wts, in, cin_act, out, res, (long)B * T * H * W, (long)T * H * W, epi, st, coef);
    ConvArgs a{};
    a.coef = coef;
    a.sS = stride;
    a.sT = stride_t;
    I2V_REQUIRE(stride_t == 1 || stride_t == 2, I2V_E_INVALID, "conv: temporal stride %d", stride_t);
    a.in = in; a.wp = wts.w.as<float>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.B = B; a.T = T; a.H = H; a.W = W; a.CinAct = cin_act;
    a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.KT = wts.KT; a.KH = wts.KH; a.KW = wts.KW;
    a.rt = res ? rt : 1; a.rs = res ? rs : 1; a.epi = epi;
    a.frames_bstride = frames_bstride ? frames_bstride : (long)T * wts.Cout * H * W;
    // brick: as cubic as the layer allows (small halo), remaining factor goes to the batch
    int TW = W < 8 ? W : 8, TH = H < 8 ? H : 8;
    int rem = CONV_BM / (TW * TH);
    int TT = T < rem ? T : rem;
    rem /= TT;
    while (rem > 1 && W >= TW * 2) { TW *= 2; rem /= 2; }
    while (rem > 1 && H >= TH * 2) { TH *= 2; rem /= 2; }
    int TB = rem;
    I2V_REQUIRE(TB * TT * TH * TW == CONV_BM && T % TT == 0 && H % TH == 0 && W % TW == 0, I2V_E_INVALID,
                "conv: cannot tile [T=%d,H=%d,W=%d] into bricks of %d positions (dims must be powers of two)", T, H, W,
                CONV_BM);
    // channel tile: the widest that divides CoutPad -- unless that leaves most of the 256 CUs without a workgroup (small
    // feature maps x many channels: the ResNet embedder's and the encoder's late stages, the first decoder levels), where
    // a narrower tile trades re-staged input for 2-4x more workgroups
    int BN = a.CoutPad % 128 == 0 ? 128 : (a.CoutPad % 64 == 0 ? 64 : 32);
    {
        const long bricks = (long)((B + TB - 1) / TB) * (T / TT) * (H / TH) * (W / TW);
        while (BN > 32 && bricks * (a.CoutPad / BN) < 512) BN /= 2;
    }
    const int pos1 = ((TT - 1) * stride_t + a.KT) * ((TH - 1) * stride + a.KH) * ((TW - 1) * stride + a.KW);  // halo rows per sample
    auto lds_of = [&](int tb) {
        return ((size_t)tb * pos1 * CONV_LDS_STRIDE + 2 * (size_t)BN * CONV_LDS_STRIDE) * 4 + (2 * CONV_BM + 160) * 4;
    };
    while (TB > 1 && lds_of(TB) > 160 * 1024) TB /= 2;  // fewer samples per brick: the tile's unused rows are masked
    a.TB = TB; a.TT = TT; a.TH = TH; a.TW = TW;
    a.nbB = (B + TB - 1) / TB; a.nbT = T / TT; a.nbH = H / TH; a.nbW = W / TW;
    const size_t lds = lds_of(TB);
    I2V_REQUIRE(lds <= 160 * 1024, I2V_E_INVALID, "conv: LDS %zu bytes exceeds 160 KiB", lds);
    if (BN == 128) return launch<2, 2, 2, 2>(a, lds, st);
    if (BN == 64) return launch<2, 2, 2, 1>(a, lds, st);
    return launch<4, 1, 1, 1>(a, lds, st);
}

}  // namespace i2v

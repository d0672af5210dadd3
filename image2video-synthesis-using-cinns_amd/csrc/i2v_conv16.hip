// 3x3x3 Conv3d as an implicit GEMM on the gfx950 fp16 matrix cores with fp32-class accuracy ("split-fp16").
//
// gfx950 has no TF32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16/bf16 rate.  This kernel keeps
// the reference's fp32 numerics to ~2^-22 per product while using v_mfma_f32_32x32x16_f16: every fp32 operand x is
// carried as the pair (hi, lo) = (fp16(x), fp16((x - hi) * 2^11)), so x = hi + lo * 2^-11 up to 2^-22 |x|, and
//     x * w  =  hi_x hi_w  +  2^-11 (hi_x lo_w + lo_x hi_w)  +  O(2^-22)
// costs three fp16 MFMAs (fp16 x fp16 products are exact in the fp32 accumulator).  Two accumulators per tile
// (hi*hi, cross terms) are combined in the epilogue.  Effective peak = 2.5 PFLOP/s / 3.
//
// Operand format "hl16" (HBM and LDS): per position, per group of 8 channels: 8 x fp16 hi (16 B) | 8 x fp16 lo (16 B),
// i.e. 4 bytes per element like fp32; one ds_read_b128 yields one MFMA operand (lane (i, kg): row i, k = 8 kg + j).
// Activations are produced in this format by the modulate kernel (the split is done once per element, not per tap);
// weights are split on the host at load time.
//
// Tiling: 512 threads = 8 wavefronts (2 per SIMD) per workgroup, 256 output positions (TB x TT x TH x TW brick) x 128
// output channels, wave tile 64 x 64.  Per 32-channel K chunk the input halo brick is staged once in LDS (rows padded
// 128 -> 144 B: consecutive rows start 4 banks apart, conflict-free ds_read_b128) and reused by all taps; the
// [128][32] weight slab of each tap is double-buffered (global loads issued before the tap's 24 MFMAs per wave,
// written to LDS after them, one barrier per tap).
#include <algorithm>

#include "i2v_conv.h"

namespace i2v {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C16_BM = 256, C16_KC = 32;
constexpr int C16_ROW = 144;  // bytes per staged row: 32 channels x 4 B + 16 B pad

struct Conv16Args {
    const char* in;   // hl16 channels-last [B][T][H][W][Cin]
    const char* wp;   // hl16 weights [tap][chunk][CoutPad][128 B]
    const float* bias;
    const float* res;
    float* out;       // fp32 channels-last [B][T][H][W][Cout]
    int B, T, H, W, Cin, Cout, CoutPad, nchunk;
    int KT, KH, KW, tap_base;
    int TB, TT, TH, TW, nbB, nbT, nbH, nbW;
    int rt, rs, epi;
};

template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(512, 2) void conv_mfma_f16x3_kernel(Conv16Args a) {
    constexpr int C16_BN = 32 * WN * WAVES_N;
    static_assert(32 * WM * WAVES_M == C16_BM && WAVES_M * WAVES_N == 8, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int kg = lane >> 5, l31 = lane & 31;

    const int pt = a.KT / 2, ph = a.KH / 2, pw = a.KW / 2;
    const int HT = a.TT + a.KT - 1, HH = a.TH + a.KH - 1, HW = a.TW + a.KW - 1;
    const int NPOS = a.TB * HT * HH * HW;
    const int ntaps = a.KT * a.KH * a.KW;

    char* in_lds = smem;
    char* w_lds = smem + NPOS * C16_ROW;
    int* rowpos = reinterpret_cast<int*>(w_lds + 2 * C16_BN * C16_ROW);
    int* rowres = rowpos + C16_BM;
    int* taplist = rowres + C16_BM;

    const int nNt = a.CoutPad / C16_BN;
    const int ntile = blockIdx.x % nNt;
    int brick = blockIdx.x / nNt;
    const int bw = brick % a.nbW; brick /= a.nbW;
    const int bh = brick % a.nbH; brick /= a.nbH;
    const int bt = brick % a.nbT; brick /= a.nbT;
    const int b0 = brick * a.TB, t0 = bt * a.TT, h0 = bh * a.TH, w0 = bw * a.TW;
    const int n0 = ntile * C16_BN;

    if (tid < C16_BM) {
        int m = tid;
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        const int b = b0 + m, t = t0 + it, h = h0 + ih, w = w0 + iw;
        const bool ok = b < a.B;
        rowpos[tid] = ok ? ((b * a.T + t) * a.H + h) * a.W + w : -1;
        rowres[tid] = ok ? ((b * (a.T / a.rt) + t / a.rt) * (a.H / a.rs) + h / a.rs) * (a.W / a.rs) + w / a.rs : 0;
    }
    if (tid == 0) {
        int cnt = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int dt = tap / (a.KH * a.KW);
            const int lo = t0 + dt - pt, hi = lo + a.TT - 1;
            if (hi < 0 || lo >= a.T) continue;
            taplist[1 + cnt++] = tap;
        }
        taplist[0] = cnt;
    }

    int aoff[WM], boff[WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        int m = wave_m * (32 * WM) + 32 * wm + l31;
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        aoff[wm] = (((m * HT + it) * HH + ih) * HW + iw) * C16_ROW + kg * 32;
    }
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) boff[wn] = (wave_n * (32 * WN) + 32 * wn + l31) * C16_ROW + kg * 32;

    f32x16 acc_h[WM][WN], acc_x[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_h[wm][wn][r] = 0.f; acc_x[wm][wn][r] = 0.f; }

    constexpr int WF4 = C16_BN * 8;            // 16-byte pieces per weight slab
    constexpr int WLD = (WF4 + 511) / 512;
    const long slab = (long)a.CoutPad * 128;  // bytes per (tap, chunk)
    __syncthreads();
    const int ntv = taplist[0];
    const long in_row = (long)a.Cin * 4;

    for (int ch = 0; ch < a.nchunk; ++ch) {
        __syncthreads();
        const int cgrp0 = ch * 4;  // first 8-channel group of this chunk
        const int ngrp = a.Cin >> 3;
        for (int idx = tid; idx < NPOS * 8; idx += 512) {
            const int q = idx & 7;
            int p = idx >> 3;
            const int iw = p % HW; p /= HW;
            const int ih = p % HH; p /= HH;
            const int it = p % HT; p /= HT;
            const int b = b0 + p, t = t0 + it - pt, h = h0 + ih - ph, w = w0 + iw - pw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < a.B && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W &&
                cgrp0 + (q >> 1) < ngrp) {
                v = *reinterpret_cast<const float4*>(a.in + ((((long)b * a.T + t) * a.H + h) * a.W + w) * in_row +
                                                     (long)ch * 128 + q * 16);
            }
            *reinterpret_cast<float4*>(in_lds + (idx >> 3) * C16_ROW + q * 16) = v;
        }
        if (ntv > 0) {
            const char* src = a.wp + ((long)(a.tap_base + taplist[1]) * a.nchunk + ch) * slab + (long)n0 * 128;
#pragma unroll
            for (int u = 0; u < WLD; ++u) {
                const int f = tid + u * 512;
                if (f < WF4) *reinterpret_cast<float4*>(w_lds + (f >> 3) * C16_ROW + (f & 7) * 16) =
                    *reinterpret_cast<const float4*>(src + (long)f * 16);
            }
        }
        __syncthreads();
        for (int ti = 0; ti < ntv; ++ti) {
            const int tap = taplist[1 + ti];
            float4 wreg[WLD];
            const bool more = ti + 1 < ntv;
            if (more) {
                const char* src = a.wp + ((long)(a.tap_base + taplist[2 + ti]) * a.nchunk + ch) * slab + (long)n0 * 128;
#pragma unroll
                for (int u = 0; u < WLD; ++u) {
                    const int f = tid + u * 512;
                    if (f < WF4) wreg[u] = *reinterpret_cast<const float4*>(src + (long)f * 16);
                }
            }
            const int dw = tap % a.KW, dh = (tap / a.KW) % a.KH, dt = tap / (a.KW * a.KH);
            const int tapoff = ((dt * HH + dh) * HW + dw) * C16_ROW;
            const char* wb = w_lds + (ti & 1) * (C16_BN * C16_ROW);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 ah[WM], al[WM], bh_[WN], bl[WN];
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) {
                    const char* p = in_lds + aoff[wm] + tapoff + s * 64;
                    ah[wm] = *reinterpret_cast<const half8*>(p);
                    al[wm] = *reinterpret_cast<const half8*>(p + 16);
                }
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) {
                    const char* p = wb + boff[wn] + s * 64;
                    bh_[wn] = *reinterpret_cast<const half8*>(p);
                    bl[wn] = *reinterpret_cast<const half8*>(p + 16);
                }
#pragma unroll
                for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) {
                        acc_h[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[wm], bh_[wn], acc_h[wm][wn], 0, 0, 0);
                        acc_x[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[wm], bl[wn], acc_x[wm][wn], 0, 0, 0);
                        acc_x[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[wm], bh_[wn], acc_x[wm][wn], 0, 0, 0);
                    }
            }
            if (more) {
                char* wd = w_lds + ((ti + 1) & 1) * (C16_BN * C16_ROW);
#pragma unroll
                for (int u = 0; u < WLD; ++u) {
                    const int f = tid + u * 512;
                    if (f < WF4) *reinterpret_cast<float4*>(wd + (f >> 3) * C16_ROW + (f & 7) * 16) = wreg[u];
                }
            }
            __syncthreads();
        }
    }

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
                const int m = wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int p = rowpos[m];
                if (p < 0) continue;
                float v = fmaf(acc_x[wm][wn][r], 1.0f / 2048.0f, acc_h[wm][wn][r]) + bias;
                if (a.res) v += a.res[(long)rowres[m] * a.Cout + n];
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                if (a.epi & EPI_FRAMES) {
                    const int bt_ = p / HWo, hw = p - bt_ * HWo;
                    a.out[((long)bt_ * a.Cout + n) * HWo + hw] = tanhf(v);
                } else {
                    a.out[(long)p * a.Cout + n] = v;
                }
            }
        }
    }
}

int Conv16Weights::pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, int kh, int kw, double scale) {
    Cin = cin; Cout = cout; KT = kt; KH = kh; KW = kw;
    CoutPad = (cout + 31) / 32 * 32;
    if (CoutPad > 64 && CoutPad % 128) CoutPad = (CoutPad + 127) / 128 * 128;
    nchunk = (cin + C16_KC - 1) / C16_KC;
    const int ntaps = kt * kh * kw;
    std::vector<_Float16> p((size_t)ntaps * nchunk * CoutPad * 64, (_Float16)0.f);
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < ntaps; ++tap) {
                const float v = (float)((double)w_src[((size_t)n * cin + c) * ntaps + tap] * scale);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                const int chunk = c / C16_KC, g = (c % C16_KC) / 8, j = c % 8;
                _Float16* row = &p[(((size_t)tap * nchunk + chunk) * CoutPad + n) * 64];
                row[g * 16 + j] = hi;
                row[g * 16 + 8 + j] = lo;
            }
    int rc = w.upload(p.data(), p.size() * 2);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN>
static int launch16(const Conv16Args& a, unsigned nblk, size_t lds, hipStream_t st) {
    auto kern = conv_mfma_f16x3_kernel<WAVES_M, WAVES_N, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        I2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int conv16_forward(const Conv16Weights& wts, const void* in_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                   int H, int W, int epi, hipStream_t st) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "conv16: weights not packed");
    I2V_REQUIRE(wts.Cin % 8 == 0, I2V_E_INVALID, "conv16: Cin %d must be a multiple of 8", wts.Cin);
    Conv16Args a{};
    a.in = static_cast<const char*>(in_hl16); a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.B = B; a.T = T; a.H = H; a.W = W; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.KT = wts.KT; a.KH = wts.KH; a.KW = wts.KW; a.tap_base = 0;
    if (T == 1 && wts.KT == 3) {  // a single frame only ever meets the centre time-slice of the kernel (rest is padding)
        a.KT = 1;
        a.tap_base = wts.KH * wts.KW;
    }
    a.rt = res ? rt : 1; a.rs = res ? rs : 1; a.epi = epi;
    int TW = W < 8 ? W : 8, TH = H < 8 ? H : 8;
    int rem = C16_BM / (TW * TH);
    int TT = T < rem ? T : rem;
    rem /= TT;
    while (rem > 1 && W >= TW * 2) { TW *= 2; rem /= 2; }
    while (rem > 1 && H >= TH * 2) { TH *= 2; rem /= 2; }
    const int TB = rem;
    I2V_REQUIRE(TB * TT * TH * TW == C16_BM && T % TT == 0 && H % TH == 0 && W % TW == 0, I2V_E_INVALID,
                "conv16: cannot tile [T=%d,H=%d,W=%d] into bricks of %d positions", T, H, W, C16_BM);
    a.TB = TB; a.TT = TT; a.TH = TH; a.TW = TW;
    a.nbB = (B + TB - 1) / TB; a.nbT = T / TT; a.nbH = H / TH; a.nbW = W / TW;
    const int npos = TB * (TT + a.KT - 1) * (TH + a.KH - 1) * (TW + a.KW - 1);
    const int BN = a.CoutPad % 128 == 0 ? 128 : (a.CoutPad % 64 == 0 ? 64 : 32);
    const size_t lds = (size_t)npos * C16_ROW + 2 * (size_t)BN * C16_ROW + (2 * C16_BM + 32) * 4;
    I2V_REQUIRE(lds <= 160 * 1024, I2V_E_INVALID, "conv16: LDS %zu bytes exceeds 160 KiB", lds);
    const long nblk = (long)a.nbB * a.nbT * a.nbH * a.nbW * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 31), I2V_E_INVALID, "conv16: grid of %ld workgroups", nblk);
    if (BN == 128) return launch16<4, 2, 2, 2>(a, (unsigned)nblk, lds, st);
    if (BN == 64) return launch16<4, 2, 2, 1>(a, (unsigned)nblk, lds, st);
    return launch16<8, 1, 1, 1>(a, (unsigned)nblk, lds, st);
}

}  // namespace i2v

// 1x1(x1) convolutions / Linear layers as a plain GEMM on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Out[M = positions][N = Cout] = In[M][K = Cin] * W[K][N] (+ bias, + residual, optional leaky_relu), with the optional
// per-(sample, channel) affine x*A + B folded into the loads (Norm3D in front of the decoder's learned shortcut,
// decoder.py:44-49).  These are the decoder's shortcut convs, two thirds of the ResNet-50 embedder's convs (AE.py:109,
// torchvision Bottleneck conv1/conv3/downsample) and every nn.Linear that goes through conv_forward.
//
// The general implicit-GEMM kernel (i2v_conv.hip) pays three barriers and a re-staging round per 16-channel chunk, which
// is amortised over 27 taps for a 3x3x3 conv but not over the single tap of a 1x1 conv (measured: 36 % of the fp32 MFMA
// peak vs 72 %).  Here one pipeline stage is 32 channels of both operands, double-buffered in LDS with ONE barrier per
// stage: the next stage's pieces are requested (unconditional, clamped loads -> they stay in registers and the vmcnt
// queue is exact) before the current stage's 64 MFMAs per wave and written to the other buffer after them.
// LDS rows are 32 floats + 4 pad (144 B): the 16 rows a ds_read_b128 lane group touches fall on 16 distinct bank quads.
// 256 threads = 4 wavefronts, wave tile 32*WM x 32*WN.
#include <cmath>

#include "i2v_conv.h"

namespace i2v {

typedef float pw_f32x16 __attribute__((ext_vector_type(16)));

constexpr int PW_KC = 32;   // channels per stage
constexpr int PW_LS = 36;   // floats per LDS row

struct PwArgs {
    const float* in;    // [M][CinAct]
    const float* wp;    // ConvWeights layout for one tap: [chunk16][CoutPad][16]
    const float* bias;
    const float* res;   // [M][Cout] or null
    const float* coef;  // [B][CinAct][2] or null
    float* out;         // [M][Cout]
    long M;
    long P;             // positions per sample (coef index b = row / P)
    int CinAct, Cin, Cout, CoutPad, nchunk16;
    int epi;
};

template <int WAVES_M, int WAVES_N, int WM, int WN, bool HAS_COEF>
__global__ __launch_bounds__(256, 2) void pw_mfma_f32_kernel(PwArgs a) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int APT = BM * 8 / 256, BPT = BN * 8 / 256;  // 16-byte pieces per thread and stage (A rows, W rows)
    static_assert(APT >= 1 && BPT >= 1, "tile too small for 256 threads");
    extern __shared__ __attribute__((aligned(16))) float pw_smem[];
    float* a_lds = pw_smem;                       // [2][BM][PW_LS]
    float* b_lds = pw_smem + 2 * BM * PW_LS;      // [2][BN][PW_LS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int half = lane >> 5, l31 = lane & 31;
    const int nNt = a.CoutPad / BN;
    const long m0 = (long)(blockIdx.x / nNt) * BM;
    const int n0 = (blockIdx.x % nNt) * BN;
    const int q4 = (tid & 7) * 4;  // channel offset of this thread's pieces inside a stage (256 % 8 == 0)

    // piece geometry (constant over the stages)
    long arow[APT];   // element offset of the A row, -1 beyond M
    int acb[APT];     // coef path: sample index * CinAct
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const long m = m0 + (tid >> 3) + u * 32;
        arow[u] = m < a.M ? m * a.CinAct : -1;
        acb[u] = HAS_COEF ? (int)((m < a.M ? m : 0) / a.P) * a.CinAct : 0;
    }
    const int nstage = (a.Cin + PW_KC - 1) / PW_KC;

    float4 pa[APT], pb[BPT], pc0[HAS_COEF ? APT : 1], pc1[HAS_COEF ? APT : 1];
#define PW_REQUEST(st_)                                                                                              \
    {                                                                                                                \
        const int c_ = (st_) * PW_KC + q4;                                                                           \
        const bool cok_ = c_ < a.CinAct;                                                                             \
        _Pragma("unroll") for (int u = 0; u < APT; ++u) {                                                            \
            const bool ok_ = arow[u] >= 0 && cok_;                                                                   \
            const float4 v_ = *reinterpret_cast<const float4*>(a.in + (ok_ ? arow[u] + c_ : 0));                     \
            pa[u] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                      \
            if (HAS_COEF) {                                                                                          \
                const long o_ = ok_ ? ((long)acb[u] + c_) * 2 : 0;                                                   \
                pc0[u] = *reinterpret_cast<const float4*>(a.coef + o_);                                              \
                pc1[u] = *reinterpret_cast<const float4*>(a.coef + o_ + 4);                                          \
            }                                                                                                        \
        }                                                                                                            \
        const int ck_ = (st_) * 2 + (q4 >> 4);  /* 16-channel chunk of the packed weights */                         \
        const bool wok_ = ck_ < a.nchunk16;                                                                          \
        _Pragma("unroll") for (int u = 0; u < BPT; ++u) {                                                            \
            const int n_ = n0 + (tid >> 3) + u * 32;                                                                 \
            const float4 v_ = *reinterpret_cast<const float4*>(                                                      \
                a.wp + (wok_ ? ((long)ck_ * a.CoutPad + n_) * 16 + (q4 & 15) : 0));                                  \
            pb[u] = wok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
        }                                                                                                            \
    }

    int aoff[WM], boff[WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) aoff[wm] = (wave_m * (32 * WM) + 32 * wm + l31) * PW_LS + 4 * half;
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) boff[wn] = (wave_n * (32 * WN) + 32 * wn + l31) * PW_LS + 4 * half;

    pw_f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    PW_REQUEST(0)
    for (int st = 0; st < nstage; ++st) {
        float* ab = a_lds + (st & 1) * (BM * PW_LS);
        float* bb = b_lds + (st & 1) * (BN * PW_LS);
        // park the requested pieces (the buffer's last readers finished before the previous stage's barrier)
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            float4 v = pa[u];
            if (HAS_COEF && arow[u] >= 0 && st * PW_KC + q4 < a.CinAct) {  // norm(x)*g + beta == x*A + B per (sample, channel)
                v.x = fmaf(v.x, pc0[u].x, pc0[u].y); v.y = fmaf(v.y, pc0[u].z, pc0[u].w);
                v.z = fmaf(v.z, pc1[u].x, pc1[u].y); v.w = fmaf(v.w, pc1[u].z, pc1[u].w);
            }
            *reinterpret_cast<float4*>(ab + ((tid >> 3) + u * 32) * PW_LS + q4) = v;
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) *reinterpret_cast<float4*>(bb + ((tid >> 3) + u * 32) * PW_LS + q4) = pb[u];
        { const int sn = st + 1 < nstage ? st + 1 : st; PW_REQUEST(sn) }
        __syncthreads();
        // MFMA k-slot (s, half) of group g maps to channel 8g + 4*half + s: one ds_read_b128 per operand feeds four MFMAs
#pragma unroll
        for (int g = 0; g < PW_KC / 8; ++g) {
            float4 av[WM], bv[WN];
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) av[wm] = *reinterpret_cast<const float4*>(ab + aoff[wm] + 8 * g);
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) bv[wn] = *reinterpret_cast<const float4*>(bb + boff[wn] + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int wm = 0; wm < WM; ++wm) {
                    const float as = s == 0 ? av[wm].x : s == 1 ? av[wm].y : s == 2 ? av[wm].z : av[wm].w;
#pragma unroll
                    for (int wn = 0; wn < WN; ++wn) {
                        const float bs = s == 0 ? bv[wn].x : s == 1 ? bv[wn].y : s == 2 ? bv[wn].z : bv[wn].w;
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs, acc[wm][wn], 0, 0, 0);
                    }
                }
            }
        }
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int n = n0 + wave_n * (32 * WN) + 32 * wn + l31;
        if (n >= a.Cout) continue;
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= a.M) continue;
                float v = acc[wm][wn][r] + bias;
                if (a.res) v += a.res[m * a.Cout + n];
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                a.out[m * a.Cout + n] = v;
            }
        }
    }
}

namespace {

template <int WAVES_M, int WAVES_N, int WM, int WN>
int pw_launch(const PwArgs& a, hipStream_t st) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    const size_t lds = (size_t)2 * (BM + BN) * PW_LS * 4;
    const long nblk = (a.M + BM - 1) / BM * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 31), I2V_E_INVALID, "pointwise conv: grid of %ld workgroups", nblk);
    if (a.coef) {
        auto kern = pw_mfma_f32_kernel<WAVES_M, WAVES_N, WM, WN, true>;
        static bool attr_set[I2V_MAX_DEV] = {};
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 96 * 1024, attr_set)) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
    } else {
        auto kern = pw_mfma_f32_kernel<WAVES_M, WAVES_N, WM, WN, false>;
        static bool attr_set[I2V_MAX_DEV] = {};
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 96 * 1024, attr_set)) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
    }
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace

// ---- split-fp16 variant (decoder shortcut convs in split-fp16 mode) ---------------------------------------------------
//
// Same GEMM, operands as fp16 (hi, lo) pairs and three v_mfma_f32_32x32x16_f16 per product (hi*hi + hi*lo + lo*hi, relative
// error ~2^-22 like the 3x3x3 convs of i2v_conv16.hip): 16x the fp32 MFMA rate, so the layer becomes HBM-bound.  The
// activations are split ONCE, by the thread that parks them in LDS (the per-(sample, channel) affine is folded in first);
// the weights arrive pre-split (Conv16Weights with one tap).  LDS rows: 32 channels = 4 groups x (8 hi | 8 lo) = 128 B + 16 pad
// (the 16 rows of a ds_read_b128 lane group fall on 16 distinct bank quads).
typedef _Float16 pw_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 pw_half4 __attribute__((ext_vector_type(4)));
typedef float pw_f32x4 __attribute__((ext_vector_type(4)));

constexpr int PW16_ROW = 144;  // bytes per LDS row

struct Pw16Args {
    const float* in;    // [M][Cin] fp32
    const char* wp;     // Conv16Weights layout, one tap: [chunk32][CoutPad][128 B]
    const float* bias;
    const float* res;   // [M][Cout] or null
    const float* coef;  // [B][Cin][2] or null
    float* out;         // [M][Cout]
    int* range_flag;    // optional: set when an activation leaves the fp16 range
    long M, P;
    int Cin, Cout, CoutPad, nchunk;
    int epi;
    float oscale;
};

// TRANS: the output is written transposed, out[n][m] (row stride M): the MFMA operands are swapped, so a lane holds one
// position and its registers the output channels, and every store instruction writes 128 contiguous bytes per half-wave
// (conv_img's 81-column GEMM, whose gather pass then reads coalesced planes).  No residual in this mode.
template <int WAVES_M, int WAVES_N, int WM, int WN, bool HAS_COEF, bool TRANS = false>
__global__ __launch_bounds__(256, 2) void pw_mfma_f16x3_kernel(Pw16Args a) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int APT = BM * 8 / 256, BPT = BN * 8 / 256;  // 16-byte pieces per thread and stage (A: 4 fp32 channels; W: 16 B)
    static_assert(APT >= 1 && BPT >= 1, "tile too small for 256 threads");
    extern __shared__ __attribute__((aligned(16))) char pw16_smem[];
    char* a_lds = pw16_smem;                          // [2][BM][PW16_ROW]
    char* b_lds = pw16_smem + 2 * BM * PW16_ROW;      // [2][BN][PW16_ROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int kg = lane >> 5, l31 = lane & 31;
    const int nNt = a.CoutPad / BN;
    const long m0 = (long)(blockIdx.x / nNt) * BM;
    const int n0 = (blockIdx.x % nNt) * BN;
    const int q4 = (tid & 7) * 4;  // channel offset of this thread's A pieces inside a stage

    const float* arow[APT];   // the A row (clamped; rows beyond M are computed and never stored)
    int acb[APT];             // coef path: sample index * Cin
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        long m = m0 + (tid >> 3) + u * 32;
        m = m < a.M ? m : a.M - 1;
        arow[u] = a.in + m * a.Cin;
        acb[u] = HAS_COEF ? (int)(m / a.P) * a.Cin : 0;
    }
    const int nstage = a.nchunk;

    pw_f32x4 pa[APT], pb[BPT], pc0[HAS_COEF ? APT : 1], pc1[HAS_COEF ? APT : 1];
#define PW16_REQUEST(st_)                                                                                            \
    {                                                                                                                \
        int c_ = (st_) * 32 + q4;                                                                                    \
        const bool cok_ = c_ < a.Cin;     /* (Cin % 4 == 0) channels beyond Cin: zero weights, finite activations */ \
        c_ = cok_ ? c_ : 0;                                                                                          \
        _Pragma("unroll") for (int u = 0; u < APT; ++u) {                                                            \
            pa[u] = *reinterpret_cast<const pw_f32x4*>(arow[u] + c_);                                                \
            if (HAS_COEF) {                                                                                          \
                const float* cp_ = a.coef + ((long)acb[u] + c_) * 2;                                                 \
                pc0[u] = *reinterpret_cast<const pw_f32x4*>(cp_);                                                    \
                pc1[u] = *reinterpret_cast<const pw_f32x4*>(cp_ + 4);                                                \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < BPT; ++u) {                                                            \
            const int n_ = n0 + (tid >> 3) + u * 32;                                                                 \
            pb[u] = *reinterpret_cast<const pw_f32x4*>(a.wp + ((long)(st_) * a.CoutPad + n_) * 128 + (tid & 7) * 16); \
        }                                                                                                            \
    }

    int aoff[WM], boff[WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) aoff[wm] = (wave_m * (32 * WM) + 32 * wm + l31) * PW16_ROW + kg * 32;
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) boff[wn] = (wave_n * (32 * WN) + 32 * wn + l31) * PW16_ROW + kg * 32;

    pw_f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    bool bad = false;
    PW16_REQUEST(0)
    for (int st = 0; st < nstage; ++st) {
        char* ab = a_lds + (st & 1) * (BM * PW16_ROW);
        char* bb = b_lds + (st & 1) * (BN * PW16_ROW);
        // park the requested pieces (the buffer's last readers finished before the previous stage's barrier): the four
        // channels q4 .. q4 + 3 of a row go to halves (q4 & 7) .. + 3 of group q4 >> 3, hi part and lo part 16 bytes apart
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            pw_f32x4 v = pa[u];
            if (HAS_COEF) {  // norm(x)*g + beta == x*A + B per (sample, channel)
                v[0] = fmaf(v[0], pc0[u][0], pc0[u][1]); v[1] = fmaf(v[1], pc0[u][2], pc0[u][3]);
                v[2] = fmaf(v[2], pc1[u][0], pc1[u][1]); v[3] = fmaf(v[3], pc1[u][2], pc1[u][3]);
            }
            pw_half4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = (_Float16)v[j];
                lo[j] = (_Float16)(v[j] - (float)hi[j]);
                bad |= !(fabsf(v[j]) <= 65504.f);
            }
            char* dst = ab + ((tid >> 3) + u * 32) * PW16_ROW + (q4 >> 3) * 32 + (q4 & 7) * 2;
            *reinterpret_cast<pw_half4*>(dst) = hi;
            *reinterpret_cast<pw_half4*>(dst + 16) = lo;
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u)
            *reinterpret_cast<pw_f32x4*>(bb + ((tid >> 3) + u * 32) * PW16_ROW + (tid & 7) * 16) = pb[u];
        { const int sn = st + 1 < nstage ? st + 1 : st; PW16_REQUEST(sn) }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {   // two 16-channel MFMA steps per stage: lane group kg reads channel group 2 s + kg
            pw_half8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) {
                ah[wm] = *reinterpret_cast<const pw_half8*>(ab + aoff[wm] + s * 64);
                al[wm] = *reinterpret_cast<const pw_half8*>(ab + aoff[wm] + s * 64 + 16);
            }
#pragma unroll
            for (int wn = 0; wn < WN; ++wn) {
                bh[wn] = *reinterpret_cast<const pw_half8*>(bb + boff[wn] + s * 64);
                bl[wn] = *reinterpret_cast<const pw_half8*>(bb + boff[wn] + s * 64 + 16);
            }
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int wn = 0; wn < WN; ++wn) {
                    if (TRANS) {
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[wn], ah[wm], acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[wn], ah[wm], acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[wn], al[wm], acc[wm][wn], 0, 0, 0);
                    } else {
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[wm], bh[wn], acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[wm], bl[wn], acc[wm][wn], 0, 0, 0);
                        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[wm], bh[wn], acc[wm][wn], 0, 0, 0);
                    }
                }
        }
    }
    if (bad && a.range_flag) atomicOr(a.range_flag, 1);

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (TRANS) {
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
            const long m = m0 + wave_m * (32 * WM) + 32 * wm + l31;
#pragma unroll
            for (int wn = 0; wn < WN; ++wn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wave_n * (32 * WN) + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float v = fmaf(acc[wm][wn][r], a.oscale, (a.bias && n < a.Cout) ? a.bias[n] : 0.f);
                    if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                    if (n < a.Cout && m < a.M) a.out[(long)n * a.M + m] = v;
                }
        }
        return;
    }
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int n = n0 + wave_n * (32 * WN) + 32 * wn + l31;
        const bool ncol = n < a.Cout;
        const float bias = (a.bias && ncol) ? a.bias[n] : 0.f;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
            float rv[16];   // residual values first, all in flight together (see i2v_conv16.hip)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg;
                rv[r] = (a.res && ncol && m < a.M) ? a.res[m * a.Cout + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg;
                float v = fmaf(acc[wm][wn][r], a.oscale, bias) + rv[r];
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                if (ncol && m < a.M) a.out[m * a.Cout + n] = v;
            }
        }
    }
}

namespace {

template <int WAVES_M, int WAVES_N, int WM, int WN, bool HAS_COEF, bool TRANS>
int pw16_launch1(const Pw16Args& a, hipStream_t st) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    const size_t lds = (size_t)2 * (BM + BN) * PW16_ROW;
    const long nblk = (a.M + BM - 1) / BM * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 31), I2V_E_INVALID, "pointwise conv: grid of %ld workgroups", nblk);
    auto kern = pw_mfma_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, HAS_COEF, TRANS>;
    static bool attr_set[I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 96 * 1024, attr_set)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN>
int pw16_launch(const Pw16Args& a, hipStream_t st) {
    return a.coef ? pw16_launch1<WAVES_M, WAVES_N, WM, WN, true, false>(a, st)
                  : pw16_launch1<WAVES_M, WAVES_N, WM, WN, false, false>(a, st);
}

}  // namespace

int pointwise16_forward(const Conv16Weights& wts, const float* in, float* out, const float* res, long M, long P, int epi,
                        hipStream_t st, const float* coef, int* range_flag, bool transposed) {
    I2V_REQUIRE(wts.w.p && wts.KT == 1 && wts.KH == 1 && wts.KW == 1 && !wts.tdup, I2V_E_STATE, "pointwise16: needs 1x1x1 split-fp16 weights");
    I2V_REQUIRE(wts.Cin % 4 == 0 && (epi & ~EPI_LRELU) == 0 && M > 0, I2V_E_INVALID, "pointwise16: Cin %d / epilogue %d", wts.Cin, epi);
    Pw16Args a{};
    a.in = in; a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.coef = coef; a.out = out;
    a.range_flag = range_flag;
    a.M = M; a.P = P; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk; a.epi = epi;
    a.oscale = (float)std::ldexp(1.0, -wts.wexp);
    if (transposed) {
        I2V_REQUIRE(!res && !coef && a.CoutPad % 32 == 0, I2V_E_INVALID, "pointwise16: transposed output takes no residual / affine");
        I2V_REQUIRE(a.CoutPad % 128 == 0, I2V_E_INVALID, "pointwise16: transposed output needs CoutPad %% 128 == 0 (have %d)", a.CoutPad);
        return pw16_launch1<2, 2, 2, 2, false, true>(a, st);                     // 128 x 128: the rows are read once
    }
    int BN = a.CoutPad % 128 == 0 ? 128 : (a.CoutPad % 64 == 0 ? 64 : 32);
    int BM = 128;
    auto blocks = [&](int bm, int bn) { return (M + bm - 1) / bm * (a.CoutPad / bn); };
    while (BN > 32 && blocks(BM, BN) < 512) BN /= 2;
    if (BN == 64 && blocks(BM, BN) < 512) BM = 64;
    if (BN == 128) return pw16_launch<2, 2, 2, 2>(a, st);                        // 128 x 128
    if (BN == 64) return BM == 128 ? pw16_launch<2, 2, 2, 1>(a, st)              // 128 x 64
                                   : pw16_launch<2, 2, 1, 1>(a, st);             //  64 x 64
    return pw16_launch<4, 1, 1, 1>(a, st);                                       // 128 x 32
}

bool pointwise_supported(const ConvWeights& wts, const float* res, int rt, int rs, int epi, int stride, int stride_t) {
    return wts.KT == 1 && wts.KH == 1 && wts.KW == 1 && stride == 1 && stride_t == 1 && (epi & ~EPI_LRELU) == 0 &&
           (!res || (rt == 1 && rs == 1));
}

int pointwise_forward(const ConvWeights& wts, const float* in, int cin_act, float* out, const float* res, long M, long P,
                      int epi, hipStream_t st, const float* coef) {
    PwArgs a{};
    a.in = in; a.wp = wts.w.as<float>(); a.bias = wts.bias.as<float>(); a.res = res; a.coef = coef; a.out = out;
    a.M = M; a.P = P; a.CinAct = cin_act; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk16 = wts.nchunk;
    a.epi = epi;
    // tile: as wide as CoutPad allows, narrowed / shortened while most of the 256 CUs would stay without a workgroup
    int BN = a.CoutPad % 128 == 0 ? 128 : (a.CoutPad % 64 == 0 ? 64 : 32);
    int BM = 128;
    auto blocks = [&](int bm, int bn) { return (M + bm - 1) / bm * (a.CoutPad / bn); };
    while (BN > 32 && blocks(BM, BN) < 512) BN /= 2;
    if (BN == 64 && blocks(BM, BN) < 512) BM = 64;
    if (BN == 128) return pw_launch<2, 2, 2, 2>(a, st);                        // 128 x 128
    if (BN == 64) return BM == 128 ? pw_launch<2, 2, 2, 1>(a, st)              // 128 x 64
                                   : pw_launch<2, 2, 1, 1>(a, st);             //  64 x 64
    return pw_launch<4, 1, 1, 1>(a, st);                                       // 128 x 32
}

}  // namespace i2v

// Stage-1 VAE decoder on gfx950: Generator.forward (reference: stage1_VAE/modules/decoder.py:97-120).
//
// All activations live channels-last ([B][T][H][W][C] fp32) in the caller's workspace.  Per GeneratorBlock
// (decoder.py:33-52) the launch sequence is
//   stats(x)                       per-(b,c) sum / sum-of-squares (fp64 accumulation, coalesced float4 rows)
//   coef                           GroupNorm / InstanceNorm statistics folded with the ADAIN / affine parameters
//                                  into one (A, B) pair per (b,c): norm(x)*g + beta == x*A + B
//   resize + conv2d + conv2d       SPADE branch on the start frame (normalization_layer.py:20-23); gamma and beta
//                                  come out of ONE 128 -> 2C implicit-GEMM conv, "+1" folded into the gamma bias
//   modulate                       a0 = lrelu((x_up*A+B)*gamma' + beta): nearest upsample folded into the read index,
//                                  gamma/beta broadcast over T instead of repeat_interleave'd (:22-23)
//   conv3d 3x3x3 (MFMA)            dx = conv_0(a0)
//   stats + coef + modulate        a1 = lrelu(ADAIN(dx, z))
//   [coef + modulate + conv 1x1x1] learned shortcut, evaluated at the LOW resolution (1x1x1 conv and GroupNorm
//                                  statistics commute with nearest upsampling)
//   conv3d 3x3x3 (MFMA)            out = conv_1(a1) + shortcut (residual read through the upsample index map)
// Spectral norm (W / sigma, signed sigma) is folded once at load time (decoder.py:20-25 recomputes it per call).
#include <algorithm>
#include <cmath>
#include <memory>

#include "i2v_conv.h"

namespace i2v {

// ------------------------------------------------------------------------------------------------ statistics
// x [B][P][C] -> sums[b][c] = (sum, sumsq) in fp64.  grid (chunks, B), block 256 = R rows x C4 float4 columns.
// Channel counts above 1024 are covered by blockIdx.z slices of 1024 channels (Ctot = row stride, C = slice width).
__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ x, double* __restrict__ sums, int P, int C,
                                                    int rows_per_block, int Ctot) {
    __shared__ double red[256][8];
    x += (long)blockIdx.z * 1024;
    sums += (long)blockIdx.z * 2048;
    const int C4 = C >> 2;
    const int tid = threadIdx.x;
    const int R = 256 / C4;            // rows handled concurrently (C4 <= 256)
    const int col = tid % C4, r = tid / C4;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * rows_per_block;
    const int p1 = min(P, p0 + rows_per_block);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (r < R) {
        const float* base = x + (long)b * P * Ctot + 4 * col;
        for (int p = p0 + r; p < p1; p += R) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)p * Ctot);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid][j] = s[j]; red[tid][4 + j] = q[j]; }
    __syncthreads();
    if (r == 0) {
        for (int rr = 1; rr < R; ++rr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += red[rr * C4 + col][j]; q[j] += red[rr * C4 + col][4 + j]; }
        }
        double* dst = sums + ((long)b * Ctot + 4 * col) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(dst + 2 * j, s[j]);
            atomicAdd(dst + 2 * j + 1, q[j]);
        }
    }
}

// (sum, sumsq) -> per-(b,c) affine (A, B) with norm(x)*gamma + beta == x*A + B.
//   groups: number of normalisation groups (C for instance norm); count = elements per channel (T*H*W)
//   gamma/beta sources: zl != null: ADAIN, gamma = zl[b][zoff + c], beta = zl[b][zoff + C + c] (normalization_layer.py:49-50)
//                       gw != null: GroupNorm affine weight/bias per channel (normalization_layer.py:31)
//                       neither: plain normalisation (Spade's GroupNorm(affine=False), :11)
__global__ void coef_kernel(const double* __restrict__ sums, float2* __restrict__ coef, int C, int groups, double count,
                            const float* __restrict__ zl, int zstride, int zoff, const float* __restrict__ gw,
                            const float* __restrict__ gb) {
    // The sample's C (sum, sumsq) pairs are staged in LDS (one memory round trip instead of a chain of dependent ones), the
    // per-GROUP totals are formed once per group (not once per channel of the group), in the same summation order.
    __shared__ double ss[1024], qq[1024], gsum[512], gsq[512];
    const int b = blockIdx.x;
    const int cpg = C / groups;
    const bool staged = C <= 1024;
    if (staged) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const double2 v = *reinterpret_cast<const double2*>(sums + ((long)b * C + c) * 2);
            ss[c] = v.x; qq[c] = v.y;
        }
        __syncthreads();
        if (cpg > 1) {   // (then groups <= 512)
            for (int g = threadIdx.x; g < groups; g += blockDim.x) {
                double s = 0, q = 0;
                for (int j = 0; j < cpg; ++j) { s += ss[g * cpg + j]; q += qq[g * cpg + j]; }
                gsum[g] = s; gsq[g] = q;
            }
            __syncthreads();
        }
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0, q = 0;
        if (staged) {
            s = cpg > 1 ? gsum[c / cpg] : ss[c];
            q = cpg > 1 ? gsq[c / cpg] : qq[c];
        } else {
            const int g0 = (c / cpg) * cpg;
            for (int j = 0; j < cpg; ++j) {
                s += sums[((long)b * C + g0 + j) * 2];
                q += sums[((long)b * C + g0 + j) * 2 + 1];
            }
        }
        const double n = count * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;  // biased variance, as F.group_norm / F.instance_norm
        var = var > 0 ? var : 0;
        const double rstd = 1.0 / sqrt(var + 1e-5);
        double gamma = 1.0, beta = 0.0;
        if (zl) { gamma = zl[(long)b * zstride + zoff + c]; beta = zl[(long)b * zstride + zoff + C + c]; }
        else if (gw) { gamma = gw[c]; beta = gb[c]; }
        coef[(long)b * C + c] = make_float2((float)(gamma * rstd), (float)(beta - gamma * mean * rstd));
    }
}

// out[b][t][h][w][c] = act( (x[b][t/ut][h/us][w/us][c] * A + B) * gamma'[b][h][w][c] + beta[b][h][w][c] )
//   gb: [B][H][W][2C] (gamma' = 1 + gamma in [0,C), beta in [C,2C)) or null.
// One thread = one (h, w) position x 8 channels (consecutive threads = consecutive channel groups), looping over the
// frames: the per-(sample, channel) coefficients and the SPADE gamma/beta of the position -- neither depends on t -- are
// loaded once and reused for all T frames, the source row once per `ut` frames; per frame 32 B are stored.
// blockIdx.y = sample, all per-sample index math in 32 bits.  HL16: write the split-fp16 operand format of
// i2v_conv16.hip (8 x fp16 hi | 8 x fp16 lo per 8 channels, lo = x - hi) instead of fp32.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// Underflow side of the range guard.  The lo part of a split-fp16 operand is an fp16 subnormal for |x| < 2^-3, i.e. the format
// has an ABSOLUTE error floor of ~2^-25: a conv whose whole operand tensor sits below ~2^-11 loses the 1e-4 gate (measured:
// INTEGRATION.md §3) although nothing overflows.  Every operand writer therefore publishes the largest |activation| it wrote
// (before the Winograd transform) into its own slot (float bits, atomicMax); status_finish_kernel turns "non-zero tensor whose
// maximum is below I2V_UNDERFLOW_MAX" into status bit 1 (value 2) at the end of the forward.
constexpr float I2V_UNDERFLOW_MAX = 0x1p-10f;
constexpr float I2V_OVERFLOW_MAX = 6400.f;   // |V| <= 10 max|d| (F(4,3): 4 + 5 + 1): below this no transformed value can leave the fp16 range
constexpr int I2V_STATUS_WORDS = 64;   // [0] flag word, [1 .. 31] per-writer maxima of the running forward, [32 + i] the last forward's (snapshot)
constexpr int I2V_STATUS_SNAP = 32;

__device__ __forceinline__ void publish_umax(int* slot, float m) {
    if (!slot) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) {
        const int bits = __float_as_int(m);   // m >= 0: the integer order of the bit patterns is the float order
        if (bits > *reinterpret_cast<volatile int*>(slot)) atomicMax(slot, bits);
    }
}

__global__ void status_finish_kernel(int* __restrict__ status) {
    int f = 0;
    for (int i = 1; i < I2V_STATUS_SNAP; ++i) {
        const int v = status[i];
        if (v != 0 && __int_as_float(v) < I2V_UNDERFLOW_MAX) f = 2;
        status[I2V_STATUS_SNAP + i] = v;   // kept for the host (mma = auto decides per layer from these)
        status[i] = 0;
    }
    if (f) atomicOr(status, f);
}

template <bool HL16>
__global__ __launch_bounds__(256) void modulate_kernel(const float* __restrict__ x, const float2* __restrict__ coef,
                                                       const float* __restrict__ gb, float* __restrict__ out, int T, int H,
                                                       int W, int C, int ut, int us, int lrelu, int* __restrict__ range_flag,
                                                       int* __restrict__ umax) {
    const int C8 = C >> 3;
    const int b = blockIdx.y;
    const int per = H * W * C8;  // threads per sample
    bool bad = false;  // HL16: a value left the fp16 range of the hi part (sticky flag, see i2v_dec_status)
    float vmax = 0.f;  // HL16: largest |activation| written (underflow guard)
    const int Hl = H / us, Wl = W / us, Tl = T / ut;
    const float2* cp0 = coef + (long)b * C;
    const float* xb = x + (long)b * Tl * Hl * Wl * C;
    const float* gbb = gb ? gb + (long)b * H * W * 2 * C : nullptr;
    char* ob = reinterpret_cast<char*>(out) + (long)b * T * per * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        const int c8 = i % C8;
        int p = i / C8;
        const int w = p % W;
        const int h = p / W;
        float ca[8], cb[8];  // norm(x) == x * ca + cb
        {
            const float4* cp = reinterpret_cast<const float4*>(cp0 + 8 * c8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 ab = cp[j];
                ca[2 * j] = ab.x; cb[2 * j] = ab.y; ca[2 * j + 1] = ab.z; cb[2 * j + 1] = ab.w;
            }
        }
        if (gbb) {  // fold SPADE's gamma / beta into the affine: (x ca + cb) ga + be
            const float* g = gbb + ((long)h * W + w) * (2 * C) + 8 * c8;
            const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(g + C), e1 = *reinterpret_cast<const float4*>(g + C + 4);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float be[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) { cb[j] = fmaf(cb[j], ga[j], be[j]); ca[j] = ca[j] * ga[j]; }
        }
        const float* xp0 = xb + ((long)(h / us) * Wl + w / us) * C + 8 * c8;
        const long xstride = (long)Hl * Wl * C;
        float r0[8];
        for (int t = 0; t < T; ++t) {
            if (t % ut == 0) {
                const float* xp = xp0 + (long)(t / ut) * xstride;
                const float4 v0 = *reinterpret_cast<const float4*>(xp), v1 = *reinterpret_cast<const float4*>(xp + 4);
                r0[0] = v0.x; r0[1] = v0.y; r0[2] = v0.z; r0[3] = v0.w; r0[4] = v1.x; r0[5] = v1.y; r0[6] = v1.z; r0[7] = v1.w;
            }
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r[j] = fmaf(r0[j], ca[j], cb[j]);
                if (lrelu) r[j] = r[j] >= 0.f ? r[j] : 0.2f * r[j];
            }
            char* o = ob + ((long)t * per + i) * 32;
            if (HL16) {
                half8_t hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const _Float16 hh = (_Float16)r[j];
                    bad |= !(fabsf(r[j]) <= 65504.f);
                    vmax = fmaxf(vmax, fabsf(r[j]));
                    hi[j] = hh;
                    lo[j] = (_Float16)(r[j] - (float)hh);
                }
                *reinterpret_cast<half8_t*>(o) = hi;
                *reinterpret_cast<half8_t*>(o + 16) = lo;
            } else {
                *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
                *reinterpret_cast<float4*>(o + 16) = make_float4(r[4], r[5], r[6], r[7]);
            }
        }
    }
    if (HL16 && bad && range_flag) atomicOr(range_flag, 1);
    if (HL16) publish_umax(umax, vmax);
}

// The same modulation, written as the Winograd-transformed operand V = B^T d of i2v_conv16w.hip:
//   V[b][t][c/16][x][h][j][c%16]  (hl16: per 8 channels 8 x fp16 hi | 8 x fp16 lo),  j = output pair (w = 2j, 2j+1),
//   V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3,  d_k = act(...)[t][h][2j-1+k]  (0 outside the row).
// One thread = one (h, j, 8-channel group), looping over the frames like modulate_kernel.  It evaluates only its OWN two
// positions (d1, d2); d0 and d3 are the neighbouring pairs' d2 / d1 and arrive by lane shuffle: thread order = channel
// group within a 32-channel (128-byte) input line fastest, then j, so lane l +- 4 holds pair j +- 1 of the same channels.
// Only the first / last pair of a 16-pair wave segment evaluates its outer neighbour itself.  (Evaluating all four
// positions per thread read every input twice: 9.1 GB instead of 5.5 GB per BAIR step.)
struct ModPos {   // affine of one position: act(x * a + b), and its source row
    float a[8], b[8];
    const float* xp;
};

__device__ __forceinline__ void mod_pos_init(ModPos& m, const float* ca, const float* cb, const float* xb, const float* gbb, int h, int w,
                                             int W, int C, int c8, int us, int Wl) {
    m.xp = xb + ((long)(h / us) * Wl + w / us) * C + 8 * c8;
    if (gbb) {  // fold SPADE's gamma' / beta of the position into the affine: (x ca + cb) ga + be
        const float* g = gbb + ((long)h * W + w) * (2 * C) + 8 * c8;
        const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(g + C), e1 = *reinterpret_cast<const float4*>(g + C + 4);
        const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float be[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) { m.b[c] = fmaf(cb[c], ga[c], be[c]); m.a[c] = ca[c] * ga[c]; }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) { m.a[c] = ca[c]; m.b[c] = cb[c]; }
    }
}

__device__ __forceinline__ void mod_pos_eval(const ModPos& m, long toff, int lrelu, float* d, float& vmax) {
    const float* p = m.xp + toff;
    const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
    const float r0[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float r = fmaf(r0[c], m.a[c], m.b[c]);
        d[c] = (lrelu && r < 0.f) ? 0.2f * r : r;
        vmax = fmaxf(vmax, fabsf(d[c]));
    }
}

__global__ __launch_bounds__(256) void modulate_wino_kernel(const float* __restrict__ x, const float2* __restrict__ coef,
                                                            const float* __restrict__ gb, char* __restrict__ out, int T, int H,
                                                            int W, int C, int ut, int us, int lrelu, int* __restrict__ range_flag,
                                                            int* __restrict__ umax) {
    bool bad = false;
    float vmax = 0.f;
    const int C8 = C >> 3, J = W >> 1;
    const int b = blockIdx.y;
    // Thread = (h, chunk, j, piece p): piece p of a 64-byte V row is [hi | lo] (p & 1) of the 8 channels c8 = 2 chunk + (p >> 1).
    // The hi and the lo lane of a channel group compute the same values (the x loads coalesce; the kernel is HBM-bound),
    // so that every store instruction of a wave writes 16 whole rows = 1 KB contiguous.
    const int per = H * J * C8 * 2;  // threads per sample (a multiple of 64: whole waves stay active for the shuffles)
    const int Hl = H / us, Wl = W / us, Tl = T / ut;
    const float2* cp0 = coef ? coef + (long)b * C : nullptr;   // null: identity (the kernel then only formats the operand)
    const float* xb = x + (long)b * Tl * Hl * Wl * C;
    const float* gbb = gb ? gb + (long)b * H * W * 2 * C : nullptr;
    const int nchunk = C >> 4;
    const long xstride = (long)Hl * Wl * C;
    const int lane = threadIdx.x & 63, jj = lane >> 2;   // jj: position of the pair inside the wave's 16-pair segment
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        // i = ((h * nchunk + chunk) * J + j) * 4 + p
        const int p = i & 3;
        int q = i >> 2;
        const int j = q % J; q /= J;
        const int chunk = q % nchunk;
        const int h = q / nchunk;
        const int c8 = chunk * 2 + (p >> 1);
        const bool is_lo = p & 1;
        float ca[8], cb[8];
        if (cp0) {
            const float4* cp = reinterpret_cast<const float4*>(cp0 + 8 * c8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 ab = cp[k];
                ca[2 * k] = ab.x; cb[2 * k] = ab.y; ca[2 * k + 1] = ab.z; cb[2 * k + 1] = ab.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { ca[k] = 1.f; cb[k] = 0.f; }
        }
        // own positions w = 2j, 2j + 1; the outer neighbours 2j - 1 / 2j + 2 come from lane -+ 4 unless this pair opens /
        // closes the wave's segment (then they are evaluated here) or the row (then they are 0: the conv's zero padding)
        ModPos m1, m2, me;
        mod_pos_init(m1, ca, cb, xb, gbb, h, 2 * j, W, C, c8, us, Wl);
        mod_pos_init(m2, ca, cb, xb, gbb, h, 2 * j + 1, W, C, c8, us, Wl);
        const bool left_row = j == 0, right_row = j == J - 1;
        const bool left_own = !left_row && jj == 0, right_own = !right_row && jj == 15;
        if (left_own || right_own)   // (an edge pair is never both: J >= 4 keeps jj == 0 and jj == 15 apart unless J >= 16)
            mod_pos_init(me, ca, cb, xb, gbb, h, left_own ? 2 * j - 1 : 2 * j + 2, W, C, c8, us, Wl);
        const bool both_own = left_own && right_own;   // impossible (jj is 0 or 15), kept for clarity
        (void)both_own;
        float d0[8], d1[8], d2[8], d3[8], de[8];
        // V row of (t, chunk, x, h, j): 64 bytes; this thread owns its 16-byte piece p
        char* ob = out + ((((long)b * T * nchunk + chunk) * 4 * H + h) * J + j) * 64 + p * 16;
        const long ostride_x = (long)H * J * 64, ostride_t = (long)nchunk * 4 * ostride_x;
        for (int t = 0; t < T; ++t) {
            if (t % ut == 0) {
                const long toff = (long)(t / ut) * xstride;
                mod_pos_eval(m1, toff, lrelu, d1, vmax);
                mod_pos_eval(m2, toff, lrelu, d2, vmax);
                if (left_own || right_own) mod_pos_eval(me, toff, lrelu, de, vmax);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float up = __shfl_up(d2[c], 4), dn = __shfl_down(d1[c], 4);
                    d0[c] = left_row ? 0.f : (left_own ? de[c] : up);
                    d3[c] = right_row ? 0.f : (right_own ? de[c] : dn);
                }
            }
            char* o = ob + (long)t * ostride_t;
#pragma unroll
            for (int xq = 0; xq < 4; ++xq) {
                half8_t piece;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float v = xq == 0 ? d0[c] - d2[c] : xq == 1 ? d1[c] + d2[c] : xq == 2 ? d2[c] - d1[c] : d1[c] - d3[c];
                    const _Float16 hh = (_Float16)v;
                    bad |= !(fabsf(v) <= 65504.f);
                    piece[c] = is_lo ? (_Float16)(v - (float)hh) : hh;
                }
                *reinterpret_cast<half8_t*>(o + xq * ostride_x) = piece;
            }
        }
    }
    if (bad && range_flag) atomicOr(range_flag, 1);
    publish_umax(umax, vmax);
}

// The operand of the F(4,3) kernel (i2v_conv16w4.hip): V[b][t][c/16][x][h][j][c%16], x = 0..5, j = tile of four output
// positions (w = 4j .. 4j+3), d_k = act(...)[t][h][4j-1+k]:
//   V0 = 4 d0 - 5 d2 + d4   V1 = -4 d1 - 4 d2 + d3 + d4   V2 = 4 d1 - 4 d2 - d3 + d4
//   V3 = -2 d1 - d2 + 2 d3 + d4   V4 = 2 d1 - d2 - 2 d3 + d4   V5 = 4 d1 - 5 d3 + d5
// Same thread mapping as modulate_wino_kernel: one thread = one 16-byte piece of the V rows of one (h, tile) column; it
// evaluates its OWN four positions (d1..d4), gets d0 / d5 from the neighbouring tiles by lane shuffle and loops over the frames.
// Thread = (h, chunk, tile j, q): the FOUR channels 4q .. 4q+3 of the chunk, hi AND lo parts: per plane it writes two 8-byte
// half-pieces (hi at byte (q >> 1) * 32 + (q & 1) * 8 of the 64-byte row, lo 16 bytes behind) with two back-to-back store
// instructions, so that the four lanes of a tile complete the row within a few cycles (round 3 gave a lane 8 channels of the hi
// OR the lo part: every value was loaded, evaluated and kept twice -- 215 VGPRs and scratch; now every element is loaded and
// evaluated once).  GB: SPADE's gamma' / beta are present -- every position then has its own affine (a, b)[4], kept in
// registers over the frame loop; without them (the ADAIN operand of conv_1, SPADE's own activation) all positions share the
// sample's (ca, cb).
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

template <bool GB>
struct ModPos4 {
    float a[GB ? 4 : 1], b[GB ? 4 : 1];
    const float* xp;
};

template <bool GB>
__device__ __forceinline__ void mod_pos4_init(ModPos4<GB>& m, const float* ca, const float* cb, const float* xb, const float* gbb, int h, int w,
                                              int W, int C, int c4, int us, int Wl) {
    m.xp = xb + ((long)(h / us) * Wl + w / us) * C + 4 * c4;
    if constexpr (GB) {  // fold SPADE's gamma' / beta of the position into the affine: (x ca + cb) ga + be
        const float* g = gbb + ((long)h * W + w) * (2 * C) + 4 * c4;
        const float4 g0 = *reinterpret_cast<const float4*>(g), e0 = *reinterpret_cast<const float4*>(g + C);
        const float ga[4] = {g0.x, g0.y, g0.z, g0.w};
        const float be[4] = {e0.x, e0.y, e0.z, e0.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { m.b[c] = fmaf(cb[c], ga[c], be[c]); m.a[c] = ca[c] * ga[c]; }
    }
}

template <bool GB>
__device__ __forceinline__ void mod_pos4_eval(const ModPos4<GB>& m, const float* ca, const float* cb, long toff, int lrelu, float* d,
                                              float& vmax) {
#ifdef MOD_NT   // measurement build: the writer's reads and writes are pure streams
    typedef float f4v_ __attribute__((ext_vector_type(4)));
    const f4v_ v0 = __builtin_nontemporal_load(reinterpret_cast<const f4v_*>(m.xp + toff));
#else
    const float4 v0 = *reinterpret_cast<const float4*>(m.xp + toff);
#endif
    const float r0[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float r = GB ? fmaf(r0[c], m.a[c], m.b[c]) : fmaf(r0[c], ca[c], cb[c]);
        d[c] = (lrelu && r < 0.f) ? 0.2f * r : r;
        vmax = fmaxf(vmax, fabsf(d[c]));
    }
}

template <bool GB>
__global__ __launch_bounds__(256) void modulate_wino4_kernel(const float* __restrict__ x, const float2* __restrict__ coef,
                                                             const float* __restrict__ gb, char* __restrict__ out, int T, int H,
                                                             int W, int C, int ut, int us, int lrelu, int* __restrict__ range_flag,
                                                             int* __restrict__ umax) {
    bool bad = false;
    float vmax = 0.f;
    const int C4 = C >> 2, J = W >> 2;
    const int b = blockIdx.y;
    const int per = H * J * C4;        // threads per sample (a multiple of 64: whole waves stay active for the shuffles)
    const int Hl = H / us, Wl = W / us, Tl = T / ut;
    const float2* cp0 = coef ? coef + (long)b * C : nullptr;
    const float* xb = x + (long)b * Tl * Hl * Wl * C;
    const float* gbb = GB ? gb + (long)b * H * W * 2 * C : nullptr;
    const int nchunk = C >> 4;
    const long xstride = (long)Hl * Wl * C;
    const int lane = threadIdx.x & 63, jj = lane >> 2;   // jj: position of the tile inside the wave's 16-tile segment
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        // i = ((h * nchunk + chunk) * J + j) * 4 + q
        // (workgroup = four chunks of one row.  Four rows of one chunk -- 4-8 KB contiguous writes per plane and frame instead of
        //  1-2 KB -- measured the same: profiles/r04_h_operand_writer_order.txt)
        const int q4 = i & 3;
        int q = i >> 2;
        const int j = q % J; q /= J;
        const int chunk = q % nchunk;
        const int h = q / nchunk;
        const int c4 = chunk * 4 + q4;
        float ca[4], cb[4];
        if (cp0) {
            const float4* cp = reinterpret_cast<const float4*>(cp0 + 4 * c4);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 ab = cp[k];
                ca[2 * k] = ab.x; cb[2 * k] = ab.y; ca[2 * k + 1] = ab.z; cb[2 * k + 1] = ab.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { ca[k] = 1.f; cb[k] = 0.f; }
        }
        ModPos4<GB> m1, m2, m3, m4, me;
        mod_pos4_init<GB>(m1, ca, cb, xb, gbb, h, 4 * j, W, C, c4, us, Wl);
        mod_pos4_init<GB>(m2, ca, cb, xb, gbb, h, 4 * j + 1, W, C, c4, us, Wl);
        mod_pos4_init<GB>(m3, ca, cb, xb, gbb, h, 4 * j + 2, W, C, c4, us, Wl);
        mod_pos4_init<GB>(m4, ca, cb, xb, gbb, h, 4 * j + 3, W, C, c4, us, Wl);
        // own positions w = 4j .. 4j+3; the outer neighbours 4j - 1 / 4j + 4 come from lane -+ 4 unless this tile opens / closes the
        // wave's segment (then they are evaluated here) or the row (then they are 0: the conv's zero padding)
        const bool left_row = j == 0, right_row = j == J - 1;
        const bool left_own = !left_row && jj == 0, right_own = !right_row && jj == 15;
        me = m1;
        if (left_own || right_own) mod_pos4_init<GB>(me, ca, cb, xb, gbb, h, left_own ? 4 * j - 1 : 4 * j + 4, W, C, c4, us, Wl);
        float d0[4], d1[4], d2[4], d3[4], d4[4], d5[4], de[4];
        // V row of (t, chunk, plane, h, j): 64 bytes [hi c0-7 | lo c0-7 | hi c8-15 | lo c8-15]; this thread's channels 4 q4 .. 4 q4 + 3
        char* ob = out + ((((long)b * T * nchunk + chunk) * 6 * H + h) * J + j) * 64 + (q4 >> 1) * 32 + (q4 & 1) * 8;
        const long ostride_x = (long)H * J * 64, ostride_t = (long)nchunk * 6 * ostride_x;
        for (int t = 0; t < T; ++t) {
            if (t % ut == 0) {
                const long toff = (long)(t / ut) * xstride;
                mod_pos4_eval<GB>(m1, ca, cb, toff, lrelu, d1, vmax);
                mod_pos4_eval<GB>(m2, ca, cb, toff, lrelu, d2, vmax);
                mod_pos4_eval<GB>(m3, ca, cb, toff, lrelu, d3, vmax);
                mod_pos4_eval<GB>(m4, ca, cb, toff, lrelu, d4, vmax);
                if (left_own || right_own) mod_pos4_eval<GB>(me, ca, cb, toff, lrelu, de, vmax);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float up = __shfl_up(d4[c], 4), dn = __shfl_down(d1[c], 4);
                    d0[c] = left_row ? 0.f : (left_own ? de[c] : up);
                    d5[c] = right_row ? 0.f : (right_own ? de[c] : dn);
                }
            }
            char* o = ob + (long)t * ostride_t;
#pragma unroll
            for (int xq = 0; xq < 6; ++xq) {
                half4_t ph, pl;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v;
                    if (xq == 0) v = fmaf(4.f, d0[c], fmaf(-5.f, d2[c], d4[c]));
                    else if (xq == 1) v = fmaf(-4.f, d1[c] + d2[c], d3[c] + d4[c]);
                    else if (xq == 2) v = fmaf(4.f, d1[c] - d2[c], d4[c] - d3[c]);
                    else if (xq == 3) v = fmaf(2.f, d3[c] - d1[c], d4[c] - d2[c]);
                    else if (xq == 4) v = fmaf(2.f, d1[c] - d3[c], d4[c] - d2[c]);
                    else v = fmaf(4.f, d1[c], fmaf(-5.f, d3[c], d5[c]));
                    const _Float16 hh = (_Float16)v;
                    bad |= !(fabsf(v) <= 65504.f);
                    ph[c] = hh;
                    pl[c] = (_Float16)(v - (float)hh);
                }
#ifdef MOD_NT
                __builtin_nontemporal_store(ph, reinterpret_cast<half4_t*>(o + xq * ostride_x));
                __builtin_nontemporal_store(pl, reinterpret_cast<half4_t*>(o + xq * ostride_x + 16));
#else
                *reinterpret_cast<half4_t*>(o + xq * ostride_x) = ph;
                *reinterpret_cast<half4_t*>(o + xq * ostride_x + 16) = pl;
#endif
            }
        }
    }
    if (bad && range_flag) atomicOr(range_flag, 1);
    publish_umax(umax, vmax);
}

// conv_img (decoder.py:117: Conv3d(nf, 3, 3, padding 1) + tanh) in split-fp16 mode.  With three output channels a tiled
// implicit GEMM wastes the matrix cores (N padded to 32) and the vector-ALU kernel is LDS-bound; instead the conv is split
// into a 1x1x1 GEMM Y[tap * 3 + n][pos] = sum_c x[pos][c] w[n][c][tap] (81 planes, written transposed by
// pointwise16_forward: HBM-bound) and this gather: out[n][pos] = tanh(bias[n] + sum_tap Y[tap * 3 + n][pos + delta_tap]),
// zero padding = skipped taps.  Consecutive lanes = consecutive w: every load and store is coalesced.
// out: frames [B][T][3][H][W].
__global__ __launch_bounds__(256) void conv_img_gather_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                              float* __restrict__ out, long total, int T, int H, int W, long obs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % W);
    long q = i / W;
    const int h = (int)(q % H); q /= H;
    const int t = (int)(q % T);
    const long b = q / T;
    float s0 = bias[0], s1 = bias[1], s2 = bias[2];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const int tt = t + kt - 1;
        if ((unsigned)tt >= (unsigned)T) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = w + kw - 1;
                if ((unsigned)ww >= (unsigned)W) continue;
                const float* p = y + (long)(((kt * 3 + kh) * 3 + kw) * 3) * total + (((b * T + tt) * H + hh) * W + ww);
                s0 += p[0]; s1 += p[total]; s2 += p[2 * total];
            }
        }
    }
    const long hw = (long)h * W + w, HW = (long)H * W;
    float* o = out + b * obs + ((long)t * 3) * HW + hw;
    o[0] = tanhf(s0); o[HW] = tanhf(s1); o[2 * HW] = tanhf(s2);
}

// F.interpolate(img, size=(h,w), mode='bilinear', align_corners=True) (normalization_layer.py:20), written
// channels-last with the 3 colour channels zero-padded to 16 (the conv kernel's K chunk).
__global__ void resize_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int Hi, int Wi, int Ho, int Wo,
                              int hl16, int* __restrict__ range_flag, long ibs) {   // ibs: floats between the samples of `img`
    bool bad = false;
    const long total = (long)B * Ho * Wo;
    const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wo);
        const int h = (int)((i / Wo) % Ho);
        const int b = (int)(i / ((long)Wo * Ho));
        const float fh = sh * h, fw = sw * w;
        const int h0 = (int)fh, w0 = (int)fw;
        const int h1 = h0 + (h0 < Hi - 1 ? 1 : 0), w1 = w0 + (w0 < Wi - 1 ? 1 : 0);
        const float lh1 = fh - h0, lh0 = 1.f - lh1, lw1 = fw - w0, lw0 = 1.f - lw1;
        float* o = out + i * 16;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = img + (long)b * ibs + (long)c * Hi * Wi;
            v[c] = lh0 * (lw0 * pl[h0 * Wi + w0] + lw1 * pl[h0 * Wi + w1]) +
                   lh1 * (lw0 * pl[h1 * Wi + w0] + lw1 * pl[h1 * Wi + w1]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) o[c] = 0.f;
        if (hl16) {  // split-fp16 operand format: per 8 channels 8 x fp16 hi | 8 x fp16 lo (64 bytes per position, as fp32)
            _Float16* oh = reinterpret_cast<_Float16*>(o);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const _Float16 hh = (_Float16)v[c];
                bad |= !(fabsf(v[c]) <= 65504.f);
                oh[c] = hh;
                oh[8 + c] = (_Float16)(v[c] - (float)hh);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = v[c];
        }
    }
    if (bad && range_flag) atomicOr(range_flag, 1);
}

// Layout conversion for the stand-alone sub-module entry points: the reference surface is [B][C][T][H][W] ("NCDHW"),
// the kernels work channels-last.  32x32 tiles through LDS, both sides coalesced.  to_cl: in [B][C][P] -> out [B][P][C].
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int P,
                                                        int to_cl) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int R = to_cl ? C : P, S = to_cl ? P : C;  // input is [R][S] per sample, output [S][R]
    const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* ip = in + (long)b * R * S;
    float* op = out + (long)b * R * S;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && s0 + tx < S) tile[i][tx] = ip[(long)(r0 + i) * S + s0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (s0 + i < S && r0 + tx < R) op[(long)(s0 + i) * R + r0 + tx] = tile[tx][i];
}

}  // namespace i2v

using namespace i2v;

namespace {

struct Block {
    std::string name;
    int n_in = 0, n_out = 0, n_mid = 0;
    bool learned = false;
    int groups_spade = 16;
    ConvWeights conv0, conv1, convs, sp_conv, sp_gb;
    Conv16Weights sp_conv16;  // SPADE's Conv2d(3, 128, 3) with the 3 input channels zero-padded to 8 (split-fp16 mode)
    Conv16Weights conv0_16, conv1_16, sp_gb16;  // split-fp16 variants (cfg.mma == 1)
    Conv16Weights convs16;      // the learned shortcut's 1x1x1 conv on split-fp16 operands (pointwise16_forward)
    Wino16Weights sp_gb_w;      // SPADE's fused gamma|beta Conv2d(128, 2C, 3) on the Winograd kernel (1x3x3 variant)
    Wino4Weights sp_gb_w4;      // ... on the F(4,3) kernel (packed INSTEAD where the shape allows: W % 16 == 0, H % 32 == 0)
    Wino16Weights conv0_w, conv1_w;             // Winograd F(2,3) variants of conv_0 / conv_1 (packed where the shape allows)
    Wino4Weights conv0_w4, conv1_w4;            // Winograd F(4,3) variants (i2v_conv16w4.hip); packed INSTEAD of the F(2,3) ones
    Wino4F32Weights conv0_wf, conv1_wf;         // exact-fp32 mode: Winograd F(4,3) on the fp32 matrix cores (i2v_wino32.hip), next to conv0 / conv1
    bool tdup0 = false;                          // conv_0 runs on the half-rate tensor (x2 temporal up-sampling in front)
    DevBuf gn_w, gn_b;
    int zoff = 0;  // offset of this block's ADAIN (gamma|beta) in the z-GEMM output
};

struct Level { int T, H, W, ut, us; };  // resolution a block runs at and the upsample factors in front of it

}  // namespace

struct i2v_dec {
    i2v_dec_cfg cfg;
    bool loaded = false;
    int nf = 0;
    Block blk[6];
    Level lvl[6];
    ConvWeights fc, zlin, conv_img;
    ConvImgWeights conv_img_v;  // vector-ALU variant (used when the output geometry tiles into 4x8x8 bricks)
    ConvImgMfmaWeights conv_img_m;  // fused matrix-core variant (split-fp16 mode, img16 == 2)
    Conv16Weights conv_img16;   // split-fp16 mode: the 81-column 1x1x1 GEMM of conv_img_gather_kernel
    DevBuf conv_img_bias;
    int Nz = 0;
    int wino = 1;  // 1: 3x3x3 convs whose shape allows it use the Winograd kernel (env I2V_DEC_WINO=0 disables)
    int img16 = 2;  // split-fp16 mode: 2 fused matrix-core kernel (i2v_convimg.hip), 1 round 2's 81-plane GEMM + gather at nf >= 64, 0 vector-ALU kernel (env I2V_DEC_IMG16)
    int wino4 = 1; // 1: F(4,3) Winograd kernel where the shape allows and one sample gives >= 32 workgroups (env I2V_DEC_WINO4=0: F(2,3); 2: wherever the shape allows)
    int spw = 1;   // 1: SPADE's gamma|beta conv uses the Winograd kernel where the shape allows (env I2V_DEC_SPW=0: direct kernel)
    int gen = 0;   // 1: the thin F(4,3) layers of the 128 x 128 configs (g_4: 32 output channels at 16 x 128 x 128) generate their operand in the
                   // conv kernel's own producer waves instead of reading a V tensor an operand-writer launch wrote (i2v_conv16w4g.hip; same
                   // bits).  env I2V_DEC_GEN.  Measured in profiles/r06_*_thin_fused_*.
    int wino32 = 1;  // exact-fp32 mode (mma = 0): 1 = 3x3x3 convs from the 8x8 level on run Winograd F(4,3) on the fp32 matrix cores (env I2V_DEC_WINO32=0: direct kernel)
    const float* prep_img = nullptr;   // i2v_dec_prepare: the start frames whose SPADE branches are in the workspace's gbs[] ...
    int prep_B = 0;                    // ... their batch, image size and the workspace they live in (consumed by the next matching forward)
    int prep_h = 0, prep_w = 0;
    long prep_bstride = 0;
    bool prep_forked = false;          // the prepared maps are being computed on the handle's side stream (ev_lvl[k] mark them complete)
    const void* prep_ws = nullptr;
    long img_bstride = 0;              // floats between the samples of the current call's start frames (0: dense [B][3][H][W])
    int sub = 0;   // samples per sub-batch of the last two levels (env I2V_DEC_SUB; 0: the whole batch per launch)
    int pw16 = 1;  // 1: split-fp16 mode runs the shortcut convs on split-fp16 operands too (env I2V_DEC_PW16=0: exact-fp32 MFMA)
    int device = 0;             // the device the packed weights live on
    int* status_dev = nullptr;  // sticky range flag of the hl16 producers (device) ...
    int* status_host = nullptr; // ... and its pinned host mirror, refreshed asynchronously at the end of every forward
    // In-call overlap (round 5): the SPADE conditioning branches of all six blocks depend on the start frames only, so a forward
    // that finds no prepared maps runs them on the handle's own side stream (forked from the caller's stream by an event) while the
    // caller's stream computes fc / ADAIN linears / head_0 / g_0 ... -- the early levels' launches leave most of the chip idle
    // (4x4 .. 16x16 maps), the branches of the late levels fill it.  Every block waits for its level's event; same kernels, same
    // bits.  env I2V_DEC_OVERLAP=0: the branches run inline on the caller's stream (round 4).
    int overlap = 1;
    int no_side_shortcut = 0;   // env I2V_DEC_OVERLAP=2: branches on the side stream, shortcuts inline (A/B of the two halves)
    hipStream_t side = nullptr;
    bool side_owned = true;   // false: `side` is a caller's stream (i2v_dec_set_side_stream), e.g. the one its cINN prefetch runs on
    hipEvent_t ev_fork = nullptr, ev_lvl[6] = {};
    // ... and the learned shortcut of a block (Norm3D + 1x1x1 conv at the low resolution: an HBM-bound GEMM that only conv_1 needs)
    // runs there too, underneath the block's modulate / conv_0 chain: ev_x[k] = block input and its statistics complete (caller's
    // stream), ev_s[k] = shortcut complete (side stream)
    hipEvent_t ev_x[6] = {}, ev_s[6] = {};
    // One handle = one workspace, one set of side-stream events: forwards / prepares on a handle are serialised.  A call that arrives
    // on another stream than the previous one first waits for the previous call (event recorded behind every call), like i2v_flow.
    // Matrix-core mode (i2v_dec_cfg.mma): 0 exact fp32, 1 split-fp16, 2 AUTO = split-fp16 with a per-layer fallback behind the range
    // guard: both weight sets are packed; every forward ends with a stream synchronisation and a look at the operand maxima the
    // writers published -- a 3x3x3 conv whose operand tensor left the window the split format holds 1e-4 in (max |activation| below
    // 2^-10 or above 6400) is switched to the exact-fp32 kernels (Winograd F(4,3) on the fp32 matrix cores where the shape allows,
    // i2v_wino32.hip) for the rest of the handle's life and the forward is run again; an overflow no slot explains (SPADE's own
    // activation, the shortcut GEMM, conv_img) switches the whole handle.  In-range checkpoints run exactly the mma = 1 launches.
    bool fp32_layer[12] = {};   // layer = 2 * block + (0: conv_0, 1: conv_1)
    bool fp32_all = false;
    int auto_reruns = 0;        // forwards that had to be run again (reporting)
    bool has16() const { return cfg.mma != 0; }                       // split-fp16 weights are packed
    bool has32() const { return cfg.mma != 1; }                       // exact-fp32 weights are packed
    bool aux16() const { return has16() && !fp32_all; }               // SPADE branch, shortcut GEMM, conv_img, resize on the split-fp16 path
    bool layer16(int layer) const { return aux16() && !fp32_layer[layer]; }
    StreamOrder order;   // (capture-aware: i2v_common.h)
    // A forked prepare (or an in-call fork that failed half-way) leaves work on the side stream that nothing on a caller's stream has
    // waited for yet: `side_unjoined`.  Whoever DROPS such a prepare (i2v_dec_prepare_cancel followed by a forward, a forward with other
    // start frames / another workspace, i2v_dec_join, the destructor) joins the side stream first, so that the lifetime of the
    // caller-owned workspace and start frames is bounded by the caller's stream again (round-5 advisor finding).
    bool side_unjoined = false;
    // `st` waits for everything enqueued on the side stream so far (a fresh record of ev_fork on the side stream: covers the SPADE
    // branches of every level AND the shortcut GEMMs).  Not while `st` captures: an event recorded outside a capture cannot be waited
    // on inside it; the flag then stays set for the next eager call.
    int join_side(hipStream_t st) {
        if (!side || !ev_fork) { side_unjoined = false; return I2V_OK; }
        if (stream_is_capturing(st)) return I2V_OK;
        I2V_HIP_CHECK(hipEventRecord(ev_fork, side));
        I2V_HIP_CHECK(hipStreamWaitEvent(st, ev_fork, 0));
        side_unjoined = false;
        return I2V_OK;
    }
    ~i2v_dec() {
        if (side) (void)hipStreamSynchronize(side);   // nothing of this handle may still write the caller's workspace once it is gone
        for (auto& e : ev_x)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_s)
            if (e) (void)hipEventDestroy(e);
        if (status_dev) (void)hipFree(status_dev);
        if (status_host) (void)hipHostFree(status_host);
        if (side && side_owned) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (auto& e : ev_lvl)
            if (e) (void)hipEventDestroy(e);
    }
    int profile = 0;
    struct ProfEv { hipEvent_t e0, e1; double flops, exec_flops; int layer; };
    std::vector<ProfEv> prof_events;
    // per-layer totals of the profiled 3x3x3 launches: layer = 2 * block + (0: conv_0, 1: conv_1)
    struct ProfLayer { double ms = 0, flops = 0, exec_flops = 0; long launches = 0; int kernel = 0; long grid = 0; };
    ProfLayer prof_layers[12];
    int prof_cur_layer = 0, prof_cur_kernel = 0;
    double prof_conv3_ms = 0, prof_conv3_flops = 0, prof_conv3_exec = 0;
    long prof_conv3_launches = 0;
    // debug tap: copy one intermediate (channels-last) of one block out of the workspace during forward
    int tap_block = -1, tap_which = -1;
    float* tap_dst = nullptr;
    size_t tap_max = 0;
};

namespace {

struct DecWs {
    size_t xA, xB, a, dx, xs_in, xs_low, y0, y1, gb, zl, sums1, sums2, sums3, coef, splitk, splitk_floats, y1v, total;
    size_t gbs[6], py0, py1, py1v;   // i2v_dec_prepare: one gamma|beta buffer per level and its own SPADE scratch
    size_t m6 = 0;                   // exact-fp32 Winograd: the six partial outputs M_x of one conv
    size_t coef_s = 0;
    bool has_y1v = false;
};

bool want_wino0(const i2v_dec* d, const Block& b, const Level& l);
bool want_wino1(const i2v_dec* d, const Block& b, const Level& l);
bool want_w4_0(const i2v_dec* d, const Block& b, const Level& l);
bool want_w4_1(const i2v_dec* d, const Block& b, const Level& l);
bool want_wf_0(const i2v_dec* d, const Block& b, const Level& l);
bool want_wf_1(const i2v_dec* d, const Block& b, const Level& l);

// SPADE's gamma|beta Conv2d(128, 2C, 3) on a Winograd kernel (F(4,3) 1x3x3 variant, else F(2,3)): the predicate of
// i2v_dec_load's packing and of the y1v workspace
bool spade_w4_wanted(const i2v_dec* d, const Block& b, const Level& l) {
    return d->has16() && d->wino && d->spw && d->wino4 && (2 * b.n_in) % 64 == 0 && wino4_supported(2 * b.n_in, 128, 1, l.H, l.W, 1);
}
bool spade_wino_wanted(const i2v_dec* d, const Block& b, const Level& l) {
    return spade_w4_wanted(d, b, l) || (d->has16() && d->wino && d->spw && wino16_supported(2 * b.n_in, 128, 1, l.H, l.W, 1));
}

DecWs dec_ws(const i2v_dec* d, int B) {
    size_t mx_x = (size_t)16 * d->blk[0].n_in, mx_a = 0, mx_dx = 0, mx_xsin = 0, mx_xslow = 0, mx_y = 0, mx_gb = 0, mx_yv = 0, mx_m6 = 0;
    int cmax = 0;
    for (int k = 0; k < 6; ++k) {
        const Block& b = d->blk[k];
        const Level& l = d->lvl[k];
        const size_t P = (size_t)l.T * l.H * l.W, Pl = P / ((size_t)l.ut * l.us * l.us);
        mx_x = std::max(mx_x, P * b.n_out);
        // conv operands: hl16 (4 B per element), or the Winograd operand V (4 values per output pair: 8 B per element)
        mx_a = std::max(mx_a, P * b.n_in * (want_wino0(d, b, l) || want_w4_0(d, b, l) ? 2 : 1));
        mx_a = std::max(mx_a, P * b.n_mid * (want_wino1(d, b, l) || want_w4_1(d, b, l) ? 2 : 1));
        // exact-fp32 Winograd: V = 6 planes per 4 positions (1.5 x the activation), M = 6 planes per 4 outputs
        if (want_wf_0(d, b, l)) { mx_a = std::max(mx_a, P * b.n_in * 3 / 2); mx_m6 = std::max(mx_m6, P * b.n_mid * 3 / 2); }
        if (want_wf_1(d, b, l)) { mx_a = std::max(mx_a, P * b.n_mid * 3 / 2); mx_m6 = std::max(mx_m6, P * b.n_out * 3 / 2); }
        mx_dx = std::max(mx_dx, P * b.n_mid);
        if (b.learned) { mx_xsin = std::max(mx_xsin, Pl * b.n_in); mx_xslow = std::max(mx_xslow, Pl * b.n_out); }
        mx_y = std::max(mx_y, (size_t)l.H * l.W);
        if (spade_wino_wanted(d, b, l)) mx_yv = std::max(mx_yv, (size_t)l.H * l.W);   // only levels whose gamma|beta conv runs a Winograd kernel
        mx_gb = std::max(mx_gb, (size_t)l.H * l.W * 2 * b.n_in);
        cmax = std::max(cmax, std::max(b.n_in, b.n_mid));
    }
    if (d->has16() && d->img16 == 1 && d->nf >= 64) mx_a = std::max(mx_a, (size_t)d->lvl[5].T * d->lvl[5].H * d->lvl[5].W * 81);  // conv_img's Y
    DecWs L;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = align_up(o + floats * 4, 256); return r; };
    L.xA = take(B * mx_x); L.xB = take(B * mx_x);
    L.a = take(B * mx_a); L.dx = take(B * mx_dx);
    L.xs_in = take(B * mx_xsin); L.xs_low = take(B * mx_xslow);
    L.y0 = take(B * mx_y * 16); L.y1 = take(B * mx_y * 128); L.gb = take(B * mx_gb);
    L.y1v = take(B * mx_yv * 256);  // Winograd operand V of SPADE's 128-channel activation (8 bytes per activation)
    L.has_y1v = mx_yv > 0;
    L.zl = take((size_t)B * d->Nz);
    L.sums1 = take((size_t)B * cmax * 4); L.sums2 = take((size_t)B * cmax * 4);  // doubles: 2 per channel
    L.sums3 = take((size_t)B * cmax * 4);   // block output statistics (sums1 / sums3 alternate as a block's input / output statistics)
    L.coef = take((size_t)B * cmax * 2);
    L.coef_s = take((size_t)B * cmax * 2);   // the shortcut's Norm3D coefficients when it runs on the side stream
    {   // split-K scratch of the direct conv kernel (the tiny feature maps of head_0 / g_0): its partial copies of the output
        size_t mx = 0;
        for (int k = 0; k < 6; ++k) {
            const long P = (long)d->lvl[k].T * d->lvl[k].H * d->lvl[k].W;
            const int f = std::max(conv16_splitk_factor(P, (d->blk[k].n_in + 31) / 32), conv16_splitk_factor(P, (d->blk[k].n_mid + 31) / 32));
            if (f > 1) mx = std::max(mx, (size_t)f * P * std::max(d->blk[k].n_mid, d->blk[k].n_out));
        }
        L.splitk_floats = (size_t)B * mx;
        L.splitk = take(L.splitk_floats);
    }
    for (int k = 0; k < 6; ++k) L.gbs[k] = take((size_t)B * d->lvl[k].H * d->lvl[k].W * 2 * d->blk[k].n_in);
    L.py0 = take(B * mx_y * 16); L.py1 = take(B * mx_y * 128); L.py1v = take(B * mx_yv * 256);
    L.m6 = take(B * mx_m6);   // exact-fp32 Winograd: the six partial outputs M_x
    L.total = o;
    return L;
}

int run_stats(const float* x, double* sums, int B, long P, int C, hipStream_t st) {
    I2V_REQUIRE(C % 4 == 0 && (C <= 1024 || C % 1024 == 0), I2V_E_INVALID, "stats: unsupported channel count %d", C);
    I2V_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)B * C * 16, st));
    const int Cs = C > 1024 ? 1024 : C, nz = C / Cs;  // channel slices
    const int R = 256 / (Cs / 4);
    long rows = R * 16;                       // at least 16 rows per thread-row
    const long want = (P + 1023) / 1024;      // at most ~1024 chunks per sample
    if (rows < want) rows = (want + R - 1) / R * R;
    const int chunks = (int)((P + rows - 1) / rows);
    hipLaunchKernelGGL(stats_kernel, dim3(chunks, B, nz), dim3(256), 0, st, x, sums, (int)P, Cs, (int)rows, C);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int run_coef(const double* sums, float* coef, int B, int C, int groups, double count, const float* zl, int zstride,
             int zoff, const float* gw, const float* gb, hipStream_t st) {
    hipLaunchKernelGGL(coef_kernel, dim3(B), dim3(256), 0, st, sums, reinterpret_cast<float2*>(coef), C, groups, count, zl,
                       zstride, zoff, gw, gb);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int run_modulate(const float* x, const float* coef, const float* gb, float* out, int B, int T, int H, int W, int C, int ut,
                 int us, int lrelu, hipStream_t st, bool hl16 = false, int* range_flag = nullptr, int* umax = nullptr) {
    I2V_REQUIRE(C % 8 == 0, I2V_E_INVALID, "modulate: channels %d not a multiple of 8", C);
    const long per = (long)H * W * (C / 8);  // threads per sample (each loops over the T frames)
    I2V_REQUIRE(per * T < (1L << 31), I2V_E_INVALID, "modulate: tensor too large");
    const unsigned gx = (unsigned)std::min<long>((per + 255) / 256, 8192);
    if (hl16)
        hipLaunchKernelGGL(modulate_kernel<true>, dim3(gx, B), dim3(256), 0, st, x, reinterpret_cast<const float2*>(coef), gb, out,
                           T, H, W, C, ut, us, lrelu, range_flag, umax);
    else
        hipLaunchKernelGGL(modulate_kernel<false>, dim3(gx, B), dim3(256), 0, st, x, reinterpret_cast<const float2*>(coef), gb, out,
                           T, H, W, C, ut, us, lrelu, range_flag, umax);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int run_modulate_wino4(const float* x, const float* coef, const float* gb, float* out, int B, int T, int H, int W, int C, int ut,
                       int us, int lrelu, hipStream_t st, int* range_flag, int* umax = nullptr) {
    I2V_REQUIRE(C % 32 == 0 && W % 4 == 0, I2V_E_INVALID, "modulate (F(4,3) operand): channels %d / width %d", C, W);
    const long per = (long)H * (W / 4) * (C / 4);   // one thread per (h, tile, 4 channels)
    I2V_REQUIRE(per % 64 == 0, I2V_E_INVALID, "modulate (F(4,3) operand): %ld threads per sample (need whole wavefronts)", per);
    I2V_REQUIRE(per * T * 6 < (1L << 31), I2V_E_INVALID, "modulate: tensor too large");
    const unsigned gx = (unsigned)std::min<long>((per + 255) / 256, 8192);
    if (gb)
        hipLaunchKernelGGL(modulate_wino4_kernel<true>, dim3(gx, B), dim3(256), 0, st, x, reinterpret_cast<const float2*>(coef), gb,
                           reinterpret_cast<char*>(out), T, H, W, C, ut, us, lrelu, range_flag, umax);
    else
        hipLaunchKernelGGL(modulate_wino4_kernel<false>, dim3(gx, B), dim3(256), 0, st, x, reinterpret_cast<const float2*>(coef), gb,
                           reinterpret_cast<char*>(out), T, H, W, C, ut, us, lrelu, range_flag, umax);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int run_modulate_wino(const float* x, const float* coef, const float* gb, float* out, int B, int T, int H, int W, int C, int ut,
                      int us, int lrelu, hipStream_t st, int* range_flag, int* umax = nullptr) {
    I2V_REQUIRE(C % 32 == 0 && W % 2 == 0, I2V_E_INVALID, "modulate (Winograd operand): channels %d / width %d", C, W);
    const long per = (long)H * (W / 2) * (C / 8) * 2;
    I2V_REQUIRE(per % 64 == 0, I2V_E_INVALID, "modulate (Winograd operand): %ld threads per sample (need whole wavefronts)", per);
    I2V_REQUIRE(per * T * 4 < (1L << 31), I2V_E_INVALID, "modulate: tensor too large");
    const unsigned gx = (unsigned)std::min<long>((per + 255) / 256, 8192);
    hipLaunchKernelGGL(modulate_wino_kernel, dim3(gx, B), dim3(256), 0, st, x, reinterpret_cast<const float2*>(coef), gb,
                       reinterpret_cast<char*>(out), T, H, W, C, ut, us, lrelu, range_flag, umax);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

// Brackets one 3x3x3 conv launch with HIP events on the launch stream WITHOUT synchronising; the pairs are
// resolved later by i2v_dec_get_profile (after the caller has synchronised the stream).
struct ProfScope {
    i2v_dec* d;
    hipStream_t st;
    hipEvent_t e0 = nullptr;
    double flops, exec_flops;
    ProfScope(i2v_dec* d_, hipStream_t st_, double flops_, double exec_) : d(d_), st(st_), flops(flops_), exec_flops(exec_) {
        if (d->profile) { (void)hipEventCreate(&e0); (void)hipEventRecord(e0, st); }
    }
    ~ProfScope() {
        if (d->profile) {
            hipEvent_t e1 = nullptr;
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e1, st);
            d->prof_events.push_back({e0, e1, flops, exec_flops, d->prof_cur_layer});
            if (d->prof_cur_layer >= 0 && d->prof_cur_layer < 12) d->prof_layers[d->prof_cur_layer].kernel = d->prof_cur_kernel;
        }
    }
};

int conv3(i2v_dec* d, const ConvWeights& w, const float* in, float* out, const float* res, int rt, int rs, int B,
          const Level& l, int epi, hipStream_t st) {
    const double fl = 2.0 * B * l.T * l.H * l.W * (double)w.Cin * w.Cout * 27.0;
    ProfScope ps(d, st, fl, fl);
    return conv_forward(w, in, w.Cin, out, res, rt, rs, B, l.T, l.H, l.W, epi, st);
}

// exact-fp32 mode, Winograd F(4,3): six 9-tap plane convs + the output transform; matrix-core FLOPs issued = 1/2 of the algorithmic
int conv3_wf(i2v_dec* d, const Wino4F32Weights& w, const float* V, float* M, float* out, const float* res, int rt, int rs, int B,
             const Level& l, int epi, hipStream_t st) {
    const double fl = 2.0 * B * l.T * l.H * l.W * (double)w.Cin * w.Cout * 27.0;
    ProfScope ps(d, st, fl, 0.5 * fl);
    return wino4f32_forward(w, V, M, out, res, rt, rs, B, l.T, l.H, l.W, epi, st);
}

int conv3_16(i2v_dec* d, const Conv16Weights& w, const float* in_hl16, float* out, const float* res, int rt, int rs, int B,
             const Level& l, int epi, hipStream_t st, double* stats = nullptr, float* splitk = nullptr, size_t splitk_floats = 0) {
    if (stats) I2V_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)B * w.Cout * 16, st));
    // algorithmic FLOPs of the reference's 3x3x3 conv; matrix-core FLOPs actually issued = 3 fp16 MFMAs per product, on
    // 18 instead of 27 taps in temporal-duplication mode
    const double fl = 2.0 * B * l.T * l.H * l.W * (double)w.Cin * w.Cout * 27.0;
    ProfScope ps(d, st, fl, 3.0 * fl * (w.tdup ? 18.0 / 27.0 : 1.0));
    return conv16_forward(w, in_hl16, out, res, rt, rs, B, l.T, l.H, l.W, epi, st, stats, nullptr, splitk, splitk_floats);
}

int conv3_w(i2v_dec* d, const Wino16Weights& w, const float* v_hl16, float* out, const float* res, int rt, int rs, int B,
            const Level& l, int epi, hipStream_t st, double* stats = nullptr) {
    if (stats) I2V_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)B * w.Cout * 16, st));
    // matrix-core FLOPs issued: 4 Winograd products per 2 outputs x 3 kw taps (x 2/3), 3 fp16 MFMAs each, 18 of 27 taps in
    // temporal-duplication mode
    const double fl = 2.0 * B * l.T * l.H * l.W * (double)w.Cin * w.Cout * 27.0;
    ProfScope ps(d, st, fl, 3.0 * fl * (2.0 / 3.0) * (w.tdup ? 18.0 / 27.0 : 1.0));
    return wino16_forward(w, v_hl16, out, res, rt, rs, B, l.T, l.H, l.W, epi, st, stats);
}

int conv3_w4(i2v_dec* d, const Wino4Weights& w, const float* v_hl16, float* out, const float* res, int rt, int rs, int B,
             const Level& l, int epi, hipStream_t st, double* stats = nullptr) {
    if (stats) I2V_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)B * w.Cout * 16, st));
    // matrix-core FLOPs issued: 6 Winograd products per 4 outputs x 3 kw taps (x 1/2), 3 fp16 MFMAs each
    const double fl = 2.0 * B * l.T * l.H * l.W * (double)w.Cin * w.Cout * 27.0;
    ProfScope ps(d, st, fl, 3.0 * fl * 0.5 * (w.tdup ? 18.0 / 27.0 : 1.0));
    return wino4_forward(w, v_hl16, out, res, rt, rs, B, l.T, l.H, l.W, epi, st, stats);
}

// the same conv with the operand generated in the kernel (no operand-writer launch in front): x = the conv's fp32 input before the
// modulation, coef / gb as the writer takes them
int conv3_w4g(i2v_dec* d, const Wino4Weights& w, const float* x, const float* coef, const float* gb, int us, float* out, const float* res, int rt,
              int rs, int B, const Level& l, int epi, hipStream_t st, double* stats, int* flag, int* umax) {
    if (stats) I2V_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)B * w.Cout * 16, st));
    const double fl = 2.0 * B * l.T * l.H * l.W * (double)w.Cin * w.Cout * 27.0;
    ProfScope ps(d, st, fl, 3.0 * fl * 0.5);
    return wino4g_forward(w, x, coef, gb, us, out, res, rt, rs, B, l.T, l.H, l.W, epi, st, stats, flag, umax);
}

// which kernel conv_0 / conv_1 of a block use at this geometry (want_*: by shape; use_*: and the weights are packed for it)
bool want_wino0(const i2v_dec* d, const Block& b, const Level& l) {
    const bool tdup = l.ut == 2;   // conv_0 behind a x2 temporal up-sampling: pair kernels on the half-rate tensor (Block::tdup0)
    return d->has16() && d->wino && wino16_supported(b.n_mid, b.n_in, tdup ? l.T / 2 : l.T, l.H, l.W, tdup ? 2 : 3);
}
bool want_wino1(const i2v_dec* d, const Block& b, const Level& l) {
    return d->has16() && d->wino && wino16_supported(b.n_out, b.n_mid, l.T, l.H, l.W, 3);
}
// F(4,3): its bricks hold 512 output positions x 64 or 32 channels (the launcher picks 32-channel workgroups when 64-channel ones
// would not fill the chip; both give the same bits) -- wherever one SAMPLE gives >= 16 workgroups of 32 channels, i.e. from the
// 16x16 level on (round 3 stopped at 32x32: g_1 ran F(2,3); measured at B = 64: g_1.conv_0 1.94 -> 1.49 ms, conv_1 1.47 -> 1.13,
// at B = 8 equal).  WHETHER a layer runs F(4,3) depends on the layer only; the workgroup width (64 or 32 channels) is chosen by the
// launcher from batch x bricks against the CU count -- it changes the schedule, not the accumulation order of any output, so
// shards reproduce the full batch bit for bit (test_f43_tile_width_switch_across_batches crosses the threshold).
bool w4_fills(const Level& l, int cout) { return (long)l.T * l.H * l.W / 512 * std::max(cout / 32, 1) >= 16; }
bool want_w4_0(const i2v_dec* d, const Block& b, const Level& l) {
    const bool tdup = l.ut == 2;
    return d->has16() && d->wino && d->wino4 && (d->wino4 == 2 || w4_fills(l, b.n_mid)) &&
           wino4_supported(b.n_mid, b.n_in, tdup ? l.T / 2 : l.T, l.H, l.W, tdup ? 2 : 3);
}
bool want_w4_1(const i2v_dec* d, const Block& b, const Level& l) {
    return d->has16() && d->wino && d->wino4 && (d->wino4 == 2 || w4_fills(l, b.n_out)) && wino4_supported(b.n_out, b.n_mid, l.T, l.H, l.W, 3);
}
bool want_wf_0(const i2v_dec* d, const Block& b, const Level& l) { return d->has32() && d->wino32 && wino4f32_supported(b.n_mid, b.n_in, l.T, l.H, l.W); }
bool want_wf_1(const i2v_dec* d, const Block& b, const Level& l) { return d->has32() && d->wino32 && wino4f32_supported(b.n_out, b.n_mid, l.T, l.H, l.W); }
bool use_wf_0(const i2v_dec* d, const Block& b, const Level& l) { return b.conv0_wf.u[0].w.p && want_wf_0(d, b, l); }
bool use_wf_1(const i2v_dec* d, const Block& b, const Level& l) { return b.conv1_wf.u[0].w.p && want_wf_1(d, b, l); }
bool use_w4_0(const i2v_dec* d, const Block& b, const Level& l) { return b.conv0_w4.w.p && want_w4_0(d, b, l); }
bool use_w4_1(const i2v_dec* d, const Block& b, const Level& l) { return b.conv1_w4.w.p && want_w4_1(d, b, l); }
bool use_wino0(const i2v_dec* d, const Block& b, const Level& l) { return b.conv0_w.w.p && want_wino0(d, b, l); }
bool use_wino1(const i2v_dec* d, const Block& b, const Level& l) { return b.conv1_w.w.p && want_wino1(d, b, l); }

}  // namespace

namespace i2v {
// exported to i2v_embed.hip
int stats_forward(const float* x, double* sums, int B, long P, int C, hipStream_t st) { return run_stats(x, sums, B, P, C, st); }
int coef_forward(const double* sums, float* coef, int B, int C, int groups, double count, hipStream_t st, const float* gw,
                 const float* gb) {
    return run_coef(sums, coef, B, C, groups, count, nullptr, 0, 0, gw, gb, st);
}
int resize_forward(const float* img, float* out, int B, int Hi, int Wi, int Ho, int Wo, hipStream_t st) {
    const long tot = (long)B * Ho * Wo;
    hipLaunchKernelGGL(resize_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 65536)), dim3(256), 0, st, img, out, B, Hi, Wi,
                       Ho, Wo, 0, static_cast<int*>(nullptr), (long)3 * Hi * Wi);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}
}  // namespace i2v

namespace {

struct BlockBufs {
    float *a, *dx, *xs_in, *xs_low, *y0, *y1, *gb, *coef;
    double *sums1, *sums2;        // sums1: statistics of the block INPUT (filled by the previous block's conv_1 epilogue or by run_stats)
    double* sums_out = nullptr;   // where conv_1's epilogue accumulates the statistics of the block OUTPUT (null: into sums1)
    float* splitk = nullptr;      // split-K scratch of conv16_forward (optional)
    size_t splitk_floats = 0;
    float* y1v = nullptr;         // Winograd operand of SPADE's 128-channel activation (2 x the size of y1; optional)
    const float* gb_ready = nullptr;   // this block's gamma | beta, already computed by i2v_dec_prepare
    float* m6 = nullptr;          // exact-fp32 Winograd scratch (six partial outputs); null: the direct kernel is used
    hipStream_t side = nullptr;   // the learned shortcut runs on this stream (events ev_x / ev_s of the handle), with coef_s
    float* coef_s = nullptr;
};

// SPADE's conditioning branch of one block (normalization_layer.py:20-23): resize(start frame) -> Conv2d(3, 128) + lrelu ->
// fused gamma | beta Conv2d(128, 2C) ("+1" folded into the gamma bias) -> gb [B][H][W][2C].  Depends on the start frame only.
int spade_branch(i2v_dec* d, Block& b, const Level& l, const float* img, int img_h, int img_w, int B, float* y0, float* y1, float* y1v_,
                 float* gb, hipStream_t st) {
    int rc;
    struct { float* y1v; } w{y1v_};
    {
        const long tot = (long)B * l.H * l.W;
        hipLaunchKernelGGL(resize_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 65536)), dim3(256), 0, st, img, y0,
                           B, img_h, img_w, l.H, l.W, d->aux16() ? 1 : 0, d->status_dev, d->img_bstride ? d->img_bstride : (long)3 * img_h * img_w);
        I2V_HIP_CHECK(hipGetLastError());
    }
    if (d->aux16() && b.sp_gb_w4.w.p && w.y1v) {
        if ((rc = conv16_forward(b.sp_conv16, y0, y1, nullptr, 1, 1, B, 1, l.H, l.W, EPI_LRELU, st))) return rc;
        if ((rc = run_modulate_wino4(y1, nullptr, nullptr, w.y1v, B, 1, l.H, l.W, 128, 1, 1, 0, st, d->status_dev))) return rc;
        if ((rc = wino4_forward(b.sp_gb_w4, w.y1v, gb, nullptr, 1, 1, B, 1, l.H, l.W, EPI_NONE, st, nullptr))) return rc;
    } else if (d->aux16() && b.sp_gb_w.w.p && w.y1v) {
        // gamma | beta conv on the Winograd kernel: the 128-channel activation goes through fp32 once more (the operand
        // writer needs the w-neighbours of every position, which the producing conv's epilogue does not hold)
        if ((rc = conv16_forward(b.sp_conv16, y0, y1, nullptr, 1, 1, B, 1, l.H, l.W, EPI_LRELU, st))) return rc;
        if ((rc = run_modulate_wino(y1, nullptr, nullptr, w.y1v, B, 1, l.H, l.W, 128, 1, 1, 0, st, d->status_dev))) return rc;
        if ((rc = wino16_forward(b.sp_gb_w, w.y1v, gb, nullptr, 1, 1, B, 1, l.H, l.W, EPI_NONE, st, nullptr))) return rc;
    } else if (d->aux16()) {
        if ((rc = conv16_forward(b.sp_conv16, y0, y1, nullptr, 1, 1, B, 1, l.H, l.W, EPI_LRELU | EPI_HL16, st, nullptr, d->status_dev)))
            return rc;
        if ((rc = conv16_forward(b.sp_gb16, y1, gb, nullptr, 1, 1, B, 1, l.H, l.W, EPI_NONE, st))) return rc;
    } else {
        if ((rc = conv_forward(b.sp_conv, y0, 16, y1, nullptr, 1, 1, B, 1, l.H, l.W, EPI_LRELU, st))) return rc;
        if ((rc = conv_forward(b.sp_gb, y1, 128, gb, nullptr, 1, 1, B, 1, l.H, l.W, EPI_NONE, st))) return rc;
    }
    return I2V_OK;
}

// One GeneratorBlock (decoder.py:33-52) on channels-last tensors: x [B][T/ut][H/us][W/us][n_in] -> xn [B][T][H][W][n_out].
// `last`: the block output only feeds conv_img(leaky_relu(.)) (decoder.py:117), so the activation is fused here.
int block_forward(i2v_dec* d, int k, Block& b, const Level& l, const float* x, float* xn, const float* img, int img_h, int img_w,
                  const float* zl, int zstride, int B, const BlockBufs& w, bool& x_stats_ready, bool last, hipStream_t st) {
    float *a = w.a, *dx = w.dx, *xs_in = w.xs_in, *xs_low = w.xs_low, *y0 = w.y0, *y1 = w.y1, *gb = w.gb, *coef = w.coef;
    double *sums1 = w.sums1, *sums2 = w.sums2;
    double* sums_out = w.sums_out ? w.sums_out : w.sums1;
    int rc;
    auto tap = [&](int k_, int which, const float* src, size_t count) -> int {
        if (d->tap_dst && d->tap_block == k_ && d->tap_which == which)
            I2V_HIP_CHECK(hipMemcpyAsync(d->tap_dst, src, std::min(count, d->tap_max) * 4, hipMemcpyDeviceToDevice, st));
        return I2V_OK;
    };
    const int Tl = l.T / l.ut, Hl = l.H / l.us, Wl = l.W / l.us;
    const long Pl = (long)Tl * Hl * Wl, P = (long)l.T * l.H * l.W;
    // GroupNorm statistics of the (virtually upsampled) block input == statistics of the low-res tensor; they are
    // already in sums1 when the previous block's conv_1 accumulated them in its epilogue
    if (!x_stats_ready && (rc = run_stats(x, sums1, B, Pl, b.n_in, st))) return rc;
    if ((rc = run_coef(sums1, coef, B, b.n_in, b.groups_spade, (double)Pl, nullptr, 0, 0, nullptr, nullptr, st))) return rc;
    // The learned shortcut depends on the block input and its statistics only: on the side stream it runs underneath the
    // modulate / conv_0 / modulate chain below (an HBM-bound GEMM next to matrix-core-bound convs); conv_1 waits for it.
    const bool side_shortcut = b.learned && w.side && w.coef_s && k < 6 && d->ev_x[k];
    if (side_shortcut) {
        I2V_HIP_CHECK(hipEventRecord(d->ev_x[k], st));
        I2V_HIP_CHECK(hipStreamWaitEvent(w.side, d->ev_x[k], 0));
        int rs_ = run_coef(sums1, w.coef_s, B, b.n_in, 16, (double)Pl, nullptr, 0, 0, b.gn_w.as<float>(), b.gn_b.as<float>(), w.side);
        if (!rs_) {
            if (d->aux16() && b.convs16.w.p) rs_ = pointwise16_forward(b.convs16, x, xs_low, nullptr, (long)B * Pl, Pl, EPI_NONE, w.side, w.coef_s, d->status_dev);
            else rs_ = conv_forward(b.convs, x, b.n_in, xs_low, nullptr, 1, 1, B, Tl, Hl, Wl, EPI_NONE, w.side, w.coef_s);
        }
        if (!rs_ && hipEventRecord(d->ev_s[k], w.side) != hipSuccess) rs_ = I2V_E_HIP;
        if (rs_) { (void)hipStreamSynchronize(w.side); return rs_; }
    }
    // SPADE branch (normalization_layer.py:20-23): depends on the start frame only -- either computed here, or already there
    // (w.gb_ready: i2v_dec_prepare ran it, typically on a side stream underneath the cINN pass)
    if (w.gb_ready) gb = const_cast<float*>(w.gb_ready);
    else if ((rc = spade_branch(d, b, l, img, img_h, img_w, B, y0, y1, w.y1v, gb, st))) return rc;
    if ((rc = tap(k, 0, gb, (size_t)B * l.H * l.W * 2 * b.n_in))) return rc;
    // per conv: split-fp16 or exact fp32 (mma = 0: all fp32; mma = auto: the layers the range guard switched, i2v_dec::fp32_layer)
    const bool f16_0 = d->layer16((2 * k) % 12), f16_1 = d->layer16((2 * k + 1) % 12);
    const bool tdup = f16_0 && b.tdup0;  // a0 is kept at the half temporal rate (its frames 2i and 2i+1 coincide)
    const bool q0 = f16_0 && use_w4_0(d, b, l), q1 = f16_1 && use_w4_1(d, b, l);          // F(4,3)
    const bool w0 = f16_0 && !q0 && use_wino0(d, b, l), w1 = f16_1 && !q1 && use_wino1(d, b, l);   // F(2,3)
    int* flag = d->status_dev;
    // range guard: the two operand tensors of this block publish their maxima in slots 1 + 2k / 2 + 2k
    int* um0 = f16_0 && flag ? flag + 1 + 2 * (k % 12) : nullptr;
    int* um1 = f16_1 && flag ? flag + 2 + 2 * (k % 12) : nullptr;
    const bool f0 = !f16_0 && w.m6 && use_wf_0(d, b, l), f1 = !f16_1 && w.m6 && use_wf_1(d, b, l);   // exact fp32: Winograd F(4,3) on the fp32 matrix cores
    // thin F(4,3) layers: the operand is generated by the conv kernel's producer waves (no writer launch, no V tensor)
    const bool g0 = d->gen == 1 && q0 && !tdup && l.ut == 1 && l.us == 2 && !d->tap_dst && wino4g_supported(b.n_mid, b.n_in, l.T, l.H, l.W, 2);
    const bool g1 = d->gen && q1 && !d->tap_dst && wino4g_supported(b.n_out, b.n_mid, l.T, l.H, l.W, 1);
    if (g0) rc = I2V_OK;
    else if (f0) rc = modulate_wino4_f32(x, coef, gb, a, B, l.T, l.H, l.W, b.n_in, l.ut, l.us, 1, st);
    else if (q0) rc = run_modulate_wino4(x, coef, gb, a, B, tdup ? l.T / 2 : l.T, l.H, l.W, b.n_in, tdup ? 1 : l.ut, l.us, 1, st, flag, um0);
    else if (w0) rc = run_modulate_wino(x, coef, gb, a, B, tdup ? l.T / 2 : l.T, l.H, l.W, b.n_in, tdup ? 1 : l.ut, l.us, 1, st, flag, um0);
    else if (tdup) rc = run_modulate(x, coef, gb, a, B, l.T / 2, l.H, l.W, b.n_in, 1, l.us, 1, st, true, flag, um0);
    else rc = run_modulate(x, coef, gb, a, B, l.T, l.H, l.W, b.n_in, l.ut, l.us, 1, st, f16_0, flag, um0);
    if (rc) return rc;
    if (!f0 && !g0 && (rc = tap(k, 1, a, (size_t)B * (tdup ? P / 2 : P) * b.n_in))) return rc;
    const bool fuse = f16_0 && conv16_can_fuse_stats(tdup ? l.T / 2 : l.T, l.H, l.W);
    d->prof_cur_layer = 2 * k;
    d->prof_cur_kernel = g0 ? 4 : q0 ? 3 : w0 ? 2 : (f16_0 ? 1 : 0);
    if (g0) rc = conv3_w4g(d, b.conv0_w4, x, coef, gb, 2, dx, nullptr, 1, 1, B, l, EPI_NONE, st, fuse ? sums2 : nullptr, flag, um0);
    else if (f0) rc = conv3_wf(d, b.conv0_wf, a, w.m6, dx, nullptr, 1, 1, B, l, EPI_NONE, st);
    else if (q0) rc = conv3_w4(d, b.conv0_w4, a, dx, nullptr, 1, 1, B, l, EPI_NONE, st, fuse ? sums2 : nullptr);
    else if (w0) rc = conv3_w(d, b.conv0_w, a, dx, nullptr, 1, 1, B, l, EPI_NONE, st, fuse ? sums2 : nullptr);
    else if (f16_0) rc = conv3_16(d, b.conv0_16, a, dx, nullptr, 1, 1, B, l, EPI_NONE, st, fuse ? sums2 : nullptr, w.splitk, w.splitk_floats);
    else rc = conv3(d, b.conv0, a, dx, nullptr, 1, 1, B, l, EPI_NONE, st);
    if (rc) return rc;
    if ((rc = tap(k, 2, dx, (size_t)B * P * b.n_mid))) return rc;
    // ADAIN (normalization_layer.py:47-51) + leaky_relu
    if (!fuse && (rc = run_stats(dx, sums2, B, P, b.n_mid, st))) return rc;
    if ((rc = run_coef(sums2, coef, B, b.n_mid, b.n_mid, (double)P, zl, zstride, b.zoff, nullptr, nullptr, st))) return rc;
    if (g1) rc = I2V_OK;
    else if (f1) rc = modulate_wino4_f32(dx, coef, nullptr, a, B, l.T, l.H, l.W, b.n_mid, 1, 1, 1, st);
    else if (q1) rc = run_modulate_wino4(dx, coef, nullptr, a, B, l.T, l.H, l.W, b.n_mid, 1, 1, 1, st, flag, um1);
    else if (w1) rc = run_modulate_wino(dx, coef, nullptr, a, B, l.T, l.H, l.W, b.n_mid, 1, 1, 1, st, flag, um1);
    else rc = run_modulate(dx, coef, nullptr, a, B, l.T, l.H, l.W, b.n_mid, 1, 1, 1, st, f16_1, flag, um1);
    if (rc) return rc;
    if (!f1 && !g1 && (rc = tap(k, 3, a, (size_t)B * P * b.n_mid))) return rc;
    // shortcut (decoder.py:44-49) at low resolution
    const float* res = x;
    if (b.learned && !side_shortcut) {
        if ((rc = run_coef(sums1, coef, B, b.n_in, 16, (double)Pl, nullptr, 0, 0, b.gn_w.as<float>(), b.gn_b.as<float>(), st)))
            return rc;
        // Norm3D folded into the 1x1x1 conv's loads (no padding taps -> exact): no normalised copy of x is written
        (void)xs_in;
        if (d->aux16() && b.convs16.w.p) rc = pointwise16_forward(b.convs16, x, xs_low, nullptr, (long)B * Pl, Pl, EPI_NONE, st, coef, d->status_dev);
        else rc = conv_forward(b.convs, x, b.n_in, xs_low, nullptr, 1, 1, B, Tl, Hl, Wl, EPI_NONE, st, coef);
        if (rc) return rc;
        res = xs_low;
        if ((rc = tap(k, 4, xs_low, (size_t)B * Pl * b.n_out))) return rc;
    } else if (b.learned) {
        res = xs_low;
        I2V_HIP_CHECK(hipStreamWaitEvent(st, d->ev_s[k], 0));   // enqueued on the side stream at the top of the block
    }
    // g_4's output only feeds conv_img(leaky_relu(x)) (decoder.py:117): fuse the activation here
    // (the shortcut's coefficients were derived from sums1 above, so conv_1 may now overwrite sums1 with the
    // statistics of the block OUTPUT = the next block's input)
    const bool fuse_out = f16_1 && conv16_can_fuse_stats(l.T, l.H, l.W) && !last;
    d->prof_cur_layer = 2 * k + 1;
    d->prof_cur_kernel = g1 ? 4 : q1 ? 3 : w1 ? 2 : (f16_1 ? 1 : 0);
    if (g1) rc = conv3_w4g(d, b.conv1_w4, dx, coef, nullptr, 1, xn, res, l.ut, l.us, B, l, last ? EPI_LRELU : EPI_NONE, st, fuse_out ? sums_out : nullptr, flag, um1);
    else if (f1) rc = conv3_wf(d, b.conv1_wf, a, w.m6, xn, res, l.ut, l.us, B, l, last ? EPI_LRELU : EPI_NONE, st);
    else if (q1) rc = conv3_w4(d, b.conv1_w4, a, xn, res, l.ut, l.us, B, l, last ? EPI_LRELU : EPI_NONE, st, fuse_out ? sums_out : nullptr);
    else if (w1) rc = conv3_w(d, b.conv1_w, a, xn, res, l.ut, l.us, B, l, last ? EPI_LRELU : EPI_NONE, st, fuse_out ? sums_out : nullptr);
    else if (f16_1) rc = conv3_16(d, b.conv1_16, a, xn, res, l.ut, l.us, B, l, last ? EPI_LRELU : EPI_NONE, st, fuse_out ? sums_out : nullptr,
                                w.splitk, w.splitk_floats);
    else rc = conv3(d, b.conv1, a, xn, res, l.ut, l.us, B, l, last ? EPI_LRELU : EPI_NONE, st);
    if (rc) return rc;
    x_stats_ready = fuse_out;
    if ((rc = tap(k, 5, xn, (size_t)B * P * b.n_out))) return rc;
    return I2V_OK;
}

// sigma = u . (W_mat v), W_mat = weight_orig.reshape(Cout, -1); signed, no abs (torch spectral_norm, eval mode)
int sn_scale(const StateDict& sd, const std::string& name, bool spectral, int cout, int64_t kk, const float** w_out,
             double* scale_out) {
    if (!spectral) {
        *w_out = sd.f32(name + ".weight", (int64_t)cout * kk);
        *scale_out = 1.0;
        return *w_out ? I2V_OK : I2V_E_MISSING;
    }
    const float* w = sd.f32(name + ".weight_orig", (int64_t)cout * kk);
    const float* u = sd.f32(name + ".weight_u", cout);
    const float* v = sd.f32(name + ".weight_v", kk);
    if (!w || !u || !v) return I2V_E_MISSING;
    double sigma = 0.0;
    for (int n = 0; n < cout; ++n) {
        double r = 0.0;
        const float* row = w + (size_t)n * kk;
        for (int64_t j = 0; j < kk; ++j) r += (double)row[j] * v[j];
        sigma += r * u[n];
    }
    I2V_REQUIRE(sigma != 0.0 && std::isfinite(sigma), I2V_E_INVALID, "spectral norm sigma of %s is %g", name.c_str(), sigma);
    *w_out = w;
    *scale_out = 1.0 / sigma;
    return I2V_OK;
}

template <class WT>
int sn_pack(const StateDict& sd, const std::string& name, bool spectral, int cout, int cin, int k, bool has_bias,
            WT& out) {
    const float* bias = nullptr;
    if (has_bias) { bias = sd.f32(name + ".bias", cout); if (!bias) return I2V_E_MISSING; }
    const float* w = nullptr;
    double scale = 1.0;
    int rc = sn_scale(sd, name, spectral, cout, (int64_t)cin * k * k * k, &w, &scale);
    if (rc) return rc;
    return out.pack(w, bias, cout, cin, k, k, k, scale);
}

int sn_pack_wf(const StateDict& sd, const std::string& name, bool spectral, int cout, int cin, Wino4F32Weights& out) {
    const float* bias = sd.f32(name + ".bias", cout);
    if (!bias) return I2V_E_MISSING;
    const float* w = nullptr;
    double scale = 1.0;
    int rc = sn_scale(sd, name, spectral, cout, (int64_t)cin * 27, &w, &scale);
    if (rc) return rc;
    return out.pack(w, bias, cout, cin, scale);
}

// conv_0 of a block that sits behind a x2 temporal up-sampling: packed for the half-rate input (Conv16Weights::pack_tdup)
int sn_pack_tdup(const StateDict& sd, const std::string& name, bool spectral, int cout, int cin, Conv16Weights& out) {
    const float* bias = sd.f32(name + ".bias", cout);
    if (!bias) return I2V_E_MISSING;
    const float* w = nullptr;
    double scale = 1.0;
    int rc = sn_scale(sd, name, spectral, cout, (int64_t)cin * 27, &w, &scale);
    if (rc) return rc;
    return out.pack_tdup(w, bias, cout, cin, scale);
}

// device binding + the sticky range flag (device word and pinned host mirror)
int init_status(i2v_dec* d) {
    I2V_HIP_CHECK(hipGetDevice(&d->device));
    { const char* zp = nullptr; if (int rcz = zero_page(&zp)) return rcz; }  // the conv kernels' zero page: allocated here, not inside a forward
    I2V_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d->status_dev), I2V_STATUS_WORDS * sizeof(int)));   // [0] flags, [1..] underflow maxima
    I2V_HIP_CHECK(hipMemset(d->status_dev, 0, I2V_STATUS_WORDS * sizeof(int)));
    I2V_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d->status_host), I2V_STATUS_WORDS * sizeof(int), hipHostMallocDefault));   // [0] flags, [32 + i] the last forward's maxima
    std::memset(d->status_host, 0, I2V_STATUS_WORDS * sizeof(int));
    return I2V_OK;
}

// entry check of every call that enqueues work: right device, and no overflow reported by an earlier call
int check_entry(i2v_dec* d, const char* what) {
    I2V_REQUIRE_DEVICE(d->device, what);
    I2V_REQUIRE(!(*static_cast<volatile int*>(d->status_host) & 1), I2V_E_RANGE,
                "%s: an earlier call on this handle produced activations outside the fp16 range of the split-fp16 operand "
                "format (|x| > 65504 or non-finite); its output is invalid.  Use the exact-fp32 mode (mma = 0 / I2V_DEC_MMA=0) "
                "for this checkpoint, or clear the flag with i2v_dec_status(reset = 1)", what);
    return I2V_OK;
}

int sn_pack_wino(const StateDict& sd, const std::string& name, bool spectral, int cout, int cin, bool tdup, Wino16Weights& out) {
    const float* bias = sd.f32(name + ".bias", cout);
    if (!bias) return I2V_E_MISSING;
    const float* w = nullptr;
    double scale = 1.0;
    int rc = sn_scale(sd, name, spectral, cout, (int64_t)cin * 27, &w, &scale);
    if (rc) return rc;
    return tdup ? out.pack_tdup(w, bias, cout, cin, scale) : out.pack(w, bias, cout, cin, 3, scale);
}

int sn_pack_wino4(const StateDict& sd, const std::string& name, bool spectral, int cout, int cin, bool tdup, Wino4Weights& out) {
    const float* bias = sd.f32(name + ".bias", cout);
    if (!bias) return I2V_E_MISSING;
    const float* w = nullptr;
    double scale = 1.0;
    int rc = sn_scale(sd, name, spectral, cout, (int64_t)cin * 27, &w, &scale);
    if (rc) return rc;
    return tdup ? out.pack_tdup(w, bias, cout, cin, scale) : out.pack(w, bias, cout, cin, scale);
}

}  // namespace

extern "C" {

int i2v_dec_create(const i2v_dec_cfg* cfg, i2v_dec** out) {
    I2V_REQUIRE(cfg && out, I2V_E_INVALID, "i2v_dec_create: null argument");
    I2V_REQUIRE(cfg->channel_factor > 0 && cfg->channel_factor % 8 == 0, I2V_E_INVALID,
                "i2v_dec_create: channel_factor must be a multiple of 8 (Norm3D uses 16 groups), got %d", cfg->channel_factor);
    I2V_REQUIRE(cfg->channel_factor * 16 <= 1024, I2V_E_INVALID, "i2v_dec_create: channel_factor %d too large", cfg->channel_factor);
    I2V_REQUIRE(cfg->z_dim > 0 && cfg->z_dim % 4 == 0, I2V_E_INVALID, "i2v_dec_create: z_dim must be a multiple of 4");
    for (int i = 0; i < 2; ++i) {
        const int s = cfg->upsample_s[i], t = cfg->upsample_t[i];
        I2V_REQUIRE((s == 1 || s == 2 || s == 4) && (t == 1 || t == 2 || t == 4), I2V_E_INVALID,
                    "i2v_dec_create: upsample factors must be 1, 2 or 4");
    }
    I2V_REQUIRE(cfg->mma == 0 || cfg->mma == 1 || cfg->mma == 2, I2V_E_INVALID, "i2v_dec_create: unknown mma mode %d (0 fp32, 1 split-fp16, 2 auto)", cfg->mma);
    int ndev = 0;
    I2V_HIP_CHECK(hipGetDeviceCount(&ndev));
    I2V_REQUIRE(ndev > 0, I2V_E_HIP, "i2v_dec_create: no HIP device");
    auto d = std::make_unique<i2v_dec>();
    d->cfg = *cfg;
    if (const char* e = std::getenv("I2V_DEC_WINO")) d->wino = std::atoi(e) != 0;
    if (const char* e = std::getenv("I2V_DEC_WINO4")) d->wino4 = std::atoi(e);
    if (const char* e = std::getenv("I2V_DEC_PW16")) d->pw16 = std::atoi(e) != 0;
    if (const char* e = std::getenv("I2V_DEC_IMG16")) d->img16 = std::atoi(e);
    if (const char* e = std::getenv("I2V_DEC_SPW")) d->spw = std::atoi(e) != 0;
    if (const char* e = std::getenv("I2V_DEC_WINO32")) d->wino32 = std::atoi(e) != 0;
    if (const char* e = std::getenv("I2V_DEC_GEN")) d->gen = std::atoi(e);   // 1: conv_0 and conv_1 of the thin level, 2: conv_1 only
    if (const char* e = std::getenv("I2V_DEC_OVERLAP")) { d->overlap = std::atoi(e) != 0; d->no_side_shortcut = std::atoi(e) == 2; }
    if (const char* e = std::getenv("I2V_DEC_SUB")) d->sub = std::max(0, std::atoi(e));
    if (int rc = init_status(d.get())) return rc;
    const int nf = d->nf = cfg->channel_factor;
    const char* names[6] = {"head_0", "g_0", "g_1", "g_2", "g_3", "g_4"};
    const int cin[6] = {16, 16, 16, 8, 4, 2}, cout[6] = {16, 16, 8, 4, 2, 1};
    int T = 1, S = 4, zoff = 0;
    for (int k = 0; k < 6; ++k) {
        Block& b = d->blk[k];
        b.name = names[k];
        b.n_in = cin[k] * nf; b.n_out = cout[k] * nf; b.n_mid = std::min(b.n_in, b.n_out);
        b.learned = b.n_in != b.n_out;
        int g = 16;
        while (b.n_in % g) --g;  // normalization_layer.py:9-10
        b.groups_spade = g;
        b.zoff = zoff;
        zoff += 2 * b.n_mid;
        int ut = 1, us = 1;
        if (k >= 1 && k <= 3) { ut = 2; us = 2; }                                  // decoder.py:102-108
        if (k == 4) { ut = cfg->upsample_t[0]; us = cfg->upsample_s[0]; }          // :111
        if (k == 5) { ut = cfg->upsample_t[1]; us = cfg->upsample_s[1]; }          // :114
        T *= ut; S *= us;
        d->lvl[k] = Level{T, S, S, ut, us};
    }
    d->Nz = zoff;
    *out = d.release();
    return I2V_OK;
}

void i2v_dec_destroy(i2v_dec* d) { delete d; }

int i2v_dec_load(i2v_dec* d, const i2v_tensor* tensors, int32_t n_tensors) {
    I2V_REQUIRE(d && tensors && n_tensors > 0, I2V_E_INVALID, "i2v_dec_load: null argument");
    I2V_REQUIRE_DEVICE(d->device, "i2v_dec_load");
    StateDict sd(tensors, n_tensors);
    const int nf = d->nf, zd = d->cfg.z_dim;
    const bool sn = d->cfg.spectral_norm != 0;
    int rc;
    {   // fc: Linear(z_dim, 4*4*16nf) (decoder.py:72); rows permuted so the output is channels-last [h][w][c]
        const int C = 16 * nf;
        const float* w = sd.f32("fc.weight", (int64_t)16 * C * zd);
        const float* b = sd.f32("fc.bias", (int64_t)16 * C);
        if (!w || !b) return I2V_E_MISSING;
        std::vector<float> wp((size_t)16 * C * zd), bp((size_t)16 * C);
        for (int c = 0; c < C; ++c)
            for (int hw = 0; hw < 16; ++hw) {
                std::memcpy(&wp[((size_t)hw * C + c) * zd], w + ((size_t)c * 16 + hw) * zd, (size_t)zd * 4);
                bp[(size_t)hw * C + c] = b[(size_t)c * 16 + hw];
            }
        if ((rc = d->fc.pack(wp.data(), bp.data(), 16 * C, zd, 1, 1, 1, 1.0))) return rc;
    }
    std::vector<float> zw((size_t)d->Nz * zd), zb((size_t)d->Nz);
    for (int k = 0; k < 6; ++k) {
        Block& b = d->blk[k];
        const std::string p = b.name + ".";
        if (d->has16()) {
            // behind a x2 up-sampling in time, SPADE's output is identical for frames 2i and 2i+1 (gamma/beta do not depend
            // on t): conv_0 runs on the half-rate tensor with two pre-summed 2-tap temporal kernels (-1/3 of its MACs)
            b.tdup0 = d->lvl[k].ut == 2;
            // layers whose shape allows it run on the Winograd kernel (1.5x fewer MFMAs), the rest on the direct one
            if (want_w4_0(d, b, d->lvl[k])) rc = sn_pack_wino4(sd, p + "conv_0", sn, b.n_mid, b.n_in, b.tdup0, b.conv0_w4);
            else if (want_wino0(d, b, d->lvl[k])) rc = sn_pack_wino(sd, p + "conv_0", sn, b.n_mid, b.n_in, b.tdup0, b.conv0_w);
            else if (b.tdup0) rc = sn_pack_tdup(sd, p + "conv_0", sn, b.n_mid, b.n_in, b.conv0_16);
            else rc = sn_pack(sd, p + "conv_0", sn, b.n_mid, b.n_in, 3, true, b.conv0_16);
            if (rc) return rc;
            if (want_w4_1(d, b, d->lvl[k])) rc = sn_pack_wino4(sd, p + "conv_1", sn, b.n_out, b.n_mid, false, b.conv1_w4);
            else if (want_wino1(d, b, d->lvl[k])) rc = sn_pack_wino(sd, p + "conv_1", sn, b.n_out, b.n_mid, false, b.conv1_w);
            else rc = sn_pack(sd, p + "conv_1", sn, b.n_out, b.n_mid, 3, true, b.conv1_16);
            if (rc) return rc;
        }
        if (d->has32()) {   // (mma = auto packs both sets)
            if ((rc = sn_pack(sd, p + "conv_0", sn, b.n_mid, b.n_in, 3, true, b.conv0))) return rc;
            if ((rc = sn_pack(sd, p + "conv_1", sn, b.n_out, b.n_mid, 3, true, b.conv1))) return rc;
            // from the 8x8 level on: Winograd F(4,3) on the fp32 matrix cores (half the MFMA work of the 27-tap kernel)
            if (want_wf_0(d, b, d->lvl[k]) && (rc = sn_pack_wf(sd, p + "conv_0", sn, b.n_mid, b.n_in, b.conv0_wf))) return rc;
            if (want_wf_1(d, b, d->lvl[k]) && (rc = sn_pack_wf(sd, p + "conv_1", sn, b.n_out, b.n_mid, b.conv1_wf))) return rc;
        }
        if (b.learned) {
            if ((rc = sn_pack(sd, p + "conv_s", sn, b.n_out, b.n_in, 1, false, b.convs))) return rc;
            if (d->has16() && d->pw16 && (rc = sn_pack(sd, p + "conv_s", sn, b.n_out, b.n_in, 1, false, b.convs16))) return rc;
            const float* gw = sd.f32(p + "norm_s.bn.weight", b.n_in);
            const float* gb = sd.f32(p + "norm_s.bn.bias", b.n_in);
            if (!gw || !gb) return I2V_E_MISSING;
            if ((rc = b.gn_w.upload(gw, (size_t)b.n_in * 4))) return rc;
            if ((rc = b.gn_b.upload(gb, (size_t)b.n_in * 4))) return rc;
        }
        // Spade: Conv2d(3,128,3) then conv_gamma | conv_beta fused as one Conv2d(128, 2C, 3)
        const float* w1 = sd.f32(p + "norm_0.conv.weight", 128 * 3 * 9);
        const float* b1 = sd.f32(p + "norm_0.conv.bias", 128);
        const float* wg = sd.f32(p + "norm_0.conv_gamma.weight", (int64_t)b.n_in * 128 * 9);
        const float* bg = sd.f32(p + "norm_0.conv_gamma.bias", b.n_in);
        const float* wb = sd.f32(p + "norm_0.conv_beta.weight", (int64_t)b.n_in * 128 * 9);
        const float* bb = sd.f32(p + "norm_0.conv_beta.bias", b.n_in);
        if (!w1 || !b1 || !wg || !bg || !wb || !bb) return I2V_E_MISSING;
        if ((rc = b.sp_conv.pack(w1, b1, 128, 3, 1, 3, 3, 1.0))) return rc;
        {   // the same conv for the split-fp16 path: input channels padded 3 -> 16 (the resize kernel's row)
            std::vector<float> w16((size_t)128 * 16 * 9, 0.f);
            for (int n = 0; n < 128; ++n)
                for (int c = 0; c < 3; ++c)
                    for (int t = 0; t < 9; ++t) w16[((size_t)n * 16 + c) * 9 + t] = w1[((size_t)n * 3 + c) * 9 + t];
            if ((rc = b.sp_conv16.pack(w16.data(), b1, 128, 16, 1, 3, 3, 1.0))) return rc;
        }
        std::vector<float> wgb((size_t)2 * b.n_in * 128 * 9), bgb((size_t)2 * b.n_in);
        std::memcpy(wgb.data(), wg, (size_t)b.n_in * 128 * 9 * 4);
        std::memcpy(wgb.data() + (size_t)b.n_in * 128 * 9, wb, (size_t)b.n_in * 128 * 9 * 4);
        for (int c = 0; c < b.n_in; ++c) { bgb[c] = bg[c] + 1.0f; bgb[b.n_in + c] = bb[c]; }  // normalized*(1+gamma)+beta
        if (d->has16() && (rc = b.sp_gb16.pack(wgb.data(), bgb.data(), 2 * b.n_in, 128, 1, 3, 3, 1.0))) return rc;
        if (d->has32() && (rc = b.sp_gb.pack(wgb.data(), bgb.data(), 2 * b.n_in, 128, 1, 3, 3, 1.0))) return rc;
        if (spade_w4_wanted(d, b, d->lvl[k])) {
            if ((rc = b.sp_gb_w4.pack(wgb.data(), bgb.data(), 2 * b.n_in, 128, 1.0, 1))) return rc;
        } else if (spade_wino_wanted(d, b, d->lvl[k]) && (rc = b.sp_gb_w.pack(wgb.data(), bgb.data(), 2 * b.n_in, 128, 1, 1.0)))
            return rc;
        // ADAIN linear rows into the shared z-GEMM
        const float* lw = sd.f32(p + "norm_1.linear.weight", (int64_t)2 * b.n_mid * zd);
        const float* lb = sd.f32(p + "norm_1.linear.bias", (int64_t)2 * b.n_mid);
        if (!lw || !lb) return I2V_E_MISSING;
        std::memcpy(&zw[(size_t)b.zoff * zd], lw, (size_t)2 * b.n_mid * zd * 4);
        std::memcpy(&zb[b.zoff], lb, (size_t)2 * b.n_mid * 4);
    }
    if ((rc = d->zlin.pack(zw.data(), zb.data(), d->Nz, zd, 1, 1, 1, 1.0))) return rc;
    {
        const float* w = sd.f32("conv_img.weight", (int64_t)3 * nf * 27);
        const float* b = sd.f32("conv_img.bias", 3);
        if (!w || !b) return I2V_E_MISSING;
        if ((rc = d->conv_img.pack(w, b, 3, nf, 3, 3, 3, 1.0))) return rc;
        if ((rc = d->conv_img_v.pack(w, b, nf))) return rc;
        // (measured: 1.3 vs 1.7 ms per B = 64 BAIR pass at nf = 64, but 0.8 ms SLOWER than the vector-ALU kernel per B = 32
        //  128x128 pass at nf = 32, where the 81 planes outweigh the 32-channel input)
        if (d->has16() && d->img16 == 2 && conv_img_mfma_supported(d->lvl[5].T, d->lvl[5].H, d->lvl[5].W, nf) &&
            (rc = d->conv_img_m.pack(w, b, nf))) return rc;
        if (d->has16() && d->img16 == 1 && nf >= 64 && nf % 4 == 0) {
            std::vector<float> w81((size_t)81 * nf);   // row tap * 3 + n = w[n][:][tap]
            for (int n = 0; n < 3; ++n)
                for (int c = 0; c < nf; ++c)
                    for (int tap = 0; tap < 27; ++tap) w81[(size_t)(tap * 3 + n) * nf + c] = w[((size_t)n * nf + c) * 27 + tap];
            if ((rc = d->conv_img16.pack(w81.data(), nullptr, 81, nf, 1, 1, 1, 1.0))) return rc;
            if ((rc = d->conv_img_bias.upload(b, 3 * 4))) return rc;
        }
    }
    d->loaded = true;
    return I2V_OK;
}

int i2v_dec_out_shape(const i2v_dec* d, int32_t* t, int32_t* h, int32_t* w) {
    I2V_REQUIRE(d && t && h && w, I2V_E_INVALID, "i2v_dec_out_shape: null argument");
    *t = d->lvl[5].T; *h = d->lvl[5].H; *w = d->lvl[5].W;
    return I2V_OK;
}

size_t i2v_dec_workspace_bytes(const i2v_dec* d, int32_t batch, int32_t img_h, int32_t img_w) {
    (void)img_h; (void)img_w;
    if (!d || batch <= 0) return 0;
    return dec_ws(d, batch).total;
}

double i2v_dec_flops_per_sample(const i2v_dec* d, int32_t img_h, int32_t img_w) {
    (void)img_h; (void)img_w;
    if (!d) return 0.0;
    double f = 2.0 * d->cfg.z_dim * (16.0 * 16 * d->nf + d->Nz);
    for (int k = 0; k < 6; ++k) {
        const Block& b = d->blk[k];
        const Level& l = d->lvl[k];
        const double P = (double)l.T * l.H * l.W, HW = (double)l.H * l.W;
        f += 2.0 * P * 27.0 * ((double)b.n_in * b.n_mid + (double)b.n_mid * b.n_out);
        if (b.learned) f += 2.0 * P * (double)b.n_in * b.n_out;  // counted at output resolution, as the reference runs it
        f += 2.0 * HW * 9.0 * (3.0 * 128 + 128.0 * 2 * b.n_in);
    }
    const Level& l = d->lvl[5];
    f += 2.0 * l.T * l.H * l.W * 27.0 * d->nf * 3;
    return f;
}

int i2v_dec_set_profile(i2v_dec* d, int32_t on) {
    I2V_REQUIRE(d, I2V_E_INVALID, "i2v_dec_set_profile: null");
    d->profile = on;
    d->prof_conv3_ms = d->prof_conv3_flops = d->prof_conv3_exec = 0;
    d->prof_conv3_launches = 0;
    for (auto& pl : d->prof_layers) pl = i2v_dec::ProfLayer{};
    return I2V_OK;
}

int i2v_dec_debug_tap(i2v_dec* d, int32_t block, int32_t which, float* dst, size_t max_floats) {
    I2V_REQUIRE(d, I2V_E_INVALID, "i2v_dec_debug_tap: null");
    d->tap_block = block; d->tap_which = which; d->tap_dst = dst; d->tap_max = max_floats;
    return I2V_OK;
}

int i2v_dec_get_profile(i2v_dec* d, double* conv3_ms, double* conv3_flops, double* conv3_mfma_flops, int64_t* conv3_launches) {
    I2V_REQUIRE(d, I2V_E_INVALID, "i2v_dec_get_profile: null");
    for (auto& ev : d->prof_events) {
        I2V_HIP_CHECK(hipEventSynchronize(ev.e1));
        float ms = 0;
        I2V_HIP_CHECK(hipEventElapsedTime(&ms, ev.e0, ev.e1));
        d->prof_conv3_ms += ms;
        d->prof_conv3_flops += ev.flops;
        d->prof_conv3_exec += ev.exec_flops;
        d->prof_conv3_launches += 1;
        if (ev.layer >= 0 && ev.layer < 12) {
            auto& pl = d->prof_layers[ev.layer];
            pl.ms += ms; pl.flops += ev.flops; pl.exec_flops += ev.exec_flops; pl.launches += 1;
        }
        (void)hipEventDestroy(ev.e0);
        (void)hipEventDestroy(ev.e1);
    }
    d->prof_events.clear();
    if (conv3_ms) *conv3_ms = d->prof_conv3_ms;
    if (conv3_flops) *conv3_flops = d->prof_conv3_flops;
    if (conv3_mfma_flops) *conv3_mfma_flops = d->prof_conv3_exec;
    if (conv3_launches) *conv3_launches = d->prof_conv3_launches;
    return I2V_OK;
}

int i2v_dec_get_layer_profile(i2v_dec* d, int32_t layer, char* name, int32_t name_len, double* ms, double* flops,
                              double* mfma_flops, int64_t* launches, int32_t* kernel) {
    I2V_REQUIRE(d && layer >= 0 && layer < 12, I2V_E_INVALID, "i2v_dec_get_layer_profile: layer index %d", layer);
    if (int rc = i2v_dec_get_profile(d, nullptr, nullptr, nullptr, nullptr)) return rc;  // resolves pending event pairs
    const auto& pl = d->prof_layers[layer];
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s.conv_%d", d->blk[layer / 2].name.c_str(), layer & 1);
    if (ms) *ms = pl.ms;
    if (flops) *flops = pl.flops;
    if (mfma_flops) *mfma_flops = pl.exec_flops;
    if (launches) *launches = pl.launches;
    if (kernel) *kernel = pl.kernel;
    return I2V_OK;
}

}  // extern "C"

// The SPADE conditioning branches of all six blocks on the handle's side stream, forked from `st` by an event (everything the
// caller enqueued on `st` before -- the start frames, an earlier forward on this workspace -- is complete before the side stream
// touches the workspace); one event per level for the consumer.  *done = false: not possible here (graph capture on `st`, debug tap
// active, overlap switched off) -- the caller runs the branches inline.
static int fork_spade(i2v_dec* d, const DecWs& L, char* ws, const float* img, int img_h, int img_w, int B, hipStream_t st, bool* done) {
    *done = false;
    if (!d->overlap || d->tap_dst) return I2V_OK;
    if (stream_is_capturing(st)) return I2V_OK;
    if (d->side && d->side == st) return I2V_OK;   // the caller runs this call ON the shared side stream: nothing to fork to
    if (!d->side) {
        I2V_HIP_CHECK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
        d->side_owned = true;
    }
    if (!d->ev_fork) {
        I2V_HIP_CHECK(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
        for (auto& e : d->ev_lvl) I2V_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : d->ev_x) I2V_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : d->ev_s) I2V_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    I2V_HIP_CHECK(hipEventRecord(d->ev_fork, st));
    I2V_HIP_CHECK(hipStreamWaitEvent(d->side, d->ev_fork, 0));
    int rc = I2V_OK;
    // A SHARED side stream (i2v_dec_set_side_stream: the caller's cINN prefetch runs on it) typically has the NEXT batch's cINN pass
    // queued in front of these branches: the two tiny first levels (4x4, 8x8 maps: the maps head_0 and g_0 wait for) then run inline on
    // the caller's stream -- with their own scratch -- so that the main chain does not stall behind that pass; the branches of the
    // later levels have the first two blocks' time to get through.  (Own side stream: everything on it, as before.)
    const int k0 = d->side_owned ? 0 : 2;
    for (int k = 0; k < k0 && !rc; ++k) {
        rc = spade_branch(d, d->blk[k], d->lvl[k], img, img_h, img_w, B, F(L.y0), F(L.y1), L.has_y1v ? F(L.y1v) : nullptr, F(L.gbs[k]), st);
        if (!rc && hipEventRecord(d->ev_lvl[k], st) != hipSuccess) rc = I2V_E_HIP;
    }
    for (int k = k0; k < 6 && !rc; ++k) {
        rc = spade_branch(d, d->blk[k], d->lvl[k], img, img_h, img_w, B, F(L.py0), F(L.py1), L.has_y1v ? F(L.py1v) : nullptr, F(L.gbs[k]), d->side);
        if (!rc && hipEventRecord(d->ev_lvl[k], d->side) != hipSuccess) rc = I2V_E_HIP;
    }
    // (an error must not leave the caller's stream ahead of work this call put on the side stream)
    if (rc) { (void)hipStreamSynchronize(d->side); return rc; }
    d->side_unjoined = true;
    *done = true;
    return I2V_OK;
}

extern "C" {

int i2v_dec_forward(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, const float* motion, float* out,
                    void* workspace, size_t workspace_bytes, int32_t batch, void* stream) {
    return i2v_dec_forward_strided(d, img, img_h, img_w, 0, motion, out, 0, workspace, workspace_bytes, batch, stream);
}

}  // extern "C"

static int dec_forward_once(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, int64_t img_bstride, const float* motion, float* out,
                            int64_t out_bstride, void* workspace, size_t workspace_bytes, int32_t batch, void* stream) {
    // One prepare serves at most the NEXT forward on the handle: whatever happens below (range error of the previous call, bad
    // argument, workspace too small), a prepared set of SPADE maps must never survive into a later call, whose start frames can
    // sit at the same address (caching allocators, buffers refilled in place).
    const float* prep_img = d ? d->prep_img : nullptr;
    if (d) d->prep_img = nullptr;
    I2V_REQUIRE(d && d->loaded, I2V_E_STATE, "i2v_dec_forward: weights not loaded");
    if (int rc0 = check_entry(d, "i2v_dec_forward")) return rc0;
    I2V_REQUIRE(img && motion && out && workspace && batch > 0 && img_h > 0 && img_w > 0, I2V_E_INVALID,
                "i2v_dec_forward: null argument or bad size");
    const int B = batch;
    {
        const Level& lo = d->lvl[5];
        const int64_t img_dense = (int64_t)3 * img_h * img_w, out_dense = (int64_t)lo.T * 3 * lo.H * lo.W;
        I2V_REQUIRE((img_bstride == 0 || img_bstride >= img_dense) && (out_bstride == 0 || out_bstride >= out_dense), I2V_E_INVALID,
                    "i2v_dec_forward_strided: sample strides %lld / %lld below the dense %lld / %lld", (long long)img_bstride,
                    (long long)out_bstride, (long long)img_dense, (long long)out_dense);
        if (img_bstride == 0) img_bstride = img_dense;
        if (out_bstride == 0) out_bstride = out_dense;
        d->img_bstride = img_bstride;
    }
    const DecWs L = dec_ws(d, B);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_dec_forward: workspace %zu < required %zu", workspace_bytes,
                L.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rco = d->order.entry(st)) return rco;
    StreamOrderMark mark{&d->order, st};   // (declared before Join: runs after the join)
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    float *xA = F(L.xA), *xB = F(L.xB), *a = F(L.a), *dx = F(L.dx), *xs_in = F(L.xs_in), *xs_low = F(L.xs_low);
    float *y0 = F(L.y0), *y1 = F(L.y1), *gb = F(L.gb), *zl = F(L.zl), *coef = F(L.coef);
    double* sums1 = reinterpret_cast<double*>(ws + L.sums1);
    double* sums2 = reinterpret_cast<double*>(ws + L.sums2);
    int rc;
    // x = fc(motion).reshape(B, 16nf, 1, 4, 4) (decoder.py:99) -- written channels-last [B][1][4][4][16nf]
    if ((rc = conv_forward(d->fc, motion, d->cfg.z_dim, xA, nullptr, 1, 1, B, 1, 1, 1, EPI_NONE, st))) return rc;
    // all six ADAIN Linear(z_dim, 2C) in one GEMM (they depend only on z)
    if ((rc = conv_forward(d->zlin, motion, d->cfg.z_dim, zl, nullptr, 1, 1, B, 1, 1, 1, EPI_NONE, st))) return rc;
    float* x = xA;
    float* xn = xB;
    bool x_stats_ready = false;
    double* sums3 = reinterpret_cast<double*>(ws + L.sums3);
    // SPADE branches computed ahead by i2v_dec_prepare for exactly these start frames (same pointer, batch, size, workspace)?
    bool prepared = prep_img == img && d->prep_B == B && d->prep_h == img_h && d->prep_w == img_w && d->prep_ws == workspace &&
                    d->prep_bstride == img_bstride;
    // Prepared on the side stream (i2v_dec_prepare), or not prepared at all: then compute the maps on the handle's side stream now,
    // underneath the first levels (see i2v_dec::overlap).  Either way every block waits for its level's event.
    // While `st` captures a graph the per-level events of a prepare forked BEFORE the capture cannot be waited on (they were recorded
    // outside it): the prepared maps are dropped and the branches run inline, inside the capture (fork_spade refuses to fork there).
    if (prepared && d->prep_forked && stream_is_capturing(st)) prepared = false;
    bool forked = prepared && d->prep_forked;
    // A forked prepare this call does NOT consume (other frames / batch / workspace, cancelled, capture): its side-stream work still
    // writes gbs / py0 / py1 of the workspace it was given -- join it before anything of this call touches a workspace.
    if (!forked && d->side_unjoined)
        if (int rcj = d->join_side(st)) return rcj;
    if (!prepared) {
        if (int rcf = fork_spade(d, L, ws, img, img_h, img_w, B, st, &forked)) return rcf;
        prepared = forked;
    }
    struct Join {   // an error return below must not leave the caller's stream ahead of the side stream's work on its buffers
        i2v_dec* d; hipStream_t st; bool on;   // (branches of every level and the shortcut GEMMs: join_side covers both)
        ~Join() { if (on) (void)d->join_side(st); }
    } join{d, st, forked};
    for (int k = 0; k < 6; ++k) {
        const Block& b = d->blk[k];
        const Level& l = d->lvl[k];
        double* s_in = (k & 1) ? sums3 : sums1;
        double* s_out = (k & 1) ? sums1 : sums3;
        // Sub-batches (I2V_DEC_SUB = samples per sub-batch, levels whose one sample already fills the chip): the block's launch
        // sequence runs once per sub-batch, so that what one launch writes (dx, the V operands: ~100 MB per sample at the last
        // two levels) is still in the 256 MB Infinity Cache when the next launch reads it.  Every op is per sample: same bits.
        int nsub = B;
        if (d->sub > 0 && k >= 4 && (long)l.T * l.H * l.W >= 65536) nsub = std::min(B, d->sub);
        const long Pl = (long)(l.T / l.ut) * (l.H / l.us) * (l.W / l.us), P = (long)l.T * l.H * l.W;
        bool ready_out = x_stats_ready;
        if (forked) {   // this level's gamma | beta maps are complete
            I2V_HIP_CHECK(hipStreamWaitEvent(st, d->ev_lvl[k], 0));
            if (k == 5) d->side_unjoined = false;   // every branch has been waited for (a shortcut enqueued below keeps join.on)
        }
        for (int s0 = 0; s0 < B; s0 += nsub) {
            const int n = std::min(nsub, B - s0);
            BlockBufs bufs{a, dx, xs_in, xs_low, y0, y1, gb, coef, s_in + (size_t)s0 * b.n_in * 2, sums2, s_out + (size_t)s0 * b.n_out * 2,
                           F(L.splitk), L.splitk_floats, L.has_y1v ? F(L.y1v) : nullptr,
                           prepared ? F(L.gbs[k]) + (size_t)s0 * l.H * l.W * 2 * b.n_in : nullptr,
                           d->has32() && d->wino32 ? F(L.m6) : nullptr,
                           forked && nsub == B && d->overlap >= 1 && !d->no_side_shortcut ? d->side : nullptr, F(L.coef_s)};
            bool ready = x_stats_ready;
            if ((rc = block_forward(d, k, d->blk[k], l, x + (size_t)s0 * Pl * b.n_in, xn + (size_t)s0 * P * b.n_out,
                                    img + (size_t)s0 * (size_t)img_bstride, img_h, img_w, zl + (size_t)s0 * d->Nz, d->Nz, n, bufs, ready, k == 5, st)))
                return rc;
            ready_out = ready;
        }
        x_stats_ready = ready_out;
        std::swap(x, xn);
    }
    join.on = false;   // every level's maps and every shortcut have been waited for by their consumers
    {
        const Level& l = d->lvl[5];
        if (d->aux16() && d->conv_img_m.w.p) rc = conv_img_mfma_forward(d->conv_img_m, x, out, B, l.T, l.H, l.W, st, d->status_dev, out_bstride);
        else if (d->aux16() && d->conv_img16.w.p) {
            const long P = (long)l.T * l.H * l.W, tot = (long)B * P;
            I2V_REQUIRE((tot + 255) / 256 < (1L << 31), I2V_E_INVALID, "conv_img: %ld positions", tot);
            if ((rc = pointwise16_forward(d->conv_img16, x, a, nullptr, tot, P, EPI_NONE, st, nullptr, d->status_dev, true))) return rc;
            hipLaunchKernelGGL(conv_img_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a,
                               d->conv_img_bias.as<float>(), out, tot, l.T, l.H, l.W, (long)out_bstride);
            I2V_HIP_CHECK(hipGetLastError());
        } else if (conv_img_supported(l.T, l.H, l.W, d->nf)) rc = conv_img_forward(d->conv_img_v, x, out, B, l.T, l.H, l.W, st, out_bstride);
        else rc = conv_forward(d->conv_img, x, d->nf, out, nullptr, 1, 1, B, l.T, l.H, l.W, EPI_FRAMES, st, nullptr, 1, 1, out_bstride);
        if (rc) return rc;
    }
    if (d->has16()) {
        hipLaunchKernelGGL(status_finish_kernel, dim3(1), dim3(1), 0, st, d->status_dev);
        I2V_HIP_CHECK(hipGetLastError());
        I2V_HIP_CHECK(hipMemcpyAsync(d->status_host, d->status_dev, I2V_STATUS_WORDS * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    return I2V_OK;
}

// mma = auto: after a forward has been synchronised, turn what the range guard saw into per-layer decisions.  Returns true when a
// layer (or the whole handle) was switched to the exact-fp32 kernels, i.e. the forward has to be run again.
static bool auto_decide(i2v_dec* d) {
    const int* h = d->status_host;
    bool changed = false;
    for (int layer = 0; layer < 12; ++layer) {
        const int bits = h[I2V_STATUS_SNAP + 1 + layer];
        if (!bits || d->fp32_layer[layer]) continue;
        float m;
        std::memcpy(&m, &bits, 4);
        if (!(m >= I2V_UNDERFLOW_MAX && m <= I2V_OVERFLOW_MAX)) { d->fp32_layer[layer] = true; changed = true; }
    }
    if ((h[0] & 1) && !changed && !d->fp32_all) { d->fp32_all = true; changed = true; }   // an overflow no operand slot explains
    return changed;
}

extern "C" {

int i2v_dec_forward_strided(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, int64_t img_bstride, const float* motion, float* out,
                            int64_t out_bstride, void* workspace, size_t workspace_bytes, int32_t batch, void* stream) {
    int rc = dec_forward_once(d, img, img_h, img_w, img_bstride, motion, out, out_bstride, workspace, workspace_bytes, batch, stream);
    if (rc || !d || d->cfg.mma != 2) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stream_is_capturing(st)) return rc;   // (a captured forward runs the layer choices made so far; it cannot look at its own flags)
    for (int round = 0; round < 14; ++round) {
        I2V_HIP_CHECK(hipStreamSynchronize(st));
        if (!auto_decide(d)) {
            // what is left in the flag word has been handled (bit 1 -- underflow -- by the per-layer switch; bit 0 cannot remain)
            if (*d->status_host & 2) { I2V_HIP_CHECK(hipMemsetAsync(d->status_dev, 0, sizeof(int), st)); *d->status_host &= ~2; }
            return I2V_OK;
        }
        I2V_HIP_CHECK(hipMemsetAsync(d->status_dev, 0, sizeof(int), st));
        *d->status_host = 0;
        d->auto_reruns += 1;
        if ((rc = dec_forward_once(d, img, img_h, img_w, img_bstride, motion, out, out_bstride, workspace, workspace_bytes, batch, stream))) return rc;
    }
    I2V_REQUIRE(false, I2V_E_RANGE, "i2v_dec_forward (mma = auto): the range guard still fires with every layer on the exact-fp32 kernels");
}

int i2v_dec_fallback_layers(i2v_dec* d, int32_t* mask, int32_t* reruns) {
    I2V_REQUIRE(d && mask, I2V_E_INVALID, "i2v_dec_fallback_layers: null argument");
    int m = 0;
    for (int layer = 0; layer < 12; ++layer)
        if (d->fp32_layer[layer] || d->fp32_all) m |= 1 << layer;
    if (d->fp32_all) m |= 1 << 30;
    *mask = m;
    if (reruns) *reruns = d->auto_reruns;
    return I2V_OK;
}

int i2v_dec_prepare(i2v_dec* d, const float* img, int32_t img_h, int32_t img_w, void* workspace, size_t workspace_bytes, int32_t batch,
                    void* stream) {
    I2V_REQUIRE(d && d->loaded, I2V_E_STATE, "i2v_dec_prepare: weights not loaded");
    if (int rc0 = check_entry(d, "i2v_dec_prepare")) return rc0;
    I2V_REQUIRE(img && workspace && batch > 0 && img_h > 0 && img_w > 0, I2V_E_INVALID, "i2v_dec_prepare: null argument or bad size");
    const int B = batch;
    const DecWs L = dec_ws(d, B);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_dec_prepare: workspace %zu < required %zu", workspace_bytes, L.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rco = d->order.entry(st)) return rco;
    StreamOrderMark mark{&d->order, st};
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    // an earlier forked prepare that was never consumed may have been given ANOTHER workspace (or start frames the caller has
    // released since): `st` joins it before this prepare replaces it
    if (d->side_unjoined && (d->prep_img == nullptr || d->prep_ws != workspace))
        if (int rcj = d->join_side(st)) return rcj;
    d->prep_img = nullptr;
    d->img_bstride = 0;
    // on the handle's side stream where possible (ordered behind everything already on `st`): the caller's stream stays free for
    // whatever it can do meanwhile, and the consuming forward waits per level; else inline on `st`
    bool forked = false;
    if (int rc = fork_spade(d, L, ws, img, img_h, img_w, B, st, &forked)) return rc;
    if (!forked)
        for (int k = 0; k < 6; ++k)
            if (int rc = spade_branch(d, d->blk[k], d->lvl[k], img, img_h, img_w, B, F(L.py0), F(L.py1), L.has_y1v ? F(L.py1v) : nullptr, F(L.gbs[k]), st))
                return rc;
    d->prep_forked = forked;
    d->prep_img = img; d->prep_B = B; d->prep_h = img_h; d->prep_w = img_w; d->prep_ws = workspace;
    d->prep_bstride = (long)3 * img_h * img_w;
    return I2V_OK;
}

int i2v_dec_prepare_cancel(i2v_dec* d) {
    I2V_REQUIRE(d, I2V_E_INVALID, "i2v_dec_prepare_cancel: null handle");
    d->prep_img = nullptr;   // (a forked prepare keeps side_unjoined: the next forward / prepare / i2v_dec_join joins it)
    return I2V_OK;
}

int i2v_dec_set_side_stream(i2v_dec* d, void* side_stream) {
    I2V_REQUIRE(d, I2V_E_INVALID, "i2v_dec_set_side_stream: null handle");
    I2V_REQUIRE_DEVICE(d->device, "i2v_dec_set_side_stream");
    hipStream_t ns = static_cast<hipStream_t>(side_stream);
    if (d->side && d->side == ns && !d->side_owned) return I2V_OK;
    // whatever the old side stream still carries for this handle completes first (rare: a host wait at configuration time)
    if (d->side) {
        I2V_HIP_CHECK(hipStreamSynchronize(d->side));
        if (d->side_owned) (void)hipStreamDestroy(d->side);
    }
    d->side_unjoined = false;
    d->prep_img = nullptr;
    d->side = ns;                 // null: the handle creates its own stream again at the next fork
    d->side_owned = ns == nullptr;
    return I2V_OK;
}

int i2v_dec_join(i2v_dec* d, void* stream) {
    I2V_REQUIRE(d, I2V_E_INVALID, "i2v_dec_join: null handle");
    I2V_REQUIRE_DEVICE(d->device, "i2v_dec_join");
    hipStream_t st = static_cast<hipStream_t>(stream);
    I2V_REQUIRE(!stream_is_capturing(st), I2V_E_STATE, "i2v_dec_join: the stream is capturing a graph (join before the capture begins)");
    d->prep_img = nullptr;
    return d->join_side(st);
}

int i2v_dec_status(i2v_dec* d, int32_t* flags, int32_t reset, void* stream) {
    I2V_REQUIRE(d && flags, I2V_E_INVALID, "i2v_dec_status: null argument");
    I2V_REQUIRE_DEVICE(d->device, "i2v_dec_status");
    hipStream_t st = static_cast<hipStream_t>(stream);
    I2V_HIP_CHECK(hipMemcpyAsync(d->status_host, d->status_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    I2V_HIP_CHECK(hipStreamSynchronize(st));
    *flags = *d->status_host;
    if (reset) {
        I2V_HIP_CHECK(hipMemsetAsync(d->status_dev, 0, sizeof(int), st));
        I2V_HIP_CHECK(hipStreamSynchronize(st));
        *d->status_host = 0;
    }
    return I2V_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Stand-alone GeneratorBlock / Spade / ADAIN / Norm3D (reference tensors [B][C][T][H][W] in and out)
// ------------------------------------------------------------------------------------------------------------------
}  // extern "C"

struct i2v_gblock {
    i2v_dec ctx;       // carries cfg.mma and the (unused) profiling / tap state for the shared block code
    Block b;
    ConvWeights zlin;  // this block's ADAIN Linear(z_dim, 2*n_mid)
    int z_dim = 0;
    bool has_convs = false, has_spade = false, has_adain = false, has_norm_s = false;
};

namespace {

struct GbWs { size_t x_cl, out_cl, a, dx, xs_in, xs_low, y0, y1, gb, zl, sums1, sums2, coef, total; };

GbWs gb_ws(const i2v_gblock* g, int B, int T, int H, int W) {
    const Block& b = g->b;
    const size_t P = (size_t)T * H * W, cm = std::max(b.n_in, std::max(b.n_mid, b.n_out));
    GbWs L;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = align_up(o + floats * 4, 256); return r; };
    L.x_cl = take(B * P * cm); L.out_cl = take(B * P * cm); L.a = take(B * P * cm * 2); L.dx = take(B * P * b.n_mid);
    L.xs_in = take(B * P * b.n_in); L.xs_low = take(B * P * b.n_out);
    L.y0 = take((size_t)B * H * W * 16); L.y1 = take((size_t)B * H * W * 128); L.gb = take((size_t)B * H * W * 2 * b.n_in);
    L.zl = take((size_t)B * 2 * b.n_mid);
    L.sums1 = take((size_t)B * cm * 4); L.sums2 = take((size_t)B * cm * 4); L.coef = take((size_t)B * cm * 2);
    L.total = o;
    return L;
}

int run_transpose(const float* in, float* out, int B, int C, long P, bool to_cl, hipStream_t st) {
    const int R = to_cl ? C : (int)P, S = to_cl ? (int)P : C;
    hipLaunchKernelGGL(transpose_kernel, dim3((S + 31) / 32, (R + 31) / 32, B), dim3(256), 0, st, in, out, C, (int)P, to_cl ? 1 : 0);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace

extern "C" {

int i2v_gblock_create(int32_t n_in, int32_t n_out, int32_t z_dim, int32_t spectral_norm, int32_t mma, i2v_gblock** out) {
    I2V_REQUIRE(out && n_in > 0 && n_out > 0 && n_in % 8 == 0 && n_out % 8 == 0 && n_in <= 1024 && n_out <= 1024, I2V_E_INVALID,
                "i2v_gblock_create: channel counts must be multiples of 8 in [8, 1024]");
    I2V_REQUIRE(z_dim > 0 && z_dim % 4 == 0 && (mma == 0 || mma == 1), I2V_E_INVALID, "i2v_gblock_create: bad z_dim / mma");
    int ndev = 0;
    I2V_HIP_CHECK(hipGetDeviceCount(&ndev));
    I2V_REQUIRE(ndev > 0, I2V_E_HIP, "i2v_gblock_create: no HIP device");
    auto g = std::make_unique<i2v_gblock>();
    g->ctx.cfg = i2v_dec_cfg{};
    g->ctx.cfg.mma = mma;
    g->ctx.cfg.spectral_norm = spectral_norm;
    g->ctx.cfg.z_dim = z_dim;
    if (const char* e = std::getenv("I2V_DEC_WINO")) g->ctx.wino = std::atoi(e) != 0;
    if (const char* e = std::getenv("I2V_DEC_PW16")) g->ctx.pw16 = std::atoi(e) != 0;
    if (const char* e = std::getenv("I2V_DEC_WINO4")) g->ctx.wino4 = std::atoi(e);
    if (int rc = init_status(&g->ctx)) return rc;
    g->z_dim = z_dim;
    Block& b = g->b;
    b.name = "";
    b.n_in = n_in; b.n_out = n_out; b.n_mid = std::min(n_in, n_out);
    b.learned = n_in != n_out;
    int grp = 16;
    while (n_in % grp) --grp;
    b.groups_spade = grp;
    b.zoff = 0;
    *out = g.release();
    return I2V_OK;
}

void i2v_gblock_destroy(i2v_gblock* g) { delete g; }

int i2v_gblock_load(i2v_gblock* g, const i2v_tensor* tensors, int32_t n_tensors) {
    I2V_REQUIRE(g && tensors && n_tensors > 0, I2V_E_INVALID, "i2v_gblock_load: null argument");
    I2V_REQUIRE_DEVICE(g->ctx.device, "i2v_gblock_load");
    StateDict sd(tensors, n_tensors);
    Block& b = g->b;
    const bool sn = g->ctx.cfg.spectral_norm != 0, f16 = g->ctx.cfg.mma == 1;
    int rc;
    g->has_convs = g->has_spade = g->has_adain = g->has_norm_s = false;
    if (sd.has(sn ? "conv_0.weight_orig" : "conv_0.weight")) {
        if (f16) {
            if ((rc = sn_pack(sd, "conv_0", sn, b.n_mid, b.n_in, 3, true, b.conv0_16))) return rc;
            if ((rc = sn_pack(sd, "conv_1", sn, b.n_out, b.n_mid, 3, true, b.conv1_16))) return rc;
            // the geometry is only known at the call: also pack the Winograd variants where the channel counts allow them
            if (g->ctx.wino && wino16_supported(b.n_mid, b.n_in, 16, 64, 64) &&
                (rc = sn_pack_wino(sd, "conv_0", sn, b.n_mid, b.n_in, false, b.conv0_w))) return rc;
            if (g->ctx.wino && wino16_supported(b.n_out, b.n_mid, 16, 64, 64) &&
                (rc = sn_pack_wino(sd, "conv_1", sn, b.n_out, b.n_mid, false, b.conv1_w))) return rc;
            // ... and the F(4,3) variants (used where the call's geometry gives a sample >= 32 workgroups; I2V_DEC_WINO4=2: always)
            if (g->ctx.wino && g->ctx.wino4 && wino4_supported(b.n_mid, b.n_in, 16, 64, 64, 3) &&
                (rc = sn_pack_wino4(sd, "conv_0", sn, b.n_mid, b.n_in, false, b.conv0_w4))) return rc;
            if (g->ctx.wino && g->ctx.wino4 && wino4_supported(b.n_out, b.n_mid, 16, 64, 64, 3) &&
                (rc = sn_pack_wino4(sd, "conv_1", sn, b.n_out, b.n_mid, false, b.conv1_w4))) return rc;
        } else {
            if ((rc = sn_pack(sd, "conv_0", sn, b.n_mid, b.n_in, 3, true, b.conv0))) return rc;
            if ((rc = sn_pack(sd, "conv_1", sn, b.n_out, b.n_mid, 3, true, b.conv1))) return rc;
        }
        if (b.learned && (rc = sn_pack(sd, "conv_s", sn, b.n_out, b.n_in, 1, false, b.convs))) return rc;
        if (b.learned && f16 && g->ctx.pw16 && (rc = sn_pack(sd, "conv_s", sn, b.n_out, b.n_in, 1, false, b.convs16))) return rc;
        g->has_convs = true;
    }
    if (sd.has("norm_s.bn.weight")) {
        const float* gw = sd.f32("norm_s.bn.weight", b.n_in);
        const float* gb = sd.f32("norm_s.bn.bias", b.n_in);
        if (!gw || !gb) return I2V_E_MISSING;
        if ((rc = b.gn_w.upload(gw, (size_t)b.n_in * 4))) return rc;
        if ((rc = b.gn_b.upload(gb, (size_t)b.n_in * 4))) return rc;
        g->has_norm_s = true;
    }
    if (sd.has("norm_0.conv.weight")) {
        const float* w1 = sd.f32("norm_0.conv.weight", 128 * 3 * 9);
        const float* b1 = sd.f32("norm_0.conv.bias", 128);
        const float* wg = sd.f32("norm_0.conv_gamma.weight", (int64_t)b.n_in * 128 * 9);
        const float* bg = sd.f32("norm_0.conv_gamma.bias", b.n_in);
        const float* wb = sd.f32("norm_0.conv_beta.weight", (int64_t)b.n_in * 128 * 9);
        const float* bb = sd.f32("norm_0.conv_beta.bias", b.n_in);
        if (!w1 || !b1 || !wg || !bg || !wb || !bb) return I2V_E_MISSING;
        if ((rc = b.sp_conv.pack(w1, b1, 128, 3, 1, 3, 3, 1.0))) return rc;
        {   // the same conv for the split-fp16 path: input channels padded 3 -> 16 (the resize kernel's row)
            std::vector<float> w16((size_t)128 * 16 * 9, 0.f);
            for (int n = 0; n < 128; ++n)
                for (int c = 0; c < 3; ++c)
                    for (int t = 0; t < 9; ++t) w16[((size_t)n * 16 + c) * 9 + t] = w1[((size_t)n * 3 + c) * 9 + t];
            if ((rc = b.sp_conv16.pack(w16.data(), b1, 128, 16, 1, 3, 3, 1.0))) return rc;
        }
        std::vector<float> wgb((size_t)2 * b.n_in * 128 * 9), bgb((size_t)2 * b.n_in);
        std::memcpy(wgb.data(), wg, (size_t)b.n_in * 128 * 9 * 4);
        std::memcpy(wgb.data() + (size_t)b.n_in * 128 * 9, wb, (size_t)b.n_in * 128 * 9 * 4);
        for (int c = 0; c < b.n_in; ++c) { bgb[c] = bg[c] + 1.0f; bgb[b.n_in + c] = bb[c]; }
        if (f16) rc = b.sp_gb16.pack(wgb.data(), bgb.data(), 2 * b.n_in, 128, 1, 3, 3, 1.0);
        else rc = b.sp_gb.pack(wgb.data(), bgb.data(), 2 * b.n_in, 128, 1, 3, 3, 1.0);
        if (rc) return rc;
        g->has_spade = true;
    }
    if (sd.has("norm_1.linear.weight")) {
        const float* lw = sd.f32("norm_1.linear.weight", (int64_t)2 * b.n_mid * g->z_dim);
        const float* lb = sd.f32("norm_1.linear.bias", (int64_t)2 * b.n_mid);
        if (!lw || !lb) return I2V_E_MISSING;
        if ((rc = g->zlin.pack(lw, lb, 2 * b.n_mid, g->z_dim, 1, 1, 1, 1.0))) return rc;
        g->has_adain = true;
    }
    I2V_REQUIRE(g->has_convs || g->has_spade || g->has_adain || g->has_norm_s, I2V_E_MISSING,
                "i2v_gblock_load: no GeneratorBlock / Spade / ADAIN / Norm3D keys found");
    return I2V_OK;
}

size_t i2v_gblock_workspace_bytes(const i2v_gblock* g, int32_t batch, int32_t t, int32_t h, int32_t w) {
    if (!g || batch <= 0 || t <= 0 || h <= 0 || w <= 0) return 0;
    return gb_ws(g, batch, t, h, w).total;
}

int i2v_gblock_forward(i2v_gblock* g, const float* x, const float* z, const float* img, int32_t img_h, int32_t img_w, float* out,
                       void* workspace, size_t workspace_bytes, int32_t batch, int32_t t, int32_t h, int32_t w, void* stream) {
    I2V_REQUIRE(g && g->has_convs && g->has_spade && g->has_adain && (!g->b.learned || g->has_norm_s), I2V_E_STATE,
                "i2v_gblock_forward: block weights not (fully) loaded");
    I2V_REQUIRE(x && z && img && out && workspace && batch > 0, I2V_E_INVALID, "i2v_gblock_forward: null argument");
    if (int rc0 = check_entry(&g->ctx, "i2v_gblock_forward")) return rc0;
    const GbWs L = gb_ws(g, batch, t, h, w);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_gblock_forward: workspace %zu < required %zu", workspace_bytes, L.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const long P = (long)t * h * w;
    Block& b = g->b;
    int rc;
    if ((rc = run_transpose(x, F(L.x_cl), batch, b.n_in, P, true, st))) return rc;
    if ((rc = conv_forward(g->zlin, z, g->z_dim, F(L.zl), nullptr, 1, 1, batch, 1, 1, 1, EPI_NONE, st))) return rc;
    BlockBufs bufs{F(L.a), F(L.dx), F(L.xs_in), F(L.xs_low), F(L.y0), F(L.y1), F(L.gb), F(L.coef),
                   reinterpret_cast<double*>(ws + L.sums1), reinterpret_cast<double*>(ws + L.sums2)};
    bool ready = false;
    const Level l{t, h, w, 1, 1};
    if ((rc = block_forward(&g->ctx, 0, b, l, F(L.x_cl), F(L.out_cl), img, img_h, img_w, F(L.zl), 2 * b.n_mid, batch, bufs, ready,
                            false, st)))
        return rc;
    if ((rc = run_transpose(F(L.out_cl), out, batch, b.n_out, P, false, st))) return rc;
    if (g->ctx.cfg.mma == 1) {
        hipLaunchKernelGGL(status_finish_kernel, dim3(1), dim3(1), 0, st, g->ctx.status_dev);
        I2V_HIP_CHECK(hipGetLastError());
        I2V_HIP_CHECK(hipMemcpyAsync(g->ctx.status_host, g->ctx.status_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    return I2V_OK;
}

int i2v_gblock_status(i2v_gblock* g, int32_t* flags, int32_t reset, void* stream) {
    I2V_REQUIRE(g && flags, I2V_E_INVALID, "i2v_gblock_status: null argument");
    return i2v_dec_status(&g->ctx, flags, reset, stream);
}

int i2v_gblock_norm(i2v_gblock* g, int32_t part, const float* x, const float* cond, int32_t img_h, int32_t img_w, float* out,
                    void* workspace, size_t workspace_bytes, int32_t batch, int32_t t, int32_t h, int32_t w, void* stream) {
    I2V_REQUIRE(g && x && out && workspace && batch > 0 && part >= 0 && part <= 2, I2V_E_INVALID, "i2v_gblock_norm: bad argument");
    if (int rc0 = check_entry(&g->ctx, "i2v_gblock_norm")) return rc0;
    const GbWs L = gb_ws(g, batch, t, h, w);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_gblock_norm: workspace %zu < required %zu", workspace_bytes, L.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const long P = (long)t * h * w;
    Block& b = g->b;
    const int B = batch;
    double* sums = reinterpret_cast<double*>(ws + L.sums1);
    float *x_cl = F(L.x_cl), *a = F(L.a), *coef = F(L.coef);
    int rc;
    const int C = part == 1 ? b.n_mid : b.n_in;
    if ((rc = run_transpose(x, x_cl, B, C, P, true, st))) return rc;
    if ((rc = run_stats(x_cl, sums, B, P, C, st))) return rc;
    if (part == 0) {        // Spade.forward(x, img), normalization_layer.py:18-24
        I2V_REQUIRE(g->has_spade && cond, I2V_E_STATE, "i2v_gblock_norm: Spade weights not loaded / no start frame");
        if ((rc = run_coef(sums, coef, B, C, b.groups_spade, (double)P, nullptr, 0, 0, nullptr, nullptr, st))) return rc;
        const long tot = (long)B * h * w;
        hipLaunchKernelGGL(resize_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 65536)), dim3(256), 0, st, cond, F(L.y0), B,
                           img_h, img_w, h, w, g->ctx.cfg.mma == 1 ? 1 : 0, g->ctx.status_dev, (long)3 * img_h * img_w);
        I2V_HIP_CHECK(hipGetLastError());
        if (g->ctx.cfg.mma == 1) {
            if ((rc = conv16_forward(b.sp_conv16, F(L.y0), reinterpret_cast<float*>(F(L.y1)), nullptr, 1, 1, B, 1, h, w,
                                     EPI_LRELU | EPI_HL16, st, nullptr, g->ctx.status_dev))) return rc;
            if ((rc = conv16_forward(b.sp_gb16, F(L.y1), F(L.gb), nullptr, 1, 1, B, 1, h, w, EPI_NONE, st))) return rc;
        } else {
            if ((rc = conv_forward(b.sp_conv, F(L.y0), 16, F(L.y1), nullptr, 1, 1, B, 1, h, w, EPI_LRELU, st))) return rc;
            if ((rc = conv_forward(b.sp_gb, F(L.y1), 128, F(L.gb), nullptr, 1, 1, B, 1, h, w, EPI_NONE, st))) return rc;
        }
        if ((rc = run_modulate(x_cl, coef, F(L.gb), a, B, t, h, w, C, 1, 1, 0, st))) return rc;
    } else if (part == 1) { // ADAIN.forward(x, z), normalization_layer.py:47-51
        I2V_REQUIRE(g->has_adain && cond, I2V_E_STATE, "i2v_gblock_norm: ADAIN weights not loaded / no latent");
        if ((rc = conv_forward(g->zlin, cond, g->z_dim, F(L.zl), nullptr, 1, 1, B, 1, 1, 1, EPI_NONE, st))) return rc;
        if ((rc = run_coef(sums, coef, B, C, C, (double)P, F(L.zl), 2 * b.n_mid, 0, nullptr, nullptr, st))) return rc;
        if ((rc = run_modulate(x_cl, coef, nullptr, a, B, t, h, w, C, 1, 1, 0, st))) return rc;
    } else {                // Norm3D.forward(x), normalization_layer.py:33-35
        I2V_REQUIRE(g->has_norm_s && C % 16 == 0, I2V_E_STATE, "i2v_gblock_norm: Norm3D weights not loaded or C %% 16 != 0");
        if ((rc = run_coef(sums, coef, B, C, 16, (double)P, nullptr, 0, 0, b.gn_w.as<float>(), b.gn_b.as<float>(), st))) return rc;
        if ((rc = run_modulate(x_cl, coef, nullptr, a, B, t, h, w, C, 1, 1, 0, st))) return rc;
    }
    return run_transpose(a, out, B, C, P, false, st);
}

}  // extern "C"

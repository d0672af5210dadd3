// conv_img: Conv3d(nf -> 3, 3x3x3, pad 1) + tanh, stored as [B][T][3][H][W]  (reference decoder.py:81,117-120).
//
// N = 3 output channels cannot feed a 32-wide matrix-core tile (the generic MFMA kernel wastes 10/11 of its columns
// here), so this layer runs on the vector ALU in exact fp32: the 4x8x8 position brick's input halo is staged in LDS per
// 16-channel chunk (rows padded to 20 floats, ds_read_b128) together with the chunk's [27][16][3(+1)] weight slab.
// (A first version read the weights through the scalar cache: the 20 KB weight set thrashes the 16 KB scalar cache and
// the kernel ran 5x slower.)
#include <algorithm>

#include "i2v_conv.h"

namespace i2v {

constexpr int CI_TT = 8, CI_TH = 8, CI_TW = 8;  // 512-position brick: halo 1000 rows = 1.95x (a 4x8x8 brick: 2.34x)
constexpr int CI_NTHR = 256;
constexpr int CI_HT = CI_TT + 2, CI_HH = CI_TH + 2, CI_HW = CI_TW + 2;
constexpr int CI_NPOS = CI_HT * CI_HH * CI_HW;  // 1000
constexpr int CI_KC = 8;                       // channels per chunk (small chunk -> 56 KB LDS -> 2 workgroups of 4 waves per CU)
constexpr int CI_LS = CI_KC + 4;               // floats per staged row (+4 pad)
constexpr int CI_Q = CI_KC / 4;                // float4 pieces per row
constexpr int CI_WCH = 27 * CI_KC * 4;         // weight floats per chunk: [tap][c][3(+1)]

// One thread = TWO output positions (w and w+4 of the same brick row): every weight float4 fetched from LDS (a broadcast
// read, all lanes the same address) feeds two FMA triplets, which keeps the LDS pipe below the VALU time.
__global__ __launch_bounds__(CI_NTHR) void conv_img_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int T,
                                                       int H, int W, int C, int nchunk, long obs) {
    __shared__ __attribute__((aligned(16))) float in_lds[CI_NPOS * CI_LS];
    __shared__ __attribute__((aligned(16))) float w_lds[CI_WCH];
    __shared__ int gpos[CI_NPOS];
    const int tid = threadIdx.x;
    int brick = blockIdx.x;
    const int nbW = W / CI_TW, nbH = H / CI_TH, nbT = T / CI_TT;
    const int bw = brick % nbW; brick /= nbW;
    const int bh = brick % nbH; brick /= nbH;
    const int bt = brick % nbT; brick /= nbT;
    const int b = brick, t0 = bt * CI_TT, h0 = bh * CI_TH, w0 = bw * CI_TW;
    for (int p0 = tid; p0 < CI_NPOS; p0 += CI_NTHR) {
        int p = p0;
        const int iw = p % CI_HW; p /= CI_HW;
        const int ih = p % CI_HH; p /= CI_HH;
        const int t = t0 + p - 1, h = h0 + ih - 1, w = w0 + iw - 1;
        const bool ok = (unsigned)t < (unsigned)T && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        gpos[p0] = ok ? ((b * T + t) * H + h) * W + w : -1;
    }
    const int iw = tid & 3, ih = (tid >> 2) % CI_TH, it = tid / (4 * CI_TH);   // positions (iw) and (iw + 4)
    const float* my = in_lds + ((it * CI_HH + ih) * CI_HW + iw) * CI_LS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
    __syncthreads();
    // Register double buffer: the global loads of chunk ch + 1 are issued right after chunk ch has been parked in LDS and
    // stay in flight while chunk ch is multiplied (one exposed load latency per WORKGROUP instead of one per chunk).
    constexpr int NS = (CI_NPOS * CI_Q + CI_NTHR - 1) / CI_NTHR;
    constexpr int NW4 = CI_WCH / 4, NWS = (NW4 + CI_NTHR - 1) / CI_NTHR;
    float4 v[NS], wv[NWS];
    auto request = [&](int ch) {   // all of a thread's pieces back to back (branch-free, clamped)
        const int c0 = ch * CI_KC;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int idx = tid + u * CI_NTHR;
            const int q = idx % CI_Q, gp = gpos[idx < CI_NPOS * CI_Q ? (idx / CI_Q) : 0];
            const bool ok = idx < CI_NPOS * CI_Q && gp >= 0 && c0 + 4 * q < C;
            const float4 t4 = *reinterpret_cast<const float4*>(in + (ok ? (long)gp * C + c0 + 4 * q : 0));
            v[u] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NWS; ++u) {
            const int f = tid + u * CI_NTHR;
            wv[u] = *reinterpret_cast<const float4*>(wp + (long)ch * CI_WCH + (f < NW4 ? f : 0) * 4);
        }
    };
    request(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();   // the previous chunk's LDS image is no longer read
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int idx = tid + u * CI_NTHR;
            if (idx < CI_NPOS * CI_Q) *reinterpret_cast<float4*>(in_lds + (idx / CI_Q) * CI_LS + 4 * (idx % CI_Q)) = v[u];
        }
#pragma unroll
        for (int u = 0; u < NWS; ++u) {
            const int f = tid + u * CI_NTHR;
            if (f < NW4) *reinterpret_cast<float4*>(w_lds + f * 4) = wv[u];
        }
        if (ch + 1 < nchunk) request(ch + 1);
        __syncthreads();
#pragma unroll 1
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const float* row = my + ((dt * CI_HH + dh) * CI_HW + dw) * CI_LS;
                    const float* wt = w_lds + ((dt * 3 + dh) * 3 + dw) * (CI_KC * 4);  // [c][4]
#pragma unroll
                    for (int q = 0; q < CI_Q; ++q) {
                        const float4 xa = *reinterpret_cast<const float4*>(row + 4 * q);
                        const float4 xb = *reinterpret_cast<const float4*>(row + 4 * CI_LS + 4 * q);
                        const float pa[4] = {xa.x, xa.y, xa.z, xa.w}, pb[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 w4 = *reinterpret_cast<const float4*>(wt + (4 * q + j) * 4);
                            a0 = fmaf(pa[j], w4.x, a0); a1 = fmaf(pa[j], w4.y, a1); a2 = fmaf(pa[j], w4.z, a2);
                            b0 = fmaf(pb[j], w4.x, b0); b1 = fmaf(pb[j], w4.y, b1); b2 = fmaf(pb[j], w4.z, b2);
                        }
                    }
                }
    }
    const int t = t0 + it, h = h0 + ih, w = w0 + iw;
    const long HWo = (long)H * W;
    float* o = out + (long)b * obs + ((long)t * 3) * HWo + (long)h * W + w;   // obs: floats between the samples of `out`
    const float c0_ = bias[0], c1_ = bias[1], c2_ = bias[2];
    o[0] = tanhf(a0 + c0_); o[HWo] = tanhf(a1 + c1_); o[2 * HWo] = tanhf(a2 + c2_);
    o[4] = tanhf(b0 + c0_); o[HWo + 4] = tanhf(b1 + c1_); o[2 * HWo + 4] = tanhf(b2 + c2_);
}

int ConvImgWeights::pack(const float* w_src, const float* bias_src, int cin) {
    Cin = cin;
    nchunk = (cin + CI_KC - 1) / CI_KC;
    std::vector<float> p((size_t)nchunk * CI_WCH, 0.f);
    for (int o = 0; o < 3; ++o)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < 27; ++tap)
                p[(((size_t)(c / CI_KC) * 27 + tap) * CI_KC + c % CI_KC) * 4 + o] = w_src[((size_t)o * cin + c) * 27 + tap];
    int rc = w.upload(p.data(), p.size() * 4);
    if (rc) return rc;
    return bias.upload(bias_src, 12);
}

bool conv_img_supported(int T, int H, int W, int C) { return T % CI_TT == 0 && H % CI_TH == 0 && W % CI_TW == 0 && C % 4 == 0; }

int conv_img_forward(const ConvImgWeights& wts, const float* in, float* out, int B, int T, int H, int W, hipStream_t st, long out_bstride) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "conv_img: weights not packed");
    I2V_REQUIRE(conv_img_supported(T, H, W, wts.Cin), I2V_E_INVALID, "conv_img: unsupported geometry");
    const long nblk = (long)B * (T / CI_TT) * (H / CI_TH) * (W / CI_TW);
    hipLaunchKernelGGL(conv_img_kernel, dim3((unsigned)nblk), dim3(CI_NTHR), 0, st, in, wts.w.as<float>(), wts.bias.as<float>(), out,
                       B, T, H, W, wts.Cin, wts.nchunk, out_bstride ? out_bstride : (long)T * 3 * H * W);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// conv_img on the matrix cores, fused (round 3).  N = 3 cannot fill a 32-wide MFMA tile as an implicit GEMM over positions,
// but the 27 (dh, dw, n) combinations of ONE temporal tap can: per dt
//     Y_dt[(dh, dw, n)][p] = sum_c w[n][c][dt][dh][dw] * x[frame t + dt - 1][p][c]        (32 x Cin) x (Cin x positions)
// is a small split-fp16 GEMM over the halo positions p of an 8 x 32 output brick (A = the padded 32 x Cin weight slab of the
// tap, B = the activations straight from global memory, converted to fp16 hi / lo in registers), Y_dt is parked in LDS
// (45 KB) and every thread gathers its own output position:  out[n][h][w] += sum_{dh,dw} Y_dt[(dh,dw,n)][h+dh][w+dw].
// Three such passes (dt = 0, 1, 2), then bias + tanh and the [B][T][3][H][W] store.  No 81-plane round trip through HBM
// (the nf >= 64 path of round 2) and no vector-ALU inner loop (nf = 32).
typedef _Float16 ci_half8 __attribute__((ext_vector_type(8)));
typedef float ci_f32x16 __attribute__((ext_vector_type(16)));
typedef float ci_f32x4 __attribute__((ext_vector_type(4)));

constexpr int CM_TH = 8, CM_TW = 32;                 // output brick (one frame)
constexpr int CM_HH = CM_TH + 2, CM_HW = CM_TW + 2;  // halo 10 x 34 = 340 positions
constexpr int CM_NP = CM_HH * CM_HW;
constexpr int CM_NB = (CM_NP + 31) / 32;             // 11 column blocks
constexpr int CM_LD = CM_NB * 32;                    // 352 columns per plane row

// Round 5: a workgroup keeps its 8 x 32 brick for TCH consecutive output frames and walks the INPUT frames t0 - 1 .. t0 + TCH:
// every input frame is loaded and split into fp16 hi / lo ONCE (the per-element conversion was the kernel's largest cost: ~350
// VALU instructions per lane and temporal tap against 18 MFMAs) and multiplied with the three temporal taps' weight slabs; tap dt
// of input frame f belongs to output frame f + 1 - dt, so three output frames are in flight per thread and frame f - 1 is
// finished behind input frame f.  An output frame still receives its taps in the order dt = 0, 1, 2 and its nine (dh, dw) terms
// in the same order: the bits are those of the one-frame-per-workgroup kernel.
template <int KS>   // KS = Cin / 16 k-steps
__global__ __launch_bounds__(256) void conv_img_mfma_kernel(const float* __restrict__ in, const ci_half8* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out, int B, int T,
                                                            int H, int W, int* __restrict__ range_flag, long obs, int TCH) {
    __shared__ float Y[32 * CM_LD];
    bool bad = false;   // an activation left the fp16 range of its hi part (sticky flag like every other hl16 producer)
    constexpr int C = 16 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    int brick = blockIdx.x;
    const int nbW = W / CM_TW, nbH = H / CM_TH, nbT = T / TCH;
    const int bw = brick % nbW; brick /= nbW;
    const int bh = brick % nbH; brick /= nbH;
    const int t0 = (brick % nbT) * TCH, b = brick / nbT;
    const int h0 = bh * CM_TH, w0 = bw * CM_TW;
    // the halo positions of this lane's column blocks (blocks wave, wave + 4, wave + 8): offset inside a frame, or -1
    int gp[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int p = (wave + 4 * u) * 32 + l31;
        const int ih = p / CM_HW, iw = p - ih * CM_HW;
        const int h = h0 + ih - 1, w = w0 + iw - 1;
        const bool ok = wave + 4 * u < CM_NB && p < CM_NP && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        gp[u] = ok ? h * W + w : -1;
    }
    const int oh = tid >> 5, ow = tid & 31;              // this thread's output position inside the brick
    const size_t HWo = (size_t)H * W;
    const float c0_ = bias[0], c1_ = bias[1], c2_ = bias[2];
    // the three output frames in flight: slot (t % 3)
    float o[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    for (int f = t0 - 1; f <= t0 + TCH; ++f) {
        if ((unsigned)f < (unsigned)T) {                 // (uniform) zero padding in time: a missing frame contributes nothing
            // this input frame's halo positions, split into fp16 hi / lo once
            const float* fr = in + ((size_t)b * T + f) * H * W * C + 8 * kg;
            ci_half8 xh[3][KS], xl[3][KS];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (wave + 4 * u >= CM_NB) continue;     // (uniform: wave 3 owns two blocks)
                ci_f32x4 xv[KS][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        xv[ks][q] = gp[u] >= 0 ? *reinterpret_cast<const ci_f32x4*>(fr + (size_t)gp[u] * C + 16 * ks + 4 * q)
                                               : ci_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = xv[ks][j >> 2][j & 3];
                        const _Float16 hi = (_Float16)v;
                        bad |= !(fabsf(v) <= 65504.f);
                        xh[u][ks][j] = hi;
                        xl[u][ks][j] = (_Float16)(v - (float)hi);
                    }
            }
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int t = f + 1 - dt;                // the output frame this tap of this input frame belongs to
                if (t < t0 || t >= t0 + TCH) continue;   // (uniform)
                // A: the tap's weight slab, fragment-major [dt][ks][hi | lo][64 lanes]
                ci_half8 ah[KS], al[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    ah[ks] = wp[((dt * KS + ks) * 2 + 0) * 64 + lane];
                    al[ks] = wp[((dt * KS + ks) * 2 + 1) * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (wave + 4 * u >= CM_NB) continue;
                    ci_f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xh[u][ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xl[u][ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], xh[u][ks], acc, 0, 0, 0);
                    }
                    const int col = (wave + 4 * u) * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                        Y[row * CM_LD + col] = acc[r];
                    }
                }
                __syncthreads();
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                const int slot = t % 3;
                // (static register indexing: pick the slot's running sums, add the nine terms in the fixed order, put them back)
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
                    if (sl == slot) { a0 = o[sl][0]; a1 = o[sl][1]; a2 = o[sl][2]; }
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const float* y = Y + ((dh * 3 + dw) * 3) * CM_LD + (oh + dh) * CM_HW + ow + dw;
                        a0 += y[0]; a1 += y[CM_LD]; a2 += y[2 * CM_LD];
                    }
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
                    if (sl == slot) { o[sl][0] = a0; o[sl][1] = a1; o[sl][2] = a2; }
                __syncthreads();   // Y is overwritten by the next tap
            }
        }
        // output frame f - 1 has received its last tap (dt = 2 of input frame f, or nothing if f lies outside the clip)
        const int td = f - 1;
        if (td >= t0 && td < t0 + TCH) {
            const int slot = td % 3;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int sl = 0; sl < 3; ++sl)
                if (sl == slot) { a0 = o[sl][0]; a1 = o[sl][1]; a2 = o[sl][2]; o[sl][0] = 0.f; o[sl][1] = 0.f; o[sl][2] = 0.f; }
            float* op = out + (size_t)b * (size_t)obs + ((size_t)td * 3) * HWo + (size_t)(h0 + oh) * W + w0 + ow;   // obs: floats between the samples of `out`
            op[0] = tanhf(a0 + c0_); op[HWo] = tanhf(a1 + c1_); op[2 * HWo] = tanhf(a2 + c2_);
        }
    }
    if (bad && range_flag) atomicOr(range_flag, 1);
}

int ConvImgMfmaWeights::pack(const float* w_src, const float* bias_src, int cin) {
    Cin = cin;
    const int KS = cin / 16;
    std::vector<_Float16> p((size_t)3 * KS * 2 * 64 * 8, (_Float16)0.f);
    for (int dt = 0; dt < 3; ++dt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = lane & 31, kgq = lane >> 5;
                if (row >= 27) continue;
                const int n = row % 3, dw = (row / 3) % 3, dh = row / 9;
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * ks + 8 * kgq + j;
                    const float v = w_src[((size_t)n * cin + c) * 27 + dt * 9 + dh * 3 + dw];
                    const _Float16 hi = (_Float16)v;
                    p[((((size_t)dt * KS + ks) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    p[((((size_t)dt * KS + ks) * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
                }
            }
    int rc = w.upload(p.data(), p.size() * 2);
    if (rc) return rc;
    return bias.upload(bias_src, 12);
}

bool conv_img_mfma_supported(int T, int H, int W, int C) {
    return T >= 1 && H % CM_TH == 0 && W % CM_TW == 0 && (C == 16 || C == 32 || C == 48 || C == 64);
}

int conv_img_mfma_forward(const ConvImgMfmaWeights& wts, const float* in, float* out, int B, int T, int H, int W, hipStream_t st,
                          int* range_flag, long out_bstride) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "conv_img (MFMA): weights not packed");
    I2V_REQUIRE(conv_img_mfma_supported(T, H, W, wts.Cin), I2V_E_INVALID, "conv_img (MFMA): unsupported geometry");
    // frames per workgroup: as many as still leave ~2 waves of workgroups per CU slot (3 workgroups of 45 KB LDS per CU); a chunk of
    // TCH output frames reads TCH + 2 input frames.  Depends on the batch, which changes the schedule only (same bits: see the kernel).
    int TCH = 1;
    {
#ifdef I2V_MEASURE   // (measurement build only: the production library reads no environment variable on a launch path)
        const char* e = getenv("I2V_CONVIMG_TCH");
#else
        const char* e = nullptr;
#endif
        const long bricks = (long)B * (H / CM_TH) * (W / CM_TW);
        for (int c = 16; c >= 2; c /= 2)
            if (T % c == 0 && bricks * (T / c) >= 2 * 768) { TCH = c; break; }
        if (e && atoi(e) > 0 && T % atoi(e) == 0) TCH = atoi(e);
    }
    const long nblk = (long)B * (T / TCH) * (H / CM_TH) * (W / CM_TW);
    I2V_REQUIRE(nblk < (1L << 31), I2V_E_INVALID, "conv_img (MFMA): %ld workgroups", nblk);
    const dim3 grid((unsigned)nblk), block(256);
    const ci_half8* wp = wts.w.as<ci_half8>();
    const long obs = out_bstride ? out_bstride : (long)T * 3 * H * W;
    switch (wts.Cin / 16) {
        case 1: hipLaunchKernelGGL(conv_img_mfma_kernel<1>, grid, block, 0, st, in, wp, wts.bias.as<float>(), out, B, T, H, W, range_flag, obs, TCH); break;
        case 2: hipLaunchKernelGGL(conv_img_mfma_kernel<2>, grid, block, 0, st, in, wp, wts.bias.as<float>(), out, B, T, H, W, range_flag, obs, TCH); break;
        case 3: hipLaunchKernelGGL(conv_img_mfma_kernel<3>, grid, block, 0, st, in, wp, wts.bias.as<float>(), out, B, T, H, W, range_flag, obs, TCH); break;
        default: hipLaunchKernelGGL(conv_img_mfma_kernel<4>, grid, block, 0, st, in, wp, wts.bias.as<float>(), out, B, T, H, W, range_flag, obs, TCH); break;
    }
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace i2v

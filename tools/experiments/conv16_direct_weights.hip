// 3x3x3 Conv3d (and the 3x3 SPADE Conv2d) as an implicit GEMM on the gfx950 fp16 matrix cores with fp32-class accuracy.
//
// gfx950 has no TF32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16/bf16 rate.  This kernel keeps
// the reference's fp32 numerics to ~2^-22 per product while using v_mfma_f32_32x32x16_f16 ("split-fp16"): every fp32
// operand x is carried as the pair (hi, lo) = (fp16(x), fp16(x - hi)), so x = hi + lo up to 2^-22 |x|, and
//     x * w  =  hi_x hi_w  +  hi_x lo_w  +  lo_x hi_w  +  O(2^-22 |x w|)
// costs three fp16 MFMAs into ONE fp32 accumulator (fp16 x fp16 products are exact in fp32; the matrix core honours fp16
// subnormals -- tools/mfma_denorm_test.hip -- so small lo parts keep an absolute precision of 2^-25).  Weights are
// pre-scaled by a per-layer power of two so that their lo parts stay in the normal fp16 range; the epilogue undoes it
// exactly.  Effective peak = 2.5 PFLOP/s / 3.
//
// Operand format "hl16" (HBM and LDS): per position, per group of 8 channels: 8 x fp16 hi (16 B) | 8 x fp16 lo (16 B),
// i.e. 4 bytes per element like fp32; one ds_read_b128 yields one MFMA operand (lane (i, kg): row i, k = 8 kg + j).
// Activations are produced in this format by the modulate kernel (the split is done once per element, not per tap);
// weights are split on the host at load time and stored in MFMA-fragment order (see Conv16Weights::pack).
//
// Tiling: 512 threads = 8 wavefronts (2 per SIMD) per workgroup, 256 output positions (TB x TT x TH x TW brick) x BN
// output channels (128/64/32), wave tile up to 64 x 64.  Per 32-channel K chunk the input halo brick is staged once in
// LDS (rows padded 128 -> 144 B, MFMA rows assigned to 4x4 (h,w) patches: conflict-free ds_read_b128) and reused by all
// taps; the next chunk's rows are requested from HBM a few taps ahead.
// The WEIGHT operands never touch LDS: every wave loads the B fragments of its own output columns straight from
// L2 / the vector L1 into registers (fragment-major packing: one fully coalesced 1 KB load per operand), PFD taps ahead.
// With the input tile read-only for a whole chunk this leaves NO barrier inside the tap loop -- the eight waves run
// free and de-phase, so one wave's LDS reads / address arithmetic hide behind its SIMD neighbour's MFMAs.  (The previous
// design double-buffered each tap's weight slab in LDS behind one barrier per tap: every barrier re-aligned the waves,
// and their common non-MFMA phases left the matrix pipe 35 % idle -- profiles/r01_e_conv16_pmc.json.)
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "i2v_conv.h"

namespace i2v {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C16_BM = 256, C16_KC = 32;
constexpr int C16_ROW = 144;  // bytes per staged row: 32 channels x 4 B + 16 B pad
constexpr int C16_SLOTS = 10; // prefetched 16-byte input pieces per thread and chunk (640 halo rows: the 6 x 10 x 10
                              // halo of a 4 x 8 x 8 brick); larger bricks stage the remainder synchronously
constexpr int C16_FRAG = 4096; // bytes of one (tap, chunk, 32-column block) of weights: 4 fragments x 64 lanes x 16 B

struct Conv16Args {
    const char* in;   // hl16 channels-last [B][T][H][W][Cin]
    const char* wp;   // hl16 weights [tap][chunk][CoutPad/32][kstep 2][hi|lo][lane 64][8 fp16]
    const float* bias;
    const float* res;
    float* out;       // fp32 channels-last [B][T][H][W][Cout]
    double* stats;    // optional [B][Cout][2]: per-(sample, channel) sum / sum of squares of the stored values (TB == 1)
    int B, T, H, W, Cin, Cout, CoutPad, nchunk;  // T,H,W: geometry of the INPUT tensor
    int tdup;            // 1: temporal-duplication mode -- the output has 2T frames, grid.y = output frame parity
    long wset_stride;    // bytes between the two parity weight sets (tdup)
    int KT, KH, KW, tap_base;
    int TB, TT, TH, TW, nbB, nbT, nbH, nbW;
    int HWp;   // halo row pitch in positions (>= TW + KW - 1; 12 for 8-wide bricks: conflict-free 4x4 patches)
    int patch; // 1: MFMA rows are assigned to brick positions in 4x4 (h,w) patches per ds_read_b128 lane group
    int rt, rs, epi;
    float oscale;  // 2^-s: undoes the power-of-two pre-scaling of the weights
};

// MFMA tile row (0..255 within the workgroup tile) -> linear brick index m = ((ib*TT + it)*TH + ih)*TW + iw.
// ds_read_b128 services a wave in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with `patch` every
// group reads one 4x4 (h,w) patch, whose 16 rows (pitch 12 positions x 144 B) fall on 16 distinct bank quads.
__device__ __forceinline__ int brick_index(int row, int TH, int TW, int patch) {
    if (!patch) return row;
    const int i = row & 31, tile = row >> 5;
    int grp, q;
    if (i < 4) { grp = 0; q = i; }
    else if (i < 12) { grp = 1; q = i - 4; }
    else if (i < 16) { grp = 0; q = i - 8; }
    else if (i < 20) { grp = 1; q = i - 8; }
    else if (i < 28) { grp = 0; q = i - 12; }
    else { grp = 1; q = i - 16; }
    const int pidx = tile * 2 + grp;           // 4x4 patch number inside the workgroup tile (TH, TW multiples of 4)
    const int pw = TW >> 2, ph = TH >> 2;
    const int px = pidx % pw, py = (pidx / pw) % ph, plane = pidx / (pw * ph);
    return (plane * TH + py * 4 + (q >> 2)) * TW + px * 4 + (q & 3);
}

// Position of a pipeline stage (= one tap of one 32-channel chunk) in the (chunk, dt, dh, dw) iteration space.  All fields
// are workgroup-uniform (scalar registers).
struct C16Cursor { int ch, si, dt, dh, dw; };
__device__ __forceinline__ void c16_next(C16Cursor& c, int ntv, int dt_lo, int KH, int KW) {
    if (++c.si == ntv) { c.si = 0; ++c.ch; c.dt = dt_lo; c.dh = 0; c.dw = 0; }
    else if (++c.dw == KW) { c.dw = 0; if (++c.dh == KH) { c.dh = 0; ++c.dt; } }
}

// PFD = how many stages ahead a wave requests its weight fragments (the narrower the wave tile, the shorter a stage).
template <int WAVES_M, int WAVES_N, int WM, int WN, int PFD>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M * WAVES_N / 4) void conv_mfma_f16x3_kernel(Conv16Args a) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int NSLOT = C16_SLOTS * 512 / NTHR;            // prefetched 16-byte input pieces per thread
    constexpr int C16_BN = 32 * WN * WAVES_N;
    constexpr int NSET = PFD + 1;                             // register sets of weight fragments (ring)
    static_assert(32 * WM * WAVES_M == C16_BM && WAVES_M * WAVES_N == 8, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int kg = lane >> 5, l31 = lane & 31;

    // Temporal-duplication mode (conv_0 behind a x2 nearest up-sampling in time): the virtual input satisfies
    // a[2i] == a[2i+1], so even output frames see (a[i-1], a[i], a[i]) and odd ones (a[i], a[i], a[i+1]): a 2-tap
    // temporal kernel on the HALF-rate tensor with pre-summed weights (W0, W1+W2) resp. (W0+W1, W2).
    const int par = a.tdup ? (int)blockIdx.y : 0;
    const int pt = a.tdup ? 1 - par : a.KT / 2, ph = a.KH / 2, pw = a.KW / 2;
    const int HT = a.TT + a.KT - 1, HH = a.TH + a.KH - 1, HW = a.HWp;
    const int NPOS = a.TB * HT * HH * HW;  // rows of the LDS tile (row pitch HW >= RW: pad columns are never touched)
    const int RW = a.TW + a.KW - 1;
    const int NREAL = a.TB * HT * HH * RW;  // halo rows that exist

    char* in_lds = smem;
    int* rowpos = reinterpret_cast<int*>(smem + NPOS * C16_ROW);
    int* rowres = rowpos + C16_BM;
    int* gpos = rowres + C16_BM;  // [NREAL] linear input position of every staged halo row, -1 = zero padding
    int* lrow = gpos + NREAL;     // [NREAL] its row in the LDS tile

    const int nNt = a.CoutPad / C16_BN;
    const int ntile = blockIdx.x % nNt;
    int brick = blockIdx.x / nNt;
    const int bw = brick % a.nbW; brick /= a.nbW;
    const int bh = brick % a.nbH; brick /= a.nbH;
    const int bt = brick % a.nbT; brick /= a.nbT;
    const int b0 = brick * a.TB, t0 = bt * a.TT, h0 = bh * a.TH, w0 = bw * a.TW;
    const int n0 = ntile * C16_BN;

    if (tid < C16_BM) {
        int m = brick_index(tid, a.TH, a.TW, a.patch);
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        const int b = b0 + m, t = t0 + it, h = h0 + ih, w = w0 + iw;
        const bool ok = b < a.B;
        const int To = a.tdup ? 2 * a.T : a.T, to = a.tdup ? 2 * t + par : t;  // output frame
        rowpos[tid] = ok ? ((b * To + to) * a.H + h) * a.W + w : -1;
        rowres[tid] = ok ? ((b * (To / a.rt) + to / a.rt) * (a.H / a.rs) + h / a.rs) * (a.W / a.rs) + w / a.rs : 0;
    }
    // temporal taps whose whole brick meets zero padding only are skipped: the valid dt form one interval
    int dt_lo = 0, dt_n = 0;
    for (int dt = 0; dt < a.KT; ++dt) {
        const int lo = t0 + dt - pt, hi = lo + a.TT - 1;
        if (hi < 0 || lo >= a.T) continue;
        if (dt_n == 0) dt_lo = dt;
        ++dt_n;
    }
    const int ntv = dt_n * a.KH * a.KW;  // stages (taps) per chunk
    const int total = ntv * a.nchunk;

    for (int p0 = tid; p0 < NREAL; p0 += NTHR) {
        int p = p0;
        const int iw = p % RW; p /= RW;
        const int ih = p % HH; p /= HH;
        const int it = p % HT; p /= HT;
        const int b = b0 + p, t = t0 + it - pt, h = h0 + ih - ph, w = w0 + iw - pw;
        const bool ok = b < a.B && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
        gpos[p0] = ok ? ((b * a.T + t) * a.H + h) * a.W + w : -1;
        lrow[p0] = ((p * HT + it) * HH + ih) * HW + iw;
    }

    int aoff[WM];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        int m = brick_index(wave_m * (32 * WM) + 32 * wm + l31, a.TH, a.TW, a.patch);
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        aoff[wm] = (((m * HT + it) * HH + ih) * HW + iw) * C16_ROW + kg * 32;
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    __syncthreads();
    const long in_row = (long)a.Cin * 4;

    // Input staging: all (<= C16_SLOTS) 16-byte pieces of a thread are requested back to back (one exposed memory
    // latency per chunk instead of one per piece) and the NEXT chunk's pieces are requested a few taps before the
    // current chunk ends, so that latency hides behind MFMA work.  Branch-free clamped loads: a predicated load would be
    // followed by an immediate s_waitcnt vmcnt(0).
    const int ngrp = a.Cin >> 3;
    float4 vin[NSLOT] = {};
#define C16_REQUEST_INPUT(ch_)                                                                                      \
    if (!(I2V_ABLATE & 4)) {                                                                                         \
        int gp_[NSLOT];                                                                                              \
        _Pragma("unroll") for (int u_ = 0; u_ < NSLOT; ++u_) {                                                       \
            const int idx = tid + u_ * NTHR;                                                                         \
            gp_[u_] = gpos[idx < NREAL * 8 ? (idx >> 3) : 0];                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int u_ = 0; u_ < NSLOT; ++u_) {                                                       \
            const int idx = tid + u_ * NTHR;                                                                         \
            const int q = idx & 7;                                                                                   \
            const bool ok = idx < NREAL * 8 && gp_[u_] >= 0 && (ch_) * 4 + (q >> 1) < ngrp;                          \
            const long off = ok ? (long)gp_[u_] * in_row + (long)(ch_) * 128 + q * 16 : 0;                           \
            const float4 v = *reinterpret_cast<const float4*>(a.in + off);                                           \
            vin[u_] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);                                                      \
        }                                                                                                            \
    }
    // A operands (activations) of one k-step (16 channels): hi and lo fragment per 32-row tile, from LDS
    struct AOps { half8 h[WM], l[WM]; };
    AOps a0 = {}, a1 = {};
#ifdef I2V_ABLATE_NONZERO  // ablation with live (pseudo-random, lane-dependent) operands instead of zeros: MFMA power
#define C16_RND(i_) ((_Float16)((float)((((unsigned)tid * 2654435761u + (unsigned)(i_) * 40503u) >> 7) & 2047) * (1.f / 1024.f) - 1.f))
    _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        a0.h[wm][j] = C16_RND(wm * 8 + j); a0.l[wm][j] = C16_RND(100 + wm * 8 + j) * (_Float16)4.8e-4f;
        a1.h[wm][j] = C16_RND(200 + wm * 8 + j); a1.l[wm][j] = C16_RND(300 + wm * 8 + j) * (_Float16)4.8e-4f;
    }
#endif
#ifndef I2V_ABLATE
#define I2V_ABLATE 0  // development builds of tools/conv16_bench: 1 = no LDS operand reads, 2 = no weight loads,
#endif                //   4 = no input staging, 8 = no MFMAs (results are garbage; timing only)
#define C16_LOAD_A(o, toff)                                                                                          \
    if (!(I2V_ABLATE & 1)) {                                                                                         \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) {                                                          \
            const char* p_ = in_lds + aoff[wm] + (toff);                                                             \
            (o).h[wm] = *reinterpret_cast<const half8*>(p_);                                                         \
            (o).l[wm] = *reinterpret_cast<const half8*>(p_ + 16);                                                    \
        }                                                                                                            \
    }
    // B operands (weights) of one stage: [column block][k-step x (hi, lo)], straight from global memory
    half8 bset[NSET][WN][4] = {};
#ifdef I2V_ABLATE_NONZERO
    _Pragma("unroll") for (int e = 0; e < NSET; ++e) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn)
        _Pragma("unroll") for (int f = 0; f < 4; ++f) _Pragma("unroll") for (int j = 0; j < 8; ++j)
            bset[e][wn][f][j] = C16_RND(1000 + ((e * WN + wn) * 4 + f) * 8 + j) * (_Float16)((f & 1) ? 0.01f : 20.f);
#endif
    const long slab = (long)(a.CoutPad / 32) * C16_FRAG;  // bytes per (tap, chunk)
    const char* wlane = a.wp + (long)par * a.wset_stride + (long)(n0 / 32 + wave_n * WN) * C16_FRAG + lane * 16;
#define C16_LOAD_B(SET, c_)                                                                                          \
    if (!(I2V_ABLATE & 2)) {                                                                                         \
        const char* p_ = wlane +                                                                                     \
            (long)((a.tap_base + ((c_).dt * a.KH + (c_).dh) * a.KW + (c_).dw) * a.nchunk + (c_).ch) * slab;          \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn) /* order of use */ \
            bset[SET][wn][f] = *reinterpret_cast<const half8*>(p_ + wn * C16_FRAG + f * 1024);                       \
    }
#define C16_TOFF(c_) ((((c_).dt * HH + (c_).dh) * HW + (c_).dw) * C16_ROW)
    // three terms, tiles interleaved so that consecutive MFMAs never chain on the same accumulator
#define C16_MFMA(o, SET, ks)                                                                                         \
    if (!(I2V_ABLATE & 8)) {                                                                                         \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn)          \
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16((o).h[wm], bset[SET][wn][2 * (ks)], acc[wm][wn], 0, 0, 0);     \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn)          \
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16((o).h[wm], bset[SET][wn][2 * (ks) + 1], acc[wm][wn], 0, 0, 0); \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn)          \
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16((o).l[wm], bset[SET][wn][2 * (ks)], acc[wm][wn], 0, 0, 0);     \
    }

    C16Cursor cc{0, 0, dt_lo, 0, 0};  // the stage being computed
    C16Cursor cb = cc;                // the stage whose weights are requested next
    C16_REQUEST_INPUT(0)
    // Every load of the steady state is UNCONDITIONAL (cursors clamp at the last stage instead): the s_waitcnt counters
    // retire in order, and after a conditional load the compiler has to assume the shortest queue, which turns the wait
    // for the current operands into a wait for the prefetches just issued.
    int nreq = 0;  // stages whose weights have been requested
#pragma unroll
    for (int u = 0; u < PFD; ++u) {
        C16_LOAD_B(u, cb)
        if (++nreq < total) c16_next(cb, ntv, dt_lo, a.KH, a.KW);
    }
    const int pf_stage = ntv > 4 ? ntv - 4 : 0;  // stage of a chunk in which the next chunk's input rows are requested

#define C16_STAGE(u)                                                                                                 \
    {                                                                                                                \
        if (cc.si == 0 && !(I2V_ABLATE & 4)) { /* new chunk: replace the input tile (the only barriers of the loop) */ \
            __syncthreads();                                                                                         \
            _Pragma("unroll") for (int v = 0; v < NSLOT; ++v) {                                                      \
                const int idx = tid + v * NTHR;                                                                      \
                if (idx < NREAL * 8) *reinterpret_cast<float4*>(in_lds + lrow[idx >> 3] * C16_ROW + (idx & 7) * 16) = vin[v]; \
            }                                                                                                        \
            for (int idx = tid + NSLOT * NTHR; idx < NREAL * 8; idx += NTHR) { /* oversized halo bricks only */      \
                const int q = idx & 7;                                                                               \
                const int gp = gpos[idx >> 3];                                                                       \
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
                if (gp >= 0 && cc.ch * 4 + (q >> 1) < ngrp)                                                          \
                    v = *reinterpret_cast<const float4*>(a.in + (long)gp * in_row + (long)cc.ch * 128 + q * 16);     \
                *reinterpret_cast<float4*>(in_lds + lrow[idx >> 3] * C16_ROW + q * 16) = v;                          \
            }                                                                                                        \
            __syncthreads();                                                                                         \
            C16_LOAD_A(a0, C16_TOFF(cc)) /* first k-step of the chunk: the only exposed LDS read */                  \
        }                                                                                                            \
        /* weights of stage s + PFD into the register set that stage s - 1 just released */                         \
        C16_LOAD_B(((u) + PFD) % NSET, cb)                                                                           \
        if (++nreq < total) c16_next(cb, ntv, dt_lo, a.KH, a.KW);                                                    \
        if (cc.si == pf_stage && cc.ch + 1 < a.nchunk) C16_REQUEST_INPUT(cc.ch + 1)                                  \
        C16_LOAD_A(a1, C16_TOFF(cc) + 64)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C16_MFMA(a0, u, 0)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        c16_next(cc, ntv, dt_lo, a.KH, a.KW);                                                                        \
        /* next tap's first k-step (after the last tap of a chunk this reads a stale row: reloaded after re-staging) */ \
        C16_LOAD_A(a0, C16_TOFF(cc))                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        C16_MFMA(a1, u, 1)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    int s0 = 0;
    for (; s0 + NSET <= total; s0 += NSET) {
        C16_STAGE(0)
        C16_STAGE(1)
        if constexpr (NSET > 2) C16_STAGE(2)
        if constexpr (NSET > 3) C16_STAGE(3)
    }
    if (s0 < total) {
        C16_STAGE(0)
        if (s0 + 1 < total) {
            C16_STAGE(1)
            if constexpr (NSET > 3) {
                if (s0 + 2 < total) C16_STAGE(2)
            }
        }
    }

    const int HWo = a.H * a.W;
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int n = n0 + wave_n * (32 * WN) + 32 * wn + l31;
        const bool ncol = n < a.Cout;
        const float bias = (a.bias && ncol) ? a.bias[n] : 0.f;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int p = rowpos[m];
                if (p < 0 || !ncol) continue;
                float v = fmaf(acc[wm][wn][r], a.oscale, bias);
                if (a.res) v += a.res[(long)rowres[m] * a.Cout + n];
                ssum += v;
                ssq = fmaf(v, v, ssq);
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                if (a.epi & EPI_FRAMES) {
                    const int bt_ = p / HWo, hw = p - bt_ * HWo;
                    a.out[((long)bt_ * a.Cout + n) * HWo + hw] = tanhf(v);
                } else {
                    a.out[(long)p * a.Cout + n] = v;
                }
            }
        }
        if (a.stats) {
            // fused normalisation statistics (InstanceNorm of conv_0's output / GroupNorm of the block output): the
            // workgroup tile lies inside one sample; lanes l and l^32 hold the same column -> wavefront shuffle, then one
            // fp64 atomic pair per (wave, column)
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (kg == 0 && ncol) {
                double* dst = a.stats + ((long)b0 * a.Cout + n) * 2;
                atomicAdd(dst, (double)ssum);
                atomicAdd(dst + 1, (double)ssq);
            }
        }
    }
}

// Fragment-major weight layout: [tap][chunk][column block of 32][k-step 2][hi | lo][lane = kg*32 + column][8 fp16]:
// lane (column n, kg) of the wave that owns the column block reads the 8 channels 16*kstep + 8*kg + j of its row as ONE
// 16-byte piece, and the 64 lanes of a load cover 1 KB contiguously.
static inline size_t c16_widx(int tap, int chunk, int n, int c_in_chunk, int lo, int nchunk, int cout_pad) {
    const int nblk = n >> 5, l31 = n & 31, ks = c_in_chunk >> 4, kg = (c_in_chunk >> 3) & 1, j = c_in_chunk & 7;
    return ((((size_t)tap * nchunk + chunk) * (cout_pad / 32) + nblk) * 4 + (ks * 2 + lo)) * 512 + (kg * 32 + l31) * 8 + j;
}

int Conv16Weights::pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, int kh, int kw, double scale) {
    Cin = cin; Cout = cout; KT = kt; KH = kh; KW = kw;
    CoutPad = (cout + 31) / 32 * 32;
    if (CoutPad > 64 && CoutPad % 128) CoutPad = (CoutPad + 127) / 128 * 128;
    nchunk = (cin + C16_KC - 1) / C16_KC;
    const int ntaps = kt * kh * kw;
    // power-of-two pre-scale: largest |w| lands in [2^13, 2^14) so every lo part of a non-negligible weight is a normal
    // fp16 number (full 2^-22 split precision) and hi stays far from the fp16 overflow threshold
    double wmax = 0.0;
    for (size_t i = 0; i < (size_t)cout * cin * ntaps; ++i) wmax = std::max(wmax, std::fabs((double)w_src[i] * scale));
    wexp = 0;
    if (wmax > 0.0 && std::isfinite(wmax)) {
        wexp = (int)std::floor(std::log2(16384.0 / wmax));
        wexp = std::max(-40, std::min(40, wexp));
    }
    const double pre = std::ldexp(1.0, wexp);
    std::vector<_Float16> p((size_t)ntaps * nchunk * CoutPad * 64, (_Float16)0.f);
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < ntaps; ++tap) {
                const float v = (float)((double)w_src[((size_t)n * cin + c) * ntaps + tap] * scale * pre);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                p[c16_widx(tap, c / C16_KC, n, c % C16_KC, 0, nchunk, CoutPad)] = hi;
                p[c16_widx(tap, c / C16_KC, n, c % C16_KC, 1, nchunk, CoutPad)] = lo;
            }
    int rc = w.upload(p.data(), p.size() * 2);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

int Conv16Weights::pack_tdup(const float* w_src, const float* bias_src, int cout, int cin, double scale) {
    // two 2x3x3 kernels from one 3x3x3 kernel: parity 0 = (W[0], W[1]+W[2]), parity 1 = (W[0]+W[1], W[2]) along time
    std::vector<float> w2((size_t)2 * cout * cin * 18);
    for (int par = 0; par < 2; ++par)
        for (size_t nc = 0; nc < (size_t)cout * cin; ++nc)
            for (int hw = 0; hw < 9; ++hw) {
                const double w0 = w_src[nc * 27 + hw], w1 = w_src[nc * 27 + 9 + hw], w2v = w_src[nc * 27 + 18 + hw];
                float* dst = &w2[((size_t)par * cout * cin + nc) * 18];
                dst[hw] = (float)(par == 0 ? w0 : w0 + w1);
                dst[9 + hw] = (float)(par == 0 ? w1 + w2v : w2v);
            }
    // both sets share one power-of-two pre-scale: pack them as one [2*cout] tensor, then split the buffer
    Cin = cin; Cout = cout; KT = 2; KH = 3; KW = 3; tdup = true;
    CoutPad = (cout + 31) / 32 * 32;
    if (CoutPad > 64 && CoutPad % 128) CoutPad = (CoutPad + 127) / 128 * 128;
    nchunk = (cin + C16_KC - 1) / C16_KC;
    const int ntaps = 18;
    double wmax = 0.0;
    for (float v : w2) wmax = std::max(wmax, std::fabs((double)v * scale));
    wexp = 0;
    if (wmax > 0.0 && std::isfinite(wmax)) wexp = std::max(-40, std::min(40, (int)std::floor(std::log2(16384.0 / wmax))));
    const double pre = std::ldexp(1.0, wexp);
    const size_t set_halfs = (size_t)ntaps * nchunk * CoutPad * 64;
    std::vector<_Float16> p(2 * set_halfs, (_Float16)0.f);
    for (int par = 0; par < 2; ++par)
        for (int n = 0; n < cout; ++n)
            for (int c = 0; c < cin; ++c)
                for (int tap = 0; tap < ntaps; ++tap) {
                    const float v = (float)((double)w2[(((size_t)par * cout + n) * cin + c) * 18 + tap] * scale * pre);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    p[par * set_halfs + c16_widx(tap, c / C16_KC, n, c % C16_KC, 0, nchunk, CoutPad)] = hi;
                    p[par * set_halfs + c16_widx(tap, c / C16_KC, n, c % C16_KC, 1, nchunk, CoutPad)] = lo;
                }
    set_bytes = (long)set_halfs * 2;
    int rc = w.upload(p.data(), p.size() * 2);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int PFD>
static int launch16(const Conv16Args& a, unsigned nblk, size_t lds, hipStream_t st) {
    auto kern = conv_mfma_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, PFD>;
    static bool attr_set = false;
    if (!attr_set) {
        I2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, a.tdup ? 2 : 1), dim3(64 * WAVES_M * WAVES_N), lds, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

bool conv16_can_fuse_stats(int T, int H, int W) {
    // the 256-position brick stays inside one sample when the sample has at least 256 positions (power-of-two dims)
    return (long)T * H * W >= C16_BM;
}

int conv16_forward(const Conv16Weights& wts, const void* in_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                   int H, int W, int epi, hipStream_t st, double* stats) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "conv16: weights not packed");
    I2V_REQUIRE(wts.Cin % 8 == 0, I2V_E_INVALID, "conv16: Cin %d must be a multiple of 8", wts.Cin);
    Conv16Args a{};
    a.in = static_cast<const char*>(in_hl16); a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.B = B; a.T = T; a.H = H; a.W = W; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.KT = wts.KT; a.KH = wts.KH; a.KW = wts.KW; a.tap_base = 0;
    a.tdup = wts.tdup ? 1 : 0;
    a.wset_stride = wts.set_bytes;
    if (wts.tdup) {  // T is the OUTPUT frame count; the (half-rate) input has T / 2 frames
        I2V_REQUIRE(T % 2 == 0 && !res, I2V_E_INVALID, "conv16: temporal-duplication mode needs an even frame count and no residual");
        T /= 2;
        a.T = T;
    }
    if (T == 1 && wts.KT == 3) {  // a single frame only ever meets the centre time-slice of the kernel (rest is padding)
        a.KT = 1;
        a.tap_base = wts.KH * wts.KW;
    }
    a.rt = res ? rt : 1; a.rs = res ? rs : 1; a.epi = epi;
    a.stats = stats;
    a.oscale = (float)std::ldexp(1.0, -wts.wexp);
    int TW = W < 8 ? W : 8, TH = H < 8 ? H : 8;
    int rem = C16_BM / (TW * TH);
    int TT = T < rem ? T : rem;
    rem /= TT;
    while (rem > 1 && W >= TW * 2) { TW *= 2; rem /= 2; }
    while (rem > 1 && H >= TH * 2) { TH *= 2; rem /= 2; }
    const int TB = rem;
    I2V_REQUIRE(TB * TT * TH * TW == C16_BM && T % TT == 0 && H % TH == 0 && W % TW == 0, I2V_E_INVALID,
                "conv16: cannot tile [T=%d,H=%d,W=%d] into bricks of %d positions", T, H, W, C16_BM);
    I2V_REQUIRE(!stats || TB == 1, I2V_E_INVALID, "conv16: fused statistics need bricks inside one sample");
    a.TB = TB; a.TT = TT; a.TH = TH; a.TW = TW;
    a.nbB = (B + TB - 1) / TB; a.nbT = T / TT; a.nbH = H / TH; a.nbW = W / TW;
    a.HWp = TW + a.KW - 1;
    a.patch = (TW % 4 == 0 && TH % 4 == 0) ? 1 : 0;
    if (a.patch) {  // a halo row pitch = 4 or 12 (mod 16) makes the 4x4 patches conflict-free; keep it if LDS allows
        int hp = a.HWp;
        while (hp % 16 != 4 && hp % 16 != 12) ++hp;
        const size_t rows = (size_t)TB * (TT + a.KT - 1) * (TH + a.KH - 1) * hp;
        const size_t need = rows * C16_ROW + 2 * C16_BM * 4 + rows * 8;
        if (need <= 160 * 1024) a.HWp = hp;
    }
    const int npos = TB * (TT + a.KT - 1) * (TH + a.KH - 1) * a.HWp;
    const int BN = a.CoutPad % 128 == 0 ? 128 : (a.CoutPad % 64 == 0 ? 64 : 32);
    const size_t lds = (size_t)npos * C16_ROW + 2 * C16_BM * 4 + (size_t)npos * 8;  // (tables sized for npos >= NREAL rows)
    I2V_REQUIRE(lds <= 160 * 1024, I2V_E_INVALID, "conv16: LDS %zu bytes exceeds 160 KiB", lds);
    const long nblk = (long)a.nbB * a.nbT * a.nbH * a.nbW * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 31), I2V_E_INVALID, "conv16: grid of %ld workgroups", nblk);
    // weight prefetch distance: one 64x64-tile stage is 24 MFMAs (768 cycles) per wave, narrower tiles need more stages
    // (a 16-wave variant -- 4 waves per SIMD, wave tile 32x64, 128 VGPRs -- of the LDS-weights design was 5 % slower)
    if (BN == 128) return launch16<4, 2, 2, 2, 1>(a, (unsigned)nblk, lds, st);
    if (BN == 64) return launch16<4, 2, 2, 1, 2>(a, (unsigned)nblk, lds, st);
    return launch16<8, 1, 1, 1, 3>(a, (unsigned)nblk, lds, st);
}

}  // namespace i2v

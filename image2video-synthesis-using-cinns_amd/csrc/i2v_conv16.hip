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
// weights are split on the host at load time.
//
// Tiling: 512 threads = 8 wavefronts (2 per SIMD) per workgroup, 256 output positions (TB x TT x TH x TW brick) x BN
// output channels (128/64/32), wave tile up to 64 x 64.  Per 32-channel K chunk the input halo brick is staged once in
// LDS (rows padded 128 -> 144 B, MFMA rows assigned to 4x4 (h,w) patches: conflict-free ds_read_b128) and reused by all
// taps; the next chunk's rows are requested from HBM a few taps ahead.  The [BN][32] weight slab of each tap is
// double-buffered in LDS and requested three stages ahead, over VIRTUAL stages that run across chunk boundaries (round 6).
// The tap loop is software-pipelined: the operands of the next k-step (second half of this tap / first half of the next
// tap) are read from LDS while the current k-step's 12 MFMAs per wave run -- one ds_read_b128 per gap between two MFMAs
// (round 6: sched_group_barrier) -- with ONE barrier per tap placed between the two k-steps (it publishes the next tap's
// weights).  A brick that spans the map's whole time extent stages no temporal halo frames (round 6, `tclip`).
// Round-6 A/Bs and ablations of the stage loop: profiles/r06_q_conv16_direct_ab.txt.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "i2v_conv.h"

namespace i2v {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C16_BM = 256, C16_KC = 32;
constexpr int C16_ROW = 144;  // bytes per staged row: 32 channels x 4 B + 16 B pad
constexpr int C16_SLOTS = 12; // prefetched 16-byte input pieces per thread and chunk (768 halo rows); larger bricks
                              // stage the remainder synchronously

typedef float c16_f32x4 __attribute__((ext_vector_type(4)));

struct Conv16Args {
    const char* in;   // hl16 channels-last [B][T][H][W][Cin]
    const char* zeros;  // >= 16 zero bytes: source of padding pieces (no select behind the prefetch loads)
    const char* wp;   // hl16 weights [tap][chunk][CoutPad][128 B]
    const float* bias;
    const float* res;
    float* out;       // fp32 channels-last [B][T][H][W][Cout]
    double* stats;    // optional [B][Cout][2]: per-(sample, channel) sum / sum of squares of the stored values (TB == 1)
    int* range_flag;  // optional (EPI_HL16): set when a stored value leaves the fp16 range
    float* part;      // split-K (ksplit > 1): raw partial sums [ksplit][positions][Cout] instead of `out`
    int ksplit;       // the K chunks are divided over `ksplit` workgroups per tile (blockIdx.y); 1 = off
    long part_stride; // floats per split slice
    int B, T, H, W, Cin, Cout, CoutPad, nchunk;  // T,H,W: geometry of the INPUT tensor
    int tdup;            // 1: temporal-duplication mode -- the output has 2T frames, two tiles (frame parities) per brick
    int tskperm;         // TSK: 1 = the permuted wave -> row slab order (always, except in measurement A/Bs)
    int tclip;           // 1: the brick spans the map's whole time extent and only its TT real frames are staged (no temporal halo:
                         //    those frames are zero padding); every (row block, tap) pair that would read them is skipped -- by the
                         //    brick-level tap list (TT == 1) or by the TSK masks, which then also gate the operand reads
    long wset_stride;    // bytes between the two parity weight sets (tdup)
    int KT, KH, KW, tap_base, ztap;  // ztap: index of the all-zero weight slab (stage padding)
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

// TPS = taps per pipeline stage: the narrower the channel tile, the more taps share one weight buffer / barrier
// (BN x TPS = 128 rows per buffer for every variant).
// TSK (round 6): temporal tap skipping per MFMA row block.  A 3-tap temporal kernel on a TWO-frame map (g_0.conv_1: 2 x 8 x 8) meets
// zero padding in one of its three temporal taps for EVERY output frame (frame 0: the tap at t - 1, frame 1: the tap at t + 1), but the
// brick holds both frames, so the brick-level tap list keeps all 27 taps and a third of the MFMAs multiply zero rows.  A 32-row MFMA
// block lies inside one frame: with TSK each tap carries its temporal offset (low bits of its LDS offset) and a row block whose frame
// meets only padding for the tap skips its MFMAs -- 18 instead of 27 taps of matrix work, the same bits (the skipped products are
// exact zeros).
template <int WAVES_M, int WAVES_N, int WM, int WN, int TPS, bool TSK = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M * WAVES_N / 4) void conv_mfma_f16x3_kernel(Conv16Args a) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;              // 512 (2 waves per SIMD) or 1024 (4 per SIMD)
    constexpr int NSLOT = C16_SLOTS * 512 / NTHR;            // prefetched 16-byte input pieces per thread
    constexpr int C16_BN = 32 * WN * WAVES_N;
    constexpr int WBUF = TPS * C16_BN * C16_ROW;  // bytes per weight buffer
    static_assert(32 * WM * WAVES_M == C16_BM && (WAVES_M * WAVES_N == 8 || WAVES_M * WAVES_N == 16), "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // TSK: a workgroup's waves w and w + 4 share a SIMD (waves go to the SIMDs in a cyclic order), and a 64-row slab of the two-frame
    // brick lies in ONE frame -- in the plain order the two waves of a SIMD hold the same frame, so a tap that only one frame meets
    // leaves two SIMDs idle and the other two with both their waves busy: the stage takes as long as an unmasked one.  The permuted
    // order gives every SIMD one wave of each frame (which rows a wave owns changes, no output's arithmetic does).
    const int wm_idx = wave / WAVES_N;
    const int wave_m = !(TSK && a.tskperm) ? wm_idx : WAVES_M == 4 ? ((0x2130 >> (4 * wm_idx)) & 3) : WAVES_M == 8 ? ((0x76325410 >> (4 * wm_idx)) & 7) : wm_idx;
    const int wave_n = wave % WAVES_N;
    const int kg = lane >> 5, l31 = lane & 31;

    // Temporal-duplication mode (conv_0 behind a x2 nearest up-sampling in time): the virtual input satisfies
    // a[2i] == a[2i+1], so even output frames see (a[i-1], a[i], a[i]) and odd ones (a[i], a[i], a[i+1]): a 2-tap
    // temporal kernel on the HALF-rate tensor with pre-summed weights (W0, W1+W2) resp. (W0+W1, W2).
    // Tile order and the XCDs.  Workgroup b runs on XCD b % 8, each XCD with its own L2.  Bricks are numbered w-fastest
    // (8 bricks per row at 64 x 64), so the plain order gives XCD x the bricks of ONE w-column: its 32 concurrent
    // workgroups cover all h-rows and t-slabs of that column and share their h- and t-halos in L2 (measured: 1.8x the
    // algorithmic input bytes; giving each XCD a contiguous tile range instead shares w/h but not t and reads MORE, and
    // it makes every XCD stream all N-tiles' weights instead of a quarter of them).  Only the two frame parities of a
    // temporal-duplication brick -- same input, different weights -- need placing: consecutive slots of the same XCD.
    const unsigned nb_ = gridDim.x;  // tiles (x 2 parities in tdup mode)
    const bool pair_ = a.tdup && (nb_ & 15) == 0;
    const int par = !a.tdup ? 0 : pair_ ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x >= (nb_ >> 1));
    const int tile_id = !a.tdup ? (int)blockIdx.x
                        : pair_ ? (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : (int)(blockIdx.x % (nb_ >> 1));
    const int pt = a.tdup ? 1 - par : a.KT / 2, ph = a.KH / 2, pw = a.KW / 2;
    const int HT = a.tclip ? a.TT : a.TT + a.KT - 1, HH = a.TH + a.KH - 1, HW = a.HWp;
    const int tsh = a.tclip ? 0 : pt;   // staged frame 0 = input frame t0 - tsh
    const int NPOS = a.TB * HT * HH * HW;
    const int ntaps = a.KT * a.KH * a.KW;

    char* in_lds = smem;
    char* w_lds = smem + NPOS * C16_ROW;
    int* rowpos = reinterpret_cast<int*>(w_lds + 2 * WBUF);
    int* rowres = rowpos + C16_BM;
    int* taplist = rowres + C16_BM;  // [0] = padded tap count, [1..32] weight-slab index, [33..64] LDS byte offset of the tap
    int* gpos = taplist + 72;  // [NPOS] linear input position of every staged halo row, -1 = zero padding

    const int nNt = a.CoutPad / C16_BN;
    const int ntile = tile_id % nNt;
    int brick = tile_id / nNt;
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
        const bool ok = b < a.B && m < a.TB;  // (a brick is only partly filled when LDS limits the samples per brick)
        const int To = a.tdup ? 2 * a.T : a.T, to = a.tdup ? 2 * t + par : t;  // output frame
        rowpos[tid] = ok ? ((b * To + to) * a.H + h) * a.W + w : -1;
        rowres[tid] = ok ? ((b * (To / a.rt) + to / a.rt) * (a.H / a.rs) + h / a.rs) * (a.W / a.rs) + w / a.rs : 0;
    }
    if (tid == 0) {
        int cnt = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int dw = tap % a.KW, dh = (tap / a.KW) % a.KH, dt = tap / (a.KH * a.KW);
            const int lo = t0 + dt - pt, hi = lo + a.TT - 1;
            if (hi < 0 || lo >= a.T) continue;  // the whole brick meets zero padding only
            taplist[1 + cnt] = a.tap_base + tap;
            // (C16_ROW is a multiple of 16: the low bits are free -- also of a negative offset, which tclip produces for dt < pt)
            taplist[33 + cnt] = (((dt - (pt - tsh)) * HH + dh) * HW + dw) * C16_ROW | (TSK ? dt : 0);
            ++cnt;
        }
        while (cnt % TPS) {  // pad the stage with the all-zero weight slab
            taplist[1 + cnt] = a.ztap;
            taplist[33 + cnt] = TSK ? 3 : 0;   // (TSK: temporal offset 3 = outside the map for every row block: skipped)
            ++cnt;
        }
        taplist[0] = cnt;
    }

    for (int p0 = tid; p0 < NPOS; p0 += NTHR) {
        int p = p0;
        const int iw = p % HW; p /= HW;
        const int ih = p % HH; p /= HH;
        const int it = p % HT; p /= HT;
        const int b = b0 + p, t = t0 + it - tsh, h = h0 + ih - ph, w = w0 + iw - pw;
        const bool ok = b < a.B && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W &&
                        iw < a.TW + a.KW - 1;
        gpos[p0] = ok ? ((b * a.T + t) * a.H + h) * a.W + w : -1;
    }

    int aoff[WM], boff[WN];
    int tbs[WM];   // TSK: input frame of the row block for temporal offset 0 (the launcher guarantees 32-row blocks inside one frame)
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        int m = brick_index(wave_m * (32 * WM) + 32 * wm + l31, a.TH, a.TW, a.patch);
        const int iw = m % a.TW; m /= a.TW;
        const int ih = m % a.TH; m /= a.TH;
        const int it = m % a.TT; m /= a.TT;
        aoff[wm] = ((((m < a.TB ? m : 0) * HT + it) * HH + ih) * HW + iw) * C16_ROW + kg * 32;
        tbs[wm] = TSK ? __builtin_amdgcn_readfirstlane(t0 + it - pt) : 0;
    }
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) boff[wn] = (wave_n * (32 * WN) + 32 * wn + l31) * C16_ROW + kg * 32;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WN; ++wn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    constexpr int WF4 = TPS * C16_BN * 8;      // 16-byte pieces per stage of weights (TPS slabs)
    constexpr int WLD = WF4 / NTHR;
    static_assert(WF4 % NTHR == 0 && WLD >= 1 && WLD <= 2, "weight pieces per thread");
    const long slab = (long)a.CoutPad * 128;  // bytes per (tap, chunk)
    __syncthreads();
    const int ntv = taplist[0];
    const long in_row = (long)a.Cin * 4;

    // Input staging: all (<= C16_SLOTS) 16-byte pieces of a thread are requested back to back (one exposed memory
    // latency per chunk instead of one per piece) and the NEXT chunk's pieces are requested a few taps before the
    // current chunk ends, so that latency hides behind MFMA work.
    const int ngrp = a.Cin >> 3;
    float4 vin[NSLOT];
#define C16_REQUEST_INPUT(ch_)                                                                                      \
    {                                                                                                                \
        int gp_[NSLOT];                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NSLOT; ++u) {                                                          \
            const int idx = tid + u * NTHR;                                                                          \
            gp_[u] = gpos[idx < NPOS * 8 ? (idx >> 3) : 0];                                                          \
        }                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < NSLOT; ++u) {                                                          \
            const int idx = tid + u * NTHR;                                                                          \
            const int q = idx & 7;                                                                                   \
            const bool ok = idx < NPOS * 8 && gp_[u] >= 0 && (ch_) * 4 + (q >> 1) < ngrp;                            \
            const c16_f32x4 t_ = *reinterpret_cast<const c16_f32x4*>(                                                  \
                ok ? a.in + (long)gp_[u] * in_row + (long)(ch_) * 128 + q * 16 : a.zeros);                          \
            vin[u] = make_float4(t_[0], t_[1], t_[2], t_[3]);                                                        \
        }                                                                                                            \
    }
    struct Ops { half8 ah[WM], al[WM], bh[WN], bl[WN]; };
    Ops o0, o1;
    // TSK: WM-bit mask of the row blocks whose frame meets data for the tap with table value tapv_ (wave-uniform); a masked row
    // block neither reads its A operands nor multiplies, and a wave whose row blocks are all masked skips the tap's B reads too
    // (with tclip the masked A address would lie outside the staged frames)
#define C16_TAPMASK(tapv_) ([&]() -> unsigned {                                                                      \
        if constexpr (!TSK) return ~0u;                                                                              \
        const int dt_ = __builtin_amdgcn_readfirstlane((tapv_) & 3);                                                 \
        unsigned m_ = 0u;                                                                                            \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) m_ |= ((unsigned)(tbs[wm] + dt_) < (unsigned)a.T ? 1u : 0u) << wm; \
        return m_; }())
#define C16_LOAD_OPS(o, aoffs, wbuf, koff, msk_)                                                                      \
    { if (!(C16_ABL & 2)) {                                                                                          \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) if (!TSK || (((msk_) >> wm) & 1u)) {                        \
            const char* p_ = in_lds + aoff[wm] + (TSK ? ((aoffs) & ~3) : (aoffs)) + (koff);                          \
            (o).ah[wm] = *reinterpret_cast<const half8*>(p_);                                                        \
            (o).al[wm] = *reinterpret_cast<const half8*>(p_ + 16);                                                   \
        }                                                                                                            \
        if (!TSK || ((msk_) & ((1u << WM) - 1u))) {                                                                  \
            _Pragma("unroll") for (int wn = 0; wn < WN; ++wn) {                                                      \
                const char* p_ = (wbuf) + boff[wn] + (koff);                                                         \
                (o).bh[wn] = *reinterpret_cast<const half8*>(p_);                                                    \
                (o).bl[wn] = *reinterpret_cast<const half8*>(p_ + 16);                                               \
            }                                                                                                        \
        }                                                                                                            \
    } }
    // three terms, tiles interleaved so that consecutive MFMAs never chain on the same accumulator
#define C16_MFMA(o, tq_)                                                                                             \
    { if (!(C16_ABL & 1)) {                                                                                          \
        bool on_[WM];                                                                                                \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) on_[wm] = !TSK || ((onmask >> ((tq_) * WM + wm)) & 1u);     \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) if (on_[wm]) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn) \
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16((o).ah[wm], (o).bh[wn], acc[wm][wn], 0, 0, 0);      \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) if (on_[wm]) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn) \
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16((o).ah[wm], (o).bl[wn], acc[wm][wn], 0, 0, 0);      \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) if (on_[wm]) _Pragma("unroll") for (int wn = 0; wn < WN; ++wn) \
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16((o).al[wm], (o).bh[wn], acc[wm][wn], 0, 0, 0);      \
    } }
    // split-K (tiny feature maps: few tiles, long K): this workgroup covers the chunks [ch0, ch1)
    const int ch0 = (int)((long)a.nchunk * blockIdx.y / a.ksplit), ch1 = (int)((long)a.nchunk * (blockIdx.y + 1) / a.ksplit);
    C16_REQUEST_INPUT(ch0)
    const int nst = ntv / TPS;  // stages per chunk
    constexpr int PF = TPS >= 4 ? 1 : 4 / TPS;
    const int pf_stage = nst > PF ? nst - PF : 0;
    const int* tapw = taplist + 1;
    const int* tapo = taplist + 33;
    // per-thread weight piece geometry (constant over the kernel): piece f of a stage = tap f / (BN*8) of the stage,
    // 16-byte piece f % (BN*8) of that tap's [BN][128 B] slab
    int wtis[WLD], wsrc[WLD], wdst[WLD];
#pragma unroll
    for (int u = 0; u < WLD; ++u) {
        const int f = tid + u * NTHR;
        const int tis = TPS == 1 ? 0 : f / (C16_BN * 8), fr = f % (C16_BN * 8);   // (TPS == 1: a compile-time 0 keeps the slab address scalar)
        wtis[u] = tis;
        wsrc[u] = fr * 16;
        wdst[u] = tis * (C16_BN * C16_ROW) + (fr >> 3) * C16_ROW + (fr & 7) * 16;
    }
    const long wtap_stride = (long)a.nchunk * slab;

#ifndef C16_ABL
#define C16_ABL 0      // measurement builds only (tools/conv16_bench_abl*): bit 0 no MFMAs, 1 no operand reads, 2 no weight park / request, 3 no per-stage barrier
#endif
#ifndef C16_ILV
#define C16_ILV 1      // 0 (measurement builds): the operand reads of the next k-step as one burst in front of the k-step's MFMAs (round 5)
#endif
    // The 2 (WM + WN) ds_read_b128 of the next k-step go ONE PER GAP between this k-step's MFMAs (sched_group_barrier: MFMA, DS read,
    // MFMA, DS read, ...).  As a burst in front of the MFMAs -- all eight waves leave the stage barrier together -- 64 reads queue up at
    // the LDS and every wave's first MFMA waits behind its own eight (in-order issue): the ablations (profiles/r06_q) put the burst at
    // 0.22 of the 0.50 ms of g_0.conv_0.  (TSK instantiations branch around masked row blocks: their k-steps are not one basic block.)
#define C16_INTERLEAVE                                                                                               \
    {                                                                                                                \
        constexpr int NR_ = 2 * (WM + WN), NM_ = 3 * WM * WN, NI_ = NR_ < NM_ ? NR_ : NM_;                           \
        _Pragma("unroll") for (int i_ = 0; i_ < NI_; ++i_) {                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                       \
        }                                                                                                            \
        if constexpr (NM_ > NI_) __builtin_amdgcn_sched_group_barrier(0x008, NM_ - NI_, 0);                          \
        if constexpr (NR_ > NI_) __builtin_amdgcn_sched_group_barrier(0x100, NR_ - NI_, 0);                          \
    }
#ifndef C16_XCHUNK
#define C16_XCHUNK 1   // 0 (measurement builds): round 5's per-chunk weight pipeline, which restarts -- two exposed fetches -- at every chunk
#endif
    // The weight pipeline runs over VIRTUAL stages v = (chunk, stage) of the workgroup's whole K range (round 6): the slab of virtual
    // stage v + 3 is requested at stage v, also across a chunk boundary, so that a new chunk finds its first slab parked in LDS and its
    // second one in registers.  Before, every chunk fetched its stage-0 slab straight into LDS and waited for the stage-1 slab at the
    // top of stage 0 -- two exposed L2 / MALL round trips per chunk, ~8 us of the 17 us a 9-tap chunk of g_0.conv_0 took.  The LDS
    // buffer and the register set of a stage follow the parity of v (a chunk may have an odd number of stages).
    const char* wbase0 = a.wp + (long)par * a.wset_stride + (long)n0 * 128;
    float4 wra[WLD], wrb[WLD];
#pragma unroll
    for (int u = 0; u < WLD; ++u) wra[u] = wrb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#define C16_REQUEST_W(WR, c_, s_)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < WLD; ++u)                                                                  \
        WR[u] = *reinterpret_cast<const float4*>(wbase0 + (long)(c_) * slab + (long)tapw[(s_) * TPS + wtis[u]] * wtap_stride + wsrc[u]);
    int rq_c = ch0, rq_s = 0;   // the next virtual stage to request: chunk, stage (past the end: the last slab again, harmless)
#define C16_RQ_NEXT(WR)                                                                                              \
    {                                                                                                                \
        const bool in_ = rq_c < ch1 && (C16_XCHUNK || rq_c == ch);                                                   \
        const int c_ = in_ ? rq_c : (C16_XCHUNK ? ch1 - 1 : ch), s_ = in_ ? rq_s : nst - 1;                          \
        C16_REQUEST_W(WR, c_, s_)                                                                                    \
        if (++rq_s == nst) { rq_s = 0; ++rq_c; }                                                                     \
    }
    int ch = ch0, sidx = 0;     // the current virtual stage: chunk, stage inside the chunk
    const int nv = nst > 0 ? (ch1 - ch0) * nst : 0;
    // LDS byte offsets of the taps of the current and of the next stage, fetched from the table a stage AHEAD: a
    // ds_read_b32 right in front of the operand reads it addresses exposes one LDS round trip per k-step
    int tcur[TPS], tnxt[TPS];
#pragma unroll
    for (int t = 0; t < TPS; ++t) tcur[t] = tnxt[t] = 0;
    // Start of a chunk (stage 0): the previous brick's readers are done (first barrier), the prefetched input rows go to LDS, and the
    // chunk's first operands are read -- the only exposed LDS read.  Only the workgroup's FIRST chunk (or every chunk with
    // C16_XCHUNK = 0) also fetches its stage-0 slab straight into LDS and starts the request queue.
#define C16_CHUNK_START(vpar_)                                                                                       \
    {                                                                                                                \
        __syncthreads();                                                                                             \
        _Pragma("unroll") for (int u = 0; u < NSLOT; ++u) {                                                          \
            const int idx = tid + u * NTHR;                                                                          \
            if (idx < NPOS * 8) *reinterpret_cast<float4*>(in_lds + (idx >> 3) * C16_ROW + (idx & 7) * 16) = vin[u]; \
        }                                                                                                            \
        for (int idx = tid + NSLOT * NTHR; idx < NPOS * 8; idx += NTHR) {  /* oversized halo bricks only */          \
            const int q = idx & 7;                                                                                   \
            const int gp = gpos[idx >> 3];                                                                           \
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            if (gp >= 0 && ch * 4 + (q >> 1) < ngrp)                                                                 \
                v = *reinterpret_cast<const float4*>(a.in + (long)gp * in_row + (long)ch * 128 + q * 16);            \
            *reinterpret_cast<float4*>(in_lds + (idx >> 3) * C16_ROW + q * 16) = v;                                  \
        }                                                                                                            \
        if (ch == ch0 || !C16_XCHUNK) {                                                                              \
            rq_c = ch; rq_s = 0;                                                                                     \
            _Pragma("unroll") for (int u = 0; u < WLD; ++u)                                                          \
                *reinterpret_cast<float4*>(w_lds + (vpar_) * WBUF + wdst[u]) =                                       \
                    *reinterpret_cast<const float4*>(wbase0 + (long)ch * slab + (long)tapw[wtis[u]] * wtap_stride + wsrc[u]); \
            if (++rq_s == nst) { rq_s = 0; ++rq_c; }                                                                 \
            if ((vpar_) == 0) { C16_RQ_NEXT(wra) C16_RQ_NEXT(wrb) } else { C16_RQ_NEXT(wrb) C16_RQ_NEXT(wra) }       \
        }                                                                                                            \
        __syncthreads();                                                                                             \
        _Pragma("unroll") for (int t = 0; t < TPS; ++t) {                                                            \
            tcur[t] = tapo[t];                                                                                       \
            tnxt[t] = tapo[(nst > 1 ? TPS : 0) + t];                                                                 \
        }                                                                                                            \
        C16_LOAD_OPS(o0, tcur[0], w_lds + (vpar_) * WBUF, 0, C16_TAPMASK(tcur[0]))                                   \
    }
    // One pipeline stage = TPS taps.  WR holds the weights of virtual stage v + 1, requested TWO stages ago (an L2/MALL miss on
    // a slab that every workgroup wants at the same moment costs more than one stage): park them in the other LDS buffer
    // -- its last readers finished before the previous barrier -- and request stage v + 3 into the same registers.
    // Then 2*TPS k-steps: the operands of k-step q+1 are read from LDS while k-step q's MFMAs run; the stage's single
    // barrier sits in front of the last k-step and publishes the next stage's weights.
    // (s_setprio around the MFMA block and dropping the scheduling fences were measured: no effect.)
#define C16_STAGE(vpar_, WR)                                                                                         \
    {                                                                                                                \
        if (sidx == 0) C16_CHUNK_START(vpar_)                                                                        \
        unsigned onmask = ~0u;   /* TSK: bit (tap of the stage) * WM + row block = the block's frame meets data for the tap */ \
        unsigned onnext = ~0u;   /* the same for tap 0 of the NEXT stage (its first operands are read behind this stage's barrier) */ \
        if constexpr (TSK) {                                                                                         \
            onmask = 0u;                                                                                             \
            _Pragma("unroll") for (int t = 0; t < TPS; ++t) onmask |= C16_TAPMASK(tcur[t]) << (t * WM);              \
            onnext = C16_TAPMASK(tnxt[0]);                                                                           \
        }                                                                                                            \
        const char* wb = w_lds + (vpar_) * WBUF;                                                                     \
        char* wnext = w_lds + (1 - (vpar_)) * WBUF;                                                                  \
        /* unconditional (after the last stage: a harmless re-park / re-request of the last slab): the vmcnt queue    \
           retires in order, and behind a conditional load the compiler must assume the shortest queue */           \
        if (!(C16_ABL & 4)) {                                                                                        \
        _Pragma("unroll") for (int u = 0; u < WLD; ++u) *reinterpret_cast<float4*>(wnext + wdst[u]) = WR[u];         \
        C16_RQ_NEXT(WR)                                                                                              \
        }                                                                                                            \
        if (sidx == pf_stage && ch + 1 < ch1) C16_REQUEST_INPUT(ch + 1)                                              \
        int tnn[TPS]; /* tap offsets of stage sidx + 2 (clamped), needed one stage from now */                       \
        _Pragma("unroll") for (int t = 0; t < TPS; ++t) tnn[t] = tapo[(sidx + 2 < nst ? sidx + 2 : nst - 1) * TPS + t]; \
        _Pragma("unroll") for (int q = 0; q < 2 * TPS; ++q) {                                                        \
            if (q + 1 < 2 * TPS) {                                                                                   \
                const int tq = (q + 1) >> 1, sq = (q + 1) & 1;                                                       \
                if ((q + 1) & 1) C16_LOAD_OPS(o1, tcur[tq], wb + tq * (C16_BN * C16_ROW), 64 * sq, onmask >> (tq * WM)) \
                else C16_LOAD_OPS(o0, tcur[tq], wb + tq * (C16_BN * C16_ROW), 64 * sq, onmask >> (tq * WM))           \
            } else {                                                                                                 \
                if (!(C16_ABL & 8)) __syncthreads();                                                                 \
                /* (interleaved form: unconditional -- behind a chunk's last stage a harmless read of valid addresses --  \
                   so that the k-step stays one basic block) */                                                      \
                if ((C16_ILV && !TSK) || sidx + 1 < nst) C16_LOAD_OPS(o0, tnxt[0], wnext, 0, onnext)                 \
            }                                                                                                        \
            if (!C16_ILV || TSK) __builtin_amdgcn_sched_barrier(0);                                                  \
            if (q & 1) C16_MFMA(o1, q >> 1) else C16_MFMA(o0, q >> 1)                                                \
            if (C16_ILV && !TSK) C16_INTERLEAVE                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        _Pragma("unroll") for (int t = 0; t < TPS; ++t) { tcur[t] = tnxt[t]; tnxt[t] = tnn[t]; }                     \
        if (++sidx == nst) { sidx = 0; ++ch; }                                                                       \
    }
    for (int v = 0; v < nv; v += 2) {
        C16_STAGE(0, wra)
        if (v + 1 < nv) C16_STAGE(1, wrb)
    }

    const int HWo = a.H * a.W;
    if (a.part) {   // split-K: raw partial sums; bias / residual / statistics are applied by the reduction pass
        float* pp = a.part + (long)blockIdx.y * a.part_stride;
#pragma unroll
        for (int wn = 0; wn < WN; ++wn) {
            const int n = n0 + wave_n * (32 * WN) + 32 * wn + l31;
            if (n >= a.Cout) continue;
#pragma unroll
            for (int wm = 0; wm < WM; ++wm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = rowpos[wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg];
                    if (p >= 0) pp[(long)p * a.Cout + n] = acc[wm][wn][r] * a.oscale;
                }
        }
        return;
    }
#pragma unroll
    for (int wn = 0; wn < WN; ++wn) {
        const int n = n0 + wave_n * (32 * WN) + 32 * wn + l31;
        const bool ncol = n < a.Cout;
        const float bias = (a.bias && ncol) ? a.bias[n] : 0.f;
        // per-lane partial statistics in fp64: the totals must not depend on how many rows a wave tile holds, or a shard
        // of the batch (which may pick a narrower tile) would not reproduce the full batch bit for bit
        double ssum = 0.0, ssq = 0.0;
        bool bad = false;
        // residual values first, all of them in flight together: inside the store loop each load would wait for the
        // previous store (`res` may alias `out` as far as the compiler knows; vmcnt counts stores too)
        float rv[WM][16];
#pragma unroll
        for (int wm = 0; wm < WM; ++wm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg;
                rv[wm][r] = (a.res && ncol && rowpos[m] >= 0) ? a.res[(long)rowres[m] * a.Cout + n] : 0.f;
            }
#pragma unroll
        for (int wm = 0; wm < WM; ++wm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wave_m * (32 * WM) + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int p = rowpos[m];
                if (p < 0 || !ncol) continue;
                float v = fmaf(acc[wm][wn][r], a.oscale, bias) + rv[wm][r];
                ssum += (double)v;
                ssq = fma((double)v, (double)v, ssq);
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                if (a.epi & EPI_FRAMES) {
                    const int bt_ = p / HWo, hw = p - bt_ * HWo;
                    a.out[((long)bt_ * a.Cout + n) * HWo + hw] = tanhf(v);
                } else if (a.epi & EPI_HL16) {  // the next conv's operand format instead of fp32 (same bytes per element)
                    const _Float16 hi = (_Float16)v;
                    bad |= !(fabsf(v) <= 65504.f);
                    char* o = reinterpret_cast<char*>(a.out) + (long)p * a.Cout * 4 + (n >> 3) * 32 + (n & 7) * 2;
                    *reinterpret_cast<_Float16*>(o) = hi;
                    *reinterpret_cast<_Float16*>(o + 16) = (_Float16)(v - (float)hi);
                } else {
                    a.out[(long)p * a.Cout + n] = v;
                }
            }
        }
        if (bad && a.range_flag) atomicOr(a.range_flag, 1);
        if (a.stats) {
            // fused normalisation statistics (InstanceNorm of conv_0's output / GroupNorm of the block output): the
            // workgroup tile lies inside one sample; lanes l and l^32 hold the same column -> wavefront shuffle, then one
            // fp64 atomic pair per (wave, column)
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (kg == 0 && ncol) {
                double* dst = a.stats + ((long)b0 * a.Cout + n) * 2;
                atomicAdd(dst, ssum);
                atomicAdd(dst + 1, ssq);
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
    std::vector<_Float16> p((size_t)(ntaps + 1) * nchunk * CoutPad * 64, (_Float16)0.f);  // + one all-zero tap
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < ntaps; ++tap) {
                const float v = (float)((double)w_src[((size_t)n * cin + c) * ntaps + tap] * scale * pre);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
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
    Conv16Weights tmp;
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
    const size_t set_halfs = (size_t)(ntaps + 1) * nchunk * CoutPad * 64;
    std::vector<_Float16> p(2 * set_halfs, (_Float16)0.f);
    for (int par = 0; par < 2; ++par)
        for (int n = 0; n < cout; ++n)
            for (int c = 0; c < cin; ++c)
                for (int tap = 0; tap < ntaps; ++tap) {
                    const float v = (float)((double)w2[(((size_t)par * cout + n) * cin + c) * 18 + tap] * scale * pre);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const int chunk = c / C16_KC, g = (c % C16_KC) / 8, j = c % 8;
                    _Float16* row = &p[par * set_halfs + (((size_t)tap * nchunk + chunk) * CoutPad + n) * 64];
                    row[g * 16 + j] = hi;
                    row[g * 16 + 8 + j] = lo;
                }
    set_bytes = (long)set_halfs * 2;
    int rc = w.upload(p.data(), p.size() * 2);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

// measurement builds (-DC16_TUNE) can switch the clipped staging off (I2V_C16_CLIP=0) for same-box A/Bs; the production build cannot
static bool c16_switch(const char* name) {
#ifdef C16_TUNE
    const char* e = getenv(name);
    return !(e && e[0] == '0');
#else
    (void)name;
    return true;
#endif
}
static bool c16_clip_enabled() { return c16_switch("I2V_C16_CLIP"); }

template <int WAVES_M, int WAVES_N, int WM, int WN, int TPS, bool TSK = false>
static int launch16(const Conv16Args& a, unsigned nblk, size_t lds, hipStream_t st) {
    auto kern = conv_mfma_f16x3_kernel<WAVES_M, WAVES_N, WM, WN, TPS, TSK>;
    static bool attr_set[I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.tdup ? 2 * nblk : nblk, a.ksplit), dim3(64 * WAVES_M * WAVES_N), lds, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

// out[p][c] = act(bias[c] + sum_s part[s][p][c] (+ res[rowres(p)][c])) -- the reduction pass of a split-K launch.
// Fixed summation order (s = 0, 1, ...): deterministic, and independent of the batch (shards stay bit-identical).
__global__ void conv16_splitk_reduce_kernel(const float* __restrict__ part, long part_stride, int ksplit,
                                            const float* __restrict__ bias, const float* __restrict__ res, float* __restrict__ out,
                                            long npos, int Cout, int T, int H, int W, int rt, int rs, int lrelu) {
    const int C4 = Cout >> 2;
    const long total = npos * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        float4 s = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ksplit; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(part + (long)k * part_stride + p * Cout + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (res) {
            long q = p;
            const int w = (int)(q % W); q /= W;
            const int h = (int)(q % H); q /= H;
            const int t = (int)(q % T); q /= T;
            const long rp = ((q * (T / rt) + t / rt) * (H / rs) + h / rs) * (W / rs) + w / rs;
            const float4 r = *reinterpret_cast<const float4*>(res + rp * Cout + c);
            s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        }
        if (lrelu) {
            s.x = s.x >= 0.f ? s.x : 0.2f * s.x; s.y = s.y >= 0.f ? s.y : 0.2f * s.y;
            s.z = s.z >= 0.f ? s.z : 0.2f * s.z; s.w = s.w >= 0.f ? s.w : 0.2f * s.w;
        }
        *reinterpret_cast<float4*>(out + p * Cout + c) = s;
    }
}

// The split factor depends on the LAYER geometry only (positions per sample, K chunks), never on the batch: the summation
// order of a sample's outputs must not change when the batch is sharded (shards are bit-identical to the full batch).
int conv16_splitk_factor(long pos_per_sample, int nchunk) {
    int s = pos_per_sample <= 16 ? 8 : pos_per_sample <= 128 ? 4 : 1;
    while (s > 1 && nchunk / s < 2) s /= 2;
    return s;
}

bool conv16_can_fuse_stats(int T, int H, int W) {
    // the 256-position brick stays inside one sample when the sample has at least 256 positions (power-of-two dims)
    return (long)T * H * W >= C16_BM;
}

int conv16_forward(const Conv16Weights& wts, const void* in_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                   int H, int W, int epi, hipStream_t st, double* stats, int* range_flag, float* splitk_ws, size_t splitk_ws_floats) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "conv16: weights not packed");
    I2V_REQUIRE(wts.Cin % 8 == 0, I2V_E_INVALID, "conv16: Cin %d must be a multiple of 8", wts.Cin);
    Conv16Args a{};
    if (int rc0 = zero_page(&a.zeros)) return rc0;
    a.in = static_cast<const char*>(in_hl16); a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.B = B; a.T = T; a.H = H; a.W = W; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.KT = wts.KT; a.KH = wts.KH; a.KW = wts.KW; a.tap_base = 0;
    a.ztap = wts.KT * wts.KH * wts.KW;
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
    a.range_flag = range_flag;
    a.ksplit = 1;
    a.oscale = (float)std::ldexp(1.0, -wts.wexp);
    int TW = W < 8 ? W : 8, TH = H < 8 ? H : 8;
    int rem = C16_BM / (TW * TH);
    int TT = T < rem ? T : rem;
    rem /= TT;
    while (rem > 1 && W >= TW * 2) { TW *= 2; rem /= 2; }
    while (rem > 1 && H >= TH * 2) { TH *= 2; rem /= 2; }
    const int TB_full = rem;
    I2V_REQUIRE(TB_full * TT * TH * TW == C16_BM && T % TT == 0 && H % TH == 0 && W % TW == 0, I2V_E_INVALID,
                "conv16: cannot tile [T=%d,H=%d,W=%d] into bricks of %d positions", T, H, W, C16_BM);
    a.TT = TT; a.TH = TH; a.TW = TW;
    a.nbT = T / TT; a.nbH = H / TH; a.nbW = W / TW;
    a.patch = (TW % 4 == 0 && TH % 4 == 0) ? 1 : 0;
    const bool can_split = splitk_ws && !stats && !(epi & (EPI_FRAMES | EPI_HL16)) && a.Cout % 4 == 0;
    const int ksplit_plan = can_split ? conv16_splitk_factor((long)(a.tdup ? 2 * T : T) * H * W, a.nchunk) : 1;
    // temporal tap skipping per row block (see the kernel): a 3-tap temporal kernel on a two-frame map whose 32-row MFMA blocks lie
    // inside one frame (g_0.conv_1: 2 x 8 x 8)
    const bool tsk_geo = !a.tdup && a.KT == 3 && a.T == 2 && TT == 2 && (TH * TW) % 32 == 0;
    // Brick plan for a given staging form.  clip (round 6): a brick that spans the map's whole time extent stages only its real frames --
    // the temporal halo of such a brick is zero padding, which a one-frame brick's tap list and the TSK masks skip anyway.  Without the
    // halo frames (half of g_0.conv_1's staged rows, half of g_0.conv_0's) the brick fits LDS with the conflict-free row pitch.
    // Nothing an output sums, and no order, changes: same bits.
    int TB = 1, BN = 32, npos = 0;
    auto plan = [&](bool clip) {
        const int ht = clip ? TT : TT + a.KT - 1;
        auto lds_of = [&](int tb, int hp) {
            const size_t rows = (size_t)tb * ht * (TH + a.KH - 1) * hp;
            return rows * C16_ROW + 2 * (size_t)128 * C16_ROW + (2 * C16_BM + 72) * 4 + rows * 4;
        };
        // tiny feature maps x many samples: the halo tile of a full brick may not fit LDS -- take fewer samples per brick
        // (the tile's unused rows are masked)
        TB = TB_full;
        while (TB > 1 && lds_of(TB, TW + a.KW - 1) > 160 * 1024) TB /= 2;
        a.HWp = TW + a.KW - 1;
        if (a.patch) {  // a halo row pitch = 4 or 12 (mod 16) makes the 4x4 patches conflict-free; keep it if LDS allows
            int hp = a.HWp;
            while (hp % 16 != 4 && hp % 16 != 12) ++hp;
            if (lds_of(TB, hp) <= 160 * 1024) a.HWp = hp;
        }
        a.TB = TB;
        a.nbB = (B + TB - 1) / TB;
        npos = TB * ht * (TH + a.KH - 1) * a.HWp;
        // channel tile: the widest that divides CoutPad, narrowed while the launch would leave most CUs without a workgroup
        // (head_0: 64 samples x 4 x 4 positions = 4 bricks x 8 tiles of 128 channels, each streaming 9 x 1024 channels of K)
        // split-K (decided by the layer geometry alone, see below): its slices are workgroups too -- round 5 narrowed the channel tile
        // without counting them, so head_0 at B = 64 (4 bricks x 8 slices) ran 32-channel tiles: 1024 workgroups that each re-read the
        // whole activation brick for 32 output channels (0.38 / 0.27 ms for 0.04 ms of matrix work).  The tile width changes the schedule
        // only, never an output's accumulation order: same bits.
        BN = a.CoutPad % 128 == 0 ? 128 : (a.CoutPad % 64 == 0 ? 64 : 32);
        const long bricks = (long)a.nbB * a.nbT * a.nbH * a.nbW * (a.tdup ? 2 : 1) * ksplit_plan;
        while (BN > 32 && bricks * (a.CoutPad / BN) < 256) BN /= 2;
    };
    a.tskperm = c16_switch("I2V_C16_PERM") ? 1 : 0;
    a.tclip = 0;
    if (a.nbT == 1 && a.KT > 1 && (TT == 1 || tsk_geo) && c16_clip_enabled()) {
        plan(true);
        a.tclip = (TT == 1 || BN != 64) ? 1 : 0;   // (the 64-channel tile has no TSK instantiation: it needs the staged padding frames)
    }
    if (!a.tclip) plan(false);
    I2V_REQUIRE(!stats || TB == 1, I2V_E_INVALID, "conv16: fused statistics need bricks inside one sample");
    const size_t lds = (size_t)npos * C16_ROW + 2 * (size_t)128 * C16_ROW + (2 * C16_BM + 72) * 4 + (size_t)npos * 4;
    I2V_REQUIRE(lds <= 160 * 1024, I2V_E_INVALID, "conv16: LDS %zu bytes exceeds 160 KiB", lds);
    const long nblk = (long)a.nbB * a.nbT * a.nbH * a.nbW * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 30), I2V_E_INVALID, "conv16: grid of %ld workgroups", nblk);
    // split-K: a launch that cannot fill the chip even with the narrowest channel tile (head_0 / g_0 at small batches: a
    // handful of bricks, K = 9 x 1024) divides its K chunks over several workgroups per tile; the partial sums go to the
    // caller's scratch and a reduction pass (fixed order) applies bias / residual / activation.
    const long npos_out = (long)B * (a.tdup ? 2 * T : T) * H * W;
    int ksplit = 1;
    if (can_split) {
        ksplit = ksplit_plan;
        I2V_REQUIRE((size_t)ksplit * npos_out * a.Cout <= splitk_ws_floats || ksplit == 1, I2V_E_WORKSPACE,
                    "conv16: split-K scratch of %zu floats is too small for %d x %ld x %d", splitk_ws_floats, ksplit, npos_out, a.Cout);
    }
    if (ksplit > 1) {
        a.ksplit = ksplit;
        a.part = splitk_ws;
        a.part_stride = npos_out * a.Cout;
    }
    // (a 16-wave variant <8,2,1,2,1> -- 4 waves per SIMD, wave tile 32x64, 128 VGPRs -- was measured 5 % slower)
    int rc;
    const bool tsk = tsk_geo;
    if (BN == 128) rc = tsk ? launch16<4, 2, 2, 2, 1, true>(a, (unsigned)nblk, lds, st) : launch16<4, 2, 2, 2, 1>(a, (unsigned)nblk, lds, st);
    else if (BN == 64) rc = launch16<4, 2, 2, 1, 2>(a, (unsigned)nblk, lds, st);   // (its TSK instantiation spills: 256 VGPRs + scratch; the 64-wide tile keeps all 27 taps)
    else rc = tsk ? launch16<8, 1, 1, 1, 4, true>(a, (unsigned)nblk, lds, st) : launch16<8, 1, 1, 1, 4>(a, (unsigned)nblk, lds, st);
    if (rc || ksplit == 1) return rc;
    const long total4 = npos_out * (a.Cout / 4);
    hipLaunchKernelGGL(conv16_splitk_reduce_kernel, dim3((unsigned)std::min<long>((total4 + 255) / 256, 4096)), dim3(256), 0, st,
                       splitk_ws, a.part_stride, ksplit, a.bias, res, out, npos_out, a.Cout, a.tdup ? 2 * T : T, H, W, a.rt, a.rs,
                       (epi & EPI_LRELU) ? 1 : 0);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace i2v

// cINN coupling chain on the fp32 matrix cores of gfx950 (reference: stage2_cINN/modules/flow_blocks.py:63-139,
// stage2_cINN/modules/modules.py:9-30).
//
// One pass = 40 coupling half-steps, each an s-net and a t-net of four Linear layers (modules.py:14-24, called at
// flow_blocks.py:90-91,103).  Every Linear is evaluated as 16 x 16 tiles of v_mfma_f32_16x16x4_f32 (exact fp32): 16 output
// rows x 16 samples per tile, tiles stored as the accumulator fragment (see i2v_flow_tile.h) so that one layer's output
// IS the next layer's B operand.  Launches per pass: 1 (embedding part of all 80 first layers) + 41 + 80:
//
//   flow_pre_tile_kernel   pre[step][st][rt] = b0 + W0[:, 32:] . embed      (the dense conditioning GEMM, off the chain)
//   flow_tail_tile_kernel  per sample tile: sum the last layer's 2 x 32 partial tiles -> (s, t) -> affine coupling
//                          x*exp(s)+t / (x-t)*exp(-s) (flow_blocks.py:91,103) and log-det (:93) -> Shuffle / ActNorm /
//                          InvLeakyRelu / half swap of the block boundary -> first Linear of the NEXT half-step (K = 32)
//   flow_hid_tile_kernel   hidden Linear + LeakyReLU(0.01), s- and t-net in one launch: workgroup = 16 rows x NS sample
//                          tiles, K split over the 8 waves (LDS reduce in fixed order); the LAST hidden layer never
//                          writes its output: it multiplies its 16 x 16 tile straight into the final Linear (H -> 32)
//                          and writes the 32 x 16 partial product instead.
//
// The summation order of every output element is fixed by the layer geometry alone, so results do not depend on the
// batch size, the sample-tile grouping NS or the position of a sample in the batch (shards == full batch, bit for bit).
//
// Measured and not kept (profiles/r03_a_*, tools/experiments/README.md): 64 extra workgroups per launch that touch the next
// launch's weight tiles of their XCD ("L2 warming") made the pass 43 us SLOWER at B = 64; with every layer reading the same
// (L2-resident) weights the pass is only 23 us faster, i.e. the weight fetch is not what a launch waits for.
#include "i2v_flow_tile.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace i2v {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#ifdef FLOW_TIMELINE   // measurement build (tools/flow_timeline.py): wall-clock stamps (100 MHz) per launch and workgroup phase
constexpr int FTL_LAUNCHES = 512, FTL_WGS = 64, FTL_ST = 8;
__device__ unsigned long long flow_tl[FTL_LAUNCHES * FTL_WGS * FTL_ST];
__device__ unsigned long long flow_tl_span[FTL_LAUNCHES * 2];   // [launch]{earliest start, latest end} over ALL workgroups
#define FTL_STAMP(seq, i) { if (threadIdx.x == 0 && (seq) < FTL_LAUNCHES && blockIdx.x < FTL_WGS) flow_tl[((seq) * FTL_WGS + blockIdx.x) * FTL_ST + (i)] = wall_clock64(); }
#define FTL_BEGIN(seq) { if (threadIdx.x == 0 && (seq) < FTL_LAUNCHES) atomicMin(&flow_tl_span[(seq) * 2], wall_clock64()); FTL_STAMP(seq, 0) }
#define FTL_END(seq) { if (threadIdx.x == 0 && (seq) < FTL_LAUNCHES) atomicMax(&flow_tl_span[(seq) * 2 + 1], wall_clock64()); }
#define FTL_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // "operands landed" needs an explicit wait to be timed
#else
#define FTL_STAMP(seq, i) {}
#define FTL_BEGIN(seq) {}
#define FTL_END(seq) {}
#define FTL_LANDED() {}
#endif

namespace {

__device__ __forceinline__ v4f ld4(const float* p) { return *reinterpret_cast<const v4f*>(p); }
__device__ __forceinline__ void st4(float* p, v4f v) { *reinterpret_cast<v4f*>(p) = v; }
// One 16 x 16 x 16 block D += A . B with A = a weight fragment (fp32: 16 bytes per lane, four v_mfma_f32_16x16x4_f32 steps; F16: 8 bytes
// per lane, ONE v_mfma_f32_16x16x16_f16 -- lane (q, m) holds A[m][4q .. 4q+3] in both) and B = an activation tile in the
// accumulator-fragment layout (fp32; F16: rounded to fp16 here, round-to-nearest-even).
template <bool F16> struct WFrag { typedef v4f T; };
template <> struct WFrag<true> { typedef h4 T; };
template <bool F16>
__device__ __forceinline__ typename WFrag<F16>::T ldw(const void* base, size_t frag, int lane) {
    if constexpr (F16) return *reinterpret_cast<const h4*>(static_cast<const char*>(base) + frag * 512 + lane * 8);
    else return *reinterpret_cast<const v4f*>(static_cast<const char*>(base) + frag * 1024 + lane * 16);
}
template <bool F16>
__device__ __forceinline__ v4f mma16(typename WFrag<F16>::T a, v4f b, v4f d) {
    if constexpr (F16) {
        const h4 bh = {(_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, bh, d, 0, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], d, 0, 0, 0);
        return d;
    }
}

__device__ __forceinline__ v4f lrelu4(v4f v, float slope) {
    v4f o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = v[r] >= 0.f ? v[r] : v[r] * slope;
    return o;
}

// ---------------------------------------------------------------------------------------------------------------------
// embedding part of ALL first layers (modules.py:14 with dim = 32 + E, columns 32..): pre = b0 + W0e . embed
struct PreTileArgs {
    const float* W0E;  // [Rtiles][KE16][256]
    const float* b0;   // [Rtiles * 16]
    const FlowIo* io;
    float* pre;        // [S][NST][NRT][256]
    int NRT, NST, KE16, E, B, Rtiles, nblk;
    int seq;           // launch number inside the pass (timeline builds)
};

constexpr int PRE_SC = 4;        // sample tiles per workgroup: the weight fragments stay in registers across them
constexpr int PRE_LD = 128 + 4;  // LDS row stride (floats): 16-byte aligned rows, 16 rows hit 16 distinct bank quads

// workgroup = 8 row tiles (one per wave) x PRE_SC sample tiles; the embeddings of the 64 samples are staged in LDS with
// coalesced loads (rows of E floats) and read back as B fragments (lane (q, n), j -> embed[n][16 i + 4 q + j])
template <bool F16>
__global__ __launch_bounds__(512) void flow_pre_tile_kernel(PreTileArgs a) {
    __shared__ __attribute__((aligned(16))) float es[PRE_SC * 16][PRE_LD];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    FTL_BEGIN(a.seq)
    const int sc = blockIdx.x / a.nblk, blk = blockIdx.x - sc * a.nblk;
    const int R = blk * 8 + w;
    const bool rok = R < a.Rtiles;
    const int Rc = rok ? R : 0;
    const int step = Rc / a.NRT, rt = Rc - step * a.NRT;
    typename WFrag<F16>::T A[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = ldw<F16>(a.W0E, (size_t)Rc * a.KE16 + (i < a.KE16 ? i : 0), lane);
    const v4f bias = ld4(a.b0 + (size_t)Rc * 16 + q * 4);
    const float* emb = a.io->embed;
    const int b0 = sc * PRE_SC * 16, K = a.KE16 * 16;
    for (int i = threadIdx.x; i < PRE_SC * 16 * K; i += 512) {   // K = E rounded up to 16: the pad columns are zeros
        const int r = i / K, k = i - r * K;
        es[r][k] = (b0 + r < a.B && k < a.E) ? emb[(size_t)(b0 + r) * a.E + k] : 0.f;
    }
    __syncthreads();
    if (!rok) return;
#pragma unroll
    for (int t = 0; t < PRE_SC; ++t) {
        const int st = sc * PRE_SC + t;
        if (st >= a.NST) break;
        v4f D = bias;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < a.KE16) {
                D = mma16<F16>(A[i], ld4(&es[t * 16 + n][16 * i + 4 * q]), D);
            }
        st4(a.pre + (((size_t)step * a.NST + st) * a.NRT + rt) * 256 + lane * 4, D);
    }
    FTL_END(a.seq)
}

// ---------------------------------------------------------------------------------------------------------------------
// hidden Linear(H, H) + LeakyReLU(0.01) (modules.py:19-22), optionally fused with the final Linear(H, 32) (:24)
struct HidTileArgs {
    const float* WT;    // this layer [NRT][HB][256]
    const float* bias;  // [2H]
    const float* in;    // [NST][NRT][256]
    float* out;         // [NST][NRT][256] or null (last hidden layer)
    const float* W3P;   // [NRT][2][256] or null
    float* P;           // [NST][NRT][2][256] partial products of the final Linear
    int NRT, NST;
    int seq;
};

// (flat parameters, 15 dwords: with -mllvm -amdgpu-kernarg-preload-count=16 they arrive in SGPRs with the wave instead of
//  through a scalar load at the head of every one of the pass's 80 dependent hidden-layer launches)
template <int KPW, int NS, bool F16>
__global__ __launch_bounds__(512) void flow_hid_tile_kernel(const float* WT_, const float* in_, const float* bias_, float* out_,
                                                            const float* W3P_, float* P_, int NRT_, int NST_, int seq_) {
    const HidTileArgs a{WT_, bias_, in_, out_, W3P_, P_, NRT_, NST_, seq_};
    __shared__ v4f red[8][NS][64];
    constexpr int HB = 8 * KPW;
    FTL_BEGIN(a.seq)
    // Workgroup b runs on XCD b % 8 (speed only).  XCDs 0-3 take the s-net's row tiles, 4-7 the t-net's: an XCD's L2 then
    // fetches the activations of ONE net (its workgroups read nothing else) and every weight tile exactly once.
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int net = xcd >> 2;
    const int sg = slot / (HB / 4), rt = net * HB + (xcd & 3) + 4 * (slot - sg * (HB / 4)), st0 = sg * NS;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // operand requests: this wave's K slice (k16 blocks w*KPW ...) of the weight tile row and of NS activation tiles
    typename WFrag<F16>::T A[KPW];
    v4f Bv[NS][KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) A[i] = ldw<F16>(a.WT, (size_t)rt * HB + w * KPW + i, lane);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int st = st0 + s < a.NST ? st0 + s : a.NST - 1;
        const float* bp = a.in + ((size_t)st * a.NRT + net * HB + w * KPW) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < KPW; ++i) Bv[s][i] = ld4(bp + i * 256);
    }
    // epilogue operands of the waves that finish a tile: (sample tile es, column block cb of the final Linear)
    const int es = w >> 1, cb = w & 1;
    const bool epi = w < 2 * NS;
    v4f bias4 = {0.f, 0.f, 0.f, 0.f};
    typename WFrag<F16>::T A3 = {};
    if (epi) {
        bias4 = ld4(a.bias + rt * 16 + (lane >> 4) * 4);
        if (a.W3P) A3 = ldw<F16>(a.W3P, (size_t)rt * 2 + cb, lane);
    }
    v4f D[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) D[s] = v4f{0.f, 0.f, 0.f, 0.f};
    FTL_STAMP(a.seq, 1)   // requests issued
    FTL_LANDED()
    FTL_STAMP(a.seq, 2)   // operands landed
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) D[s] = mma16<true>(A[i], Bv[s][i], D[s]);
    } else {
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) D[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i][j], Bv[s][i][j], D[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) red[w][s][lane] = D[s];
    FTL_STAMP(a.seq, 3)   // MFMAs done, partial tiles parked
    __syncthreads();
    FTL_STAMP(a.seq, 4)   // barrier passed
    if (!epi) return;
    const int st = st0 + es;
    if (st >= a.NST) return;
    v4f h = bias4;
#pragma unroll
    for (int v = 0; v < 8; ++v) h += red[v][es][lane];
    h = lrelu4(h, 0.01f);
    if (a.out && cb == 0) st4(a.out + ((size_t)st * a.NRT + rt) * 256 + lane * 4, h);
    if (a.W3P) {
        // this tile's share of the final Linear: the accumulator fragment h IS the B operand (k = 16 rt + 4 q + j)
        const v4f d3 = mma16<F16>(A3, h, v4f{0.f, 0.f, 0.f, 0.f});
        st4(a.P + (((size_t)st * a.NRT + rt) * 2 + cb) * 256 + lane * 4, d3);
    }
    FTL_STAMP(a.seq, 5)   // reduce + epilogue MFMA + stores issued
    FTL_END(a.seq)
}

// ---------------------------------------------------------------------------------------------------------------------
struct TailTileArgs {
    const float* P;     // [NST][NRT][2][256] or null (launch in front of the first half-step: no coupling)
    const float* b3;    // [64]: s bias 0..31, t bias 32..63
    const float* x;     // workspace state [NST * 16][64] this launch reads ...
    float* xo;          // ... and the one it writes: NOT the same buffer -- the eight row-group workgroups of a sample tile all
                        // read the old state, one of them writes the new one, and nothing orders them inside a launch
    float* logdet;      // workspace [NST * 16] or null (reverse pass)
    const FlowIo* io;
    int io_in, io_out;  // read the caller's x instead of the workspace state / also write the caller's outputs
    int ld_init;        // log-det: start from 0 instead of accumulating
    int B, NST, NRT, HB2, reverse;
    // elementwise ops between the coupling and the next half-step
    const int* shuf;        // [64] gather indices or null
    const float* an_loc;    // [64] or null
    const float* an_scale;
    float an_logdet;
    int do_lrelu, do_swap;
    // first layer of the NEXT half-step: 0 none, 1 K = 32 state channels, 2 K = 0 (mode 'cond', flow_blocks.py:89,102)
    int l1;
    const float* W0T;   // [NRT][2][256]
    const float* pre;   // [NST][NRT][256]
    float* h0;          // [NST][NRT][256]
    int seq;
};

// (flat parameters ordered by first use: the 16 dwords that the kernel's first vector loads need -- partial tiles, state, first-layer
//  weights and biases, geometry, flags, the caller's pointer block -- are preloaded into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16);
//  what the later phases need arrives by scalar load underneath those vector loads)
template <bool F16>
__global__ __launch_bounds__(512) void flow_tail_tile_kernel(const float* P_, const float* x_, const float* W0T_, const float* pre_,
                                                             const float* b3_, int NST_, int NRT_, int HB2_, int flags_, const FlowIo* io_,
                                                             float* xo_, float* logdet_, const int* shuf_, const float* an_loc_,
                                                             const float* an_scale_, float an_logdet_, float* h0_, int B_, int seq_) {
    TailTileArgs a{};
    a.P = P_; a.x = x_; a.W0T = W0T_; a.pre = pre_; a.b3 = b3_; a.NST = NST_; a.NRT = NRT_; a.HB2 = HB2_; a.io = io_;
    a.io_in = flags_ & 1; a.io_out = (flags_ >> 1) & 1; a.ld_init = (flags_ >> 2) & 1; a.reverse = (flags_ >> 3) & 1;
    a.do_lrelu = (flags_ >> 4) & 1; a.do_swap = (flags_ >> 5) & 1; a.l1 = (flags_ >> 6) & 3;
    a.xo = xo_; a.logdet = logdet_; a.shuf = shuf_; a.an_loc = an_loc_; a.an_scale = an_scale_; a.an_logdet = an_logdet_; a.h0 = h0_;
    a.B = B_; a.seq = seq_;
    __shared__ v4f ps[8][64];
    __shared__ __attribute__((aligned(16))) float xs[16][68];
    __shared__ __attribute__((aligned(16))) float xs2[16][68];
    __shared__ float ld[2][16];
    __shared__ float anl[64], ans[64];
    __shared__ int sidx[64];
    const int id = blockIdx.x;
    // all row groups of a sample tile on one XCD (they sum the same 2 x 32 partial tiles)
    const int st = (id >> 6) * 8 + (id & 7), rq = (id >> 3) & 7;
    if (st >= a.NST) return;
    FTL_BEGIN(a.seq)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int HB = 2 * a.HB2;
    // ---- requests: partial tiles of the final Linear: wave = (net, column block, half of the 2 HB2 row tiles)
    v4f pp[16];
    if (a.P) {
        const int net = w >> 2, cb = (w >> 1) & 1, half = w & 1;
        const float* base = a.P + (((size_t)st * a.NRT + net * HB + half * a.HB2) * 2 + cb) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < a.HB2) pp[i] = ld4(base + (size_t)i * 512);
    }
    const int rpg = a.NRT >> 3;  // row tiles of the next first layer per row group
    const bool l0 = a.l1 != 0 && w < rpg;
    const int rt = rq * rpg + w;
    typename WFrag<F16>::T A0[2] = {};
    v4f D = {0.f, 0.f, 0.f, 0.f};
    if (l0) {
        if (a.l1 == 1) {
            A0[0] = ldw<F16>(a.W0T, (size_t)rt * 2, lane);
            A0[1] = ldw<F16>(a.W0T, (size_t)rt * 2 + 1, lane);
        }
        D = ld4(a.pre + ((size_t)st * a.NRT + rt) * 256 + lane * 4);
    }
    v4f bs = {0.f, 0.f, 0.f, 0.f}, bt = bs;
    if (a.P && w < 2) { bs = ld4(a.b3 + 16 * w + 4 * q); bt = ld4(a.b3 + 32 + 16 * w + 4 * q); }
    if (tid < 256) {  // old state of the 16 samples -> LDS
        const int nn = tid >> 4, c4 = tid & 15, b = st * 16 + nn;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (a.io_in) { if (b < a.B) v = ld4(a.io->xin + (size_t)b * 64 + 4 * c4); }
        else v = ld4(a.x + (size_t)b * 64 + 4 * c4);
        st4(&xs[nn][4 * c4], v);
    } else if (tid < 320) {
        const int c = tid - 256;
        anl[c] = a.an_loc ? a.an_loc[c] : 0.f;
        ans[c] = a.an_loc ? a.an_scale[c] : 1.f;
    } else if (tid < 384) {
        const int c = tid - 320;
        sidx[c] = a.shuf ? a.shuf[c] : c;
    }
    FTL_STAMP(a.seq, 1)   // requests issued
    FTL_LANDED()
    FTL_STAMP(a.seq, 2)   // partial tiles, state, weights landed
    if (a.P) {
        v4f s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < a.HB2) s += pp[i];
        ps[w][lane] = s;
    }
    __syncthreads();
    FTL_STAMP(a.seq, 3)   // partial sums parked, barrier
    // ---- affine coupling of the transformed half x[32..63] (flow_blocks.py:91-93 / 103): waves 0, 1 = channel blocks
    if (a.P && w < 2) {
        const int cb = w;
        const v4f s = (bs + ps[cb * 2][lane]) + ps[cb * 2 + 1][lane];
        const v4f t = (bt + ps[4 + cb * 2][lane]) + ps[4 + cb * 2 + 1][lane];
        v4f xv = ld4(&xs[n][32 + 16 * cb + 4 * q]);
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = a.reverse ? (xv[r] - t[r]) * expf(-s[r]) : fmaf(xv[r], expf(s[r]), t[r]);
        st4(&xs[n][32 + 16 * cb + 4 * q], xv);
        if (a.logdet && !a.reverse) {  // log-det of the coupling: sum of s over the 32 channels (wavefront shuffles)
            float l = (s[0] + s[1]) + (s[2] + s[3]);
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            if (q == 0) ld[cb][n] = l;
        }
    }
    __syncthreads();
    // ---- block boundary: Shuffle gather (:152-154), ActNorm (modules.py:80/100), InvLeakyRelu (:180-187), half swap (:86,99)
    {
        const int nn = tid >> 5, e2 = (tid & 31) * 2, b = st * 16 + nn;
        float o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cp = e2 + e;
            const int c1 = a.do_swap ? cp ^ 32 : cp;
            const int c0 = sidx[c1];
            float v = xs[nn][c0];
            if (!a.reverse) {
                if (a.an_loc) v = ans[c1] * (v + anl[c1]);
                if (a.do_lrelu) v = v * (v >= 0.f ? 1.0f : 0.9f);
            } else {
                if (a.do_lrelu) v = v / (v >= 0.f ? 1.0f : 0.9f);
                if (a.an_loc) v = v / ans[c0] - anl[c0];
            }
            o[e] = v;
        }
        *reinterpret_cast<float2*>(&xs2[nn][e2]) = make_float2(o[0], o[1]);
        if (rq == 0) {
            *reinterpret_cast<float2*>(a.xo + (size_t)b * 64 + e2) = make_float2(o[0], o[1]);
            if (a.io_out && b < a.B) *reinterpret_cast<float2*>(a.io->xout + (size_t)b * 64 + e2) = make_float2(o[0], o[1]);
        }
        if (rq == 0 && a.logdet && tid < 16) {
            const int bb = st * 16 + tid;
            float v = a.ld_init ? 0.f : a.logdet[bb];
            if (a.P) v += ld[0][tid] + ld[1][tid];
            if (a.an_loc) v += a.an_logdet;  // ActNorm.forward: sum log|scale| (modules.py:86-87, H = W = 1)
            a.logdet[bb] = v;
            if (a.io_out && bb < a.B && a.io->logdet_out) a.io->logdet_out[bb] = v;
        }
    }
    FTL_STAMP(a.seq, 4)   // coupling + block boundary done
    if (!a.l1) { FTL_END(a.seq) return; }  // (uniform)
    __syncthreads();
    // ---- first Linear of the next half-step (modules.py:14-17, slope 0.01): K = the 32 passive state channels
    if (l0) {
        if (a.l1 == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) D = mma16<F16>(A0[i], ld4(&xs2[n][16 * i + 4 * q]), D);
        }
        st4(a.h0 + ((size_t)st * a.NRT + rt) * 256 + lane * 4, lrelu4(D, 0.01f));
    }
    FTL_STAMP(a.seq, 5)   // first Linear of the next half-step stored
    FTL_END(a.seq)
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the tail FOLDED into the first hidden layer's launch (2 instead of 3 dependent launches per half-step: 1 + 2 S + 1 = 82
// launches per pass instead of 122).  Every workgroup of the first hidden Linear(H, H) of half-step k+1 needs ALL of h0 = lrelu(pre +
// W0x . x_passive) of its net for its sample tiles, and x_passive needs the coupling of half-step k.  Instead of waiting for a launch
// that computes them once (1.5 us of kernel boundary + 1-2 us from kernel entry to the first requests + the launch's own span, 41
// times per pass), every workgroup recomputes them for its own sample tiles: sum the 2 x 32 partial tiles of the final Linear (128 KB
// per sample tile, L2 hits after the first workgroup of the XCD), coupling, log-det, Shuffle / ActNorm / InvLeakyRelu / half swap in
// LDS, then each wave evaluates the K = 32 first Linear for exactly the KPW row tiles of h0 that are ITS K slice of the hidden
// layer -- they never leave the registers.  The hidden layer's own weight fragments are requested at kernel entry and travel
// underneath all of that.  One workgroup per sample-tile group (row tile 0) stores the new state and the log-det.
// Same operations in the same order as flow_tail_tile_kernel + flow_hid_tile_kernel: the bits do not change (tested).
template <int KPW, int NS, bool F16>
__global__ __launch_bounds__(512) void flow_first_tile_kernel(const float* WT_, const float* P_, const float* x_, const float* W0T_,
                                                              const float* pre_, const float* bias_, int NRT_, int NST_, int flags_,
                                                              const float* b3_, const FlowIo* io_, float* xo_, float* logdet_,
                                                              const int* shuf_, const float* an_loc_, const float* an_scale_,
                                                              float an_logdet_, float* out_, const float* W3P_, float* Pout_, int B_,
                                                              int seq_) {
    constexpr int HB = 8 * KPW;       // row tiles per net
    constexpr int HB2 = HB / 2;
    const int NRT = NRT_, NST = NST_;
    const bool io_in = flags_ & 1, ld_init = (flags_ >> 2) & 1, reverse = (flags_ >> 3) & 1, do_lrelu = (flags_ >> 4) & 1,
               do_swap = (flags_ >> 5) & 1;
    const int l1 = (flags_ >> 6) & 3;
    __shared__ v4f red[8][NS][64];
    __shared__ v4f ps[8][64];
    __shared__ __attribute__((aligned(16))) float xs[16][68];
    __shared__ __attribute__((aligned(16))) float xs2[16][68];
    __shared__ float ld[2][16];
    __shared__ float anl[64], ans[64];
    __shared__ int sidx[64];
    FTL_BEGIN(seq_)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int net = xcd >> 2;
    const int sg = slot / (HB / 4), rt = net * HB + (xcd & 3) + 4 * (slot - sg * (HB / 4)), st0 = sg * NS;
    const bool writer = rt == 0;      // one workgroup per sample-tile group stores the new state and the log-det
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n = lane & 15;
    // Request order = the order in which the chain of this launch needs the operands (vector loads return in order, so a wait for an
    // early request leaves the later ones in flight): block-boundary tables and the final Linear's biases, then per sample tile the
    // partial tiles and the old state (coupling), the embedding part of the first Linear; behind the first sample tile's requests
    // the fragments of the first Linear for the KPW row tiles of h0 that ARE this wave's K slice, the hidden layer's weight row
    // and the epilogue operands -- those travel underneath the whole tail.
    v4f bs = {0.f, 0.f, 0.f, 0.f}, bt = bs;
    if (P_ && w < 2) { bs = ld4(b3_ + 16 * w + 4 * q); bt = ld4(b3_ + 32 + 16 * w + 4 * q); }
    float tab0 = 0.f, tab1 = 1.f;
    int tabi = 0;
    if (tid >= 256 && tid < 320) {
        if (an_loc_) { tab0 = an_loc_[tid - 256]; tab1 = an_scale_[tid - 256]; }
    } else if (tid >= 320 && tid < 384) {
        tabi = shuf_ ? shuf_[tid - 320] : tid - 320;
    }
    typename WFrag<F16>::T A[KPW];
    typename WFrag<F16>::T A0[KPW][2] = {};
    const int es = w >> 1, cb = w & 1;
    const bool epi = w < 2 * NS;
    v4f bias4 = {0.f, 0.f, 0.f, 0.f};
    typename WFrag<F16>::T A3 = {};
    v4f Bv[NS][KPW];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool live = st0 + s < NST;
        const int st = live ? st0 + s : NST - 1;
        // ---- requests of this sample tile: partial tiles of the final Linear (wave = (net, column block, half of the row tiles)),
        // the old state, the embedding part of the first Linear for this wave's KPW row tiles
        v4f pp[HB2];
        if (P_) {
            const int pnet = w >> 2, pcb = (w >> 1) & 1, half = w & 1;
            const float* base = P_ + (((size_t)st * NRT + pnet * HB + half * HB2) * 2 + pcb) * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < HB2; ++i) pp[i] = ld4(base + (size_t)i * 512);
        }
        v4f xv0 = {0.f, 0.f, 0.f, 0.f};
        if (tid < 256) {
            const int nn = tid >> 4, c4 = tid & 15, b = st * 16 + nn;
            if (io_in) { if (b < B_) xv0 = ld4(io_->xin + (size_t)b * 64 + 4 * c4); }
            else xv0 = ld4(x_ + (size_t)b * 64 + 4 * c4);
        }
        v4f D0[KPW];
#pragma unroll
        for (int i = 0; i < KPW; ++i) D0[i] = ld4(pre_ + ((size_t)st * NRT + net * HB + w * KPW + i) * 256 + lane * 4);
        if (s == 0) {
            if (l1 == 1) {
#pragma unroll
                for (int i = 0; i < KPW; ++i) {
                    const size_t r0 = (size_t)(net * HB + w * KPW + i) * 2;
                    A0[i][0] = ldw<F16>(W0T_, r0, lane);
                    A0[i][1] = ldw<F16>(W0T_, r0 + 1, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < KPW; ++i) A[i] = ldw<F16>(WT_, (size_t)rt * HB + w * KPW + i, lane);
            if (epi) {
                bias4 = ld4(bias_ + rt * 16 + q * 4);
                if (W3P_) A3 = ldw<F16>(W3P_, (size_t)rt * 2 + cb, lane);
            }
            if (tid >= 256 && tid < 320) { anl[tid - 256] = tab0; ans[tid - 256] = tab1; }
            else if (tid >= 320 && tid < 384) sidx[tid - 320] = tabi;
        }
        if (tid < 256) st4(&xs[tid >> 4][4 * (tid & 15)], xv0);
        if (s == 0) { FTL_STAMP(seq_, 1) FTL_LANDED() FTL_STAMP(seq_, 2) }
        if (P_) {
            v4f sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < HB2; ++i) sum += pp[i];
            ps[w][lane] = sum;
        }
        __syncthreads();
        // ---- affine coupling of the transformed half (flow_blocks.py:91-93 / 103) and its log-det: waves 0, 1 = channel blocks
        if (P_ && w < 2) {
            const v4f sv = (bs + ps[w * 2][lane]) + ps[w * 2 + 1][lane];
            const v4f tv = (bt + ps[4 + w * 2][lane]) + ps[4 + w * 2 + 1][lane];
            v4f xv = ld4(&xs[n][32 + 16 * w + 4 * q]);
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = reverse ? (xv[r] - tv[r]) * expf(-sv[r]) : fmaf(xv[r], expf(sv[r]), tv[r]);
            st4(&xs[n][32 + 16 * w + 4 * q], xv);
            if (logdet_ && !reverse) {
                float l = (sv[0] + sv[1]) + (sv[2] + sv[3]);
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
                if (q == 0) ld[w][n] = l;
            }
        }
        __syncthreads();
        // ---- block boundary: Shuffle gather, ActNorm, InvLeakyRelu, half swap (as flow_tail_tile_kernel)
        {
            const int nn = tid >> 5, e2 = (tid & 31) * 2, b = st * 16 + nn;
            float o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cp = e2 + e;
                const int c1 = do_swap ? cp ^ 32 : cp;
                const int c0 = sidx[c1];
                float v = xs[nn][c0];
                if (!reverse) {
                    if (an_loc_) v = ans[c1] * (v + anl[c1]);
                    if (do_lrelu) v = v * (v >= 0.f ? 1.0f : 0.9f);
                } else {
                    if (do_lrelu) v = v / (v >= 0.f ? 1.0f : 0.9f);
                    if (an_loc_) v = v / ans[c0] - anl[c0];
                }
                o[e] = v;
            }
            *reinterpret_cast<float2*>(&xs2[nn][e2]) = make_float2(o[0], o[1]);
            if (writer && live) {
                *reinterpret_cast<float2*>(xo_ + (size_t)b * 64 + e2) = make_float2(o[0], o[1]);
                if (logdet_ && tid < 16) {
                    const int bb = st * 16 + tid;
                    float v = ld_init ? 0.f : logdet_[bb];
                    if (P_) v += ld[0][tid] + ld[1][tid];
                    if (an_loc_) v += an_logdet_;
                    logdet_[bb] = v;
                }
            }
        }
        __syncthreads();
        // ---- first Linear of this half-step (modules.py:14-17) for the wave's own K slice of the hidden layer: stays in registers
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            v4f D = D0[i];
            if (l1 == 1) {
                D = mma16<F16>(A0[i][0], ld4(&xs2[n][4 * q]), D);
                D = mma16<F16>(A0[i][1], ld4(&xs2[n][16 + 4 * q]), D);
            }
            Bv[s][i] = lrelu4(D, 0.01f);
        }
        if (NS > 1) __syncthreads();   // xs / xs2 / ps are rewritten for the next sample tile
    }
    // ---- the hidden Linear(H, H) + LeakyReLU of flow_hid_tile_kernel on the h0 tiles in registers
    v4f D[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) D[s] = v4f{0.f, 0.f, 0.f, 0.f};
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) D[s] = mma16<true>(A[i], Bv[s][i], D[s]);
    } else {
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) D[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i][j], Bv[s][i][j], D[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) red[w][s][lane] = D[s];
    FTL_STAMP(seq_, 3)
    __syncthreads();
    FTL_STAMP(seq_, 4)
    if (!epi) return;
    const int st = st0 + es;
    if (st >= NST) return;
    v4f h = bias4;
#pragma unroll
    for (int v = 0; v < 8; ++v) h += red[v][es][lane];
    h = lrelu4(h, 0.01f);
    if (out_ && cb == 0) st4(out_ + ((size_t)st * NRT + rt) * 256 + lane * 4, h);
    if (W3P_) {
        const v4f d3 = mma16<F16>(A3, h, v4f{0.f, 0.f, 0.f, 0.f});
        st4(Pout_ + (((size_t)st * NRT + rt) * 2 + cb) * 256 + lane * 4, d3);
    }
    FTL_STAMP(seq_, 5)
    FTL_END(seq_)
}

__global__ void flow_set_io_kernel(FlowIo* dst, FlowIo v) { *dst = v; }

template <int KPW, bool F16>
void launch_hid(const HidTileArgs& a, int ns, int groups, hipStream_t st) {
    const dim3 grid(a.NRT * groups), block(512);
    if (ns == 1) hipLaunchKernelGGL((flow_hid_tile_kernel<KPW, 1, F16>), grid, block, 0, st, a.WT, a.in, a.bias, a.out, a.W3P, a.P, a.NRT, a.NST, a.seq);
    else if (ns == 2) hipLaunchKernelGGL((flow_hid_tile_kernel<KPW, 2, F16>), grid, block, 0, st, a.WT, a.in, a.bias, a.out, a.W3P, a.P, a.NRT, a.NST, a.seq);
    else hipLaunchKernelGGL((flow_hid_tile_kernel<KPW, 4, F16>), grid, block, 0, st, a.WT, a.in, a.bias, a.out, a.W3P, a.P, a.NRT, a.NST, a.seq);
}

struct FirstTileArgs {
    HidTileArgs m;      // the hidden layer (m.in unused)
    TailTileArgs t;     // the tail in front of it (t.h0 unused; t.W0T / t.pre / t.l1: the first Linear of the hidden layer's half-step)
};

template <int KPW, bool F16>
void launch_first(const FirstTileArgs& a, int ns, int groups, hipStream_t st) {
    const dim3 grid(a.m.NRT * groups), block(512);
    const TailTileArgs& t = a.t;
    const int flags = t.io_in | t.ld_init << 2 | t.reverse << 3 | t.do_lrelu << 4 | t.do_swap << 5 | t.l1 << 6;
#define I2V_FIRST_ARGS a.m.WT, t.P, t.x, t.W0T, t.pre, a.m.bias, a.m.NRT, a.m.NST, flags, t.b3, t.io, t.xo, t.logdet, t.shuf, t.an_loc, t.an_scale, \
                       t.an_logdet, a.m.out, a.m.W3P, a.m.P, t.B, a.m.seq
    if (ns == 1) hipLaunchKernelGGL((flow_first_tile_kernel<KPW, 1, F16>), grid, block, 0, st, I2V_FIRST_ARGS);
    else if (ns == 2) hipLaunchKernelGGL((flow_first_tile_kernel<KPW, 2, F16>), grid, block, 0, st, I2V_FIRST_ARGS);
    else hipLaunchKernelGGL((flow_first_tile_kernel<KPW, 4, F16>), grid, block, 0, st, I2V_FIRST_ARGS);
#undef I2V_FIRST_ARGS
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atoi(e) : dflt;
}

}  // namespace

static int upload_frags(DevBuf& dst, const std::vector<float>& w, bool f16) {
    if (!f16) return dst.upload(w.data(), w.size() * 4);
    std::vector<_Float16> h(w.size());
    for (size_t i = 0; i < w.size(); ++i) h[i] = (_Float16)w[i];   // round to nearest even, once, at load
    return dst.upload(h.data(), h.size() * 2);
}

int flow_tile_pack(FlowTilePack& p, int S, int H, int depth, int E, const float* W0, const float* Wmid, const float* W3T, bool f16) {
    p.ok = false;
    p.f16 = f16;
    p.force_ns = env_int("I2V_FLOW_NS", 0);
    p.force_fold = env_int("I2V_FLOW_FOLD", -1);
    p.S = S; p.H = H; p.depth = depth; p.E = E;
    p.HB = H / 16; p.NRT = 2 * p.HB; p.KE16 = (E + 15) / 16;
    const int HB = p.HB, NRT = p.NRT, KE16 = p.KE16, N2 = 2 * H, ld0 = 32 + E;
    std::vector<float> wt((size_t)S * depth * NRT * HB * 256), w3((size_t)S * NRT * 2 * 256), w0t((size_t)S * NRT * 2 * 256),
        w0e((size_t)S * NRT * KE16 * 256, 0.f);
    for (int l = 0; l < S * depth; ++l)
        for (int rt = 0; rt < NRT; ++rt)
            for (int kb = 0; kb < HB; ++kb) {
                float* dst = &wt[(((size_t)l * NRT + rt) * HB + kb) * 256];
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, m = lane & 15;
                    const float* src = Wmid + ((size_t)l * N2 + 16 * rt + m) * H + 16 * kb + 4 * q;
                    for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = src[j];
                }
            }
    for (int s = 0; s < S; ++s)
        for (int rt = 0; rt < NRT; ++rt) {
            const int net = rt / HB, rtn = rt % HB;
            for (int cb = 0; cb < 2; ++cb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, m = lane & 15;
                    for (int j = 0; j < 4; ++j)
                        w3[((((size_t)s * NRT + rt) * 2 + cb) * 64 + lane) * 4 + j] =
                            W3T[((size_t)s * H + 16 * rtn + 4 * q + j) * 64 + net * 32 + 16 * cb + m];
                }
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, m = lane & 15;
                const float* row = W0 + ((size_t)s * N2 + 16 * rt + m) * ld0;
                for (int kb = 0; kb < 2; ++kb)
                    for (int j = 0; j < 4; ++j) w0t[((((size_t)s * NRT + rt) * 2 + kb) * 64 + lane) * 4 + j] = row[16 * kb + 4 * q + j];
                for (int kb = 0; kb < KE16; ++kb)
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * kb + 4 * q + j;
                        w0e[((((size_t)s * NRT + rt) * KE16 + kb) * 64 + lane) * 4 + j] = k < E ? row[32 + k] : 0.f;
                    }
            }
        }
    int rc;
    if ((rc = upload_frags(p.WT, wt, f16))) return rc;
    if ((rc = upload_frags(p.W3P, w3, f16))) return rc;
    if ((rc = upload_frags(p.W0T, w0t, f16))) return rc;
    if ((rc = upload_frags(p.W0E, w0e, f16))) return rc;
    FlowIo zero{};
    if ((rc = p.io.upload(&zero, sizeof(zero)))) return rc;
    p.io_host = zero;
    p.ok = true;
    return I2V_OK;
}

FlowTileWs flow_tile_ws(const FlowTilePack& p, int B) {
    FlowTileWs L;
    const size_t NST = (size_t)(B + 15) / 16;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n, 256); return r; };
    L.x = take(NST * 16 * 64 * 4);
    L.x2 = take(NST * 16 * 64 * 4);
    L.logdet = take(NST * 16 * 4);
    L.pre = take((size_t)p.S * NST * p.NRT * 1024);
    L.hA = take(NST * p.NRT * 1024);
    L.hB = take(NST * p.NRT * 1024);
    L.P = take(NST * p.NRT * 2 * 1024);
    L.P2 = take(NST * p.NRT * 2 * 1024);
    L.total = o;
    return L;
}

int flow_tile_set_io(FlowTilePack& p, const FlowIo& io, hipStream_t st) {
    if (io.xin == p.io_host.xin && io.embed == p.io_host.embed && io.xout == p.io_host.xout && io.logdet_out == p.io_host.logdet_out)
        return I2V_OK;
    hipLaunchKernelGGL(flow_set_io_kernel, dim3(1), dim3(1), 0, st, p.io.as<FlowIo>(), io);
    I2V_HIP_CHECK(hipGetLastError());
    p.io_host = io;
    return I2V_OK;
}

int flow_tile_enqueue(const FlowTileChain& c, bool reverse, char* ws, int B, hipStream_t st) {
    const FlowTilePack& p = *c.pack;
    const FlowTileWs L = flow_tile_ws(p, B);
    const int NST = (B + 15) / 16, NRT = p.NRT, HB = p.HB, S = p.S, D = p.depth, N2 = 2 * p.H, nf = c.n_flows;
    float* xbuf[2] = {reinterpret_cast<float*>(ws + L.x), reinterpret_cast<float*>(ws + L.x2)};   // state, ping-pong per tail
    int xcur = 0;
    float* logdet = reinterpret_cast<float*>(ws + L.logdet);
    float* pre = reinterpret_cast<float*>(ws + L.pre);
    float* hA = reinterpret_cast<float*>(ws + L.hA);
    float* hB = reinterpret_cast<float*>(ws + L.hB);
    float* Pbuf[2] = {reinterpret_cast<float*>(ws + L.P), reinterpret_cast<float*>(ws + L.P2)};   // half-step `it` writes Pbuf[it & 1]
    const FlowIo* io = p.io.as<FlowIo>();
    int ns = NST <= 4 ? 1 : NST <= 8 ? 2 : 4;   // sample tiles per hidden-layer workgroup: keep ~256 workgroups
    if (const int e = p.force_ns) ns = e >= 4 ? 4 : e >= 2 ? 2 : 1;
    const int groups = (NST + ns - 1) / ns;
    // Folded chain (the tail travels with the first hidden layer's launch: 82 launches per pass) or round 4's 122-launch chain.
    // Same bits either way (test_flow_fold_keeps_the_bits).  Every workgroup of a folded launch redoes the tail of its own sample
    // tiles (128 KB of partial tiles each), so it pays while a workgroup holds ONE sample tile (B <= 64: 508 -> 473 us at B = 64,
    // 460 -> 439 at B = 8) and loses with four (B = 256: 765 -> 1083 us): default = folded iff ns == 1.  I2V_FLOW_FOLD=0|1 (read when the weights are packed) forces.
    const bool fold = (p.force_fold >= 0 ? p.force_fold : (ns == 1 ? 1 : 0)) != 0;
    int seq = 0;   // launch number inside the pass
    const size_t fb = p.f16 ? 512 : 1024;   // bytes per weight fragment

    {   // embedding part of every first layer of the pass
        PreTileArgs a{};
        a.seq = seq++;
        a.W0E = p.W0E.as<float>(); a.b0 = c.b0; a.io = io; a.pre = pre;
        a.NRT = NRT; a.NST = NST; a.KE16 = p.KE16; a.E = p.E; a.B = B; a.Rtiles = S * NRT; a.nblk = (a.Rtiles + 7) / 8;
        if (p.f16) hipLaunchKernelGGL(flow_pre_tile_kernel<true>, dim3(a.nblk * ((NST + PRE_SC - 1) / PRE_SC)), dim3(512), 0, st, a);
        else hipLaunchKernelGGL(flow_pre_tile_kernel<false>, dim3(a.nblk * ((NST + PRE_SC - 1) / PRE_SC)), dim3(512), 0, st, a);
        I2V_HIP_CHECK(hipGetLastError());
    }
    auto step_of = [&](int it) {  // forward visits (fl, i) = (0,0),(0,1),(1,0)...; reverse visits (nf-1,1),(nf-1,0),(nf-2,1)...
        const int fl = reverse ? nf - 1 - it / 2 : it / 2;
        const int i = reverse ? 1 - it % 2 : it % 2;
        return fl * 2 + i;
    };
    // hidden layer d of half-step `step`; its input in `cur`, its output in `nxt` (the last one writes partial products into Pout)
    auto hid_args = [&](int step, int d, const float* cur, float* nxt, float* Pout) {
        HidTileArgs m{};
        m.WT = reinterpret_cast<const float*>(p.WT.as<char>() + ((size_t)step * D + d) * NRT * HB * fb);
        m.bias = c.bmid + ((size_t)step * D + d) * N2;
        m.in = cur;
        m.out = d == D - 1 ? nullptr : nxt;
        m.W3P = d == D - 1 ? reinterpret_cast<const float*>(p.W3P.as<char>() + (size_t)step * NRT * 2 * fb) : nullptr;
        m.P = Pout;
        m.NRT = NRT; m.NST = NST;
        return m;
    };
    auto launch_hidden = [&](HidTileArgs m) -> int {
        m.seq = seq++;
        if (p.f16) {
            switch (HB / 8) {
                case 1: launch_hid<1, true>(m, ns, groups, st); break;
                case 2: launch_hid<2, true>(m, ns, groups, st); break;
                case 3: launch_hid<3, true>(m, ns, groups, st); break;
                default: launch_hid<4, true>(m, ns, groups, st); break;
            }
        } else {
            switch (HB / 8) {
                case 1: launch_hid<1, false>(m, ns, groups, st); break;
                case 2: launch_hid<2, false>(m, ns, groups, st); break;
                case 3: launch_hid<3, false>(m, ns, groups, st); break;
                default: launch_hid<4, false>(m, ns, groups, st); break;
            }
        }
        I2V_HIP_CHECK(hipGetLastError());
        return I2V_OK;
    };
    // the tail between two half-steps: coupling of `step` (partial products in Pin; null in front of the first half-step), block
    // boundary ops, first Linear of next_step.  folded_hidden: the first hidden layer of next_step runs in the same launch.
    auto tail = [&](const float* Pin, int step, int shuf_block, int an_block, bool lrelu, bool swap, int next_step, bool first, bool last,
                    const HidTileArgs* folded_hidden) -> int {
        TailTileArgs t{};
        t.P = Pin;
        t.b3 = c.b3 + (size_t)step * 64;
        t.x = xbuf[xcur]; t.xo = xbuf[xcur ^ 1];
        xcur ^= 1;
        t.logdet = reverse ? nullptr : logdet;
        t.io = io; t.io_in = first ? 1 : 0; t.io_out = last ? 1 : 0; t.ld_init = first ? 1 : 0;
        t.B = B; t.NST = NST; t.NRT = NRT; t.HB2 = HB / 2; t.reverse = reverse ? 1 : 0;
        t.shuf = shuf_block >= 0 ? (reverse ? c.shuf_b : c.shuf_f) + shuf_block * 64 : nullptr;
        t.an_loc = an_block >= 0 ? c.an_loc + an_block * 64 : nullptr;
        t.an_scale = an_block >= 0 ? c.an_scale + an_block * 64 : nullptr;
        t.an_logdet = an_block >= 0 ? c.an_logdet_host[an_block] : 0.f;
        t.do_lrelu = lrelu ? 1 : 0; t.do_swap = swap ? 1 : 0;
        if (next_step >= 0) {
            t.l1 = c.step_cond[next_step] ? 2 : 1;
            t.W0T = reinterpret_cast<const float*>(p.W0T.as<char>() + (size_t)next_step * NRT * 2 * fb);
            t.pre = pre + (size_t)next_step * NST * NRT * 256;
            t.h0 = hA;
        }
        if (folded_hidden) {
            FirstTileArgs fa{*folded_hidden, t};
            fa.m.seq = seq++;
            if (p.f16) {
                switch (HB / 8) {
                    case 1: launch_first<1, true>(fa, ns, groups, st); break;
                    case 2: launch_first<2, true>(fa, ns, groups, st); break;
                    case 3: launch_first<3, true>(fa, ns, groups, st); break;
                    default: launch_first<4, true>(fa, ns, groups, st); break;
                }
            } else {
                switch (HB / 8) {
                    case 1: launch_first<1, false>(fa, ns, groups, st); break;
                    case 2: launch_first<2, false>(fa, ns, groups, st); break;
                    case 3: launch_first<3, false>(fa, ns, groups, st); break;
                    default: launch_first<4, false>(fa, ns, groups, st); break;
                }
            }
            I2V_HIP_CHECK(hipGetLastError());
            return I2V_OK;
        }
        t.seq = seq++;
        const int flags = t.io_in | t.io_out << 1 | t.ld_init << 2 | t.reverse << 3 | t.do_lrelu << 4 | t.do_swap << 5 | t.l1 << 6;
        auto tk = p.f16 ? flow_tail_tile_kernel<true> : flow_tail_tile_kernel<false>;
        hipLaunchKernelGGL(tk, dim3((NST + 7) / 8 * 64), dim3(512), 0, st, t.P, t.x, t.W0T, t.pre, t.b3, t.NST, t.NRT, t.HB2,
                           flags, t.io, t.xo, t.logdet, t.shuf, t.an_loc, t.an_scale, t.an_logdet, t.h0, t.B, t.seq);
        I2V_HIP_CHECK(hipGetLastError());
        return I2V_OK;
    };
    const bool act = c.use_act, an = c.use_an, sh = c.use_shuf;
    int rc;
    // (folded: the hidden layer 0 of half-step step_of(it) travels with the tail in front of it; its output goes where the
    //  unfolded chain's layer 0 puts it -- hB -- or, when it is also the last hidden layer, into Pbuf[it & 1])
    HidTileArgs h0a = hid_args(step_of(0), 0, hA, hB, Pbuf[0]);
    if (!reverse) rc = tail(nullptr, 0, -1, an ? 0 : -1, act, false, step_of(0), true, false, fold ? &h0a : nullptr);
    else rc = tail(nullptr, 0, sh ? nf - 1 : -1, -1, false, false, step_of(0), true, false, fold ? &h0a : nullptr);
    if (rc) return rc;
    for (int it = 0; it < S; ++it) {
        const int fl = reverse ? nf - 1 - it / 2 : it / 2;
        const int i = reverse ? 1 - it % 2 : it % 2;
        const int step = fl * 2 + i;
        const int next_step = it + 1 < S ? step_of(it + 1) : -1;
        float* cur = hA;
        float* nxt = hB;
        for (int d = 0; d < D; ++d) {
            if (!(fold && d == 0) && (rc = launch_hidden(hid_args(step, d, cur, nxt, Pbuf[it & 1])))) return rc;
            std::swap(cur, nxt);
        }
        int shuf_block = -1, an_block = -1;
        bool lrelu = false, swap = false;
        if (!reverse) {
            if (i == 0) swap = true;  // before half-step 1: cat(chunk[::-1]), flow_blocks.py:86
            else {
                if (sh) shuf_block = fl;
                if (fl + 1 < nf) { if (an) an_block = fl + 1; lrelu = act; }
            }
        } else {
            if (i == 1) swap = true;  // before half-step 0 (flow_blocks.py:98-99)
            else {
                lrelu = act;
                if (an) an_block = fl;
                if (fl - 1 >= 0 && sh) shuf_block = fl - 1;
            }
        }
        HidTileArgs hn{};
        const bool folded = fold && next_step >= 0;
        if (folded) hn = hid_args(next_step, 0, hA, hB, Pbuf[(it + 1) & 1]);
        if ((rc = tail(Pbuf[it & 1], step, shuf_block, an_block, lrelu, swap, next_step, false, it == S - 1, folded ? &hn : nullptr))) return rc;
    }
    return I2V_OK;
}

#ifdef FLOW_TIMELINE
}  // namespace i2v
// measurement build only: kind[launch]: 0 pre-GEMM, 1 tail, 2 hidden layer (the launch order of one pass is fixed: pre, tail, then
// (hid x depth, tail) per half-step; FOLDED chain (n_launches == 2 + half-steps x depth): pre, then (first = tail + hidden 0, hid x
// (depth - 1)) per half-step, final tail -- the folded launches are reported under the tail's phase names, their stamps 1-2 =
// requests / landed, 3-5 = the hidden layer's); prints mean phase durations per kind and the launch-to-launch gaps
extern "C" int i2v_flow_timeline_report(int n_launches, int depth, int reset) {
    using namespace i2v;
    std::vector<unsigned long long> tl((size_t)FTL_LAUNCHES * FTL_WGS * FTL_ST), span((size_t)FTL_LAUNCHES * 2);
    if (reset) {
        std::fill(tl.begin(), tl.end(), 0ull);
        for (int i = 0; i < FTL_LAUNCHES; ++i) { span[2 * i] = ~0ull; span[2 * i + 1] = 0ull; }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(flow_tl), tl.data(), tl.size() * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(flow_tl_span), span.data(), span.size() * 8);
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(flow_tl), tl.size() * 8);
    (void)hipMemcpyFromSymbol(span.data(), HIP_SYMBOL(flow_tl_span), span.size() * 8);
    if (n_launches > FTL_LAUNCHES) n_launches = FTL_LAUNCHES;
    const bool folded = depth > 0 && (n_launches - 2) % depth == 0 && (n_launches - 2) / depth * (depth + 1) + 2 != n_launches;
    auto kind = [&](int l) {
        if (l == 0) return 0;
        if (folded) return l == n_launches - 1 ? 1 : ((l - 1) % depth == 0 ? 1 : 2);
        return (l - 1) % (depth + 1) == 0 ? 1 : 2;
    };
    const char* kn[3] = {"flow_pre_tile_kernel", "flow_tail_tile_kernel", "flow_hid_tile_kernel"};
    const char* ph[2][5] = {{"entry -> requests issued", "requests -> operands landed (P tiles, state, weights)", "partial sums + barrier",
                             "coupling, log-det, block boundary", "first Linear of the next half-step + store"},
                            {"entry -> requests issued", "requests -> operands landed (weights, activations)", "MFMAs + partials to LDS",
                             "barrier", "reduce, bias, lrelu, final-Linear MFMA, store"}};
    double dur[3] = {}, gap[3] = {}, phs[3][5] = {};
    int cnt[3] = {}, gcnt[3] = {}, pcnt[3] = {};
    for (int l = 0; l < n_launches; ++l) {
        const int k = kind(l);
        if (span[2 * l] == ~0ull) continue;
        dur[k] += (double)(span[2 * l + 1] - span[2 * l]); cnt[k]++;
        if (l + 1 < n_launches && span[2 * (l + 1)] != ~0ull) { gap[k] += (double)((long long)span[2 * (l + 1)] - (long long)span[2 * l + 1]); gcnt[k]++; }
        if (k == 0) continue;
        for (int w = 0; w < FTL_WGS; ++w) {
            const unsigned long long* t = &tl[((size_t)l * FTL_WGS + w) * FTL_ST];
            if (!t[0] || !t[5]) continue;
            for (int i = 0; i < 5; ++i) phs[k][i] += (double)(t[i + 1] - t[i]);
            pcnt[k]++;
        }
    }
    printf("cINN launch timeline (us, 100 MHz wall clock; one pass = %d launches)\n", n_launches);
    for (int k = 0; k < 3; ++k) {
        if (!cnt[k]) continue;
        printf("  %-22s x %3d: first workgroup start -> last workgroup end %.2f; gap to the next launch's first start %.2f\n", kn[k], cnt[k],
               dur[k] / cnt[k] / 100.0, gcnt[k] ? gap[k] / gcnt[k] / 100.0 : 0.0);
        if (k && pcnt[k])
            for (int i = 0; i < 5; ++i) printf("      %-58s %.2f\n", ph[k - 1][i], phs[k][i] / pcnt[k] / 100.0);
    }
    double total = 0;
    if (n_launches > 1 && span[0] != ~0ull) total = (double)(span[2 * (n_launches - 1) + 1] - span[0]) / 100.0;
    printf("  pass, first start -> last end: %.1f us\n", total);
    return 0;
}
#else
}  // namespace i2v
#endif

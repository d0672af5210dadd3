// 3x3x3 Conv3d with a Winograd F(2,3) transform along W on the gfx950 fp16 matrix cores (split-fp16 operands).
//
// The split-fp16 implicit GEMM of i2v_conv16.hip is bound by the NUMBER of MFMAs (three per product; the matrix cores
// run power-limited at 65-67 % of their sustained rate whatever the schedule).  This kernel issues 1.5x fewer of them:
// for every pair of output positions (w = 2j, 2j+1) -- a "tile" -- and every (kt, kh) tap
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3          d_k = a[t+kt-1][h+kh-1][2j-1+k]  (zero padded)
//     U0 = g0        U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2     g_k = w[kt][kh][k]
//     M_x = sum over (kt, kh, c) of V_x * U_x   (x = 0..3: four GEMMs instead of the six of two outputs x three kw taps)
//     y[2j] = M0 + M1 + M2        y[2j+1] = M1 - M2 - M3
// The input transform is done ONCE per element by the producer (the modulate kernel writes V = B^T d, computed in fp32
// and then split into fp16 hi/lo, as [B][T][C/16][4][H][W/2][16 channels]: K-chunk-major, so that the 64-byte rows a
// workgroup stages per chunk are contiguous along w); the weight transform G g is done in fp64 at load time.  Both
// operands therefore keep the 2^-22 relative precision of the hl16 format and the products the three-term form
// hi*hi + hi*lo + lo*hi of i2v_conv16.hip; the output transform A^T M runs in fp32 in the epilogue.
//
// Workgroup = 512 threads = 8 wavefronts: 128 tiles (256 output positions, TT x TH x 2TJ brick) x 64 output channels for
// all four x.  Wave w owns x = w & 3 and the 32-channel half w >> 2: four MFMA row blocks (32 tiles each) x one column block
// of v_mfma_f32_32x32x16_f16 = 64 accumulator registers, so no two waves ever add into the same accumulator; the four
// partial results of a tile meet in the epilogue (through LDS).  K chunk = 16 channels = one MFMA k-step per tap.
//   * V halo brick [x][TT+KT-1][TH+2][TJ] (no halo along w -- it is inside V) of 64-byte rows (2 groups x (8 hi | 8 lo)),
//     DOUBLE-buffered in LDS, unpadded: the 16-byte piece p of row r sits at p ^ ((r >> 2) & 3), which makes every
//     ds_read_b128 lane group hit 16 distinct bank quads.  The brick arrives by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no ds_write pass; the swizzle is applied to the per-lane SOURCE address, padding rows read a zero page);
//     the next chunk's brick is requested in two halves at the chunk's first two taps and published by the chunk's ONE
//     barrier.
//   * Weights never touch LDS: fragment-major U, each wave loads the B operand of a tap with two coalesced 1 KB loads into
//     a register ring (nine slots = eight taps ahead in the 9-tap kernel, six elsewhere), one continuous stream across the
//     chunk boundaries.
//   * The loop body is a pair of chunks (ring slot, A register set and V buffer are all compile-time, the body is
//     branch-free); the loads are asm statements and EVERY wait is counted by hand (tools/check_asm_waits.py replays the
//     compiled loops against the in-order VMEM queue, and checks that the loop is entered with nothing in flight).
//   * Within a tap the wave's other work (next tap's LDS address arithmetic and reads, the weight request: scalar base + the
//     lane's 16 bytes) sits one piece per 32-cycle gap between the MFMAs, and wave priority falls with the tap index inside a
//     chunk (round 3; the timing behind it was taken on the F(4,3) kernel, i2v_conv16w4.hip).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "i2v_conv.h"

namespace i2v {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // (HIP's float4 is a struct: copies through it become memcpys that pin register arrays in scratch)

constexpr int W16_TILES = 128;  // Winograd tiles per workgroup
constexpr int W16_KC = 16;      // input channels per K chunk
constexpr int W16_SLOTS = 8;    // prefetched 16-byte V pieces per thread and chunk
constexpr int W16_VROWS = W16_SLOTS * 512 / 4;  // staged V rows (64 B each): the halo brick, padded to 1024 rows


struct WinoArgs {
    const char* in;   // V: hl16 [B][T][Cin/16][4][H][J][16 channels = 64 B]
    const char* zeros;  // >= 64 zero bytes: source of the padding rows (no select behind the prefetch loads)
    const char* wp;   // U: [parity][tap][chunk][4][CoutPad][64 B]
    const float* bias;
    const float* res;
    float* out;       // fp32 channels-last [B][To][H][W][Cout]
    double* stats;
    int B, T, H, W, J, Cin, Cout, CoutPad, nchunk;  // T = frames of the INPUT tensor
    int tdup;
    long wset_stride;
    int TT, TH, TJ, nbT, nbH, nbJ;
    int rt, rs, epi;
    float oscale;
    int wofs, tofs;   // LDS byte offsets of the weight buffers / the index tables
};

// number of taps U - k, k = 0 .. R - 1, that are tap `h` of their chunk (t = U % NT)
constexpr int w16_count(int t, int R, int NT, int h) {
    int n = 0;
    for (int k = 0; k < R; ++k) n += ((t - k - h) % NT + NT) % NT == 0;
    return n;
}

constexpr int w16_prio(int t, int NT) { return NT == 3 ? 3 - t : (t < 6 ? 3 - t / 2 : 0); }   // 9 taps: 3 3 2 2 1 1 0 0 0

#ifndef W16_RING9
#define W16_RING9 1   // measurement builds: 0 = six-slot B ring for the 9-tap kernel too
#endif

// NT: (kt, kh) taps: 9 = 3x3x3, 6 = temporal-duplication pair kernels (2x3x3), 3 = one time-slice (1x3x3).
// BN: output channels per workgroup.  64: wave = (x, 32-channel half), all 128 tiles (4 MFMA row blocks); 32 (layers with
// 32 output channels): wave = (x, tile half), 64 tiles (2 row blocks) -- the two waves of an x load the same B operand.
template <int NT, int BN>
__global__ __launch_bounds__(512, 2) void conv_wino_f16x3_kernel(WinoArgs a) {
    constexpr int KT = NT / 3;
    constexpr int WM = BN == 64 ? 4 : 2;  // MFMA row blocks (32 tiles each) per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave & 3;                      // Winograd position of this wave
    const int nh = BN == 64 ? wave >> 2 : 0;      // its 32-channel half (BN = 64)
    const int mh = BN == 64 ? 0 : wave >> 2;      // its tile half (BN = 32)
    const int kg = lane >> 5, l31 = lane & 31;

    // Tile order and the XCDs.  Workgroup b runs on XCD b % 8 (observed, not promised: only speed depends on it), each XCD
    // with its own L2.  All workgroups that read the SAME V brick -- the channel tiles of a brick and, in temporal-
    // duplication mode, its two frame parities -- are given consecutive dispatch slots of ONE XCD, and XCD x owns the
    // bricks x, x + 8, ...: with 8 bricks per row that is one w-column, whose h- and t-neighbours share their halos in
    // the same L2.  (Numbering the channel tile fastest put the tiles of a brick on different XCDs: every L2 fetched
    // the whole V tensor, 2.9x its size from HBM per launch.)
    const int nNt_ = a.CoutPad / BN;
    const int npar_ = a.tdup ? 2 : 1;
    const int per_brick = nNt_ * npar_;
    const int nbrick = (int)(gridDim.x / per_brick);
    int par, tile_id;
    if ((nbrick & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int sub = slot % per_brick, brick_ = (slot / per_brick) * 8 + xcd;
        par = a.tdup ? sub & 1 : 0;
        tile_id = brick_ * nNt_ + (a.tdup ? sub >> 1 : sub);
    } else {
        par = a.tdup ? (int)(blockIdx.x >= (gridDim.x >> 1)) : 0;
        tile_id = a.tdup ? (int)(blockIdx.x % (gridDim.x >> 1)) : (int)blockIdx.x;
    }
    const int pt = a.tdup ? 1 - par : KT / 2;
    const int HT = a.TT + KT - 1, HH = a.TH + 2;
    const int plane = HT * HH * a.TJ;
    const int NROW = 4 * plane;

    char* v_lds = smem;                                  // two V bricks of W16_VROWS rows
    int* gpos = reinterpret_cast<int*>(smem + a.tofs);   // [W16_VROWS] global V row of every staged row, -1 = zero padding
    int* tpos = gpos + W16_VROWS;                        // [128] output position of a tile's first column
    int* tres = tpos + W16_TILES;                        // [128][2] residual rows of the tile's two columns

    const int nNt = a.CoutPad / BN;
    const int ntile = tile_id % nNt;
    int brick = tile_id / nNt;
    const int bj = brick % a.nbJ; brick /= a.nbJ;
    const int bh = brick % a.nbH; brick /= a.nbH;
    const int bt = brick % a.nbT; brick /= a.nbT;
    const int b0 = brick, t0 = bt * a.TT, h0 = bh * a.TH, j0 = bj * a.TJ;
    const int n0 = ntile * BN;

    if (tid < W16_TILES) {
        int m = tid;
        const int ij = m % a.TJ; m /= a.TJ;
        const int ih = m % a.TH; m /= a.TH;
        const int t = t0 + m, h = h0 + ih, w = 2 * (j0 + ij);
        const int To = a.tdup ? 2 * a.T : a.T, to = a.tdup ? 2 * t + par : t;
        tpos[tid] = ((b0 * To + to) * a.H + h) * a.W + w;
        const int rbase = ((b0 * (To / a.rt) + to / a.rt) * (a.H / a.rs) + h / a.rs) * (a.W / a.rs);
        tres[2 * tid] = rbase + w / a.rs;
        tres[2 * tid + 1] = rbase + (w + 1) / a.rs;
    }
    for (int r = tid; r < W16_VROWS; r += 512) {  // (rows >= NROW: never read, staged as zeros so that no access is conditional)
        const int x = r / plane;
        int q = r - x * plane;
        const int ij = q % a.TJ; q /= a.TJ;
        const int ih = q % HH; q /= HH;
        const int t = t0 + q - pt, h = h0 + ih - 1, j = j0 + ij;
        const bool ok = r < NROW && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H;
        gpos[r] = ok ? ((((b0 * a.T + t) * a.nchunk * 4 + x) * a.H + h) * a.J + j) : -1;  // chunk 0; 64-byte rows
    }

    // LDS row of this lane's tile (tap (0,0)) in its x plane, per MFMA row block (4 blocks of 32 tiles)
    int arow[WM];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm) {
        int m = mh * 64 + wm * 32 + l31;
        const int ij = m % a.TJ; m /= a.TJ;
        const int ih = m % a.TH; m /= a.TH;
        arow[wm] = xi * plane + (m * HH + ih) * a.TJ + ij;
    }

    f32x16 acc[WM];
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wm][r] = 0.f;

    // weights: fragment-major [tap][chunk][x][32-channel block][hi | lo][64 lanes][16 B]: a wave's B operand of one tap is
    // two coalesced 1 KB loads straight into registers (no LDS staging, no barrier)
    const long cstride = (long)a.CoutPad * 256;          // bytes per (tap, chunk): 4 x x CoutPad x 64
    const long wtap_stride = (long)a.nchunk * cstride;
    // (the fragment address is scalar -- it goes into the loads' SGPR base operand -- plus the lane's 16 bytes)
    const char* wfrag;
    {
        const unsigned long w_ = (unsigned long)(a.wp + (long)par * a.wset_stride + ((long)xi * (a.CoutPad >> 5) + (n0 >> 5) + nh) * 2048);
        const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)w_);          // (the builtin returns int:
        const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w_ >> 32));  //  no sign extension)
        wfrag = reinterpret_cast<const char*>((unsigned long)lo_ | ((unsigned long)hi_ << 32));
    }
    const unsigned wofs = lane * 16;

    // V staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  Piece idx = tid + 512 u
    // (u < 8) -> LDS row (tid >> 2) + 128 u; a wave's instruction fills 1 KB = 16 rows, lane i at base + 16 i, so the LDS
    // image is lane-linear and the XOR swizzle is applied on the SOURCE side: physical piece (tid & 3) of a row holds its
    // logical piece (tid & 3) ^ ((row >> 2) & 3), and ((row >> 2) & 3) == ((tid >> 4) & 3) does not depend on u.
    // Padding rows read the zero page.  hipcc does not count asm memory operations, so EVERY wait of the tap loop is
    // written by hand (W16_WAIT_B, the wait in front of the chunk barrier).
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    const int* gq = gpos + (tid >> 2);
    const long vpiece = (long)((tid & 3) ^ ((tid >> 4) & 3)) * 16;
    const long vchunk = (long)4 * a.H * a.J * 64;  // bytes between the K chunks of one frame
#define W16_GLDS(src_, dst_)                                                                                         \
    {                                                                                                                \
        unsigned keep_;                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                                              \
    }
    // half HF (0 / 1) of a brick: pieces u = 4 HF .. 4 HF + 3 (two batches of four keep the address registers few)
#define W16_REQUEST_V(ch_, VB, HF)                                                                                   \
    {                                                                                                                \
        int gp_[4];                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) gp_[u] = gq[128 * (4 * (HF) + u)];                             \
        const char* vb_ = a.in + (long)(ch_) * vchunk + vpiece;                                                      \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                              \
            const char* s_ = gp_[u] >= 0 ? vb_ + (long)gp_[u] * 64 : a.zeros; \
            W16_GLDS(s_, vdst + (unsigned)((VB) * (W16_VROWS * 64) + (4 * (HF) + u) * 8192))                         \
        }                                                                                                            \
    }

    struct AOps { half8 ah[WM], al[WM]; };
    struct BOps { half8 bh, bl; };
    AOps a0, a1;
    int adn[WM];
    // ring of the B operands of R consecutive taps (requested R - 1 taps ahead).  R = 9 for the 9-tap kernel: no B request
    // younger than a chunk's V request is consumed before the chunk's barrier, so the V brick has the whole chunk to land.
    constexpr int R = (NT == 9 && W16_RING9) ? 9 : 6;
    BOps bq0, bq1, bq2, bq3, bq4, bq5, bq6, bq7, bq8;
    // A operands of tap TAP (compile-time) from V brick VB: LDS address of one row block, and its two ds_read_b128
#define W16_ADDR_A(TAP, VB, wm)                                                                                      \
    {                                                                                                                \
        const int r_ = arow[wm] + (((TAP) / 3) * HH + ((TAP) % 3)) * a.TJ + (VB) * W16_VROWS;                        \
        adn[wm] = (r_ << 6) + (((kg << 1) ^ ((r_ >> 2) & 3)) << 4);                                                  \
    }
#define W16_READ_A(o, wm)                                                                                            \
    {                                                                                                                \
        (o).ah[wm] = *reinterpret_cast<const half8*>(v_lds + adn[wm]);                                               \
        (o).al[wm] = *reinterpret_cast<const half8*>(v_lds + (adn[wm] ^ 16));                                        \
    }
#define W16_LOAD_A(o, TAP, VB)                                                                                       \
    {                                                                                                                \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) {                                                          \
            W16_ADDR_A(TAP, VB, wm)                                                                                  \
            W16_READ_A(o, wm)                                                                                        \
        }                                                                                                            \
    }
    // B operands of tap TAP of chunk CH (clamped to the last chunk: past the end the stream re-requests harmlessly)
#define W16_REQUEST_B(q, TAP, CH)                                                                                    \
    {                                                                                                                \
        const int c_ = (CH) < a.nchunk ? (CH) : a.nchunk - 1;                                                        \
        const char* p_ = wfrag + (long)(TAP) * wtap_stride + (long)c_ * cstride;                                     \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"((q).bh) : "v"(wofs), "s"(p_));                         \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"((q).bl) : "v"(wofs), "s"(p_));             \
    }
    // B operands of the current tap have landed when at most N younger loads are outstanding (loads return in order)
#define W16_WAIT_B(q, N) asm volatile("s_waitcnt vmcnt(%2)" : "+v"((q).bh), "+v"((q).bl) : "n"(N));
#define W16_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");

    // The 3 WM MFMAs of a tap with everything else the wave issues for the NEXT taps in the 32-cycle shadows between them, one
    // small piece per gap (see i2v_conv16w4.hip, where the per-tap timing behind this schedule was taken): per row block the
    // LDS address arithmetic and the two ds_read_b128 of the next tap's A operands, then the weight request of tap U + R - 1.
#define W16_MFMA_SPREAD(o, q, onxt, TAPN, VBN, QREQ, TAPR, CHR)                                                      \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 3 * WM; ++i) {                                                         \
            const int wm_ = i % WM, term_ = i / WM;                                                                  \
            acc[wm_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term_ == 2 ? (o).al[wm_] : (o).ah[wm_],                \
                                                              term_ == 1 ? (q).bl : (q).bh, acc[wm_], 0, 0, 0);      \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (i < 2 * WM && (i & 1) == 0) W16_ADDR_A(TAPN, VBN, i / 2)                                             \
            if (i < 2 * WM && (i & 1) == 1) W16_READ_A(onxt, i / 2)                                                  \
            if (i == 2 * WM) W16_REQUEST_B(QREQ, TAPR, CHR)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    }

    // prologue: the first R-1 weight requests do not need the index tables; everything requested here has landed before the
    // loop starts (the wait counts inside the loop assume the steady state and would under-wait in the first taps otherwise)
    W16_REQUEST_B(bq0, 0 % NT, 0 / NT)
    W16_REQUEST_B(bq1, 1 % NT, 1 / NT)
    W16_REQUEST_B(bq2, 2 % NT, 2 / NT)
    W16_REQUEST_B(bq3, 3 % NT, 3 / NT)
    W16_REQUEST_B(bq4, 4 % NT, 4 / NT)
    if constexpr (R == 9) {
        W16_REQUEST_B(bq5, 5 % NT, 5 / NT)
        W16_REQUEST_B(bq6, 6 % NT, 6 / NT)
        W16_REQUEST_B(bq7, 7 % NT, 7 / NT)
    }
    __syncthreads();  // tables
    W16_REQUEST_V(0, 0, 0)
    W16_REQUEST_V(0, 0, 1)
    if constexpr (R == 9) {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(bq0.bh), "+v"(bq0.bl), "+v"(bq1.bh), "+v"(bq1.bl), "+v"(bq2.bh), "+v"(bq2.bl), "+v"(bq3.bh),
                       "+v"(bq3.bl), "+v"(bq4.bh), "+v"(bq4.bl), "+v"(bq5.bh), "+v"(bq5.bl), "+v"(bq6.bh), "+v"(bq6.bl),
                       "+v"(bq7.bh), "+v"(bq7.bl)
                     :
                     : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(bq0.bh), "+v"(bq0.bl), "+v"(bq1.bh), "+v"(bq1.bl), "+v"(bq2.bh), "+v"(bq2.bl), "+v"(bq3.bh),
                       "+v"(bq3.bl), "+v"(bq4.bh), "+v"(bq4.bl)
                     :
                     : "memory");
    }
    __syncthreads();
    W16_LOAD_A(a0, 0, 0)

    // The loop body is a PAIR of chunks = 2 NT taps, numbered U = 0 .. 2 NT - 1 (a multiple of 6).  Tap U multiplies the A
    // operands in register set U & 1 (read from LDS during the previous tap) with the B operands in ring slot U % R
    // (requested R - 1 taps ago); meanwhile it reads the A operands of tap U + 1 and requests the B operands of tap U + R - 1.
    // The V brick is double-buffered: chunk c reads buffer c & 1; the next chunk's brick is requested (LDS-DMA) at the
    // chunk's first two taps and published by the chunk's ONE barrier, which sits in front of the last tap's MFMAs (the first
    // A read of the next chunk follows it).  Everything is compile-time and branch-free; after the last chunk the stream
    // re-requests harmlessly.
    // Program order of a tap: [V half-request (taps 0, 1)] [chunk barrier (last tap)] [wait for this tap's weights] [MFMAs,
    // between them the next tap's A operands and then the weight request of tap U + R - 1].  Wait counts (loads return in
    // order): younger than the weight request of tap U (issued in the middle of tap U - R + 1) are the weight requests of taps
    // U-R+2 .. U-1 (2 (R-2) loads) and the V half-requests (4 loads each) of every chunk's taps 0 and 1 among taps
    // U-R+2 .. U.  In front of the chunk barrier the brick requested at taps 0 and 1 must have landed: younger than its second
    // half are the weight requests of taps 1 .. NT-2.
    // Priority (s_setprio) falls with the tap index inside a chunk, so that whichever of the two waves of a SIMD is behind
    // gets the matrix pipe (at equal priority the older wave wins every arbitration and the younger one finishes the chunk alone).
#define W16_TAP(U, ACUR, ANXT, BCUR, BREQ)                                                                           \
    {                                                                                                                \
        constexpr int cp_ = (U) / NT, t_ = (U) % NT;          /* chunk of the pair, tap of the chunk */             \
        constexpr int un_ = (U) + R - 1, cn_ = un_ / NT, tn_ = un_ % NT;                                             \
        constexpr int ng_ = w16_count(t_, R - 1, NT, 0) + w16_count(t_, R - 1, NT, 1 % NT);   /* V half-requests among them */ \
        constexpr int nb_ = 2 * (R - 2) + 4 * ng_;                                                                   \
        if constexpr (t_ == 0 || w16_prio(t_, NT) != w16_prio(t_ - 1, NT)) __builtin_amdgcn_s_setprio(w16_prio(t_, NT)); \
        asm volatile("" : "+v"(arow[0]), "+v"(arow[1]), "+v"(arow[WM - 2]), "+v"(arow[WM - 1]));                     \
        if constexpr (t_ < 2)                                                                                        \
            W16_REQUEST_V(ch + cp_ + 1 < a.nchunk ? ch + cp_ + 1 : ch + cp_, 1 - cp_, t_)                            \
        if constexpr (t_ == NT - 1) {                                                                                \
            W16_WAIT_VM(2 * (NT - 2))                                                                                \
            __syncthreads();                                                                                         \
        }                                                                                                            \
        W16_WAIT_B(BCUR, nb_)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if constexpr (t_ < NT - 1) W16_MFMA_SPREAD(ACUR, BCUR, ANXT, t_ + 1, cp_, BREQ, tn_, ch + cn_)               \
        else W16_MFMA_SPREAD(ACUR, BCUR, ANXT, 0, 1 - cp_, BREQ, tn_, ch + cn_)                                      \
    }
#define W16_TAP6(U0)                                                                                                 \
    {                                                                                                                \
        W16_TAP((U0) + 0, a0, a1, bq0, bq5)                                                                          \
        W16_TAP((U0) + 1, a1, a0, bq1, bq0)                                                                          \
        W16_TAP((U0) + 2, a0, a1, bq2, bq1)                                                                          \
        W16_TAP((U0) + 3, a1, a0, bq3, bq2)                                                                          \
        W16_TAP((U0) + 4, a0, a1, bq4, bq3)                                                                          \
        W16_TAP((U0) + 5, a1, a0, bq5, bq4)                                                                          \
    }
#define W16_TAP18R9()                                                                                                \
    {                                                                                                                \
        W16_TAP(0, a0, a1, bq0, bq8)                                                                                 \
        W16_TAP(1, a1, a0, bq1, bq0)                                                                                 \
        W16_TAP(2, a0, a1, bq2, bq1)                                                                                 \
        W16_TAP(3, a1, a0, bq3, bq2)                                                                                 \
        W16_TAP(4, a0, a1, bq4, bq3)                                                                                 \
        W16_TAP(5, a1, a0, bq5, bq4)                                                                                 \
        W16_TAP(6, a0, a1, bq6, bq5)                                                                                 \
        W16_TAP(7, a1, a0, bq7, bq6)                                                                                 \
        W16_TAP(8, a0, a1, bq8, bq7)                                                                                 \
        W16_TAP(9, a1, a0, bq0, bq8)                                                                                 \
        W16_TAP(10, a0, a1, bq1, bq0)                                                                                \
        W16_TAP(11, a1, a0, bq2, bq1)                                                                                \
        W16_TAP(12, a0, a1, bq3, bq2)                                                                                \
        W16_TAP(13, a1, a0, bq4, bq3)                                                                                \
        W16_TAP(14, a0, a1, bq5, bq4)                                                                                \
        W16_TAP(15, a1, a0, bq6, bq5)                                                                                \
        W16_TAP(16, a0, a1, bq7, bq6)                                                                                \
        W16_TAP(17, a1, a0, bq8, bq7)                                                                                \
    }
    for (int ch = 0; ch < a.nchunk; ch += 2) {
        if constexpr (R == 9) {
            W16_TAP18R9()
        } else {
            W16_TAP6(0)
            if constexpr (NT >= 6) W16_TAP6(6)
            if constexpr (NT >= 9) W16_TAP6(12)
        }
    }

    __builtin_amdgcn_s_setprio(0);
    // The stream's harmless last requests (LDS-DMA included) must land before LDS and the ring's registers are reused: the
    // compiler does not know that the ring slots are still being written, so they stay operands of the wait.
    if constexpr (R == 9) {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(bq0.bh), "+v"(bq0.bl), "+v"(bq1.bh), "+v"(bq1.bl), "+v"(bq2.bh), "+v"(bq2.bl), "+v"(bq3.bh),
                       "+v"(bq3.bl), "+v"(bq4.bh), "+v"(bq4.bl), "+v"(bq5.bh), "+v"(bq5.bl), "+v"(bq6.bh), "+v"(bq6.bl),
                       "+v"(bq7.bh), "+v"(bq7.bl), "+v"(bq8.bh), "+v"(bq8.bl)
                     :
                     : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(bq0.bh), "+v"(bq0.bl), "+v"(bq1.bh), "+v"(bq1.bl), "+v"(bq2.bh), "+v"(bq2.bl), "+v"(bq3.bh),
                       "+v"(bq3.bl), "+v"(bq4.bh), "+v"(bq4.bl), "+v"(bq5.bh), "+v"(bq5.bl)
                     :
                     : "memory");
    }

    // ---- epilogue: the four partial GEMMs of a tile meet in LDS; y0 = M0 + M1 + M2, y1 = M1 - M2 - M3
    constexpr int NQ = BN / 4;            // float4 channel groups per tile
    constexpr int TPI = 512 / NQ;         // tiles per pass of the workgroup
    constexpr int NIT = W16_TILES / TPI;
    const int n4 = tid % NQ;
    const int n = n0 + 4 * n4;
    const bool ncol = n < a.Cout;
    // The residual rows are requested FIRST, all of them, so that their latency hides behind the LDS exchange: left inside
    // the store loop every load would wait for the previous store (`res` may alias `out` as far as the compiler knows;
    // vmcnt counts stores too), eight serialised memory round trips per workgroup.
    f32x4 rres[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            rres[it][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.res && ncol)
                rres[it][c] = *reinterpret_cast<const f32x4*>(a.res + (long)tres[2 * (tid / NQ + TPI * it) + c] * a.Cout + n);
        }
    __syncthreads();
    float* E = reinterpret_cast<float*>(smem);  // [4][128 tiles][BN channels]
#pragma unroll
    for (int wm = 0; wm < WM; ++wm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mh * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            E[(xi * W16_TILES + m) * BN + nh * 32 + l31] = acc[wm][r];
        }
    __syncthreads();
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && ncol) bias = *reinterpret_cast<const float4*>(a.bias + n);
    double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
    // all values first (in place of the residuals), then the stores in one predicated block: a per-iteration `if (ncol)`
    // makes the compiler re-synchronise vmcnt(0) -- i.e. wait for the previous stores -- at every join
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int tile = tid / NQ + TPI * it;
        const float4 m0 = *reinterpret_cast<const float4*>(E + (0 * W16_TILES + tile) * BN + 4 * n4);
        const float4 m1 = *reinterpret_cast<const float4*>(E + (1 * W16_TILES + tile) * BN + 4 * n4);
        const float4 m2 = *reinterpret_cast<const float4*>(E + (2 * W16_TILES + tile) * BN + 4 * n4);
        const float4 m3 = *reinterpret_cast<const float4*>(E + (3 * W16_TILES + tile) * BN + 4 * n4);
        const float y[2][4] = {{m0.x + m1.x + m2.x, m0.y + m1.y + m2.y, m0.z + m1.z + m2.z, m0.w + m1.w + m2.w},
                               {m1.x - m2.x - m3.x, m1.y - m2.y - m3.y, m1.z - m2.z - m3.z, m1.w - m2.w - m3.w}};
        const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = fmaf(y[c][j], a.oscale, bv[j]) + rres[it][c][j];
                if (ncol) {
                    ssum[j] += (double)v;
                    ssq[j] = fma((double)v, (double)v, ssq[j]);
                }
                if (a.epi & EPI_LRELU) v = v >= 0.f ? v : 0.2f * v;
                rres[it][c][j] = v;
            }
    }
    if (ncol) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const long p = tpos[tid / NQ + TPI * it];
#pragma unroll
            for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(a.out + (p + c) * a.Cout + n) = rres[it][c];
        }
    }
    if (a.stats) {
        // fused normalisation statistics.  The lanes of a wave that share (lane % NQ) hold the same four channels ->
        // wavefront shuffles; the eight waves' partials meet in LDS and ONE wave issues the workgroup's 2 * BN fp64 atomics
        // (2 instructions): all workgroups of a sample hit the same few lines of one L2 channel, and the atomic unit's cost
        // is per (instruction, line) -- per-wave atomics (512 line requests per workgroup) cost 1.2 ms per launch at
        // B = 8.  fp64 partials keep the totals independent of the tiling.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (NQ <= 8) { ssum[j] = wave_xor_add_f64<8>(ssum[j]); ssq[j] = wave_xor_add_f64<8>(ssq[j]); }
            ssum[j] = wave_xor_add_f64<16>(ssum[j]); ssq[j] = wave_xor_add_f64<16>(ssq[j]);
            ssum[j] = wave_xor_add_f64<32>(ssum[j]); ssq[j] = wave_xor_add_f64<32>(ssq[j]);
        }
        __syncthreads();  // E is free
        double* S = reinterpret_cast<double*>(smem);  // [8 waves][BN channels][2]
        if (lane < NQ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                S[(wave * BN + 4 * lane + j) * 2] = ssum[j];
                S[(wave * BN + 4 * lane + j) * 2 + 1] = ssq[j];
            }
        }
        __syncthreads();
        if (wave == 0 && lane < BN && n0 + lane < a.Cout) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                s0 += S[(w * BN + lane) * 2];
                s1 += S[(w * BN + lane) * 2 + 1];
            }
            double* dst = a.stats + ((long)b0 * a.Cout + n0 + lane) * 2;
            atomicAdd(dst, s0);
            atomicAdd(dst + 1, s1);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

// w3: [nset][Cout][Cin][KT][3][3] already multiplied by `scale`-free fp64 pre-sums; packs U = G g per (kt, kh).
static int wino_pack_sets(Wino16Weights& o, const std::vector<double>& w3, int nset, int cout, int cin, int kt) {
    o.Cin = cin; o.Cout = cout; o.KT = kt;
    o.CoutPad = (cout + 31) / 32 * 32;
    o.nchunk = cin / W16_KC;
    const int NT = kt * 3;
    std::vector<double> u((size_t)nset * cout * cin * NT * 4);
    double wmax = 0.0;
    for (size_t i = 0; i < (size_t)nset * cout * cin * NT; ++i) {
        const double g0 = w3[i * 3], g1 = w3[i * 3 + 1], g2 = w3[i * 3 + 2];
        double* d = &u[i * 4];
        d[0] = g0; d[1] = 0.5 * (g0 + g1 + g2); d[2] = 0.5 * (g0 - g1 + g2); d[3] = g2;
        for (int x = 0; x < 4; ++x) wmax = std::max(wmax, std::fabs(d[x]));
    }
    o.wexp = 0;
    if (wmax > 0.0 && std::isfinite(wmax)) o.wexp = std::max(-40, std::min(40, (int)std::floor(std::log2(16384.0 / wmax))));
    const double pre = std::ldexp(1.0, o.wexp);
    const size_t set_halfs = (size_t)NT * o.nchunk * 4 * o.CoutPad * 32;
    std::vector<_Float16> p((size_t)nset * set_halfs, (_Float16)0.f);
    for (int s = 0; s < nset; ++s)
        for (int n = 0; n < cout; ++n)
            for (int c = 0; c < cin; ++c)
                for (int tap = 0; tap < NT; ++tap)
                    for (int x = 0; x < 4; ++x) {
                        const float v = (float)(u[((((size_t)s * cout + n) * cin + c) * NT + tap) * 4 + x] * pre);
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        // fragment-major: [tap][chunk][x][32-channel block][hi | lo][lane = kg * 32 + n % 32][8 halfs]
                        const int chunk = c / W16_KC, kgq = (c % W16_KC) / 8, j = c % 8;
                        _Float16* blk = &p[s * set_halfs + ((((size_t)tap * o.nchunk + chunk) * 4 + x) * (o.CoutPad / 32) + n / 32) * 1024];
                        blk[(kgq * 32 + n % 32) * 8 + j] = hi;
                        blk[512 + (kgq * 32 + n % 32) * 8 + j] = lo;
                    }
    o.set_bytes = (long)set_halfs * 2;
    return o.w.upload(p.data(), p.size() * 2);
}

// Brick of 128 Winograd tiles = TT frames x TH rows x TJ = 4 output pairs (the only brick width every shipped and tested
// shape uses: an MFMA row block is then 8 consecutive rows of one frame).  False when [T, H, W] with a KT-tap temporal
// kernel cannot be tiled that way or its halo brick does not fit the staged rows -- the caller then runs the direct kernel.
static bool wino16_tiling(int T, int H, int W, int KT, int* TT_, int* TH_, int* TJ_) {
    if (T < 1 || W % 8 || H < 8) return false;
    const int J = W / 2, TJ = 4;
    int TT = 1;
    while (TT < 4 && T % (TT * 2) == 0) TT *= 2;
    const int TH = W16_TILES / (TT * TJ);
    if (TH > H || H % TH || J % TJ || TH * TJ % 32) return false;
    if (4 * (TT + KT - 1) * (TH + 2) * TJ > W16_VROWS) return false;   // halo brick, 64-byte rows
    *TT_ = TT; *TH_ = TH; *TJ_ = TJ;
    return true;
}

bool wino16_supported(int cout, int cin, int T, int H, int W, int KT) {
    if (cout % 32 || cin % (2 * W16_KC)) return false;  // (the kernel's loop body is a chunk pair)
    int TT, TH, TJ;
    return wino16_tiling(T, H, W, KT, &TT, &TH, &TJ);
}

int Wino16Weights::pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, double scale) {
    I2V_REQUIRE(kt == 3 || kt == 1, I2V_E_INVALID, "wino16: temporal kernel size %d", kt);
    tdup = false;
    std::vector<double> w3((size_t)cout * cin * kt * 9);
    for (size_t i = 0; i < w3.size(); ++i) w3[i] = (double)w_src[i] * scale;
    int rc = wino_pack_sets(*this, w3, 1, cout, cin, kt);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

int Wino16Weights::pack_tdup(const float* w_src, const float* bias_src, int cout, int cin, double scale) {
    // parity 0 = (W[0], W[1]+W[2]), parity 1 = (W[0]+W[1], W[2]) along time (see Conv16Weights::pack_tdup)
    std::vector<double> w3((size_t)2 * cout * cin * 18);
    for (int par = 0; par < 2; ++par)
        for (size_t nc = 0; nc < (size_t)cout * cin; ++nc)
            for (int hw = 0; hw < 9; ++hw) {
                const double w0 = w_src[nc * 27 + hw], w1 = w_src[nc * 27 + 9 + hw], w2 = w_src[nc * 27 + 18 + hw];
                double* dst = &w3[((size_t)par * cout * cin + nc) * 18];
                dst[hw] = (par == 0 ? w0 : w0 + w1) * scale;
                dst[9 + hw] = (par == 0 ? w1 + w2 : w2) * scale;
            }
    tdup = true;
    int rc = wino_pack_sets(*this, w3, 2, cout, cin, 2);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

template <int NT, int BN>
static int launch_wino(const WinoArgs& a, unsigned nblk, size_t lds, hipStream_t st) {
    auto kern = conv_wino_f16x3_kernel<NT, BN>;
    static bool attr_set[I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.tdup ? 2 * nblk : nblk), dim3(512), lds, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int wino16_forward(const Wino16Weights& wts, const void* v_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                   int H, int W, int epi, hipStream_t st, double* stats) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "wino16: weights not packed");
    I2V_REQUIRE((epi & ~EPI_LRELU) == 0, I2V_E_INVALID, "wino16: unsupported epilogue %d", epi);
    WinoArgs a{};
    if (int rc0 = zero_page(&a.zeros)) return rc0;
    a.in = static_cast<const char*>(v_hl16); a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.stats = stats;
    a.B = B; a.H = H; a.W = W; a.J = W / 2; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.tdup = wts.tdup ? 1 : 0;
    a.wset_stride = wts.set_bytes;
    if (wts.tdup) {  // T is the OUTPUT frame count; the half-rate input has T / 2 frames
        I2V_REQUIRE(T % 2 == 0 && !res, I2V_E_INVALID, "wino16: temporal-duplication mode needs an even frame count and no residual");
        T /= 2;
    }
    a.T = T;
    I2V_REQUIRE(wino16_supported(wts.Cout, wts.Cin, T, H, W, wts.KT), I2V_E_INVALID, "wino16: unsupported shape [%d,%d,%d] %d -> %d (kt = %d)",
                T, H, W, wts.Cin, wts.Cout, wts.KT);
    a.rt = res ? rt : 1; a.rs = res ? rs : 1; a.epi = epi;
    a.oscale = (float)std::ldexp(1.0, -wts.wexp);
    const int J = W / 2, KT = wts.KT;
    int TT = 1, TH = 1, TJ = 4;
    (void)wino16_tiling(T, H, W, KT, &TT, &TH, &TJ);
    a.TT = TT; a.TH = TH; a.TJ = TJ; a.nbT = T / TT; a.nbH = H / TH; a.nbJ = J / TJ;
    a.wofs = 0;
    const int BN = a.CoutPad % 64 == 0 ? 64 : 32;  // output channels per workgroup
    const int body = std::max(2 * W16_VROWS * 64, 4 * W16_TILES * BN * 4);  // two V bricks; the epilogue's exchange buffer
    a.tofs = body;
    const size_t lds = (size_t)body + (size_t)W16_VROWS * 4 + W16_TILES * 12;
    I2V_REQUIRE(lds <= 160 * 1024, I2V_E_INVALID, "wino16: LDS %zu bytes", lds);
    I2V_REQUIRE(!stats || (long)TT * TH * TJ <= (long)T * H * J, I2V_E_INVALID, "wino16: fused statistics need bricks inside one sample");
    const long nblk = (long)B * a.nbT * a.nbH * a.nbJ * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 30), I2V_E_INVALID, "wino16: grid of %ld workgroups", nblk);
    I2V_REQUIRE((long)B * T * a.nchunk * 4 * H * J < (1L << 31) && (long)B * (wts.tdup ? 2 * T : T) * H * W < (1L << 31), I2V_E_INVALID,
                "wino16: batch %d too large for the 32-bit row indices of this kernel ([%d,%d,%d] x %d chunks)", B, T, H, W, a.nchunk);
    if (BN == 64) {
        if (KT == 3) return launch_wino<9, 64>(a, (unsigned)nblk, lds, st);
        if (KT == 2) return launch_wino<6, 64>(a, (unsigned)nblk, lds, st);
        return launch_wino<3, 64>(a, (unsigned)nblk, lds, st);
    }
    if (KT == 3) return launch_wino<9, 32>(a, (unsigned)nblk, lds, st);
    if (KT == 2) return launch_wino<6, 32>(a, (unsigned)nblk, lds, st);
    return launch_wino<3, 32>(a, (unsigned)nblk, lds, st);
}

}  // namespace i2v

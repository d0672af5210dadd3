// Device side of the Winograd F(4,3) split-fp16 convolution (see i2v_conv16w4.hip for the design): argument blocks, the tap-loop
// pass (w4_pass), the brick decode, the index tables and the kernel template -- shared by i2v_conv16w4.hip (the kernels that read
// the operand V a producer launch wrote) and i2v_conv16w4g.hip (the kernel that generates it in its own producer waves).
// A translation unit that defines W4_NO_INSTRUMENT before including this file gets no measurement hooks (timeline stamps,
// per-tap timing): their device-side buffers live in i2v_conv16w4.hip only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "i2v_conv.h"

#ifdef W4_NO_INSTRUMENT
#undef W4_TIMELINE
#undef W4_TAPTIME
#endif

namespace i2v {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef W4_TIMELINE   // measurement builds (tools/conv16w_check): per-workgroup phase stamps, 100 MHz wall clock
__device__ unsigned long long w4_tl[8192 * 16];
#define W4_STAMP(i) { if (tid == 0 && w4_tlv_ < 8192) w4_tl[w4_tlv_ * 16 + (i)] = wall_clock64(); }   // w4_tlv_: the virtual workgroup
#else
#define W4_STAMP(i) {}
#endif
#ifdef W4_ABLATE_AL   // measurement builds only (wrong results): the lo halves of the A operands are not read from LDS
#define W4_ABL_AL(x, y) y
#else
#define W4_ABL_AL(x, y) x
#endif
#ifdef W4_ABLATE_BL   // measurement builds only (wrong results): the lo weight fragment is loaded from the hi fragment's lines
#define W4_ABL_BL(x, y) y
#else
#define W4_ABL_BL(x, y) x
#endif
#ifdef W4_TAPTIME   // measurement builds: time between the starts of consecutive taps, per wave and tap slot (s_memtime, 100 MHz)
__device__ unsigned long long w4_tt[2 * 8 * 18 * 2];   // [pass][wave][tap slot]{ticks, count}
#define W4_TT(U)                                                                                                     \
    {                                                                                                                \
        const unsigned long long now_ = __builtin_readcyclecounter();                                                \
        if (lane == 0) { tt_lds[(U)] += (unsigned)(now_ - tt_prev); tt_cnt[(U)] += 1; }                              \
        tt_prev = now_;                                                                                              \
    }
#else
#define W4_TT(U)
#endif
// Structure of the kernel (I2V_W4_PIPE, measurement build only -- see w4_switches): 0 one workgroup per brick (round 3's structure, the default); 1 the software-pipelined
// persistent kernel; 2 its "lite" form (see the kernel's comment).  All three give the same bits and pass the static checks.
// Measured (profiles/r04_b_*, r04_l_*): 1 removes 1.9 us of prologue and 1.0 us of pass A per workgroup of the 128 -> 128 layer
// and pays 0.7 us (tables under pass B's prologue), 1.5 us (pass B with the extra loads) and 1.4 us (four-quarter epilogue):
// 81.9 -> 82.7 us.  2 gains 2.3 % there (82.3 -> 80.4 us) and loses 11 % on 64 -> 64, where both tap loops slow down: the 256
// persistent workgroups run the same phase at the same time, which the hardware dispatcher of the default kernel avoids.
// Whole steps +-1 % either way, so the default stays 0.
constexpr int W4_DEFAULT_PIPE = 0;
constexpr int W4_DEFAULT_ORDER = 2;   // brick -> XCD order (kernel comment); I2V_W4_ORDER overrides for A/B runs
// Cache-policy experiments (measurement builds, tools/build_measurement_libs.sh nt): -DW4_V_NT marks the V stream (LDS-DMA loads)
// non-temporal so that it does not turn the weight fragments out of the 4 MB L2; -DW4_OUT_NT stores the output non-temporally.
#ifdef W4_V_NT
#define W4_V_POLICY " nt"
#else
#define W4_V_POLICY ""
#endif
constexpr int W4_TILES = 128;   // tiles (of four output positions) per workgroup of the 512-thread kernels
constexpr int W4_KC = 16;       // input channels per K chunk
constexpr int W4_ROWS_A = 1024; // staged V rows per buffer, pass A (4 planes); pass B stages 512 (2 planes)  [512-thread kernels]

// Workgroup geometry.  NTH = 512: the kernel as described above (8 waves, 128 tiles, one workgroup per CU: 138 KB of LDS).
// NTH = 256 (round 5, 32-channel layers only): 4 waves, 64 tiles (4 frames x 4 rows x 16 positions), TWO workgroups per CU (2 x 78 KB).
// A 32-channel workgroup of the 512-thread kernel is two independent 64-tile halves that share nothing but the V brick (pass A:
// wave = (plane, tile half) with two row blocks per weight fragment; pass B: (plane, tile quarter) with one): splitting it into two
// workgroups keeps every wave's loop exactly as it was -- same fragments per MFMA, same accumulation order, same bits -- and lets
// the CU run one workgroup's tables / first brick / hand-over / epilogue (10 of 22 us per 128 tiles on the 64 -> 32 layer at 128 x 128,
// matrix pipe idle) underneath the other one's tap loops, which one workgroup per CU cannot do and the software-pipelined
// persistent variants did not manage to do by hand.  The halo brick of 64 tiles has 4 x 144 rows per chunk (36 KB) instead of
// 4 x 240: 1.2 x the V bytes through L2 per tile.
// A workgroup load instruction (global_load_lds_dwordx4, one per thread) stages NTH / 4 rows of 64 bytes; a chunk's brick is
// requested in two half-requests of V0 and V1 such instructions.
template <int NTH> struct W4Geo;
template <> struct W4Geo<512> { static constexpr int TILES = 128, ROWS_A = 1024, ROWS_B = 512, VA0 = 4, VA1 = 4, VB0 = 2, VB1 = 2; };
template <> struct W4Geo<256> { static constexpr int TILES = 64, ROWS_A = 576, ROWS_B = 320, VA0 = 5, VA1 = 4, VB0 = 3, VB1 = 2; };
template <int NTH> constexpr int w4_table_bytes() { return (2 * W4Geo<NTH>::ROWS_A + 5 * W4Geo<NTH>::TILES) * 4; }
// Which tile of its 32-tile row block an MFMA row (= A-operand lane l31) holds.  512-thread kernels: tile = row (32 consecutive
// tiles = 32 consecutive V rows, which the XOR key (row >> 2) & 3 spreads over the 16-byte slots without conflicts for the four
// 16-lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} ... of ds_read_b128).  The 64-tile brick is 4 frames x 4 rows x 4 tiles: a row
// block is two 16-row runs 24 rows apart (keys k .. k+3 and k+2 .. k+5), and with tile = row both lane groups would read two
// pairs of rows with equal keys (2-way conflicts: 41 % conflict cycles when round 3 tried this brick).  So the eight lane quads
// take the tile quads 0 2 3 1 6 4 5 7: group {0,3,5,6} -> tile quads {0,1,4,5} = keys {k,k+1,k+2,k+3}, group {1,2,4,7} -> {2,3,6,7} =
// keys {k+2,k+3,k,k+1}.  Only the row -> tile labels move (arow here, the accumulator scatter in the epilogue): no loop changes,
// and no output's accumulation order either.
// The accumulator scatter that goes with it (256-thread kernels).  Register r of lane (l31, kg) is MFMA row (r & 3) + 8 (r >> 2) + 4 kg =
// lane quad j = 2 (r >> 2) + kg, i.e. tile quad tq = {0 2 3 1 6 4 5 7}[j], tile 4 tq + (r & 3), kept in E row tile ^ (tq & 1):
//   r >> 2 = 0: tq = 0 | 2 -> row (r & 3)           + 8 kg        r >> 2 = 1: tq = 3 | 1 -> row 12 + ((r & 3) ^ 1) - 8 kg
//   r >> 2 = 2: tq = 6 | 4 -> row 24 + (r & 3)      - 8 kg        r >> 2 = 3: tq = 5 | 7 -> row 20 + ((r & 3) ^ 1) + 8 kg
// w4_escatter: the row for kg = 0 (compile-time in r); w4_escatter_up: whether the upper lanes sit 8 rows above (else below).
__device__ __forceinline__ constexpr int w4_escatter(int base, int r) {
    return base + ((r >> 2) == 0 ? 0 : (r >> 2) == 1 ? 12 : (r >> 2) == 2 ? 24 : 20) + ((r & 3) ^ ((r >> 2) & 1));
}
__device__ __forceinline__ constexpr bool w4_escatter_up(int r) { return (r >> 2) == 0 || (r >> 2) == 3; }
template <int NTH> __device__ __forceinline__ int w4_row_tile(int l31) {
    if constexpr (NTH == 256) return 4 * ((0x75461320u >> (4 * (l31 >> 2))) & 7) + (l31 & 3);
    else return l31;
}

struct W4Args {
    const char* in;     // V: hl16 [B][T][Cin/16][6][H][J][64 B], J = W / 4
    const char* zeros;  // >= 64 zero bytes
    const char* wp;     // U: [parity][tap][chunk][6][CoutPad/32][hi | lo][64 lanes][16 B]
    const float* bias;
    const float* res;
    float* out;         // fp32 channels-last [B][To][H][W][Cout]
    double* stats;
    int B, T, H, W, J, Cin, Cout, CoutPad, nchunk;   // T = frames of the INPUT tensor
    int tdup;
    long wset_stride;
    int TT, TH, TJ, nbT, nbH, nbJ;
    int th_shift, rt_shift, rs_shift, hh_magic;   // TH, rt, rs are powers of two; hh_magic = ceil(2^20 / (TH + 2)): n / HH == n * hh_magic >> 20 for n * HH < 2^20
    int rt, rs, epi;
    float oscale;
    int tofs;           // LDS byte offset of the index tables
    int order;          // brick -> XCD order (see w4_decode): 0 round-robin over the flat brick index, 1 / 2 one w-column per XCD
    int skew;           // persistent kernels: start delay of workgroup i in units of ~5 us x ((i >> 3) & 3) (measurement, I2V_W4_SKEW)
    int nvirt;          // virtual workgroups = bricks x channel tiles x frame parities (PIPE: looped over by gridDim.x workgroups)
};

// Wave priority inside a chunk.  The two waves of a SIMD share the matrix pipe, arbitrated by priority, then age: at equal
// priority the older wave (0..3) runs its taps at full speed, the younger one gets the leftover slots, falls ~4 taps behind per
// chunk, finishes the chunk alone at half the pipe rate while the older one waits ~2000 cycles at the chunk barrier
// (per-tap timing, -DW4_TAPTIME).  A priority that FALLS with the tap index hands the pipe to whichever wave is behind.
#ifndef W4_PRIO
#define W4_PRIO 1
#endif
constexpr int w4_prio(int t, int NT) {
    return NT == 3 ? 3 - t : (t < 6 ? 3 - t / 2 : 0);   // 9 taps: 3 3 2 2 1 1 0 0 0
}
constexpr int w4_count(int t, int R, int NT, int h) {
    int n = 0;
    for (int k = 0; k < R; ++k) n += ((t - k - h) % NT + NT) % NT == 0;
    return n;
}

// One pass of the K loop over NPL = VH planes... (VH = 16-byte V pieces per thread and half-request: 4 -> 1024 staged rows =
// four planes, 2 -> 512 rows = two planes).  WM = MFMA row blocks of this wave.  arow[wm]: LDS row of the lane's tile (tap
// (0,0)) inside the pass's brick; gpos: global V row (chunk 0) of every staged row, -1 = zero padding; wlane: this wave's
// weight fragments (tap 0, chunk 0).
// gposN / PRE: hand-over between the passes.  The request a chunk issues for "the next chunk" is a harmless repeat behind the
// LAST chunk; pass A instead requests chunk 0 of pass B's brick there (table gposN: its rows in front, -1 behind), into the
// buffer pass B reads first, so that pass B (PRE = true) starts without a V round trip.
// PIPE (software-pipelined persistent kernel): the pass's two V buffers start at the LDS rows rb0 / rb1 (they alternate
// between the two 64 KB regions from brick to brick), chunk 0 of pass A's brick is already in LDS when the pass starts (it was
// requested during the PREVIOUS brick's pass B), `between` runs between the prologue's weight requests and their wait (the next
// brick's index tables are built there), and pass B carries the next brick's first V brick as two extra LDS-DMA loads per
// half-request: in its first two chunks they are the eight 16-byte pieces per thread of that brick (table nq, destination
// ndst = the region this brick's pass B does not use), behind them -- and when there is no next brick -- zero-page reads into a
// 1 KB dump row, so that the loop body and its wait counts stay the same for every chunk.
struct W4Next {
    const int* nq;      // next brick's gposA table + (tid >> 2), or the current one when there is no next brick
    unsigned ndst;      // LDS byte address of the wave's slice of the free region
    unsigned dump;      // LDS byte address of the dump row
    int valid;          // there is a next brick
};

// BUF (round 5; the one-brick-per-workgroup kernels): the V requests go through a raw buffer descriptor of the brick's SAMPLE
// (`vrsrc`: base = the sample's V tensor, num_records = its bytes < 2^31) instead of 64-bit addresses: the table holds the row index
// inside the sample (padding rows: W4_PAD_ROW = 2^25, i.e. byte offset 2^31, out of range under any reading of the range check, and
// an out-of-range lane of `buffer_load ... lds` writes ZEROS into LDS: tools/bufload_lds_test), the lane's offset is ONE
// v_lshl_or_b32, the chunk goes into the scalar offset -- 5 instead of 13 instructions per load in front of the first MFMAs of
// taps 0 / 1, where profiles/r05_n_f43_inloop_idle_analysis.md finds most of the tap loops' idle cycles.  Same loads, same order,
// same wait counts.
constexpr int W4_PAD_ROW = 1 << 25;
template <int NT, int WM, int VH0, int VH1, int NTH, bool PRE, bool PREL, int VXP, bool BUF, class Between>
__device__ __forceinline__ void w4_pass(const W4Args& a, char* smem, const int* gpos, const int* gposN, f32x16 (&acc)[WM],
                                        int (&arow)[WM], const char* wfrag, int HH, int tid, int lane, int wave, int rb0, int rb1,
                                        const W4Next& nxt, Between&& between, int w4_tlv_, __amdgpu_buffer_rsrc_t vrsrc) {
    constexpr int RPL = NTH / 4;          // V rows staged by one load instruction of the workgroup (4 threads per 64-byte row)
    // GEN (VH0 = VH1 = 0): this pass requests no V at all -- the producer waves of conv_wino4g_f16x3_kernel generate every chunk's
    // brick into the buffer the chunk barrier publishes (same buffers, same barriers); the wait counts then hold only weight loads
    constexpr bool GEN = VH0 == 0 && VH1 == 0;
    static_assert(VXP == 0 || (VH0 == VH1 && NTH == 512), "PIPE needs the 512-thread geometry");
    constexpr int VX = PRE ? VXP : 0;   // extra LDS-DMA loads per half-request (the next brick's first V brick; PREL: this brick's was preloaded)
    const int kg = lane >> 5;
    char* v_lds = smem;
    const long cstride = (long)a.CoutPad * 384;          // bytes per (tap, chunk): 6 planes x CoutPad x 64
    const long wtap_stride = (long)a.nchunk * cstride;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    const unsigned voff0 = __builtin_amdgcn_readfirstlane(rb0 * 64), voff1 = __builtin_amdgcn_readfirstlane(rb1 * 64);   // byte offsets of the two V buffers
    const int* gq = gpos + (tid >> 2);
    const int* gqn = gposN + (tid >> 2);
    const long vpiece = (long)((tid & 3) ^ ((tid >> 4) & 3)) * 16;
    const long vchunk = (long)6 * a.H * a.J * 64;        // bytes between the K chunks of one frame
    const unsigned vpiece32 = (unsigned)vpiece;
    const unsigned vchunk32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)vchunk);
    const unsigned wofs = lane * 16;                     // the lane's piece of a weight fragment (the rest of the address is scalar)
    {   // the fragment base goes into the loads' scalar address operand: make its uniformity explicit
        const unsigned long w_ = (unsigned long)wfrag;
        const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)w_);          // (the builtin returns int:
        const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w_ >> 32));  //  no sign extension)
        wfrag = reinterpret_cast<const char*>((unsigned long)lo_ | ((unsigned long)hi_ << 32));
    }
#define W4_GLDS(src_, dst_)                                                                                          \
    {                                                                                                                \
        unsigned keep_;                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" W4_V_POLICY "\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                                              \
    }
#define W4_BLDS(off_, soff_, dst_)                                                                                   \
    {                                                                                                                \
        unsigned keep_;                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds" W4_V_POLICY "\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(off_), "s"(vrsrc), "s"(dst_), "s"(soff_) : "memory");                      \
    }
#define W4_REQUEST_V(ch_, VB, HF)                                                                                    \
    if constexpr (!GEN) {                                                                                            \
        const bool nx_ = (ch_) >= a.nchunk;              /* behind the last chunk: the hand-over table, chunk 0 */    \
        const int* gt_ = nx_ ? gqn : gq;                                                                             \
        constexpr int nv_ = (HF) ? VH1 : VH0, v0_ = (HF) ? VH0 : 0;   /* this half-request's load instructions */     \
        int gp_[nv_];                                                                                                \
        _Pragma("unroll") for (int u = 0; u < nv_; ++u) gp_[u] = gt_[RPL * (v0_ + u)];                               \
        if constexpr (BUF) {                                                                                         \
            const unsigned so_ = (unsigned)(nx_ ? 0 : (ch_)) * vchunk32;                                             \
            _Pragma("unroll") for (int u = 0; u < nv_; ++u) {                                                        \
                const unsigned o_ = ((unsigned)gp_[u] << 6) | vpiece32;                                              \
                W4_BLDS(o_, so_, vdst + ((VB) ? voff1 : voff0) + (unsigned)((v0_ + u) * (NTH * 16)))                  \
            }                                                                                                        \
        } else {                                                                                                     \
        const char* vb_ = a.in + (long)(nx_ ? 0 : (ch_)) * vchunk + vpiece;                                          \
        _Pragma("unroll") for (int u = 0; u < nv_; ++u) {                                                            \
            const char* s_ = gp_[u] >= 0 ? vb_ + (long)gp_[u] * 64 : a.zeros;                     \
            W4_GLDS(s_, vdst + ((VB) ? voff1 : voff0) + (unsigned)((v0_ + u) * (NTH * 16)))                           \
        }                                                                                                            \
        }                                                                                                            \
        if constexpr (VX > 0) {   /* the next brick's first V brick, pieces (chunk parity, half, u); real in chunks 0, 1 */ \
            constexpr int pc_ = ((1 - (VB)) * 2 + (HF)) * 2;   /* (the requesting chunk's parity is 1 - VB) */         \
            const bool real_ = nxt.valid && (ch_) <= 2;        /* requested by chunks 0 and 1 */                       \
            int gx_[2];                                                                                              \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) gx_[u] = nxt.nq[128 * (pc_ + u)];                          \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                          \
                const char* s_ = real_ && gx_[u] >= 0 ? a.in + vpiece + (long)gx_[u] * 64 : a.zeros;                 \
                const unsigned d_ = real_ ? nxt.ndst + (unsigned)((pc_ + u) * 8192) : nxt.dump;                      \
                W4_GLDS(s_, d_)                                                                                      \
            }                                                                                                        \
        }                                                                                                            \
    }
    struct AOps { half8 ah[WM], al[WM]; };
    struct BOps { half8 bh, bl; };
    AOps a0, a1;
    int adn[WM];
    constexpr int R = NT == 9 ? 9 : 6;
    // tap at whose start the second half of the next chunk's V brick is requested (the first half: tap 0).  A 3-tap chunk
    // (SPADE's 2-D convs) requests both at tap 0: the brick then has two taps instead of one to arrive before the chunk barrier.
    constexpr int VT1 = NT == 3 ? 0 : 1;
    BOps bq0, bq1, bq2, bq3, bq4, bq5, bq6, bq7, bq8;
    /* LDS address of one row block of the A operands, and its two ds_read_b128 */
#define W4_ADDR_A(TAP, VB, wm)                                                                                       \
    {                                                                                                                \
        const int r_ = arow[wm] + (((TAP) / 3) * HH + ((TAP) % 3)) * 4 + ((VB) ? rb1 : rb0);                       \
        adn[wm] = (r_ << 6) + (((kg << 1) ^ ((r_ >> 2) & 3)) << 4);                                                  \
    }
#define W4_READ_A(o, wm)                                                                                             \
    {                                                                                                                \
        (o).ah[wm] = *reinterpret_cast<const half8*>(v_lds + adn[wm]);                                               \
        W4_ABL_AL((o).al[wm] = *reinterpret_cast<const half8*>(v_lds + (adn[wm] ^ 16)), (o).al[wm] = (o).ah[wm]);    \
    }
#define W4_LOAD_A(o, TAP, VB)                                                                                        \
    {                                                                                                                \
        _Pragma("unroll") for (int wm = 0; wm < WM; ++wm) {                                                          \
            W4_ADDR_A(TAP, VB, wm)                                                                                   \
            W4_READ_A(o, wm)                                                                                         \
        }                                                                                                            \
    }
    /* weight fragments of one tap: scalar base + the lane's 16 bytes */                                            \
#define W4_REQUEST_B(q, TAP, CH)                                                                                     \
    {                                                                                                                \
        const int c_ = (CH) < a.nchunk ? (CH) : a.nchunk - 1;                                                        \
        const char* p_ = wfrag + (long)(TAP) * wtap_stride + (long)c_ * cstride;                                     \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"((q).bh) : "v"(wofs), "s"(p_));                         \
        W4_ABL_BL(asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"((q).bl) : "v"(wofs), "s"(p_)),   \
                  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"((q).bl) : "v"(wofs), "s"(p_)));              \
    }
#define W4_WAIT_B(q, N) asm volatile("s_waitcnt vmcnt(%2)" : "+v"((q).bh), "+v"((q).bl) : "n"(N));
#define W4_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    // The 3 WM MFMAs of a tap with everything else this wave has to issue for the NEXT taps in the 32-cycle shadows between
    // them, one small piece per gap: an in-order wave that issues its MFMAs back to back sits blocked on the pipe, and whatever
    // it issues outside the MFMA block is time the pipe idles unless the partner wave happens to have an MFMA ready (per-tap
    // timing: a wave running alone reached 54 % of the pipe, the pair 66-76 %).  Pieces: per row block the LDS address
    // arithmetic and the two ds_read_b128 of the next tap's A operands, then the weight request of tap U + R - 1.
#define W4_MFMA_SPREAD(o, q, onxt, TAPN, VBN, QREQ, TAPR, CHR)                                                       \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 3 * WM; ++i) {                                                         \
            const int wm_ = i % WM, term_ = i / WM;                                                                  \
            acc[wm_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term_ == 2 ? (o).al[wm_] : (o).ah[wm_],                \
                                                              term_ == 1 ? (q).bl : (q).bh, acc[wm_], 0, 0, 0);      \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (WM >= 2) {                                                                                           \
                if (i < 2 * WM && (i & 1) == 0) W4_ADDR_A(TAPN, VBN, i / 2)                                          \
                if (i < 2 * WM && (i & 1) == 1) W4_READ_A(onxt, i / 2)                                               \
                if (i == 2 * WM) W4_REQUEST_B(QREQ, TAPR, CHR)                                 \
            } else {                                                                                                 \
                if (i == 0) W4_ADDR_A(TAPN, VBN, 0)                                                                  \
                if (i == 1) W4_READ_A(onxt, 0)                                                                       \
                if (i == 2) W4_REQUEST_B(QREQ, TAPR, CHR)                                      \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    }

    // prologue: the first R-1 weight requests do not need the index tables; everything requested here has landed before the
    // loop starts (the wait counts inside the loop assume the steady state and would under-wait in the first taps otherwise)
    W4_REQUEST_B(bq0, 0 % NT, 0 / NT)
    W4_REQUEST_B(bq1, 1 % NT, 1 / NT)
    W4_REQUEST_B(bq2, 2 % NT, 2 / NT)
    W4_REQUEST_B(bq3, 3 % NT, 3 / NT)
    W4_REQUEST_B(bq4, 4 % NT, 4 / NT)
    if constexpr (R == 9) {
        W4_REQUEST_B(bq5, 5 % NT, 5 / NT)
        W4_REQUEST_B(bq6, 6 % NT, 6 / NT)
        W4_REQUEST_B(bq7, 7 % NT, 7 / NT)
    }
    if constexpr (!PRE && !PREL && !GEN) {
        __syncthreads();  // tables written
        W4_REQUEST_V(0, 0, 0)
        W4_REQUEST_V(0, 0, 1)
    }
    between();
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
    W4_STAMP(PRE ? 3 : 1)
    W4_LOAD_A(a0, 0, 0)
#ifdef W4_TAPTIME
    unsigned* tt_lds = reinterpret_cast<unsigned*>(smem + 150 * 1024) + ((PRE ? 8 : 0) + wave) * 36;
    unsigned* tt_cnt = tt_lds + 18;
    if (lane < 36) tt_lds[lane] = 0;
    unsigned long long tt_prev = __builtin_readcyclecounter();
#endif

    // Tap U of a chunk pair.  Program order of a tap: [V half-request of the next chunk (taps 0, 1)] [chunk barrier (last tap)]
    // [wait for this tap's weights] [MFMAs, between them: next tap's A operands, then the weight request of tap U + R - 1].
    // Younger than the weight request of tap U (issued in the middle of tap U - R + 1): the weight requests of taps
    // U-R+2 .. U-1 (2 (R-2) loads) and the V half-requests (VH loads each) of every chunk's taps 0 and 1 among taps
    // U-R+2 .. U.  At a chunk's last tap the next chunk's brick must have landed: younger than its second half-request (start
    // of tap VT1) are the weight requests of taps VT1 .. NT-2.
#define W4_TAP(U, ACUR, ANXT, BCUR, BREQ)                                                                            \
    {                                                                                                                \
        constexpr int cp_ = (U) / NT, t_ = (U) % NT;                                                                 \
        constexpr int un_ = (U) + R - 1, cn_ = un_ / NT, tn_ = un_ % NT;                                             \
        constexpr int nb_ = 2 * (R - 2) + (VH0 + VX) * w4_count(t_, R - 1, NT, 0) + (VH1 + VX) * w4_count(t_, R - 1, NT, VT1);  \
        W4_TT(U)                                                                                                     \
        if constexpr (W4_PRIO && (t_ == 0 || w4_prio(t_, NT) != w4_prio(t_ - 1, NT)))                                \
            __builtin_amdgcn_s_setprio(w4_prio(t_, NT));                                                             \
        if constexpr (WM >= 2) asm volatile("" : "+v"(arow[0]), "+v"(arow[1]), "+v"(arow[WM - 2]), "+v"(arow[WM - 1])); \
        else asm volatile("" : "+v"(arow[0]));                                                                       \
        if constexpr (t_ == 0)                                                                                       \
            W4_REQUEST_V(ch + cp_ + 1, 1 - cp_, 0)                                                                   \
        if constexpr (t_ == VT1)                                                                                     \
            W4_REQUEST_V(ch + cp_ + 1, 1 - cp_, 1)                                                                   \
        if constexpr (t_ == NT - 1) {                                                                                \
            W4_WAIT_VM(2 * (NT - 1 - VT1))                                                                           \
            __syncthreads();                                                                                         \
        }                                                                                                            \
        W4_WAIT_B(BCUR, nb_)                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if constexpr (t_ < NT - 1) W4_MFMA_SPREAD(ACUR, BCUR, ANXT, t_ + 1, cp_, BREQ, tn_, ch + cn_)                \
        else W4_MFMA_SPREAD(ACUR, BCUR, ANXT, 0, 1 - cp_, BREQ, tn_, ch + cn_)                                       \
    }
#define W4_TAP6(U0)                                                                                                  \
    {                                                                                                                \
        W4_TAP((U0) + 0, a0, a1, bq0, bq5)                                                                           \
        W4_TAP((U0) + 1, a1, a0, bq1, bq0)                                                                           \
        W4_TAP((U0) + 2, a0, a1, bq2, bq1)                                                                           \
        W4_TAP((U0) + 3, a1, a0, bq3, bq2)                                                                           \
        W4_TAP((U0) + 4, a0, a1, bq4, bq3)                                                                           \
        W4_TAP((U0) + 5, a1, a0, bq5, bq4)                                                                           \
    }
#define W4_TAP18R9()                                                                                                 \
    {                                                                                                                \
        W4_TAP(0, a0, a1, bq0, bq8)                                                                                  \
        W4_TAP(1, a1, a0, bq1, bq0)                                                                                  \
        W4_TAP(2, a0, a1, bq2, bq1)                                                                                  \
        W4_TAP(3, a1, a0, bq3, bq2)                                                                                  \
        W4_TAP(4, a0, a1, bq4, bq3)                                                                                  \
        W4_TAP(5, a1, a0, bq5, bq4)                                                                                  \
        W4_TAP(6, a0, a1, bq6, bq5)                                                                                  \
        W4_TAP(7, a1, a0, bq7, bq6)                                                                                  \
        W4_TAP(8, a0, a1, bq8, bq7)                                                                                  \
        W4_TAP(9, a1, a0, bq0, bq8)                                                                                  \
        W4_TAP(10, a0, a1, bq1, bq0)                                                                                 \
        W4_TAP(11, a1, a0, bq2, bq1)                                                                                 \
        W4_TAP(12, a0, a1, bq3, bq2)                                                                                 \
        W4_TAP(13, a1, a0, bq4, bq3)                                                                                 \
        W4_TAP(14, a0, a1, bq5, bq4)                                                                                 \
        W4_TAP(15, a1, a0, bq6, bq5)                                                                                 \
        W4_TAP(16, a0, a1, bq7, bq6)                                                                                 \
        W4_TAP(17, a1, a0, bq8, bq7)                                                                                 \
    }
    for (int ch = 0; ch < a.nchunk; ch += 2) {
        if constexpr (R == 9) {
            W4_TAP18R9()
        } else {
            W4_TAP6(0)
            if constexpr (NT >= 6) W4_TAP6(6)
        }
    }
    if constexpr (W4_PRIO) __builtin_amdgcn_s_setprio(0);
    // the stream's harmless last requests (LDS-DMA included) must land before LDS and the ring's registers are reused
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
#ifdef W4_TAPTIME
    if (lane < 18) {
        atomicAdd(&w4_tt[(((PRE ? 8 : 0) + wave) * 18 + lane) * 2], (unsigned long long)tt_lds[lane]);
        atomicAdd(&w4_tt[(((PRE ? 8 : 0) + wave) * 18 + lane) * 2 + 1], (unsigned long long)tt_cnt[lane]);
    }
#endif
#undef W4_GLDS
#undef W4_BLDS
#undef W4_REQUEST_V
#undef W4_LOAD_A
#undef W4_ADDR_A
#undef W4_READ_A
#undef W4_MFMA_SPREAD
#undef W4_REQUEST_B
#undef W4_WAIT_B
#undef W4_WAIT_VM
#undef W4_TAP
#undef W4_TAP6
#undef W4_TAP18R9
}

// Workgroup -> (brick, channel tile, frame parity).  All (virtual) workgroups that read the same V brick (channel tiles, frame
// parities) take consecutive slots of ONE XCD (workgroup i runs on XCD i % 8).  Which bricks an XCD gets decides what its 4 MB L2
// can share between them (every brick re-reads a t-halo of 2 / TT and an h-halo of 2 / TH of its rows):
//   order 0  XCD x owns the flat brick indices x, x + 8, ... (bj fastest, then bh): one w-column and every SECOND bh -- the
//            h-neighbours of a brick always sit on another XCD;
//   order 1  XCD x owns whole (sample, w-column) columns x, x + 8, ...; inside a column bh runs fastest, then bt: the ~16
//            bricks an XCD has in flight form a contiguous (t, h) slab whose inner halos are shared through its L2;
//   order 2  the same with bt fastest.
struct W4Brick { int par, ntile, b0, t0, h0, j0; };

template <int BN>
__host__ __device__ __forceinline__ W4Brick w4_decode(const W4Args& a, int v) {
    const int nNt = a.CoutPad / BN;
    const int npar = a.tdup ? 2 : 1;
    const int per_brick = nNt * npar;
    const int nbrick = a.nvirt / per_brick;
    int par, ntile, b0, bt, bh, bj;
    if ((nbrick & 7) == 0) {
        const int xcd = v & 7, slot = v >> 3;
        const int sub = slot % per_brick, q = slot / per_brick;   // q: this XCD's q-th brick
        par = a.tdup ? sub & 1 : 0;
        ntile = a.tdup ? sub >> 1 : sub;
        const int ncol = a.B * a.nbJ;
        if (a.order && (ncol & 7) == 0) {
            const int bpc = a.nbT * a.nbH;
            const int colq = q / bpc, r = q - colq * bpc;
            const int col = colq * 8 + xcd;
            b0 = col / a.nbJ; bj = col - b0 * a.nbJ;
            if (a.order == 1) { bt = r / a.nbH; bh = r - bt * a.nbH; }
            else { bh = r / a.nbT; bt = r - bh * a.nbT; }
        } else {
            int brick = q * 8 + xcd;
            bj = brick % a.nbJ; brick /= a.nbJ;
            bh = brick % a.nbH; brick /= a.nbH;
            bt = brick % a.nbT; b0 = brick / a.nbT;
        }
    } else {
        par = a.tdup ? (int)(v >= (a.nvirt >> 1)) : 0;
        int brick = a.tdup ? v % (a.nvirt >> 1) : v;
        ntile = brick % nNt; brick /= nNt;
        bj = brick % a.nbJ; brick /= a.nbJ;
        bh = brick % a.nbH; brick /= a.nbH;
        bt = brick % a.nbT; b0 = brick / a.nbT;
    }
    return W4Brick{par, ntile, b0, bt * a.TT, bh * a.TH, bj * 4};
}

// index tables of one brick: gposA [1024] (planes 0..3), gposB [1024] (planes 4, 5 in rows 0..511, -1 = zero page behind),
// tpos [128] (output position of a tile's first column), tres [128][4] (residual rows of the tile's four columns)
// REL: V rows relative to the brick's sample and W4_PAD_ROW for padding (the buffer-descriptor requests of the one-brick kernels);
// else global row indices and -1
template <int KT, int NTH, bool REL>
__device__ __forceinline__ void w4_tables(const W4Args& a, const W4Brick& k, int* gposA, int tid) {
    constexpr int ROWS_A = W4Geo<NTH>::ROWS_A, TILES = W4Geo<NTH>::TILES;
    int* gposB = gposA + ROWS_A;
    int* tpos = gposB + ROWS_A;
    int* tres = tpos + TILES;
    const int pt = a.tdup ? 1 - k.par : KT / 2;
    const int HT = a.TT + KT - 1, HH = a.TH + 2;
    const int plane = HT * HH * 4;        // (TJ = 4 tiles along w in every brick of this kernel)
    if (tid < TILES) {
        int m = tid;
        const int ij = m & 3; m >>= 2;       // (no integer divisions in the index tables: they cost a workgroup ~1.5 us)
        const int ih = m & (a.TH - 1); m >>= a.th_shift;
        const int t = k.t0 + m, h = k.h0 + ih, w = 4 * (k.j0 + ij);
        const int To = a.tdup ? 2 * a.T : a.T, to = a.tdup ? 2 * t + k.par : t;
        tpos[tid] = ((k.b0 * To + to) * a.H + h) * a.W + w;
        const int rbase = ((k.b0 * (To >> a.rt_shift) + (to >> a.rt_shift)) * (a.H >> a.rs_shift) + (h >> a.rs_shift)) * (a.W >> a.rs_shift);
#pragma unroll
        for (int c = 0; c < 4; ++c) tres[4 * tid + c] = rbase + ((w + c) >> a.rs_shift);
    }
    for (int r = tid; r < 2 * ROWS_A; r += NTH) {
        const bool pb = r >= ROWS_A;                    // row of pass B's brick
        const int rr = pb ? r - ROWS_A : r;
        const int x = (rr >= plane) + (rr >= 2 * plane) + (rr >= 3 * plane) + (rr >= 4 * plane);   // (>= 4: not a row of the brick)
        int q = rr - x * plane;
        const int ij = q & 3; q >>= 2;
        const int qh = (int)(((unsigned)q * (unsigned)a.hh_magic) >> 20);   // q / HH
        const int ih = q - qh * HH; q = qh;
        const int t = k.t0 + q - pt, h = k.h0 + ih - 1, j = k.j0 + ij;
        const bool ok = x < (pb ? 2 : 4) && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H;
        const int xg = pb ? 4 + x : x;
        if constexpr (REL) (pb ? gposB : gposA)[rr] = ok ? (((t * a.nchunk * 6 + xg) * a.H + h) * a.J + j) : W4_PAD_ROW;   // chunk 0; 64-byte rows
        else (pb ? gposB : gposA)[rr] = ok ? ((((k.b0 * a.T + t) * a.nchunk * 6 + xg) * a.H + h) * a.J + j) : -1;
    }
}

constexpr int W4_TABLE_BYTES = w4_table_bytes<512>();   // one table set of the 512-thread kernels

// NT: (kt, kh) taps: 9 = 3x3x3, 6 = temporal-duplication pair kernels (2x3x3), 3 = one time slice (1x3x3: Conv2d).
// BN: output channels per workgroup.  64: as described above.  32 (layers with 32 output channels): pass A wave = (plane,
// tile half) with 2 row blocks, pass B wave = (plane, tile quarter) with 1 row block.
// PIPE: software-pipelined persistent kernel -- one workgroup per CU loops over the virtual workgroups v = blockIdx.x + i *
// gridDim.x (gridDim.x a multiple of 8: the XCD of a virtual workgroup does not change).  What a brick's workgroup used to do
// between its loops with the matrix pipe idle (20-33 % of its time) is moved underneath the loops of its neighbours in time:
//   * the index tables of brick i + 1 are built while brick i's pass B waits for its first weight fragments,
//   * the first V brick of brick i + 1 travels as two extra LDS-DMA loads per half-request of brick i's pass B into the 64 KB
//     region that pass B (two 30 KB buffers) does not use; the two regions swap roles from brick to brick (rb0 / rb1),
//   * the epilogue therefore works in FOUR passes (32-channel half x 64-tile half: E = 6 x 64 x 32 fp32 = 48 KB) inside pass
//     B's region and leaves the other one alone.
// LDS (PIPE): [0, 64 K) [64 K, 128 K) the two regions, then two table sets; statistics partials behind E; one dump row.
// PIPE = 2 ("lite"): the persistent loop with only what was free in the measurement of PIPE = 1: the next brick's tables under pass
// B's prologue, and its first V brick requested right behind the epilogue's last read of the exchange buffer (in front of the
// statistics tail), into the fixed first region -- pass B and the two-half epilogue are those of the default kernel.
// NTH: threads per workgroup (W4Geo): 512, or 256 = the 32-channel kernel as two workgroups per CU.
template <int NT, int BN, int PIPE, int NTH>
__global__ __launch_bounds__(NTH, 2) void conv_wino4_f16x3_kernel(W4Args a) {
    constexpr bool FULL = PIPE == 1, LITE = PIPE == 2, PERSIST = PIPE != 0;
    static_assert(NTH == 512 || (NTH == 256 && BN == 32 && PIPE == 0), "the 256-thread geometry exists for 32-channel one-brick workgroups");
    using Geo = W4Geo<NTH>;
    constexpr bool BUF = PIPE == 0;   // V requests through a buffer descriptor of the brick's sample (see w4_pass)
    constexpr int NW = NTH / 64;
    constexpr int WMA = BN == 64 ? 4 : 2, WMB = BN == 64 ? 2 : 1;
    constexpr int KT = NT / 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int HH = a.TH + 2;
    const int plane = (a.TT + KT - 1) * HH * 4;
    const int nblk = a.CoutPad >> 5;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

    int flip = 0, set = 0;                        // (PIPE) region of pass A's chunk 0 / table set of the current brick
    int v = (int)blockIdx.x;
    W4Brick bk = w4_decode<BN>(a, v);
    int w4_tlv_ = v;   // (timeline builds index their stamps by the virtual workgroup)
    {
        const int tid = tid0;
        W4_STAMP(0)
    }
    w4_tables<KT, NTH, BUF>(a, bk, reinterpret_cast<int*>(smem + a.tofs), tid0);
    // a brick's first V brick (pass A, chunk 0) into the first region: 8 LDS-DMA loads per thread from the table gq0
    auto request_chunk0 = [&](const int* gq0, int tid) {
        const int* gq = gq0 + (tid >> 2);
        const long vpiece = (long)((tid & 3) ^ ((tid >> 4) & 3)) * 16;
        const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = gq[128 * u];
            const char* src = g >= 0 ? a.in + vpiece + (long)g * 64 : a.zeros;
            unsigned keep_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" W4_V_POLICY "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_) : "v"(src), "s"(vdst + (unsigned)(u * 8192)) : "memory");
        }
    };
    if constexpr (PERSIST) {
        // (measurement: de-synchronise the persistent workgroups -- the CUs of an XCD start a quarter of a brick apart)
        for (int i = 0; i < a.skew * (int)((blockIdx.x >> 3) & 3); ++i) __builtin_amdgcn_s_sleep(127);
        // the first brick of this workgroup: its first V brick is requested here (every later one during the previous brick)
        __syncthreads();
        request_chunk0(reinterpret_cast<const int*>(smem + a.tofs), tid0);
    }
#pragma unroll 1
    for (;;) {
        // (persistent kernels: everything derived from the thread index is re-derived per brick -- hoisted out of the brick loop it
        //  stays live across both tap loops and the epilogue and costs the 9-tap kernel more registers than it has)
        int tid = tid0;
        if constexpr (PERSIST) asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int kg = lane >> 5, l31 = lane & 31;
        int* gposA = reinterpret_cast<int*>(smem + a.tofs + (PERSIST ? set * W4_TABLE_BYTES : 0));
        int* gposB = gposA + Geo::ROWS_A;
        const int* tpos = gposB + Geo::ROWS_A;
        const int* tres = tpos + Geo::TILES;
        const int n0 = bk.ntile * BN, b0 = bk.b0;
        const char* wbase = a.wp + (long)bk.par * a.wset_stride;   // wave-uniform; the lane's 16 bytes are added by the load
        // (BUF) descriptor of this brick's sample of V: base + b0 * bytes per sample, num_records = bytes per sample (< 2^31, checked by the launcher)
        const long vsample = (long)a.T * a.nchunk * 6 * a.H * a.J * 64;
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in) + (BUF ? (long)b0 * vsample : 0), 0,
                                                                               BUF ? (int)vsample : 0, 0x00020000);
        const int vn = v + (int)gridDim.x;
        const bool more = PERSIST && vn < a.nvirt;
        const W4Brick bn_ = more ? w4_decode<BN>(a, vn) : bk;
        int* gposAn = reinterpret_cast<int*>(smem + a.tofs + (set ^ 1) * W4_TABLE_BYTES);
        // V buffers (LDS rows): pass A alternates between the two 64 KB regions starting at `flip`; pass B's two 32 KB buffers
        // live in region `flip` (pass A's last chunk -- an odd one -- reads the other region)
        const int rA0 = FULL ? flip * 1024 : 0, rA1 = FULL ? (flip ^ 1) * 1024 : Geo::ROWS_A;
        const int rB0 = rA0, rB1 = rA0 + Geo::ROWS_B;

        // ---- pass A: planes 0..3, wave = (plane, 32-channel half), all 128 tiles   [BN = 32: (plane, tile half)]
        const int xa = wave & 3, nha = BN == 64 ? wave >> 2 : 0, mha = BN == 64 ? 0 : (wave >> 2) * 64;
        f32x16 accA[WMA];
        {
            int arow[WMA];
#pragma unroll
            for (int wm = 0; wm < WMA; ++wm) {
                int m = mha + wm * 32 + w4_row_tile<NTH>(l31);
                const int ij = m & 3; m >>= 2;
                const int ih = m & (a.TH - 1); m >>= a.th_shift;
                arow[wm] = xa * plane + (m * HH + ih) * 4 + ij;
#pragma unroll
                for (int r = 0; r < 16; ++r) accA[wm][r] = 0.f;
            }
            const W4Next none{gposA, 0u, 0u, 0};
            w4_pass<NT, WMA, Geo::VA0, Geo::VA1, NTH, false, PERSIST, 0, BUF>(a, smem, gposA, gposB, accA, arow, wbase + ((long)xa * nblk + (n0 >> 5) + nha) * 2048, HH, tid,
                                             lane, wave, rA0, rA1, none, [] {}, w4_tlv_, vrsrc);
        }
        W4_STAMP(2)
        // ---- pass B: planes 4, 5, wave = (plane, 32-channel half, tile half)   [BN = 32: (plane, tile quarter)]
        const int xb = wave & 1, nhb = BN == 64 ? (wave >> 1) & 1 : 0, mhb = BN == 64 ? (wave >> 2) * 64 : (wave >> 1) * 32;
        f32x16 accB[WMB];
        {
            int arow[WMB];
#pragma unroll
            for (int wm = 0; wm < WMB; ++wm) {
                int m = mhb + wm * 32 + w4_row_tile<NTH>(l31);
                const int ij = m & 3; m >>= 2;
                const int ih = m & (a.TH - 1); m >>= a.th_shift;
                arow[wm] = xb * plane + (m * HH + ih) * 4 + ij;
#pragma unroll
                for (int r = 0; r < 16; ++r) accB[wm][r] = 0.f;
            }
            // (chunk 0 of this brick was requested by pass A behind its last chunk and published by its last barrier; pass B's
            //  own request behind ITS last chunk re-reads its chunk 0 harmlessly)
            const W4Next nxt{(more ? gposAn : gposA) + (tid >> 2),
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(flip ^ 1) * 65536u + (unsigned)wave * 1024u)),
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)flip * 65536u + 30720u)), more ? 1 : 0};   // dump: rows 480..495 of pass B's first buffer (zero padding, never read)
            w4_pass<NT, WMB, Geo::VB0, Geo::VB1, NTH, true, PERSIST, FULL ? 2 : 0, BUF>(a, smem, gposB, gposB, accB, arow, wbase + ((long)(4 + xb) * nblk + (n0 >> 5) + nhb) * 2048, HH, tid,
                                            lane, wave, rB0, rB1, nxt, [&] {
                                                // (PIPE) the next brick's tables, built while this pass's first weight fragments
                                                // travel; published by the barrier in front of the loop
                                                if constexpr (PERSIST) { if (more) w4_tables<KT, NTH, BUF>(a, bn_, gposAn, tid); }
                                            }, w4_tlv_, vrsrc);
        }
        W4_STAMP(4)

        // ---- epilogue: E = [6 planes][tiles][32 channels] fp32.  A wave's ds_write_b32 stores the rows m (lanes 0..31) and m + 4
        // (lanes 32..63) of an accumulator register: 512 bytes apart = the same 32 banks.  Tile m is therefore kept in row
        // m ^ ((m >> 2) & 1), which puts the two halves of the wave on the two halves of the banks.
        // !PIPE: one 32-channel half at a time, all 128 tiles (98 KB over both V regions).  PIPE: (32-channel half, 64-tile half)
        // quarters of 48 KB inside pass B's region -- the other region holds the next brick's first V brick already.
        constexpr int NQ = 8, TPI = NTH / NQ;         // a thread owns four channels of one tile per iteration
        constexpr int NTHALF = FULL ? 2 : 1;          // tile halves per channel half
        constexpr int ET = Geo::TILES / NTHALF;       // tiles in E
        constexpr int NIT = ET / TPI;
        float* E = reinterpret_cast<float*>(smem + (FULL ? flip * 65536 : 0));
        double* S = reinterpret_cast<double*>(reinterpret_cast<char*>(E) + 6 * ET * 32 * 4);   // [2 halves][NW waves][32 channels][2] behind E
        const int n4 = tid % NQ;
        const int e3 = kg * 96, e5 = kg * 160;   // row offsets (in floats) of the wave's upper lanes, see the E writes
        const int e8 = kg * 256;                 // (256-thread kernels: the upper lanes' tile quad is 2 tile quads = 8 rows away, see w4_escatter)
#pragma unroll 1
        for (int half = 0; half < BN / 32; ++half) {
            const int n = n0 + half * 32 + 4 * n4;
            const bool ncol = n < a.Cout;
            double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
            float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && ncol) bias = *reinterpret_cast<const float4*>(a.bias + n);
            const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll 1
            for (int th = 0; th < NTHALF; ++th) {
                const int tb = th * ET;               // first tile of this E
                // residual rows first, all of them, so that their latency hides behind the LDS exchange.  (PIPE: requesting both
                // tile halves' rows in front of the first one costs 16 spilled registers in the 64-channel kernels; per quarter
                // the loads queue behind the previous quarter's stores, which the exchange's two barriers mostly cover.)
                f32x4 rres[NIT][4];
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        rres[it][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (a.res && ncol)
                            rres[it][c] = *reinterpret_cast<const f32x4*>(a.res + (long)tres[4 * (tb + tid / NQ + TPI * it) + c] * a.Cout + n);
                    }
                __syncthreads();   // the V bricks / the previous E are no longer read
                if (half == 0 && th == 0) W4_STAMP(8)
                if (nha == half) {
#pragma unroll
                    for (int wm = 0; wm < WMA; ++wm) {
                        const int m0 = mha + wm * 32 - tb;    // first tile of this row block inside E (wave-uniform)
                        if (m0 >= 0 && m0 < ET) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if constexpr (NTH == 256) E[w4_escatter(xa * ET + m0, r) * 32 + l31 + (w4_escatter_up(r) ? e8 : -e8)] = accA[wm][r];
                                else {
                                    // tile m = c + 4 kg sits in row m ^ ((m >> 2) & 1) = c + (r odd ? 3 : 5) kg: two base addresses + immediates
                                    const int c = m0 + (r & 3) + 8 * (r >> 2);
                                    E[(xa * ET + c) * 32 + l31 + ((r & 1) ? e3 : e5)] = accA[wm][r];
                                }
                            }
                        }
                    }
                }
                if (nhb == half) {
#pragma unroll
                    for (int wm = 0; wm < WMB; ++wm) {
                        const int m0 = mhb + wm * 32 - tb;
                        if (m0 >= 0 && m0 < ET) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if constexpr (NTH == 256) E[w4_escatter((4 + xb) * ET + m0, r) * 32 + l31 + (w4_escatter_up(r) ? e8 : -e8)] = accB[wm][r];
                                else {
                                    const int c = m0 + (r & 3) + 8 * (r >> 2);
                                    E[((4 + xb) * ET + c) * 32 + l31 + ((r & 1) ? e3 : e5)] = accB[wm][r];
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (half == 0 && th == 0) W4_STAMP(9)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int tile = tid / NQ + TPI * it;
                    float mx[6][4];
#pragma unroll
                    for (int x = 0; x < 6; ++x) {
                        const float4 vv = *reinterpret_cast<const float4*>(E + (x * ET + (tile ^ ((tile >> 2) & 1))) * 32 + 4 * n4);
                        mx[x][0] = vv.x; mx[x][1] = vv.y; mx[x][2] = vv.z; mx[x][3] = vv.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float s12 = mx[1][j] + mx[2][j], d12 = mx[1][j] - mx[2][j];
                        const float s34 = mx[3][j] + mx[4][j], d34 = mx[3][j] - mx[4][j];
                        const float y[4] = {mx[0][j] + s12 + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + mx[5][j]};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float vv = fmaf(y[c], a.oscale, bv[j]) + rres[it][c][j];
                            if (ncol) {
                                ssum[j] += (double)vv;
                                ssq[j] = fma((double)vv, (double)vv, ssq[j]);
                            }
                            if (a.epi & EPI_LRELU) vv = vv >= 0.f ? vv : 0.2f * vv;
                            rres[it][c][j] = vv;
                        }
                    }
                }
                if (ncol) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const long p = tpos[tb + tid / NQ + TPI * it];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
#ifdef W4_OUT_NT
                            __builtin_nontemporal_store(rres[it][c], reinterpret_cast<f32x4*>(a.out + (p + c) * a.Cout + n));
#else
                            *reinterpret_cast<f32x4*>(a.out + (p + c) * a.Cout + n) = rres[it][c];
#endif
                        }
                    }
                }
                if (half == 0 && th == 0) W4_STAMP(10)
            }
            if (a.stats) {
                // lanes of a wave that share (lane % NQ) hold the same four channels -> wavefront shuffles; the eight waves'
                // partials meet in LDS (behind E) and one wave per channel half issues its 2 x 32 fp64 atomics
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (NQ <= 8) { ssum[j] = wave_xor_add_f64<8>(ssum[j]); ssq[j] = wave_xor_add_f64<8>(ssq[j]); }
                    ssum[j] = wave_xor_add_f64<16>(ssum[j]); ssq[j] = wave_xor_add_f64<16>(ssq[j]);
                    ssum[j] = wave_xor_add_f64<32>(ssum[j]); ssq[j] = wave_xor_add_f64<32>(ssq[j]);
                }
                // (each half has its own 4 KB of S: the cross-wave sums and the atomics of both halves wait until after the loop,
                //  one barrier and two waves instead of a barrier and a serial section of wave 0 per half)
                double* Sh = S + half * (NW * 32 * 2);
                if (lane < NQ) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Sh[(wave * 32 + 4 * lane + j) * 2] = ssum[j];
                        Sh[(wave * 32 + 4 * lane + j) * 2 + 1] = ssq[j];
                    }
                }
            }
            W4_STAMP(5 + half)
        }
        if constexpr (LITE) {
            // every wave has read the exchange buffer for the last time: the first region may take the next brick's first V brick
            // (its tables were written under pass B's prologue); the statistics tail and the loop-back hide part of its latency
            __syncthreads();
            if (more) request_chunk0(gposAn, tid);
        }
        if (a.stats) {
            if constexpr (!LITE) __syncthreads();
            if (wave < BN / 32 && lane < 32 && n0 + wave * 32 + lane < a.Cout) {   // wave h sums channel half h
                const double* Sh = S + wave * (NW * 32 * 2);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    s0 += Sh[(w * 32 + lane) * 2];
                    s1 += Sh[(w * 32 + lane) * 2 + 1];
                }
                double* dst = a.stats + ((long)b0 * a.Cout + n0 + wave * 32 + lane) * 2;
                atomicAdd(dst, s0);
                atomicAdd(dst + 1, s1);
            }
        }
        W4_STAMP(7)
        if (!more) break;
        v = vn; bk = bn_; flip ^= FULL ? 1 : 0; set ^= 1;
        w4_tlv_ = v;
        W4_STAMP(0)
    }
}


// brick of `tiles` (128: 512-thread kernels, 64: 256-thread kernels) = TT frames x TH rows x 4 tiles (16 output positions); rows_a /
// rows_b: V rows one buffer of pass A / pass B can stage
inline bool wino4_tiling(int T, int H, int W, int KT, int* TT_, int* TH_, int tiles = W4_TILES, int rows_a = W4_ROWS_A, int rows_b = W4_ROWS_A / 2 - 16) {
    if (T < 1 || (T < 2 && KT != 1) || W % 16 || H < 8) return false;
    int TT = 1;
    while (TT < 4 && T % (TT * 2) == 0) TT *= 2;
    const int TH = tiles / (TT * 4);
    if (TH < 4 || TH > H || H % TH || TH * 4 % 16) return false;       // (16 consecutive tiles = 16 consecutive V rows: conflict-free ds_read_b128)
    if (4 * (TT + KT - 1) * (TH + 2) * 4 > rows_a) return false;       // pass A's halo brick
    if (2 * (TT + KT - 1) * (TH + 2) * 4 > rows_b) return false;       // pass B's (512-thread kernels keep 16 padding rows: PIPE's dump row)
    *TT_ = TT; *TH_ = TH;
    return true;
}


// the LOADER form of the 32-channel 3x3x3 kernel (i2v_conv16w4g.hip): four extra waves issue the V requests
bool wino4_loader_supported(const W4Args& a, int KT);
int wino4_loader_launch(W4Args& a, unsigned nblk, hipStream_t st, int form = 1);

}  // namespace i2v

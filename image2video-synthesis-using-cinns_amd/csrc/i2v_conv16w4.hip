// 3x3x3 Conv3d with a Winograd F(4,3) transform along W on the gfx950 fp16 matrix cores (split-fp16 operands).
//
// F(2,3) (i2v_conv16w.hip) multiplies 4 transformed planes per 2 outputs; F(4,3) multiplies 6 per 4: 0.75x the MFMAs and a
// V operand of 12 instead of 16 bytes per activation.  Per tile of four output positions (w = 4j .. 4j+3) and (kt, kh) tap
//     d_k = a[t+kt-1][h+kh-1][4j-1+k], k = 0..5 (zero padded)
//     V0 = 4 d0 - 5 d2 + d4        V1 = -4 d1 - 4 d2 + d3 + d4     V2 = 4 d1 - 4 d2 - d3 + d4
//     V3 = -2 d1 - d2 + 2 d3 + d4  V4 = 2 d1 - d2 - 2 d3 + d4      V5 = 4 d1 - 5 d3 + d5
//     U0 = g0/4   U1 = -(g0+g1+g2)/6   U2 = -(g0-g1+g2)/6   U3 = g0/24 + g1/12 + g2/6   U4 = g0/24 - g1/12 + g2/6   U5 = g2
//     M_x = sum over (kt, kh, c) of V_x U_x
//     y0 = M0+M1+M2+M3+M4   y1 = M1-M2+2M3-2M4   y2 = M1+M2+4M3+4M4   y3 = M1-M2+8M3-8M4+M5
// V = B^T d is written once by the producer (modulate_wino4_kernel, fp32 then split into fp16 hi / lo) as
// [B][T][C/16][6][H][W/4][16 channels = 64 B]; U = G g is computed in fp64 at load time.
//
// What shapes the kernel: 160 KB of LDS and the weight (B-operand) traffic.  A wave must multiply ONE weight fragment with
// 128 tiles (4 MFMA row blocks) -- at 64 tiles the weight stream from L2 doubles per MFMA, which is what holds the
// 32-channel variant of the F(2,3) kernel at 60 % of the 64-channel one -- and the double-buffered halo brick of SIX planes
// of 128 tiles would need 184 KB.  So the six planes are multiplied in TWO passes over the K loop with the accumulators of
// both passes kept in registers:
//   pass A  planes 0..3: exactly the loop of the F(2,3) kernel (wave = (plane, 32-channel half), 4 row blocks, 64
//           accumulator registers, nine-slot weight ring, V brick of 4 planes double-buffered by LDS-DMA);
//   pass B  planes 4, 5: wave = (plane, 32-channel half, tile half), 2 row blocks, 32 more accumulator registers, V brick of
//           2 planes (this third of the MFMAs sees the doubled weight stream);
//   epilogue: per 32-channel half the six partial GEMMs of a tile meet in LDS (98 KB), y = A^T M in fp32, then bias,
//           residual, optional lrelu, fused per-(b,c) statistics, stores of four positions per tile.
// Workgroup = 512 threads, 128 tiles = 512 output positions (TT x TH x 16 brick) x 64 output channels.
// Schedule (round 3, from per-tap and per-workgroup timing: -DW4_TAPTIME / -DW4_TIMELINE builds of tools/conv16w_check): an
// in-order wave that issues its MFMAs back to back sits blocked on the matrix pipe, and whatever it issues outside the MFMA
// block is time the pipe idles unless the partner wave of the SIMD happens to have an MFMA ready.  So everything a tap has to
// issue for the NEXT taps -- per row block the LDS address arithmetic and the two ds_read_b128 of the next tap's A operands,
// then the weight request of tap U + R - 1 (scalar base + the lane's 16 bytes: no vector address arithmetic) -- sits one small
// piece per 32-cycle gap between the tap's MFMAs; wave priority falls with the tap index inside a chunk, so that whichever of
// the two waves of a SIMD is behind gets the pipe; pass A requests pass B's first brick behind its own last chunk (no V
// round trip between the passes).  MFMA pipe busy 0.53 -> 0.62 at 1.71 -> 1.63 GHz (the chip is power-limited: DESIGN.md).
// Loads are asm statements with hand-counted waits exactly as in i2v_conv16w.hip; every pass waits for ALL of its prologue
// requests before the loop (the loop's counts assume the steady state).  tools/check_asm_waits.py replays both compiled loops
// of every instantiation, checks that each is entered with nothing in flight and that no asm load reads a freshly
// VALU-written SGPR.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "i2v_conv.h"

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

// ---- host side ------------------------------------------------------------------------------------------------------

// w3: [nset][Cout][Cin][NT][3] (fp64, already scaled); packs U = G g per (kt, kh)
static int wino4_pack_sets(Wino4Weights& o, const std::vector<double>& w3, int nset, int cout, int cin, int kt) {
    o.Cin = cin; o.Cout = cout; o.KT = kt;
    o.CoutPad = (cout + 31) / 32 * 32;
    o.nchunk = cin / W4_KC;
    const int NT = kt * 3;
    std::vector<double> u((size_t)nset * cout * cin * NT * 6);
    double wmax = 0.0;
    for (size_t i = 0; i < (size_t)nset * cout * cin * NT; ++i) {
        const double g0 = w3[i * 3], g1 = w3[i * 3 + 1], g2 = w3[i * 3 + 2];
        double* d = &u[i * 6];
        d[0] = g0 / 4.0;
        d[1] = -(g0 + g1 + g2) / 6.0;
        d[2] = -(g0 - g1 + g2) / 6.0;
        d[3] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
        d[4] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
        d[5] = g2;
        for (int x = 0; x < 6; ++x) wmax = std::max(wmax, std::fabs(d[x]));
    }
    o.wexp = 0;
    if (wmax > 0.0 && std::isfinite(wmax)) o.wexp = std::max(-40, std::min(40, (int)std::floor(std::log2(16384.0 / wmax))));
    const double pre = std::ldexp(1.0, o.wexp);
    const size_t set_halfs = (size_t)NT * o.nchunk * 6 * o.CoutPad * 32;
    std::vector<_Float16> p((size_t)nset * set_halfs, (_Float16)0.f);
    for (int s = 0; s < nset; ++s)
        for (int n = 0; n < cout; ++n)
            for (int c = 0; c < cin; ++c)
                for (int tap = 0; tap < NT; ++tap)
                    for (int x = 0; x < 6; ++x) {
                        const float v = (float)(u[((((size_t)s * cout + n) * cin + c) * NT + tap) * 6 + x] * pre);
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        // fragment-major: [tap][chunk][x][32-channel block][hi | lo][lane = kg * 32 + n % 32][8 halfs]
                        const int chunk = c / W4_KC, kgq = (c % W4_KC) / 8, j = c % 8;
                        _Float16* blk = &p[s * set_halfs + ((((size_t)tap * o.nchunk + chunk) * 6 + x) * (o.CoutPad / 32) + n / 32) * 1024];
                        blk[(kgq * 32 + n % 32) * 8 + j] = hi;
                        blk[512 + (kgq * 32 + n % 32) * 8 + j] = lo;
                    }
    o.set_bytes = (long)set_halfs * 2;
    return o.w.upload(p.data(), p.size() * 2);
}

// brick of `tiles` (128: 512-thread kernels, 64: 256-thread kernels) = TT frames x TH rows x 4 tiles (16 output positions); rows_a /
// rows_b: V rows one buffer of pass A / pass B can stage
static bool wino4_tiling(int T, int H, int W, int KT, int* TT_, int* TH_, int tiles = W4_TILES, int rows_a = W4_ROWS_A, int rows_b = W4_ROWS_A / 2 - 16) {
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

bool wino4_supported(int cout, int cin, int T, int H, int W, int KT) {
    if (cout % 32 || cin % (2 * W4_KC) || (KT != 3 && KT != 2 && KT != 1)) return false;
    int TT, TH;
    return wino4_tiling(T, H, W, KT, &TT, &TH);
}

int Wino4Weights::pack(const float* w_src, const float* bias_src, int cout, int cin, double scale, int kt) {
    I2V_REQUIRE(kt == 3 || kt == 1, I2V_E_INVALID, "wino4: temporal kernel size %d", kt);
    tdup = false;
    std::vector<double> w3((size_t)cout * cin * kt * 9);
    for (size_t i = 0; i < w3.size(); ++i) w3[i] = (double)w_src[i] * scale;
    int rc = wino4_pack_sets(*this, w3, 1, cout, cin, kt);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

int Wino4Weights::pack_tdup(const float* w_src, const float* bias_src, int cout, int cin, double scale) {
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
    int rc = wino4_pack_sets(*this, w3, 2, cout, cin, 2);
    if (rc) return rc;
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

template <int NT, int BN, int PIPE, int NTH = 512>
static int launch_wino4_(const W4Args& a, unsigned grid, size_t lds, hipStream_t st) {
    auto kern = conv_wino4_f16x3_kernel<NT, BN, PIPE, NTH>;
    static bool attr_set[I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), lds, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

// compute units of the current device (one persistent workgroup each)
static int device_cus() {
    static int cus[I2V_MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= I2V_MAX_DEV) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// the 32-channel kernel as two 256-thread workgroups per CU (W4Geo<256>): 2 V regions of 576 rows + one table set = 79 616 bytes
template <int NT>
static int launch_wino4_thin(W4Args& a, unsigned nblk, hipStream_t st) {
    a.nvirt = (int)(a.tdup ? 2 * nblk : nblk);
    a.tofs = 2 * W4Geo<256>::ROWS_A * 64;
    return launch_wino4_<NT, 32, 0, 256>(a, (unsigned)a.nvirt, (size_t)a.tofs + (size_t)w4_table_bytes<256>(), st);
}

// Measurement switches of this kernel.  The PRODUCTION library (no -DI2V_MEASURE) reads no environment variable on a launch path
// and carries only the one-workgroup-per-brick kernels; the structure switches -- the software-pipelined persistent kernels
// (I2V_W4_PIPE), the start skew of their workgroups (I2V_W4_SKEW), forced tile widths / workgroup sizes (I2V_W4_BN, I2V_W4_NTH),
// the brick -> XCD order (I2V_W4_ORDER) and the launch trace (I2V_W4_TRACE) -- exist in the measurement build only
// (tools/build_measurement_libs.sh measure -> tools/_tl/libi2v_hip_measure.so, loaded through I2V_LIB_PATH; tools/conv16w_check*
// are built with the flag too), where they are read per launch so that tests and A/B runs can flip them inside one process.
struct W4Switches { int pipe, bn, order, nth, skew, trace; };
static W4Switches w4_switches() {
    W4Switches w{W4_DEFAULT_PIPE, 0, W4_DEFAULT_ORDER, 0, 0, 0};
#ifdef I2V_MEASURE
    if (const char* e = getenv("I2V_W4_PIPE")) w.pipe = atoi(e);
    if (const char* e = getenv("I2V_W4_BN")) w.bn = atoi(e);
    if (const char* e = getenv("I2V_W4_ORDER")) w.order = atoi(e);
    if (const char* e = getenv("I2V_W4_NTH")) w.nth = atoi(e);
    if (const char* e = getenv("I2V_W4_SKEW")) w.skew = atoi(e);
    w.trace = getenv("I2V_W4_TRACE") != nullptr;
#endif
    return w;
}

template <int NT, int BN>
static int launch_wino4(W4Args& a, unsigned nblk, hipStream_t st, int env_pipe) {
    a.nvirt = (int)(a.tdup ? 2 * nblk : nblk);
    const int body = 2 * W4_ROWS_A * 64;   // two V regions (pass B and the epilogue's exchange buffer reuse them)
    a.tofs = body;
#if defined(I2V_MEASURE) && !defined(W4_TAPTIME)
    if (env_pipe != 0) {
        // one workgroup per CU, a multiple of 8 so that a virtual workgroup keeps its XCD
        int grid = std::min(a.nvirt, device_cus());
        if (grid >= 8) grid &= ~7;
        const size_t lds = (size_t)body + 2 * (size_t)W4_TABLE_BYTES;
        if (env_pipe == 2) return launch_wino4_<NT, BN, 2>(a, (unsigned)grid, lds, st);
        return launch_wino4_<NT, BN, 1>(a, (unsigned)grid, lds, st);
    }
#else
    (void)env_pipe;
#endif
#ifdef W4_TAPTIME
    const size_t lds = 160 * 1024;
#else
    const size_t lds = (size_t)body + (size_t)W4_TABLE_BYTES;
#endif
    return launch_wino4_<NT, BN, 0>(a, (unsigned)a.nvirt, lds, st);
}

int wino4_forward(const Wino4Weights& wts, const void* v_hl16, float* out, const float* res, int rt, int rs, int B, int T, int H,
                  int W, int epi, hipStream_t st, double* stats) {
    I2V_REQUIRE(wts.w.p, I2V_E_STATE, "wino4: weights not packed");
    I2V_REQUIRE((epi & ~EPI_LRELU) == 0, I2V_E_INVALID, "wino4: unsupported epilogue %d", epi);
    W4Args a{};
    if (int rc0 = zero_page(&a.zeros)) return rc0;
    a.in = static_cast<const char*>(v_hl16); a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.stats = stats;
    a.B = B; a.H = H; a.W = W; a.J = W / 4; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.tdup = wts.tdup ? 1 : 0;
    a.wset_stride = wts.set_bytes;
    if (wts.tdup) {  // T is the OUTPUT frame count; the half-rate input has T / 2 frames
        I2V_REQUIRE(T % 2 == 0 && !res, I2V_E_INVALID, "wino4: temporal-duplication mode needs an even frame count and no residual");
        T /= 2;
    }
    a.T = T;
    I2V_REQUIRE(wino4_supported(wts.Cout, wts.Cin, T, H, W, wts.KT), I2V_E_INVALID, "wino4: unsupported shape [%d,%d,%d] %d -> %d (kt = %d)",
                T, H, W, wts.Cin, wts.Cout, wts.KT);
    a.rt = res ? rt : 1; a.rs = res ? rs : 1; a.epi = epi;
    I2V_REQUIRE((a.rt == 1 || a.rt == 2 || a.rt == 4) && (a.rs == 1 || a.rs == 2 || a.rs == 4), I2V_E_INVALID,
                "wino4: residual up-sampling factors %d / %d (1, 2 or 4)", a.rt, a.rs);
    a.rt_shift = a.rt >> 1 == 2 ? 2 : a.rt >> 1; a.rs_shift = a.rs >> 1 == 2 ? 2 : a.rs >> 1;
    a.oscale = (float)std::ldexp(1.0, -wts.wexp);
    int TT = 1, TH = 1;
    (void)wino4_tiling(T, H, W, wts.KT, &TT, &TH);
    int BN = a.CoutPad % 64 == 0 ? 64 : 32;  // output channels per workgroup
    const W4Switches sw = w4_switches();      // (defaults unless built with -DI2V_MEASURE)
    const int env_bn = sw.bn, env_order = sw.order, env_nth = sw.nth;
    // 64-channel workgroups that would leave CUs idle (16x16 maps at small batches) become twice as many 32-channel ones: the
    // accumulation order of every output does not depend on the tile width, so the bits are the same
    if (BN == 64 && wts.KT != 1 && (long)B * (T / TT) * (H / TH) * (a.J / 4) * (a.CoutPad / 64) * (wts.tdup ? 2 : 1) < device_cus()) BN = 32;
    if (env_bn == 32 && wts.KT != 1) BN = 32;
    if (env_bn == 64 && a.CoutPad % 64 == 0) BN = 64;
    // 32-channel layers: two 256-thread workgroups of 64 tiles per CU instead of one 512-thread workgroup of 128 (W4Geo; same
    // bits: neither the brick shape nor the workgroup size enters the accumulation order of an output).  I2V_W4_NTH=512 restores
    // round 4's geometry for A/B runs.
    bool thin = false;
    {
        int TT2 = 1, TH2 = 1;
        // Default: only the layers that HAVE 32 output channels (g_4 of the 128 x 128 configs: +3 % on 32 -> 32, +-0 on 64 -> 32,
        // profiles/r05_c_*); 64-channel layers narrowed for a small grid keep the 512-thread geometry (-2 % at B = 8 with 256).
        // I2V_W4_NTH=256 forces the 256-thread geometry wherever the brick fits, 512 forbids it.
        if (BN == 32 && wts.KT != 1 && env_nth != 512 && (a.CoutPad % 64 != 0 || env_nth == 256) && sw.pipe == 0 &&
            wino4_tiling(T, H, W, wts.KT, &TT2, &TH2, W4Geo<256>::TILES, W4Geo<256>::ROWS_A, W4Geo<256>::ROWS_B)) {
            thin = true; TT = TT2; TH = TH2;
        }
    }
    a.TT = TT; a.TH = TH; a.TJ = 4; a.nbT = T / TT; a.nbH = H / TH; a.nbJ = a.J / 4;
    a.th_shift = 0;
    while ((1 << a.th_shift) < TH) ++a.th_shift;
    I2V_REQUIRE((1 << a.th_shift) == TH, I2V_E_INVALID, "wino4: brick height %d is not a power of two", TH);
    a.hh_magic = ((1 << 20) + TH + 1) / (TH + 2);
    I2V_REQUIRE(!stats || (long)TT * TH * 4 <= (long)T * H * a.J, I2V_E_INVALID, "wino4: fused statistics need bricks inside one sample");
    a.order = env_order;
    a.skew = sw.skew;
    const long nblk = (long)B * a.nbT * a.nbH * a.nbJ * (a.CoutPad / BN);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 30), I2V_E_INVALID, "wino4: grid of %ld workgroups", nblk);
    // the kernel's index tables (gpos: V rows, tpos / tres: output and residual positions) are 32-bit
    I2V_REQUIRE((long)T * a.nchunk * 6 * H * a.J * 64 < (1L << 31), I2V_E_INVALID, "wino4: the V operand of one sample ([%d,%d,%d] x %d chunks) exceeds the 2 GB a buffer descriptor offset can address", T, H, W, a.nchunk);
    I2V_REQUIRE((long)B * T * a.nchunk * 6 * H * a.J < (1L << 31) && (long)B * (wts.tdup ? 2 * T : T) * H * W < (1L << 31), I2V_E_INVALID,
                "wino4: batch %d too large for the 32-bit row indices of this kernel ([%d,%d,%d] x %d chunks)", B, T, H, W, a.nchunk);
    if (sw.trace) {
        fprintf(stderr, "wino4: B %d T %d H %d W %d Cin %d Cout %d pad %d KT %d tdup %d TT %d TH %d res %p rt %d rs %d stats %p epi %d nblk %ld\n", B, T, H, W,
                a.Cin, a.Cout, a.CoutPad, wts.KT, a.tdup, TT, TH, (const void*)res, a.rt, a.rs, (void*)stats, epi, nblk);
        (void)hipDeviceSynchronize();
    }
    if (BN == 64) {
        if (wts.KT == 3) return launch_wino4<9, 64>(a, (unsigned)nblk, st, sw.pipe);
        if (wts.KT == 2) return launch_wino4<6, 64>(a, (unsigned)nblk, st, sw.pipe);
        return launch_wino4<3, 64>(a, (unsigned)nblk, st, sw.pipe);   // one time slice: SPADE's 2-D convs
    }
    I2V_REQUIRE(wts.KT != 1, I2V_E_INVALID, "wino4: the 1x3x3 variant exists for 64-channel tiles only");
#ifndef W4_TAPTIME
    if (thin) return wts.KT == 3 ? launch_wino4_thin<9>(a, (unsigned)nblk, st) : launch_wino4_thin<6>(a, (unsigned)nblk, st);
#endif
    if (wts.KT == 3) return launch_wino4<9, 32>(a, (unsigned)nblk, st, sw.pipe);
    return launch_wino4<6, 32>(a, (unsigned)nblk, st, sw.pipe);
}

// =====================================================================================================================
// GEN (round 6): the F(4,3) kernel for the THIN layers of the 128 x 128 configs (g_4: 64 -> 32 and 32 -> 32 channels at 16 x 128 x 128)
// with the operand generated IN the kernel.  Round 5 ran these two layers at 0.19 / 0.15 of the data sheet, each behind an operand
// writer (modulate_wino4_kernel) that reads the fp32 conv input and writes V = B^T d -- 6 bytes per activation -- to HBM for the conv
// to read back 2.3 x: per 128-tile brick the writer costs 18 us next to 30 us of conv.  Here the workgroup has TWELVE waves:
//   waves 0..7   the MFMA role: exactly the 512-thread 32-channel kernel above (pass A: wave = (plane, tile half), pass B: (plane,
//                tile quarter); same accumulation order, same bits), with every V request removed from its tap loops;
//   waves 8..11  the PRODUCER role, one wave per SIMD: they read the conv's fp32 INPUT (+ the (b,c) coefficients of the
//                normalisation, + SPADE's gamma' | beta maps) for the halo brick, form d = lrelu(x a + b), V = B^T d in fp32, the
//                fp16 hi / lo split, and store the 64-byte rows into the V buffer the next chunk barrier publishes -- the same
//                arithmetic, expression for expression, as modulate_wino4_kernel (bit-identical V, hence bit-identical frames).
// The matrix pipe and the vector ALU of a SIMD are separate pipes (MI355X_MICROARCH.md: an MFMA wave and a VALU wave run
// concurrently), so the producer's ~900 VALU instructions per chunk run underneath the two MFMA waves' 108 + 54 MFMAs; buffers,
// barriers and the hand-over between the passes are those of the kernel above: while the MFMA waves multiply chunk c out of
// buffer c & 1 the producers fill buffer (c + 1) & 1, and pass A's last chunk is the time in which pass B's first brick is made.
// A producer lane owns (frame pair, halo row, tile, channel quad): 480 slots per chunk = two rounds of the 256 producer lanes, two
// frames per round; the loads of a round are issued one round ahead (registers).
struct W4GenArgs {
    const float* x;      // the conv's input BEFORE normalisation / activation: fp32 channels-last [B][T][H / us][W / us][Cin]
    const float2* coef;  // per-(b,c) affine of the normalisation, norm(x) == x * A + B: [B][Cin]
    const float* gb;     // SPADE: gamma' | beta maps [B][H][W][2 Cin]; null: ADAIN (the (b,c) affine is the whole modulation)
    int us;              // nearest up-sampling in front of the conv along H and W (1 or 2)
    int* range_flag;     // sticky overflow flag of the split-fp16 format (bit 0) or null
    int* umax;           // underflow guard slot (largest |activation| written, float bits) or null
};

constexpr int W4G_THREADS = 768, W4G_PROD = 256;
constexpr int W4G_TT = 4, W4G_TH = 8, W4G_HT = W4G_TT + 2, W4G_HH = W4G_TH + 2;   // brick of the 512-thread geometry, 3 temporal taps
constexpr int W4G_PLANE = W4G_HT * W4G_HH * 4;                                     // 240 V rows per plane
constexpr int W4G_SLOTS = (W4G_HT / 2) * W4G_HH * 4 * 4;                           // (frame pair, halo row, tile, channel quad) = 480

typedef _Float16 w4g_half4 __attribute__((ext_vector_type(4)));
typedef float w4g_f4 __attribute__((ext_vector_type(4)));

// what one producer lane keeps for ONE frame of its slot: the raw inputs of the six positions 4j-1 .. 4j+4
template <bool SPADE> struct W4GenIn {
    w4g_f4 x[SPADE ? 4 : 6];   // SPADE (us = 2): the six positions read four low-resolution columns 2j-1 .. 2j+2
};

// a producer lane's slot in round `round` of a chunk (the second round has 224 slots: act = false beyond)
struct W4GenSlot { int q, ij, ih, third; bool act; };
__device__ __forceinline__ W4GenSlot w4g_slot(int ptid, int round) {
    const int slot = round * W4G_PROD + ptid;
    W4GenSlot s;
    s.act = slot < W4G_SLOTS;
    const int sl = s.act ? slot : 0;
    s.q = sl & 3; s.ij = (sl >> 2) & 3;
    const int rest = sl >> 4;                       // 0 .. 29 = third * 10 + ih
    s.third = (rest >= W4G_HH) + (rest >= 2 * W4G_HH);
    s.ih = rest - s.third * W4G_HH;
    return s;
}

// frame `fi` of a chunk = (round fi >> 1, frame fi & 1 of the slot's pair): request its inputs
template <bool SPADE>
__device__ __forceinline__ void w4g_load(const W4Args& a, const W4Brick& k, const float* xs, int ptid, int ch, int fi, W4GenIn<SPADE>& in) {
    const W4GenSlot s = w4g_slot(ptid, fi >> 1);
    if (!s.act) return;
    const int h = min(max(k.h0 - 1 + s.ih, 0), a.H - 1);
    const int t = min(max(k.t0 - 1 + 2 * s.third + (fi & 1), 0), a.T - 1);
    const int c0 = ch * W4_KC + 4 * s.q;
    const int j = k.j0 + s.ij;
    if constexpr (SPADE) {
        const int Hl = a.H >> 1, Wl = a.W >> 1;
        const unsigned rowb = (unsigned)((t * Hl + (h >> 1)) * Wl) * (unsigned)a.Cin + (unsigned)c0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int wl = min(max(2 * j - 1 + p, 0), Wl - 1);
            in.x[p] = *reinterpret_cast<const w4g_f4*>(xs + rowb + (unsigned)wl * (unsigned)a.Cin);
        }
    } else {
        const unsigned rowb = (unsigned)((t * a.H + h) * a.W) * (unsigned)a.Cin + (unsigned)c0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int w = min(max(4 * j - 1 + p, 0), a.W - 1);
            in.x[p] = *reinterpret_cast<const w4g_f4*>(xs + rowb + (unsigned)w * (unsigned)a.Cin);
        }
    }
}

// One frame of one chunk: PASS 0 writes the planes 0..3 of the lane's V row, PASS 1 the planes 4, 5.  vbuf: LDS address of the
// target buffer's row 0.  The arithmetic follows modulate_wino4_kernel (i2v_dec.hip) expression for expression.
template <bool SPADE, int PASS>
__device__ __forceinline__ void w4g_frame(const W4Args& a, const W4GenArgs& g, const W4Brick& k, int ptid, int ch, int fi,
                                          const W4GenIn<SPADE>& in, char* vbuf, float& vmaxd, float& vmaxv) {
    const W4GenSlot s = w4g_slot(ptid, fi >> 1);
    if (!s.act) return;
    const int c0 = ch * W4_KC + 4 * s.q;
    const int hq = k.h0 - 1 + s.ih;
    const int j = k.j0 + s.ij;
    const int fr = 2 * s.third + (fi & 1);                 // halo frame 0 .. 5
    const bool ok = (unsigned)hq < (unsigned)a.H && (unsigned)(k.t0 - 1 + fr) < (unsigned)a.T;
    const bool w0ok = j > 0, w5ok = j < a.J - 1;          // positions 4j - 1 / 4j + 4 inside the row (else the conv's zero padding)
    float ca[4], cb[4];
    {
        const float4* cp = reinterpret_cast<const float4*>(g.coef + (long)k.b0 * a.Cin + c0);
        const float4 ab0 = cp[0], ab1 = cp[1];
        ca[0] = ab0.x; cb[0] = ab0.y; ca[1] = ab0.z; cb[1] = ab0.w; ca[2] = ab1.x; cb[2] = ab1.y; ca[3] = ab1.z; cb[3] = ab1.w;
    }
    float d[6][4];
    if constexpr (SPADE) {
        const int h = min(max(hq, 0), a.H - 1);
        const float* gbs = g.gb + ((long)k.b0 * a.H + h) * a.W * 2 * a.Cin + c0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int w = min(max(4 * j - 1 + p, 0), a.W - 1);
            const float4 ga = *reinterpret_cast<const float4*>(gbs + (unsigned)w * (unsigned)(2 * a.Cin));
            const float4 be = *reinterpret_cast<const float4*>(gbs + (unsigned)w * (unsigned)(2 * a.Cin) + a.Cin);
            const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
            const w4g_f4 xv = in.x[(p + 1) >> 1];   // position 4j-1+p reads the low-res column (4j-1+p) >> 1 = 2j-1 + ((p+1) >> 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (x ca + cb) ga + be as the writer folds it: a = ca ga, b = cb ga + be
                const float r = fmaf(xv[c], ca[c] * gav[c], fmaf(cb[c], gav[c], bev[c]));
                d[p][c] = fmaxf(r, 0.2f * r);   // == (r < 0 ? 0.2 r : r) for every finite r
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float r = fmaf(in.x[p][c], ca[c], cb[c]);
                d[p][c] = fmaxf(r, 0.2f * r);   // == (r < 0 ? 0.2 r : r) for every finite r
            }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!w0ok) d[0][c] = 0.f;
        if (!w5ok) d[5][c] = 0.f;
    }
    const int rrow = (fr * W4G_HH + s.ih) * 4 + s.ij;   // row inside a plane
    const int key = (rrow >> 2) & 3;
    const int qsl = 2 * (s.q >> 1);
    char* rp = vbuf + rrow * 64 + (s.q & 1) * 8;
    if (!ok) {   // a halo row outside the tensor (t or h): the conv's zero padding
#pragma unroll
        for (int xq = 0; xq < (PASS ? 2 : 4); ++xq) {
            char* o = rp + xq * (W4G_PLANE * 64);
            *reinterpret_cast<uint2*>(o + ((qsl ^ key) << 4)) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(o + (((qsl ^ key) ^ 1) << 4)) = make_uint2(0u, 0u);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        vmaxd = fmaxf(vmaxd, fmaxf(fabsf(d[1][c]), fabsf(d[2][c])));    // own positions (the neighbours' d1..d4 cover d0 / d5)
        vmaxd = fmaxf(vmaxd, fmaxf(fabsf(d[3][c]), fabsf(d[4][c])));
    }
#pragma unroll
    for (int xq = (PASS ? 4 : 0); xq < (PASS ? 6 : 4); ++xq) {
        w4g_half4 ph, pl;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v;
            if (xq == 0) v = fmaf(4.f, d[0][c], fmaf(-5.f, d[2][c], d[4][c]));
            else if (xq == 1) v = fmaf(-4.f, d[1][c] + d[2][c], d[3][c] + d[4][c]);
            else if (xq == 2) v = fmaf(4.f, d[1][c] - d[2][c], d[4][c] - d[3][c]);
            else if (xq == 3) v = fmaf(2.f, d[3][c] - d[1][c], d[4][c] - d[2][c]);
            else if (xq == 4) v = fmaf(2.f, d[1][c] - d[3][c], d[4][c] - d[2][c]);
            else v = fmaf(4.f, d[1][c], fmaf(-5.f, d[3][c], d[5][c]));
            vmaxv = fmaxf(vmaxv, fabsf(v));
            const _Float16 hh = (_Float16)v;
            ph[c] = hh;
            pl[c] = (_Float16)(v - (float)hh);
        }
        char* o = rp + (xq - (PASS ? 4 : 0)) * (W4G_PLANE * 64);
        *reinterpret_cast<w4g_half4*>(o + ((qsl ^ key) << 4)) = ph;
        *reinterpret_cast<w4g_half4*>(o + (((qsl ^ key) ^ 1) << 4)) = pl;
    }
}

// a whole chunk (two rounds x two frames) of one pass into `vbuf`.  `ina` holds frame 0's inputs on entry (requested one frame
// ahead); on exit it holds frame 0 of chunk `nch` (-1: none): the request of a frame is always in flight while the frame in front
// of it is computed
template <bool SPADE, int PASS>
__device__ __forceinline__ void w4g_chunk(const W4Args& a, const W4GenArgs& g, const W4Brick& k, const float* xs, int ptid, int ch, int nch,
                                          W4GenIn<SPADE>& ina, char* vbuf, float& vmaxd, float& vmaxv) {
    W4GenIn<SPADE> inb;
    w4g_load<SPADE>(a, k, xs, ptid, ch, 1, inb);
    w4g_frame<SPADE, PASS>(a, g, k, ptid, ch, 0, ina, vbuf, vmaxd, vmaxv);
    w4g_load<SPADE>(a, k, xs, ptid, ch, 2, ina);
    w4g_frame<SPADE, PASS>(a, g, k, ptid, ch, 1, inb, vbuf, vmaxd, vmaxv);
    w4g_load<SPADE>(a, k, xs, ptid, ch, 3, inb);
    w4g_frame<SPADE, PASS>(a, g, k, ptid, ch, 2, ina, vbuf, vmaxd, vmaxv);
    if (nch >= 0) w4g_load<SPADE>(a, k, xs, ptid, nch, 0, ina);
    w4g_frame<SPADE, PASS>(a, g, k, ptid, ch, 3, inb, vbuf, vmaxd, vmaxv);
}

template <int NT, bool SPADE>
__global__ __launch_bounds__(W4G_THREADS, 1) void conv_wino4g_f16x3_kernel(W4Args a, W4GenArgs g) {
    static_assert(NT == 9, "the generating kernel exists for the 3x3x3 convs of the last level (no temporal up-sampling in front)");
    using Geo = W4Geo<512>;
    constexpr int NTH = 512, WMA = 2, WMB = 1, KT = NT / 3, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HH = a.TH + 2;
    const int plane = (a.TT + KT - 1) * HH * 4;
    const int nblk = a.CoutPad >> 5;
    const W4Brick bk = w4_decode<32>(a, (int)blockIdx.x);
    int* tpos = reinterpret_cast<int*>(smem + a.tofs);
    int* tres = tpos + Geo::TILES;
    const int rA0 = 0, rA1 = Geo::ROWS_A, rB0 = 0, rB1 = Geo::ROWS_B;
    int w4_tlv_ = (int)blockIdx.x;
    (void)w4_tlv_;

    if (wave >= NW) {
        // ------------------------------------------------------------------------------------------------ producer role
        const int ptid = tid - NTH;
        const float* xs = g.x + (long)bk.b0 * a.T * (a.H / g.us) * (a.W / g.us) * a.Cin;   // this brick's sample
        float vmaxd = 0.f, vmaxv = 0.f;
        W4GenIn<SPADE> in0;
        w4g_load<SPADE>(a, bk, xs, ptid, 0, 0, in0);
        // pass A: chunk 0 in front of the first barrier, then chunk c + 1 (or pass B's chunk 0) underneath the MFMA waves' chunk c
        w4g_chunk<SPADE, 0>(a, g, bk, xs, ptid, 0, a.nchunk > 1 ? 1 : 0, in0, smem + rA0 * 64, vmaxd, vmaxv);
        __syncthreads();
        for (int c = 0; c < a.nchunk; ++c) {
            if (c + 1 < a.nchunk) w4g_chunk<SPADE, 0>(a, g, bk, xs, ptid, c + 1, c + 2 < a.nchunk ? c + 2 : 0, in0, smem + (((c + 1) & 1) ? rA1 : rA0) * 64, vmaxd, vmaxv);
            else w4g_chunk<SPADE, 1>(a, g, bk, xs, ptid, 0, a.nchunk > 1 ? 1 : -1, in0, smem + rB0 * 64, vmaxd, vmaxv);
            __syncthreads();
        }
        // pass B
        __syncthreads();
        for (int c = 0; c < a.nchunk; ++c) {
            if (c + 1 < a.nchunk) w4g_chunk<SPADE, 1>(a, g, bk, xs, ptid, c + 1, c + 2 < a.nchunk ? c + 2 : -1, in0, smem + (((c + 1) & 1) ? rB1 : rB0) * 64, vmaxd, vmaxv);
            __syncthreads();
        }
        // range / underflow guard of the operand format (what modulate_wino4_kernel publishes)
        if (g.range_flag && !(vmaxv <= 65504.f)) atomicOr(g.range_flag, 1);
        if (g.umax) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) vmaxd = fmaxf(vmaxd, __shfl_xor(vmaxd, o));
            if ((tid & 63) == 0) {
                const int bits = __float_as_int(vmaxd);
                if (bits > *reinterpret_cast<volatile int*>(g.umax)) atomicMax(g.umax, bits);
            }
        }
        // the epilogue's barriers (the exchange through LDS and the statistics tail belong to the MFMA-role threads)
        __syncthreads();
        __syncthreads();
        if (a.stats) __syncthreads();
        return;
    }

    // ---------------------------------------------------------------------------------------------------- MFMA role
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    if (tid < Geo::TILES) {   // output / residual positions of the brick's tiles (w4_tables' first part)
        int m = tid;
        const int ij = m & 3; m >>= 2;
        const int ih = m & (a.TH - 1); m >>= a.th_shift;
        const int t = bk.t0 + m, h = bk.h0 + ih, w = 4 * (bk.j0 + ij);
        tpos[tid] = ((bk.b0 * a.T + t) * a.H + h) * a.W + w;
        const int rbase = ((bk.b0 * (a.T >> a.rt_shift) + (t >> a.rt_shift)) * (a.H >> a.rs_shift) + (h >> a.rs_shift)) * (a.W >> a.rs_shift);
#pragma unroll
        for (int c = 0; c < 4; ++c) tres[4 * tid + c] = rbase + ((w + c) >> a.rs_shift);
    }
    const int n0 = bk.ntile * 32, b0 = bk.b0;
    const char* wbase = a.wp;
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wp), 0, 0, 0x00020000);   // (unused: no V requests)
    const W4Next none{nullptr, 0u, 0u, 0};
    const int xa = wave & 3, mha = (wave >> 2) * 64;
    f32x16 accA[WMA];
    {
        int arow[WMA];
#pragma unroll
        for (int wm = 0; wm < WMA; ++wm) {
            int m = mha + wm * 32 + l31;
            const int ij = m & 3; m >>= 2;
            const int ih = m & (a.TH - 1); m >>= a.th_shift;
            arow[wm] = xa * plane + (m * HH + ih) * 4 + ij;
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[wm][r] = 0.f;
        }
        w4_pass<NT, WMA, 0, 0, NTH, false, false, 0, false>(a, smem, nullptr, nullptr, accA, arow, wbase + ((long)xa * nblk + (n0 >> 5)) * 2048, HH, tid, lane,
                                                            wave, rA0, rA1, none, [] {}, w4_tlv_, vrsrc);
    }
    const int xb = wave & 1, mhb = (wave >> 1) * 32;
    f32x16 accB[WMB];
    {
        int arow[WMB];
#pragma unroll
        for (int wm = 0; wm < WMB; ++wm) {
            int m = mhb + wm * 32 + l31;
            const int ij = m & 3; m >>= 2;
            const int ih = m & (a.TH - 1); m >>= a.th_shift;
            arow[wm] = xb * plane + (m * HH + ih) * 4 + ij;
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[wm][r] = 0.f;
        }
        w4_pass<NT, WMB, 0, 0, NTH, true, false, 0, false>(a, smem, nullptr, nullptr, accB, arow, wbase + ((long)(4 + xb) * nblk + (n0 >> 5)) * 2048, HH, tid, lane,
                                                           wave, rB0, rB1, none, [] {}, w4_tlv_, vrsrc);
    }
    // ---- epilogue of the 512-thread 32-channel kernel (one channel half): E = [6 planes][128 tiles][32 channels] fp32 over both V regions
    constexpr int NQ = 8, TPI = NTH / NQ, ET = Geo::TILES, NIT = ET / TPI;
    float* E = reinterpret_cast<float*>(smem);
    double* S = reinterpret_cast<double*>(reinterpret_cast<char*>(E) + 6 * ET * 32 * 4);
    const int n4 = tid % NQ;
    const int e3 = kg * 96, e5 = kg * 160;
    {
        const int n = n0 + 4 * n4;
        const bool ncol = n < a.Cout;
        double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && ncol) bias = *reinterpret_cast<const float4*>(a.bias + n);
        const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
        f32x4 rres[NIT][4];
        __syncthreads();   // (tpos / tres are published by the tap loops' barriers long ago; this one: the V bricks are no longer read)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                rres[it][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.res && ncol) rres[it][c] = *reinterpret_cast<const f32x4*>(a.res + (long)tres[4 * (tid / NQ + TPI * it) + c] * a.Cout + n);
            }
#pragma unroll
        for (int wm = 0; wm < WMA; ++wm) {
            const int m0 = mha + wm * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = m0 + (r & 3) + 8 * (r >> 2);
                E[(xa * ET + c) * 32 + l31 + ((r & 1) ? e3 : e5)] = accA[wm][r];
            }
        }
#pragma unroll
        for (int wm = 0; wm < WMB; ++wm) {
            const int m0 = mhb + wm * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = m0 + (r & 3) + 8 * (r >> 2);
                E[((4 + xb) * ET + c) * 32 + l31 + ((r & 1) ? e3 : e5)] = accB[wm][r];
            }
        }
        __syncthreads();
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
                const long p = tpos[tid / NQ + TPI * it];
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(a.out + (p + c) * a.Cout + n) = rres[it][c];
            }
        }
        if (a.stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum[j] = wave_xor_add_f64<8>(ssum[j]); ssq[j] = wave_xor_add_f64<8>(ssq[j]);
                ssum[j] = wave_xor_add_f64<16>(ssum[j]); ssq[j] = wave_xor_add_f64<16>(ssq[j]);
                ssum[j] = wave_xor_add_f64<32>(ssum[j]); ssq[j] = wave_xor_add_f64<32>(ssq[j]);
            }
            if (lane < NQ) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S[(wave * 32 + 4 * lane + j) * 2] = ssum[j];
                    S[(wave * 32 + 4 * lane + j) * 2 + 1] = ssq[j];
                }
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        if (wave == 0 && lane < 32 && n0 + lane < a.Cout) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s0 += S[(w * 32 + lane) * 2];
                s1 += S[(w * 32 + lane) * 2 + 1];
            }
            double* dst = a.stats + ((long)b0 * a.Cout + n0 + lane) * 2;
            atomicAdd(dst, s0);
            atomicAdd(dst + 1, s1);
        }
    }
}

bool wino4g_supported(int cout, int cin, int T, int H, int W, int us) {
    if (cout != 32 || cin % (2 * W4_KC) || (us != 1 && us != 2) || T % W4G_TT || H % W4G_TH || W % 16 || (us == 2 && (H & 1))) return false;
    int TT, TH;
    return wino4_tiling(T, H, W, 3, &TT, &TH) && TT == W4G_TT && TH == W4G_TH;
}

int wino4g_forward(const Wino4Weights& wts, const float* x, const float* coef, const float* gb, int us, float* out, const float* res, int rt, int rs,
                   int B, int T, int H, int W, int epi, hipStream_t st, double* stats, int* range_flag, int* umax) {
    I2V_REQUIRE(wts.w.p && !wts.tdup && wts.KT == 3, I2V_E_STATE, "wino4g: needs 3x3x3 weights packed for the F(4,3) kernel");
    I2V_REQUIRE((epi & ~EPI_LRELU) == 0, I2V_E_INVALID, "wino4g: unsupported epilogue %d", epi);
    I2V_REQUIRE(x && coef && wino4g_supported(wts.Cout, wts.Cin, T, H, W, us) && (gb != nullptr) == (us == 2), I2V_E_INVALID,
                "wino4g: unsupported shape [%d,%d,%d] %d -> %d (us = %d, gb %p)", T, H, W, wts.Cin, wts.Cout, us, (const void*)gb);
    W4Args a{};
    a.in = nullptr; a.zeros = nullptr; a.wp = wts.w.as<char>(); a.bias = wts.bias.as<float>(); a.res = res; a.out = out;
    a.stats = stats;
    a.B = B; a.T = T; a.H = H; a.W = W; a.J = W / 4; a.Cin = wts.Cin; a.Cout = wts.Cout; a.CoutPad = wts.CoutPad; a.nchunk = wts.nchunk;
    a.tdup = 0;
    a.wset_stride = wts.set_bytes;
    a.rt = res ? rt : 1; a.rs = res ? rs : 1; a.epi = epi;
    I2V_REQUIRE((a.rt == 1 || a.rt == 2 || a.rt == 4) && (a.rs == 1 || a.rs == 2 || a.rs == 4), I2V_E_INVALID,
                "wino4g: residual up-sampling factors %d / %d (1, 2 or 4)", a.rt, a.rs);
    a.rt_shift = a.rt >> 1 == 2 ? 2 : a.rt >> 1; a.rs_shift = a.rs >> 1 == 2 ? 2 : a.rs >> 1;
    a.oscale = (float)std::ldexp(1.0, -wts.wexp);
    a.TT = W4G_TT; a.TH = W4G_TH; a.TJ = 4; a.nbT = T / a.TT; a.nbH = H / a.TH; a.nbJ = a.J / 4;
    a.th_shift = 3;
    a.hh_magic = ((1 << 20) + a.TH + 1) / (a.TH + 2);
    a.order = W4_DEFAULT_ORDER;
    const long nblk = (long)B * a.nbT * a.nbH * a.nbJ * (a.CoutPad / 32);
    I2V_REQUIRE(nblk > 0 && nblk < (1L << 30), I2V_E_INVALID, "wino4g: grid of %ld workgroups", nblk);
    I2V_REQUIRE((long)T * H * W * wts.Cin < (1L << 31) && (long)H * W * 2 * wts.Cin < (1L << 31) && (long)B * T * H * W < (1L << 31), I2V_E_INVALID,
                "wino4g: tensor too large for the 32-bit offsets of this kernel");
    a.nvirt = (int)nblk;
    a.tofs = 2 * W4_ROWS_A * 64;
    W4GenArgs g{x, reinterpret_cast<const float2*>(coef), gb, us, range_flag, umax};
    const size_t lds = (size_t)a.tofs + 5 * W4Geo<512>::TILES * 4;
    static bool attr_set[2][I2V_MAX_DEV] = {};
    if (gb) {
        auto kern = conv_wino4g_f16x3_kernel<9, true>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set[1])) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(W4G_THREADS), lds, st, a, g);
    } else {
        auto kern = conv_wino4g_f16x3_kernel<9, false>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set[0])) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(W4G_THREADS), lds, st, a, g);
    }
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

#ifdef W4_TAPTIME
void w4_taptime_report() {
    std::vector<unsigned long long> h(2 * 8 * 18 * 2);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(w4_tt), h.size() * 8);
    for (int p = 0; p < 2; ++p) {
        printf("   pass %c: mean ticks (10 ns) from the start of tap slot U-1 to the start of tap slot U, per wave\n", p ? 'B' : 'A');
        for (int w = 0; w < 8; ++w) {
            printf("      wave %d:", w);
            for (int u = 0; u < 18; ++u) {
                const unsigned long long t = h[((p * 8 + w) * 18 + u) * 2], c = h[((p * 8 + w) * 18 + u) * 2 + 1];
                printf(" %5.1f", c ? (double)t / (double)c : 0.0);
            }
            printf("\n");
        }
    }
}
#endif

#ifdef W4_TIMELINE
void w4_timeline_report(unsigned nwg) {
    if (nwg > 8192) nwg = 8192;
    std::vector<unsigned long long> h(8192 * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(w4_tl), h.size() * 8);
    const char* nm[6] = {"tables + first V brick", "pass A loop", "hand-over to pass B", "pass B loop", "epilogue half 0", "epilogue half 1"};
    double sum[6] = {}, sub[5] = {}, tot = 0;
    unsigned long long lo = ~0ull, hi = 0;
    unsigned cnt = 0;
    for (unsigned w = 0; w < nwg; ++w) {
        unsigned long long t[16];
        for (int i = 0; i < 16; ++i) t[i] = h[w * 16 + i];
        if (!t[6]) t[6] = t[5];   // 32-channel workgroups have ONE epilogue half: stamp 6 is never written (it used to wrap to 1.8e17)
        if (!t[0] || !t[7] || t[7] < t[0]) continue;   // workgroup not stamped
        ++cnt;
        for (int i = 0; i < 6; ++i) sum[i] += (double)(t[i + 1] - t[i]);
        sub[0] += (double)(t[8] - t[4]); sub[1] += (double)(t[9] - t[8]); sub[2] += (double)(t[10] - t[9]); sub[3] += (double)(t[5] - t[10]);
        tot += (double)(t[7] - t[0]);
        lo = std::min(lo, t[0]); hi = std::max(hi, t[7]);
        sub[4] += (double)(t[7] - t[6]);
    }
    if (!cnt) { printf("   F(4,3) timeline: no stamped workgroups\n"); return; }
    const unsigned nall = cnt;
    nwg = cnt;   // (means over the stamped workgroups)
    printf("   F(4,3) timeline over %u workgroups (us, 100 MHz clock): total %.2f per workgroup; kernel span %.1f = %.2f per workgroup slot of 256 CUs\n",
           nwg, tot / nwg / 100.0, (double)(hi - lo) / 100.0, (double)(hi - lo) / 100.0 / (nall / 256.0));
    for (int i = 0; i < 6; ++i) printf("      %-24s %7.2f\n", nm[i], sum[i] / nwg / 100.0);
    printf("      epilogue half 0 = residual requests + first barrier %.2f | accumulators -> LDS + barrier %.2f | transform, bias, residual, stores issued %.2f | statistics (per-wave part) %.2f; cross-wave sums + atomics of both halves %.2f\n",
           sub[0] / nwg / 100.0, sub[1] / nwg / 100.0, sub[2] / nwg / 100.0, sub[3] / nwg / 100.0, sub[4] / nwg / 100.0);
}
#endif

#ifdef W4_DECODE_SELFTEST
// Host-side self-test of the virtual-workgroup -> (brick, channel tile, frame parity) map (tests/test_host_cpu.py compiles this
// file with -DW4_DECODE_SELFTEST for the host only): for every order and a sweep of geometries the map must be a bijection onto
// {samples} x {t bricks} x {h bricks} x {w bricks} x {channel tiles} x {parities}.
template <int BN>
static long w4_decode_check(int B, int nbT, int nbH, int nbJ, int coutpad, int tdup, int order) {
    W4Args a{};
    a.B = B; a.nbT = nbT; a.nbH = nbH; a.nbJ = nbJ; a.CoutPad = coutpad; a.tdup = tdup; a.order = order; a.TT = 4; a.TH = 8;
    const int nNt = coutpad / BN, npar = tdup ? 2 : 1;
    a.nvirt = B * nbT * nbH * nbJ * nNt * npar;
    std::vector<char> seen((size_t)a.nvirt, 0);
    long bad = 0;
    for (int v = 0; v < a.nvirt; ++v) {
        const W4Brick k = w4_decode<BN>(a, v);
        const int bt = k.t0 / a.TT, bh = k.h0 / a.TH, bj = k.j0 / 4;
        if (k.par < 0 || k.par >= npar || k.ntile < 0 || k.ntile >= nNt || k.b0 < 0 || k.b0 >= B || bt < 0 || bt >= nbT || bh < 0 || bh >= nbH ||
            bj < 0 || bj >= nbJ) { ++bad; continue; }
        const size_t id = (((((size_t)k.b0 * nbT + bt) * nbH + bh) * nbJ + bj) * nNt + k.ntile) * npar + k.par;
        if (seen[id]) ++bad;
        seen[id] = 1;
    }
    return bad;
}
}  // namespace i2v
int main() {
    long bad = 0, cases = 0;
    for (int order = 0; order < 3; ++order)
        for (int B : {1, 2, 3, 8, 13, 64})
            for (int nbT : {1, 2, 4})
                for (int nbH : {1, 2, 8, 16})
                    for (int nbJ : {1, 2, 4, 8})
                        for (int tdup = 0; tdup < 2; ++tdup)
                            for (int cp : {32, 64, 128, 512}) {
                                bad += i2v::w4_decode_check<32>(B, nbT, nbH, nbJ, cp, tdup, order);
                                if (cp % 64 == 0) bad += i2v::w4_decode_check<64>(B, nbT, nbH, nbJ, cp, tdup, order);
                                ++cases;
                            }
    printf("w4_decode self-test: %ld geometries, %ld bad\n", cases, bad);
    return bad ? 1 : 0;
}
#else
}  // namespace i2v
#endif
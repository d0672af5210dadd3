// The F(4,3) kernel for the thin layers of the 128 x 128 configs with the operand generated in the kernel (round 6).  Own translation
// unit: the producer role is plain fp32 vector code, compiled without the SLP vectoriser (packed fp32 instructions cost more than
// the two scalar ones they replace next to MFMAs: MI355X_MICROARCH.md), which must not touch the epilogues of i2v_conv16w4.hip.
#define W4_NO_INSTRUMENT
#include "i2v_conv16w4_dev.h"

namespace i2v {

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

// What a producer lane keeps for the whole brick, per round (its slot (frame pair, halo row, tile, channel quad) of that round):
// byte offsets inside the sample (chunk 0; the chunk is added to the scalar base), the V row, validity bits.
template <bool SPADE> struct W4GenLane {
    unsigned xo[2][2];   // [round][frame]: byte offset of position p = 1 (w = 4j; SPADE: low-res column 2j) in the input, t / h clamped
    unsigned go[2];      // SPADE: byte offset of (h, w = 4j) in the gamma' | beta maps (h clamped)
    int e0[2], e5[2];    // byte offsets of the outer positions relative to p = 1: p = 0 -> -CIN floats, p = 5 -> 4 CIN floats; at the
                         // row ends they point at a valid neighbour and the value is zeroed (SPADE: low-res columns 2j-1 / 2j+2)
    int lds[2];          // byte offset of the V row of frame 0 of the pair inside a plane, + the quad's 8-byte half (frame 1: + 40 rows)
    int fl[2];           // bit 0 slot active, 1 / 2 frame 0 / 1 inside the tensor (t and h), 3 / 4 positions 4j-1 / 4j+4 inside the row,
                         // 8..9 swizzle key of frame 0's row (frame 1: key + 2), 12..13 = 2 * (q >> 1)
};

template <int CIN, bool SPADE>
__device__ __forceinline__ void w4g_lane_init(const W4Args& a, const W4Brick& k, int ptid, W4GenLane<SPADE>& L) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int slot = r * W4G_PROD + ptid;
        const bool act = slot < W4G_SLOTS;
        const int sl = act ? slot : 0;
        const int q = sl & 3, ij = (sl >> 2) & 3, rest = sl >> 4;   // rest = third * 10 + ih
        const int third = (rest >= W4G_HH) + (rest >= 2 * W4G_HH);
        const int ih = rest - third * W4G_HH;
        const int hq = k.h0 - 1 + ih, h = min(max(hq, 0), a.H - 1);
        const int j = k.j0 + ij;
        const bool hok = (unsigned)hq < (unsigned)a.H;
        const bool w0ok = j > 0, w5ok = j < a.J - 1;
        int fl = (act ? 1 : 0) | (w0ok ? 8 : 0) | (w5ok ? 16 : 0) | ((2 * (q >> 1)) << 12);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int tq = k.t0 - 1 + 2 * third + f, t = min(max(tq, 0), a.T - 1);
            if (hok && (unsigned)tq < (unsigned)a.T) fl |= 2 << f;
            if constexpr (SPADE) L.xo[r][f] = (unsigned)((((t * (a.H >> 1) + (h >> 1)) * (a.W >> 1) + 2 * j) * CIN + 4 * q) * 4);
            else L.xo[r][f] = (unsigned)((((t * a.H + h) * a.W + 4 * j) * CIN + 4 * q) * 4);
        }
        if constexpr (SPADE) {
            L.go[r] = (unsigned)((((h * a.W) + 4 * j) * 2 * CIN + 4 * q) * 4);
            L.e0[r] = w0ok ? -CIN * 4 : 0;          // low-res column 2j - 1 (position 4j - 1), else column 2j (zeroed)
            L.e5[r] = w5ok ? 2 * CIN * 4 : CIN * 4; // low-res column 2j + 2 (position 4j + 4), else column 2j + 1 (zeroed)
        } else {
            L.go[r] = 0;
            L.e0[r] = w0ok ? -CIN * 4 : 0;
            L.e5[r] = w5ok ? 4 * CIN * 4 : 3 * CIN * 4;
        }
        const int rrow = ((2 * third) * W4G_HH + ih) * 4 + ij;
        fl |= ((rrow >> 2) & 3) << 8;
        L.lds[r] = rrow * 64 + (q & 1) * 8;
        L.fl[r] = fl;
    }
}

// the raw inputs of ONE frame of a slot.  ADAIN: the six positions 4j-1 .. 4j+4; SPADE (us = 2): the four low-resolution columns 2j-1 .. 2j+2
template <bool SPADE> struct W4GenIn { w4g_f4 x[SPADE ? 4 : 6]; };

// request frame `fi` (round fi >> 1, frame fi & 1) of the chunk whose channels start at `xc` (= sample base + 16 floats per chunk)
// Measurement builds of tools/conv16w_check (results WRONG): -DW4G_ABLATE=1 the producers generate nothing (the MFMA role alone in the
// 12-wave workgroup), 2 no global loads (the inputs are whatever the registers hold), 3 no LDS stores, 4 loads only (no arithmetic)
#ifndef W4G_ABLATE
#define W4G_ABLATE 0
#endif
template <int CIN, bool SPADE>
__device__ __forceinline__ void w4g_load(const char* xc, const W4GenLane<SPADE>& L, int fi, W4GenIn<SPADE>& in) {
    const int r = fi >> 1, f = fi & 1;
    if (!(L.fl[r] & 1)) return;
    if constexpr (W4G_ABLATE == 1 || W4G_ABLATE == 2) {
#pragma unroll
        for (int p = 0; p < (SPADE ? 4 : 6); ++p) asm volatile("" : "+v"(in.x[p]));
        return;
    }
    const char* p1 = xc + L.xo[r][f];
    if constexpr (SPADE) {
        in.x[0] = *reinterpret_cast<const w4g_f4*>(p1 + L.e0[r]);
        in.x[1] = *reinterpret_cast<const w4g_f4*>(p1);
        in.x[2] = *reinterpret_cast<const w4g_f4*>(p1 + CIN * 4);
        in.x[3] = *reinterpret_cast<const w4g_f4*>(p1 + L.e5[r]);
    } else {
        in.x[0] = *reinterpret_cast<const w4g_f4*>(p1 + L.e0[r]);
#pragma unroll
        for (int p = 1; p < 5; ++p) in.x[p] = *reinterpret_cast<const w4g_f4*>(p1 + (p - 1) * CIN * 4);
        in.x[5] = *reinterpret_cast<const w4g_f4*>(p1 + L.e5[r]);
    }
}

// One frame of one chunk: PASS 0 writes the planes 0..3 of the lane's V row, PASS 1 the planes 4, 5.  vbuf: LDS address of the
// target buffer's row 0; cf = the chunk's (A, B) pairs of the lane's four channels; gc = the sample's gamma' | beta maps at the
// chunk's channels (SPADE).  The arithmetic follows modulate_wino4_kernel (i2v_dec.hip) expression for expression.
template <int CIN, bool SPADE, int PASS>
__device__ __forceinline__ void w4g_frame(const W4GenLane<SPADE>& L, int fi, const float (&ca)[4], const float (&cb)[4], const char* gc,
                                          const W4GenIn<SPADE>& in, char* vbuf, float& vmaxd, float& vmaxv) {
    const int r = fi >> 1, f = fi & 1;
    const int fl = L.fl[r];
    if (!(fl & 1)) return;
    if constexpr (W4G_ABLATE == 1) return;
    if constexpr (W4G_ABLATE == 4) {
#pragma unroll
        for (int p = 0; p < (SPADE ? 4 : 6); ++p) asm volatile("" ::"v"(in.x[p]));
        return;
    }
    const int key = ((fl >> 8) + 2 * f) & 3, qsl = (fl >> 12) & 3;
    char* rp = vbuf + L.lds[r] + f * (W4G_HH * 4 * 64);
    if (!(fl & (2 << f))) {   // a halo row outside the tensor (t or h): the conv's zero padding
#pragma unroll
        for (int xq = 0; xq < (PASS ? 2 : 4); ++xq) {
            char* o = rp + xq * (W4G_PLANE * 64);
            *reinterpret_cast<uint2*>(o + ((qsl ^ key) << 4)) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(o + (((qsl ^ key) ^ 1) << 4)) = make_uint2(0u, 0u);
        }
        return;
    }
    float d[6][4];
    if constexpr (SPADE) {
        const char* g1 = gc + L.go[r];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            // position 4j-1+p: maps at byte offset (p - 1) * 2 CIN floats from w = 4j (the row ends read a valid neighbour, zeroed below)
            const int po = p == 0 ? ((fl & 8) ? -2 * CIN * 4 : 0) : p == 5 ? ((fl & 16) ? 4 * 2 * CIN * 4 : 3 * 2 * CIN * 4) : (p - 1) * 2 * CIN * 4;
            const float4 ga = *reinterpret_cast<const float4*>(g1 + po);
            const float4 be = *reinterpret_cast<const float4*>(g1 + po + CIN * 4);
            const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
            const w4g_f4 xv = in.x[(p + 1) >> 1];   // position 4j-1+p reads the low-res column (4j-1+p) >> 1 = 2j-1 + ((p+1) >> 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float rr = fmaf(xv[c], ca[c] * gav[c], fmaf(cb[c], gav[c], bev[c]));   // (x ca + cb) ga + be as the writer folds it
                d[p][c] = fmaxf(rr, 0.2f * rr);   // == (r < 0 ? 0.2 r : r) for every finite r
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float rr = fmaf(in.x[p][c], ca[c], cb[c]);
                d[p][c] = fmaxf(rr, 0.2f * rr);
            }
    }
    if (!(fl & 8)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) d[0][c] = 0.f;
    }
    if (!(fl & 16)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) d[5][c] = 0.f;
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
            // The fp32 value goes through a register the compiler cannot see into.  Otherwise hipcc (-ffp-contract=fast) fuses the
            // transform's last fma with the conversion into ONE v_fma_mixlo_f16: hi = fp16(4 d0 + t) rounded ONCE, where the writer
            // (and the source) round twice, fp16(fp32(4 d0 + t)).  Both are valid splits -- lo = v - hi absorbs the difference -- but
            // the hi parts then differ in 1 of ~10^4 values and ~0.5 % of the conv outputs move by one ulp against the writer path.
            asm volatile("" : "+v"(v));
            vmaxv = fmaxf(vmaxv, fabsf(v));
            const _Float16 hh = (_Float16)v;
            ph[c] = hh;
            pl[c] = (_Float16)(v - (float)hh);
        }
        char* o = rp + (xq - (PASS ? 4 : 0)) * (W4G_PLANE * 64);
        if constexpr (W4G_ABLATE == 3) { asm volatile("" ::"v"(ph), "v"(pl), "v"(o)); continue; }
        *reinterpret_cast<w4g_half4*>(o + ((qsl ^ key) << 4)) = ph;
        *reinterpret_cast<w4g_half4*>(o + (((qsl ^ key) ^ 1) << 4)) = pl;
    }
}

// a whole chunk (two rounds x two frames) of one pass into `vbuf`.  `ina` holds frame 0's inputs on entry (requested one frame
// ahead); on exit it holds frame 0 of chunk `nch` (-1: none): the request of a frame is always in flight while the frame in front
// of it is computed.  xs / cfs / gbs: the sample's input, (A, B) pairs and gamma' | beta maps (chunk 0, the lane's quad folded in).
template <int CIN, bool SPADE, int PASS>
__device__ __forceinline__ void w4g_chunk(const W4GenLane<SPADE>& L, const char* xs, const float4* cfs, const char* gbs, int ch, int nch,
                                          W4GenIn<SPADE>& ina, char* vbuf, float& vmaxd, float& vmaxv) {
    const char* xc = xs + ch * (W4_KC * 4);
    const char* gc = gbs + ch * (W4_KC * 4);
    float ca[4], cb[4];
    {
        const float4 ab0 = cfs[ch * (W4_KC / 2)], ab1 = cfs[ch * (W4_KC / 2) + 1];
        ca[0] = ab0.x; cb[0] = ab0.y; ca[1] = ab0.z; cb[1] = ab0.w; ca[2] = ab1.x; cb[2] = ab1.y; ca[3] = ab1.z; cb[3] = ab1.w;
    }
    W4GenIn<SPADE> inb;
    w4g_load<CIN, SPADE>(xc, L, 1, inb);
    w4g_frame<CIN, SPADE, PASS>(L, 0, ca, cb, gc, ina, vbuf, vmaxd, vmaxv);
    w4g_load<CIN, SPADE>(xc, L, 2, ina);
    w4g_frame<CIN, SPADE, PASS>(L, 1, ca, cb, gc, inb, vbuf, vmaxd, vmaxv);
    w4g_load<CIN, SPADE>(xc, L, 3, inb);
    w4g_frame<CIN, SPADE, PASS>(L, 2, ca, cb, gc, ina, vbuf, vmaxd, vmaxv);
    if (nch >= 0) w4g_load<CIN, SPADE>(xs + nch * (W4_KC * 4), L, 0, ina);
    w4g_frame<CIN, SPADE, PASS>(L, 3, ca, cb, gc, inb, vbuf, vmaxd, vmaxv);
}

// MODE 0 / 1: the producer waves GENERATE the operand (ADAIN / SPADE form, above).  MODE 2 (LOADER): the operand is the V tensor
// modulate_wino4_kernel wrote, as for conv_wino4_f16x3_kernel, but it is REQUESTED by the four extra waves: the ablation of the
// generating kernel showed that the thin layers' tap loops run 21 % / 34 % faster when they issue no V request themselves (an LDS-DMA
// piece costs an MFMA wave 100-185 cycles of issue inside a phase that already carries its operand reads, MI355X_MICROARCH.md; a
// 32-channel wave has half the MFMAs per chunk of a 64-channel one to hide the same eight pieces under).  The loader waves do
// nothing else: 16 (pass A) / 8 (pass B) buffer_load ... lds per chunk from per-lane row offsets computed once per brick, one
// s_waitcnt, the chunk barrier.  Same V, same tap loops, same bits as the kernels of i2v_conv16w4.hip.
template <int NT, int CIN, int MODE>
__global__ __launch_bounds__(W4G_THREADS, 1) void conv_wino4g_f16x3_kernel(W4Args a, W4GenArgs g) {
    constexpr bool SPADE = MODE == 1, LOADER = MODE >= 2, VLOADER = MODE == 3;   // (MODE 3: the loader's requests go through VGPRs + ds_write_b128 instead of LDS-DMA)
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
        const int ptid = tid - NTH;
        if constexpr (LOADER) {
            // -------------------------------------------------------------------------------------------- loader role
            const int pw = wave - NW;                         // loader wave 0..3: rows [16 pw, 16 pw + 16) of every 64-row instruction group
            const int lrow = ptid >> 2;
            const unsigned vpiece = (unsigned)(((ptid & 3) ^ ((ptid >> 4) & 3)) * 16);   // 16-byte piece of the row (XOR swizzle: w4_pass)
            auto rowoff = [&](int r, int nplanes, int plane0) -> unsigned {   // byte offset of staged row r inside the sample's V (chunk 0)
                const int x = (r >= plane) + (r >= 2 * plane) + (r >= 3 * plane) + (r >= 4 * plane);
                int q = r - x * plane;
                const int ij = q & 3; q >>= 2;
                const int it = (int)(((unsigned)q * (unsigned)a.hh_magic) >> 20);   // q / HH
                const int ih = q - it * HH;
                const int t = bk.t0 + it - 1, h = bk.h0 + ih - 1, j = bk.j0 + ij;
                const bool ok = x < nplanes && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H;
                const int row = ((t * a.nchunk * 6 + plane0 + x) * a.H + h) * a.J + j;
                return (ok ? (unsigned)row << 6 : 0x80000000u) | vpiece;            // (out of range -> the load writes zeros: W4_PAD_ROW)
            };
            unsigned offA[16], offB[8];
#pragma unroll
            for (int u = 0; u < 16; ++u) offA[u] = rowoff(u * 64 + lrow, 4, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) offB[u] = rowoff(u * 64 + lrow, 2, 4);
            const long vsample = (long)a.T * a.nchunk * 6 * a.H * a.J * 64;
            const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in) + (long)bk.b0 * vsample, 0, (int)vsample, 0x00020000);
            const unsigned vchunk = (unsigned)__builtin_amdgcn_readfirstlane((int)(6u * (unsigned)a.H * (unsigned)a.J * 64u));
            const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
            const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)pw * 1024u));
#define W4L_LOAD(off_, soff_, dst_)                                                                                  \
    {                                                                                                                \
        unsigned keep_;                                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(off_), "s"(vr), "s"(dst_), "s"(soff_) : "memory");                          \
    }
            auto request_dma = [&](bool passB, int ch, int rbuf) {   // one chunk's brick into the buffer that starts at LDS row rbuf
                const unsigned so = (unsigned)ch * vchunk, db = dst0 + (unsigned)rbuf * 64u;
                if (!passB) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) W4L_LOAD(offA[u], so, db + (unsigned)(u * 4096))
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) W4L_LOAD(offB[u], so, db + (unsigned)(u * 4096))
                }
            };
            auto landed_dma = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
            // MODE 3: the same rows through the vector registers -- 16 / 8 buffer_load_dwordx4 in flight per loader lane, then one
            // ds_write_b128 each to the address the LDS-DMA form writes (buffer base + 16 x lane).  Is the thin layers' operand stream
            // bound by the LDS-DMA path (6-7 TB/s over the chip) or by what feeds it?
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 vbuf[16];
            int vcount = 0;
            unsigned vdst = 0;
            auto request_v = [&](bool passB, int ch, int rbuf) {
                const unsigned so = (unsigned)ch * vchunk;
                vdst = dst0 + (unsigned)rbuf * 64u + (unsigned)(tid & 63) * 16u;
                vcount = passB ? 8 : 16;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (u < 8 || !passB) vbuf[u] = __builtin_amdgcn_raw_buffer_load_b128(vr, passB ? offB[u & 7] : offA[u], so, 0);
            };
            auto landed_v = [&] {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (u < vcount)
                        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((unsigned long)(vdst + (unsigned)(u * 4096))) = vbuf[u];
            };
            auto request = [&](bool passB, int ch, int rbuf) { if constexpr (VLOADER) request_v(passB, ch, rbuf); else request_dma(passB, ch, rbuf); };
            auto landed = [&] { if constexpr (VLOADER) landed_v(); else landed_dma(); };
            request(false, 0, rA0);
            landed();
            __syncthreads();
            for (int c = 0; c < a.nchunk; ++c) {
                if (c + 1 < a.nchunk) request(false, c + 1, ((c + 1) & 1) ? rA1 : rA0);
                else request(true, 0, rB0);
                landed();
                __syncthreads();
            }
            __syncthreads();
            for (int c = 0; c < a.nchunk; ++c) {
                if (c + 1 < a.nchunk) request(true, c + 1, ((c + 1) & 1) ? rB1 : rB0);
                landed();
                __syncthreads();
            }
#undef W4L_LOAD
            __syncthreads();
            __syncthreads();
            if (a.stats) __syncthreads();
            return;
        }
        // ------------------------------------------------------------------------------------------------ producer role
        W4GenLane<SPADE> L;
        w4g_lane_init<CIN, SPADE>(a, bk, ptid, L);
        // the sample's tensors (uniform bases; the lane's byte offsets are 32-bit): input, (A, B) pairs of the lane's quad, SPADE maps
        const char* xs = reinterpret_cast<const char*>(g.x + (long)bk.b0 * a.T * (a.H / g.us) * (a.W / g.us) * CIN);
        const float4* cfs = reinterpret_cast<const float4*>(g.coef + (long)bk.b0 * CIN + 4 * (ptid & 3));
        const char* gbs = SPADE ? reinterpret_cast<const char*>(g.gb + (long)bk.b0 * a.H * a.W * 2 * CIN) : nullptr;
        float vmaxd = 0.f, vmaxv = 0.f;
        W4GenIn<SPADE> in0;
        w4g_load<CIN, SPADE>(xs, L, 0, in0);
        // pass A: chunk 0 in front of the first barrier, then chunk c + 1 (or pass B's chunk 0) underneath the MFMA waves' chunk c
        w4g_chunk<CIN, SPADE, 0>(L, xs, cfs, gbs, 0, a.nchunk > 1 ? 1 : 0, in0, smem + rA0 * 64, vmaxd, vmaxv);
        __syncthreads();
        for (int c = 0; c < a.nchunk; ++c) {
            if (c + 1 < a.nchunk) w4g_chunk<CIN, SPADE, 0>(L, xs, cfs, gbs, c + 1, c + 2 < a.nchunk ? c + 2 : 0, in0, smem + (((c + 1) & 1) ? rA1 : rA0) * 64, vmaxd, vmaxv);
            else w4g_chunk<CIN, SPADE, 1>(L, xs, cfs, gbs, 0, a.nchunk > 1 ? 1 : -1, in0, smem + rB0 * 64, vmaxd, vmaxv);
            __syncthreads();
        }
        // pass B
        __syncthreads();
        for (int c = 0; c < a.nchunk; ++c) {
            if (c + 1 < a.nchunk) w4g_chunk<CIN, SPADE, 1>(L, xs, cfs, gbs, c + 1, c + 2 < a.nchunk ? c + 2 : -1, in0, smem + (((c + 1) & 1) ? rB1 : rB0) * 64, vmaxd, vmaxv);
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
    if (cout != 32 || (cin != 32 && cin != 64) || (us != 1 && us != 2) || T % W4G_TT || H % W4G_TH || W % 16 || (us == 2 && (H & 1))) return false;
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
    static bool attr_set[4][I2V_MAX_DEV] = {};
    auto launch = [&](auto kern, bool* done) -> int {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, done)) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(W4G_THREADS), lds, st, a, g);
        return I2V_OK;
    };
    int rcl;
    if (gb) rcl = wts.Cin == 64 ? launch(conv_wino4g_f16x3_kernel<9, 64, 1>, attr_set[0]) : launch(conv_wino4g_f16x3_kernel<9, 32, 1>, attr_set[1]);
    else rcl = wts.Cin == 64 ? launch(conv_wino4g_f16x3_kernel<9, 64, 0>, attr_set[2]) : launch(conv_wino4g_f16x3_kernel<9, 32, 0>, attr_set[3]);
    if (rcl) return rcl;
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

// The LOADER form for the kernels of i2v_conv16w4.hip: `a` as wino4_forward fills it for the 512-thread geometry of a 32-channel 3x3x3
// layer (TT = 4, TH = 8, no temporal duplication); same V, same bits.
bool wino4_loader_supported(const W4Args& a, int KT) {
    return KT == 3 && !a.tdup && a.CoutPad == 32 && a.TT == W4G_TT && a.TH == W4G_TH && a.th_shift == 3;
}

int wino4_loader_launch(W4Args& a, unsigned nblk, hipStream_t st, int form) {
    a.nvirt = (int)nblk;
    a.tofs = 2 * W4_ROWS_A * 64;
    const size_t lds = (size_t)a.tofs + 5 * W4Geo<512>::TILES * 4;
    auto kern = form == 2 ? conv_wino4g_f16x3_kernel<9, 32, 3> : conv_wino4g_f16x3_kernel<9, 32, 2>;
    static bool attr_set[2][I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_set[form == 2])) return rc;
    W4GenArgs g{};
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(W4G_THREADS), lds, st, a, g);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace i2v

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
#include "i2v_conv16w4_dev.h"

namespace i2v {

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
// 32-channel 3x3x3 layers with the V requests issued by four extra waves (i2v_conv16w4g.hip, MODE 2; I2V_W4_LOADER=1 in the measurement
// build).  Measured neutral to negative (profiles/r06_e_thin_loader_ab.txt: 32 -> 32 0.380 vs 0.384 ms, 64 -> 32 0.561 vs 0.545 ms at B = 8;
// 1.33 vs 1.31-1.33 and 2.05 vs 1.93-1.97 ms at B = 32): what the thin layers' tap loops gain without ANY operand traffic (-21 % / -34 %) is
// not the issue cost of the requests but the traffic itself -- 92 KB of V per chunk into an LDS that the operand reads already keep
// 65 % busy.  Off.
constexpr int W4_DEFAULT_LOADER = 0;
struct W4Switches { int pipe, bn, order, nth, skew, trace, loader; };
static W4Switches w4_switches() {
    W4Switches w{W4_DEFAULT_PIPE, 0, W4_DEFAULT_ORDER, 0, 0, 0, W4_DEFAULT_LOADER};
#ifdef I2V_MEASURE
    if (const char* e = getenv("I2V_W4_PIPE")) w.pipe = atoi(e);
    if (const char* e = getenv("I2V_W4_BN")) w.bn = atoi(e);
    if (const char* e = getenv("I2V_W4_ORDER")) w.order = atoi(e);
    if (const char* e = getenv("I2V_W4_NTH")) w.nth = atoi(e);
    if (const char* e = getenv("I2V_W4_SKEW")) w.skew = atoi(e);
    w.trace = getenv("I2V_W4_TRACE") != nullptr;
    if (const char* e = getenv("I2V_W4_LOADER")) w.loader = atoi(e);
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
    // 32-channel 3x3x3 layers whose map tiles into the 512-thread brick: the loader form (12 waves: the tap loops issue no V request)
    if (sw.loader && BN == 32 && a.CoutPad == 32 && wts.KT == 3 && !wts.tdup && sw.pipe == 0 && env_nth == 0 && TT == 4 && TH == 8) {
        a.TT = TT; a.TH = TH; a.TJ = 4; a.nbT = T / TT; a.nbH = H / TH; a.nbJ = a.J / 4;
        a.th_shift = 3;
        a.hh_magic = ((1 << 20) + TH + 1) / (TH + 2);
        a.order = env_order;
        const long nb = (long)B * a.nbT * a.nbH * a.nbJ;
        if (wino4_loader_supported(a, wts.KT) && nb > 0 && nb < (1L << 30) && (long)T * a.nchunk * 6 * H * a.J * 64 < (1L << 31) &&
            (long)B * T * H * W < (1L << 31) && (!stats || (long)TT * TH * 4 <= (long)T * H * a.J))
            return wino4_loader_launch(a, (unsigned)nb, st, sw.loader);
    }
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
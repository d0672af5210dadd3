// Stand-alone check + benchmark of the Winograd split-fp16 conv kernel (csrc/i2v_conv16w.hip) against the direct
// split-fp16 kernel (csrc/i2v_conv16.hip) on one layer shape.
//   conv16w_check B T H W Cin Cout tdup res      (T = output frames; tdup: conv_0 behind a x2 temporal up-sampling)
// and, where the shape allows it, of the F(4,3) kernel (csrc/i2v_conv16w4.hip) next to them.
// Build: hipcc -O3 --offload-arch=gfx950 -I<csrc> tools/conv16w_check.hip <csrc>/i2v_conv16w.hip <csrc>/i2v_conv16w4.hip
//        <csrc>/i2v_conv16.hip <csrc>/i2v_common.hip -o tools/conv16w_check
// Measurement builds of the F(4,3) kernel (same command plus):
//   -DW4_TIMELINE   wall-clock stamps per workgroup phase (tables + first brick, pass A, hand-over, pass B, epilogue halves and
//                   the sub-phases of the first half), printed as means over the workgroups
//   -DW4_TAPTIME    s_memtime between the starts of consecutive taps, per wave and tap slot of the chunk pair
//   -DW4_ABLATE_AL / -DW4_ABLATE_BL   half of the LDS operand reads / of the weight cache lines removed (results wrong)
//   -DW4_PRIO=0     no s_setprio in the tap loop
// T = 1 without tdup runs the 1x3x3 variants (SPADE's 2-D convs) of all three kernels.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "i2v_conv.h"

using namespace i2v;
#ifdef W4_TAPTIME
namespace i2v { void w4_taptime_report(); }
#endif
#ifdef W4_TIMELINE
namespace i2v { void w4_timeline_report(unsigned nwg); }
#endif

static inline void split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2, T = argc > 2 ? atoi(argv[2]) : 16, H = argc > 3 ? atoi(argv[3]) : 64,
              W = argc > 4 ? atoi(argv[4]) : 64, Cin = argc > 5 ? atoi(argv[5]) : 256, Cout = argc > 6 ? atoi(argv[6]) : 128;
    const int tdup = argc > 7 ? atoi(argv[7]) : 0, use_res = argc > 8 ? atoi(argv[8]) : 0, nostats = argc > 9 ? atoi(argv[9]) : 0;
    // gen: 1 = also run the operand-GENERATING kernel (conv_wino4g_f16x3_kernel) on the raw fp32 input + per-(b,c) coefficients
    // (ADAIN form, conv_1 of g_4); 2 = with SPADE gamma' | beta maps and a x2 spatial up-sampling in front (conv_0 of g_4).  The
    // activations the other kernels see are then a = lrelu(modulate(x)), formed on the host with the writer's expressions.
    const int gen = argc > 10 ? atoi(argv[10]) : 0;
    const int Ti = tdup ? T / 2 : T, J = W / 2;
    std::vector<float> w((size_t)Cout * Cin * 27), bias(Cout);
    srand(1);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (auto& v : bias) v = rand() / (float)RAND_MAX - 0.5f;
    Conv16Weights cw;
    Wino16Weights ww;
    int rc;
    if (T == 1 && !tdup) {   // single frame: only the middle temporal slice ever meets data -> 1x3x3 kernels (the 3-tap variant)
        std::vector<float> w1((size_t)Cout * Cin * 9);
        for (size_t nc = 0; nc < (size_t)Cout * Cin; ++nc)
            for (int k = 0; k < 9; ++k) w1[nc * 9 + k] = w[nc * 27 + 9 + k];
        rc = cw.pack(w1.data(), bias.data(), Cout, Cin, 1, 3, 3, 0.7);
        if (rc) { printf("pack16: %s\n", i2v_last_error()); return 1; }
        rc = ww.pack(w1.data(), bias.data(), Cout, Cin, 1, 0.7);
    } else {
        rc = tdup ? cw.pack_tdup(w.data(), bias.data(), Cout, Cin, 0.7) : cw.pack(w.data(), bias.data(), Cout, Cin, 3, 3, 3, 0.7);
        if (rc) { printf("pack16: %s\n", i2v_last_error()); return 1; }
        rc = tdup ? ww.pack_tdup(w.data(), bias.data(), Cout, Cin, 0.7) : ww.pack(w.data(), bias.data(), Cout, Cin, 3, 0.7);
    }
    if (rc) { printf("packw: %s\n", i2v_last_error()); return 1; }
    if (!wino16_supported(Cout, Cin, Ti, H, W, T == 1 && !tdup ? 1 : tdup ? 2 : 3)) { printf("shape not supported by the Winograd kernel\n"); return 1; }
    const bool one = T == 1 && !tdup;
    const bool do4 = wino4_supported(Cout, Cin, Ti, H, W, one ? 1 : tdup ? 2 : 3);
    Wino4Weights w4;
    if (do4) {
        if (one) {
            std::vector<float> w1((size_t)Cout * Cin * 9);
            for (size_t nc = 0; nc < (size_t)Cout * Cin; ++nc)
                for (int k = 0; k < 9; ++k) w1[nc * 9 + k] = w[nc * 27 + 9 + k];
            rc = w4.pack(w1.data(), bias.data(), Cout, Cin, 0.7, 1);
        } else
        rc = tdup ? w4.pack_tdup(w.data(), bias.data(), Cout, Cin, 0.7) : w4.pack(w.data(), bias.data(), Cout, Cin, 0.7);
        if (rc) { printf("pack4: %s\n", i2v_last_error()); return 1; }
    }

    const size_t npi = (size_t)B * Ti * H * W, npo = (size_t)B * T * H * W;
    std::vector<float> a(npi * Cin);
    for (auto& v : a) {
        v = (rand() / (float)RAND_MAX - 0.3f) * 2.f;
        if (v < 0) v *= 0.2f;  // leaky-relu-like distribution
    }
    // gen: raw input x (low resolution for gen = 2), coefficients, SPADE maps; a = what modulate_wino4_kernel would feed the conv
    const int us = gen == 2 ? 2 : 1;
    std::vector<float> xraw, coefh, gbh;
    if (gen) {
        if (tdup || !wino4g_supported(Cout, Cin, T, H, W, us)) { printf("gen: shape not supported by the generating kernel\n"); return 1; }
        const int Hl = H / us, Wl = W / us;
        xraw.resize((size_t)B * T * Hl * Wl * Cin);
        coefh.resize((size_t)B * Cin * 2);
        for (auto& v : xraw) v = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
        for (size_t i = 0; i < (size_t)B * Cin; ++i) { coefh[2 * i] = 0.5f + rand() / (float)RAND_MAX; coefh[2 * i + 1] = rand() / (float)RAND_MAX - 0.4f; }
        // I2V_CHECK_GEN (diagnosis of the last-bit differences between the generating kernel and the writer path): 1 = identity
        // coefficients (d = lrelu(x)), 2 = identity coefficients and x >= 0 (d = x: no modulation arithmetic at all), 3 = x small integers,
        // identity coefficients (every V exact: hi exact, lo = 0)
        if (const char* e = getenv("I2V_CHECK_GEN")) {
            const int m = atoi(e);
            if (m >= 1) for (size_t i = 0; i < (size_t)B * Cin; ++i) { coefh[2 * i] = 1.f; coefh[2 * i + 1] = 0.f; }
            if (m == 2) for (auto& v : xraw) v = std::fabs(v);
            if (m == 3) for (auto& v : xraw) v = (float)(rand() % 4);
        }
        if (gen == 2) {
            gbh.resize((size_t)B * H * W * 2 * Cin);
            for (size_t i = 0; i < gbh.size(); ++i) gbh[i] = ((i / Cin) & 1) ? rand() / (float)RAND_MAX - 0.5f : 0.6f + rand() / (float)RAND_MAX;
        }
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t)
                for (int h = 0; h < H; ++h)
                    for (int w_ = 0; w_ < W; ++w_)
                        for (int c = 0; c < Cin; ++c) {
                            const float xv = xraw[((((size_t)b * T + t) * Hl + h / us) * Wl + w_ / us) * Cin + c];
                            const float ca = coefh[((size_t)b * Cin + c) * 2], cb = coefh[((size_t)b * Cin + c) * 2 + 1];
                            float r;
                            if (gen == 2) {
                                const float* gp = &gbh[(((size_t)b * H + h) * W + w_) * 2 * Cin];
                                r = fmaf(xv, ca * gp[c], fmaf(cb, gp[c], gp[Cin + c]));
                            } else r = fmaf(xv, ca, cb);
                            a[((((size_t)b * T + t) * H + h) * W + w_) * Cin + c] = r < 0.f ? 0.2f * r : r;
                        }
    }
    // power probe: I2V_CHECK_DATA=zero (all-zero activations), =pow2 (powers of two: empty lo parts)
    if (const char* e = getenv("I2V_CHECK_DATA")) {
        for (auto& v : a) v = e[0] == 'z' ? 0.f : (v == 0.f ? 0.f : std::ldexp(1.f, (int)std::floor(std::log2(std::fabs(v)))));
    }
    std::vector<_Float16> hl(npi * Cin * 2), V((size_t)B * Ti * H * J * 4 * Cin * 2);
    for (size_t p = 0; p < npi; ++p)
        for (int c = 0; c < Cin; ++c) {
            _Float16 hi, lo;
            split(a[p * Cin + c], hi, lo);
            hl[p * Cin * 2 + (c >> 3) * 16 + (c & 7)] = hi;
            hl[p * Cin * 2 + (c >> 3) * 16 + 8 + (c & 7)] = lo;
        }
    for (size_t row = 0; row < (size_t)B * Ti * H; ++row)
        for (int j = 0; j < J; ++j)
            for (int c = 0; c < Cin; ++c) {
                float d[4];
                for (int k = 0; k < 4; ++k) {
                    const int wq = 2 * j - 1 + k;
                    d[k] = (wq >= 0 && wq < W) ? a[(row * W + wq) * Cin + c] : 0.f;
                }
                const float v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
                for (int x = 0; x < 4; ++x) {
                    _Float16 hi, lo;
                    split(v[x], hi, lo);
                    // [B][T][Cin/16][4][H][J][32 halfs]
                    const size_t bt = row / H, h = row % H;
                    _Float16* dst = &V[((((bt * (Cin / 16) + c / 16) * 4 + x) * H + h) * J + j) * 32];
                    dst[((c & 15) >> 3) * 16 + (c & 7)] = hi;
                    dst[((c & 15) >> 3) * 16 + 8 + (c & 7)] = lo;
                }
            }
    // F(4,3) operand: [B][T][Cin/16][6][H][W/4][32 halfs]
    std::vector<_Float16> V4;
    if (do4) {
        const int J4 = W / 4;
        V4.resize((size_t)B * Ti * H * J4 * 6 * Cin * 2);
        for (size_t row = 0; row < (size_t)B * Ti * H; ++row)
            for (int j = 0; j < J4; ++j)
                for (int c = 0; c < Cin; ++c) {
                    float d[6];
                    for (int k = 0; k < 6; ++k) {
                        const int wq = 4 * j - 1 + k;
                        d[k] = (wq >= 0 && wq < W) ? a[(row * W + wq) * Cin + c] : 0.f;
                    }
                    // (the expressions of modulate_wino4_kernel, so that this operand is the production writer's bit for bit)
                    const float v[6] = {fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4])), fmaf(-4.f, d[1] + d[2], d[3] + d[4]), fmaf(4.f, d[1] - d[2], d[4] - d[3]),
                                        fmaf(2.f, d[3] - d[1], d[4] - d[2]), fmaf(2.f, d[1] - d[3], d[4] - d[2]), fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]))};
                    for (int x = 0; x < 6; ++x) {
                        _Float16 hi, lo;
                        split(v[x], hi, lo);
                        const size_t bt = row / H, h = row % H;
                        _Float16* dst = &V4[((((bt * (Cin / 16) + c / 16) * 6 + x) * H + h) * J4 + j) * 32];
                        dst[((c & 15) >> 3) * 16 + (c & 7)] = hi;
                        dst[((c & 15) >> 3) * 16 + 8 + (c & 7)] = lo;
                    }
                }
    }
    std::vector<float> res;
    if (use_res) {
        res.resize(npo * Cout);
        for (auto& v : res) v = rand() / (float)RAND_MAX - 0.5f;
    }
    void *dhl, *dV, *dV4 = nullptr;
    float *o0, *o1, *o2 = nullptr, *dres = nullptr;
    double *s0, *s1, *s2 = nullptr;
    hipMalloc(&dhl, hl.size() * 2);
    hipMalloc(&dV, V.size() * 2);
    hipMalloc(&o0, npo * Cout * 4);
    hipMalloc(&o1, npo * Cout * 4);
    hipMalloc(&s0, (size_t)B * Cout * 16);
    hipMalloc(&s1, (size_t)B * Cout * 16);
    hipMemcpy(dhl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dV, V.data(), V.size() * 2, hipMemcpyHostToDevice);
    if (use_res) {
        hipMalloc(&dres, res.size() * 4);
        hipMemcpy(dres, res.data(), res.size() * 4, hipMemcpyHostToDevice);
    }
    if (do4) {
        hipMalloc(&dV4, V4.size() * 2);
        hipMemcpy(dV4, V4.data(), V4.size() * 2, hipMemcpyHostToDevice);
        hipMalloc(&o2, npo * Cout * 4);
        hipMemset(o2, 0xff, npo * Cout * 4);
        hipMalloc(&s2, (size_t)B * Cout * 16);
        hipMemset(s2, 0, (size_t)B * Cout * 16);
    }
    hipMemset(o1, 0xff, npo * Cout * 4);
    hipMemset(s0, 0, (size_t)B * Cout * 16);
    hipMemset(s1, 0, (size_t)B * Cout * 16);
    const bool fuse = !nostats && conv16_can_fuse_stats(Ti, H, W);
    if (conv16_forward(cw, dhl, o0, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, fuse ? s0 : nullptr)) { printf("conv16: %s\n", i2v_last_error()); return 1; }
    if (wino16_forward(ww, dV, o1, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, nostats ? nullptr : s1)) { printf("wino16: %s\n", i2v_last_error()); return 1; }
    if (do4 && wino4_forward(w4, dV4, o2, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, nostats ? nullptr : s2)) { printf("wino4: %s\n", i2v_last_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<float> h0(npo * Cout), h1(npo * Cout);
    std::vector<double> hs0((size_t)B * Cout * 2), hs1((size_t)B * Cout * 2);
    hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs0.data(), s0, hs0.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hs1.data(), s1, hs1.size() * 8, hipMemcpyDeviceToHost);
    double num = 0, den = 0, mx = 0;
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i) {
        const double d = (double)h1[i] - h0[i];
        if (!(std::fabs(d) < 1e30)) { ++bad; continue; }
        num += d * d; den += (double)h0[i] * h0[i];
        mx = std::max(mx, std::fabs(d));
    }
    double smx = 0;
    if (fuse)
        for (size_t i = 0; i < hs0.size(); ++i) smx = std::max(smx, std::fabs(hs1[i] - hs0[i]) / (std::fabs(hs0[i]) + 1.0));
    // exact fp64 reference on a sample of outputs
    std::vector<float> h2;
    std::vector<double> hs2;
    double num4 = 0, mx4 = 0, smx4 = 0;
    size_t bad4 = 0;
    if (do4) {
        h2.resize(npo * Cout); hs2.resize((size_t)B * Cout * 2);
        hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hs2.data(), s2, hs2.size() * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h0.size(); ++i) {
            const double d = (double)h2[i] - h0[i];
            if (!(std::fabs(d) < 1e30)) { ++bad4; continue; }
            num4 += d * d; mx4 = std::max(mx4, std::fabs(d));
        }
        if (fuse)
            for (size_t i = 0; i < hs0.size(); ++i) smx4 = std::max(smx4, std::fabs(hs2[i] - hs0[i]) / (std::fabs(hs0[i]) + 1.0));
    }
    double rnum = 0, rden = 0, rnum4 = 0;
    for (int s = 0; s < 400; ++s) {
        const size_t p = ((size_t)rand() * 7919u + s) % npo;
        const int n = rand() % Cout;
        size_t q = p;
        const int wq = q % W; q /= W;
        const int hq = q % H; q /= H;
        const int tq = q % T; q /= T;
        const int b = (int)q;
        double ref = bias[n];
        for (int kt = 0; kt < 3; ++kt)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const int t = tq + kt - 1, h = hq + kh - 1, ww_ = wq + kw - 1;
                    if (t < 0 || t >= T || h < 0 || h >= H || ww_ < 0 || ww_ >= W) continue;
                    const int ti = tdup ? t / 2 : t;
                    const float* ap = &a[((((size_t)b * Ti + ti) * H + h) * W + ww_) * Cin];
                    for (int c = 0; c < Cin; ++c) ref += (double)ap[c] * (double)w[((size_t)n * Cin + c) * 27 + kt * 9 + kh * 3 + kw] * 0.7;
                }
        if (use_res) ref += res[p * Cout + n];
        const double d = h1[p * Cout + n] - ref;
        rnum += d * d; rden += ref * ref;
        if (do4) { const double d4 = h2[p * Cout + n] - ref; rnum4 += d4 * d4; }
    }
    printf("[B=%d,T=%d,%dx%d] %d -> %d tdup=%d res=%d: wino vs direct rel-L2 %.3e max|d| %.3e nonfinite %zu  stats rel %.2e | wino vs fp64 (400 samples) rel-L2 %.3e\n",
           B, T, H, W, Cin, Cout, tdup, use_res, std::sqrt(num / (den + 1e-30)), mx, bad, smx, std::sqrt(rnum / (rden + 1e-30)));

    if (do4)
        printf("   F(4,3) vs direct rel-L2 %.3e max|d| %.3e nonfinite %zu  stats rel %.2e | F(4,3) vs fp64 (400 samples) rel-L2 %.3e\n",
               std::sqrt(num4 / (den + 1e-30)), mx4, bad4, smx4, std::sqrt(rnum4 / (rden + 1e-30)));
    float *dx = nullptr, *dcoef = nullptr, *dgb = nullptr, *o3 = nullptr;
    double* s3 = nullptr;
    int* dflag = nullptr;
    if (gen && do4) {
        hipMalloc(&dx, xraw.size() * 4); hipMemcpy(dx, xraw.data(), xraw.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&dcoef, coefh.size() * 4); hipMemcpy(dcoef, coefh.data(), coefh.size() * 4, hipMemcpyHostToDevice);
        if (gen == 2) { hipMalloc(&dgb, gbh.size() * 4); hipMemcpy(dgb, gbh.data(), gbh.size() * 4, hipMemcpyHostToDevice); }
        hipMalloc(&o3, npo * Cout * 4); hipMemset(o3, 0xff, npo * Cout * 4);
        hipMalloc(&s3, (size_t)B * Cout * 16); hipMemset(s3, 0, (size_t)B * Cout * 16);
        hipMalloc(&dflag, 64 * 4); hipMemset(dflag, 0, 64 * 4);
        if (wino4g_forward(w4, dx, dcoef, dgb, us, o3, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, nostats ? nullptr : s3, dflag, dflag + 1)) {
            printf("wino4g: %s\n", i2v_last_error()); return 1;
        }
        if (hipDeviceSynchronize() != hipSuccess) { printf("gen kernel fault: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<float> h3(npo * Cout);
        std::vector<double> hs3((size_t)B * Cout * 2);
        int hflag[2] = {0, 0};
        hipMemcpy(h3.data(), o3, h3.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hs3.data(), s3, hs3.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hflag, dflag, 8, hipMemcpyDeviceToHost);
        size_t ndiff = 0, first = (size_t)-1;
        double mxd = 0;
        for (size_t i = 0; i < h3.size(); ++i)
            if (memcmp(&h3[i], &h2[i], 4)) { if (!ndiff) first = i; ++ndiff; mxd = std::max(mxd, std::fabs((double)h3[i] - h2[i])); }
        if (ndiff) {   // where do the differing outputs sit?  (brick = 4 frames x 8 rows x 16 columns; channel quads of the producer lanes)
            long hw[16] = {}, hh[8] = {}, ht[4] = {}, hc[32] = {};
            for (size_t i = 0; i < h3.size(); ++i)
                if (memcmp(&h3[i], &h2[i], 4)) {
                    const int c = (int)(i % Cout); size_t p_ = i / Cout;
                    const int w_ = (int)(p_ % W); p_ /= W;
                    const int h_ = (int)(p_ % H); p_ /= H;
                    const int t_ = (int)(p_ % T);
                    ++hw[w_ % 16]; ++hh[h_ % 8]; ++ht[t_ % 4]; ++hc[c % 32];
                }
            printf("      differing outputs by w %% 16:"); for (long v : hw) printf(" %ld", v);
            printf("\n      by h %% 8:"); for (long v : hh) printf(" %ld", v);
            printf("\n      by t %% 4:"); for (long v : ht) printf(" %ld", v);
            printf("\n      by output channel:"); for (long v : hc) printf(" %ld", v);
            printf("\n");
        }
        double smx3 = 0;
        if (!nostats) for (size_t i = 0; i < hs3.size(); ++i) smx3 = std::max(smx3, std::fabs(hs3[i] - hs2[i]) / (std::fabs(hs2[i]) + 1.0));
        float amax = 0.f;
        for (float v : a) amax = std::max(amax, std::fabs(v));
        float um; memcpy(&um, &hflag[1], 4);
        printf("   GEN (mode %d) vs F(4,3) on the writer's operand: %zu of %zu outputs differ (first at %zu, max|d| %.3e)  stats rel %.2e | range flag %d, published max|d| %.6g (host %.6g)\n",
               gen, ndiff, h3.size(), first, mxd, smx3, hflag[0], um, amax);
    }
    const double flops = 2.0 * npo * Cin * Cout * 27.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 5;
    for (int which = 0; which < (do4 ? (gen ? 4 : 3) : 2); ++which) {
        auto go = [&]() {
            if (which == 3) wino4g_forward(w4, dx, dcoef, dgb, us, o3, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, nostats ? nullptr : s3, dflag, dflag + 1);
            else if (which == 2) wino4_forward(w4, dV4, o2, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, nostats ? nullptr : s2);
            else if (which) wino16_forward(ww, dV, o1, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, nostats ? nullptr : s1);
            else conv16_forward(cw, dhl, o0, dres, 1, 1, B, T, H, W, EPI_NONE, nullptr, fuse ? s0 : nullptr);
        };
        for (int it = 0; it < 2; ++it) go();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int it = 0; it < n; ++it) go();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= n;
        printf("   %-8s %8.3f ms  %7.1f TFLOP/s algorithmic\n", which == 3 ? "GEN" : which == 2 ? "F(4,3)" : which ? "F(2,3)" : "direct", ms, flops / ms / 1e9);
#ifdef W4_TAPTIME
        if (which == 2) w4_taptime_report();
#endif
#ifdef W4_TIMELINE
        if (which == 2) w4_timeline_report(8192u);   // (the report averages over the workgroups that left stamps)
#endif
    }
    return 0;
}

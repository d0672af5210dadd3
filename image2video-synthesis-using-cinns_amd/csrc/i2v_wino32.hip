// 3x3x3 Conv3d in exact-fp32 mode (cfg.mma == 0) with a Winograd F(4,3) transform along W on the fp32 matrix cores.
// Replaces conv_0 / conv_1 of GeneratorBlock (decoder.py:14-15, 37-40) from the 8x8 level on when the decoder runs with mma = 0,
// the mode a checkpoint needs whose activations leave the fp16 range of the split-fp16 operands: no fp16
// value exists anywhere on this path.  The direct fp32 kernel (i2v_conv.hip) costs 9.3x the split-fp16 step; F(4,3) halves its
// MFMA work.
//
// Per tile of four output positions (w = 4j .. 4j+3), (kt, kh) tap and channel, with d_k = a[t+kt-1][h+kh-1][4j-1+k] (zero padded):
//     V0 = 4 d0 - 5 d2 + d4        V1 = -4 d1 - 4 d2 + d3 + d4     V2 = 4 d1 - 4 d2 - d3 + d4
//     V3 = -2 d1 - d2 + 2 d3 + d4  V4 = 2 d1 - d2 - 2 d3 + d4      V5 = 4 d1 - 5 d3 + d5
//     U0 = g0/4   U1 = -(g0+g1+g2)/6   U2 = -(g0-g1+g2)/6   U3 = g0/24 + g1/12 + g2/6   U4 = g0/24 - g1/12 + g2/6   U5 = g2
//     M_x = sum over (kt, kh, c) of V_x U_x
//     y0 = M0+M1+M2+M3+M4   y1 = M1-M2+2M3-2M4   y2 = M1+M2+4M3+4M4   y3 = M1-M2+8M3-8M4+M5
// (the same matrices as the split-fp16 kernel, i2v_conv16w4.hip).  Structure -- deliberately plain, this is the fallback mode:
//   1. modulate_wino4_f32_kernel   norm / SPADE / ADAIN apply + lrelu + nearest up-sampling (normalization_layer.py:19-23, 47-51,
//                                  decoder.py:37-40) and V = B^T d, written as SIX channels-last fp32 tensors [x][B][T][H][W/4][C]
//   2. six launches of conv_mfma_f32_kernel (i2v_conv.hip) with the 3x3x1 kernels U_x: M_x = conv(V_x, U_x), 9 taps instead of 27
//      over a quarter of the positions = 54 instead of 108 tap-GEMMs per four outputs
//   3. wino4_out_f32_kernel        y = A^T M + bias + residual (nearest up-sampled shortcut, decoder.py:44-49) [+ lrelu]
// U = G g is computed in fp64 and rounded to fp32 once at load; V and y are evaluated in fp32: rel-L2 vs fp64 ~1e-6 (the transform's
// rounding, measured in the parity tests), against ~2e-7 for the direct kernel -- both far inside the 1e-4 gate.
// I2V_DEC_WINO32=0 keeps the direct kernel everywhere.
#include <algorithm>
#include <cmath>
#include <vector>

#include "i2v_conv.h"

namespace i2v {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ldf4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void stf4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// One thread = one (h, tile j, 4 channels) of one sample, looping over the frames: the per-(sample, channel) coefficients and the
// SPADE gamma' / beta of its six positions do not depend on t.  x: [B][T/ut][H/us][W/us][C]; coef: per (b, c) pairs (A, B) with
// norm(x) = x A + B (null: identity); gb: [B][H][W][2C] (gamma' | beta) or null; V: [6][B][T][H][J][C].
__global__ __launch_bounds__(256) void modulate_wino4_f32_kernel(const float* __restrict__ x, const float2* __restrict__ coef,
                                                                 const float* __restrict__ gb, float* __restrict__ V, int T, int H, int W,
                                                                 int C, int ut, int us, int lrelu, long plane) {
    const int C4 = C >> 2, J = W >> 2;
    const int b = blockIdx.y;
    const int per = H * J * C4;
    const int Hl = H / us, Wl = W / us, Tl = T / ut;
    const float* xb = x + (long)b * Tl * Hl * Wl * C;
    const float* gbb = gb ? gb + (long)b * H * W * 2 * C : nullptr;
    const long xstride = (long)Hl * Wl * C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        const int c4 = i % C4;
        int q = i / C4;
        const int j = q % J;
        const int h = q / J;
        f4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
        if (coef) {
            const f4 ab0 = ldf4(reinterpret_cast<const float*>(coef + (long)b * C + 4 * c4));
            const f4 ab1 = ldf4(reinterpret_cast<const float*>(coef + (long)b * C + 4 * c4 + 2));
            ca = f4{ab0[0], ab0[2], ab1[0], ab1[2]};
            cb = f4{ab0[1], ab0[3], ab1[1], ab1[3]};
        }
        f4 pa[6], pb[6];
        const float* xp[6];
        bool in[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int w = 4 * j - 1 + k;
            in[k] = (unsigned)w < (unsigned)W;
            const int wc = in[k] ? w : 0;
            pa[k] = ca; pb[k] = cb;
            if (gbb) {   // fold SPADE's gamma' / beta of the position into the affine: (x A + B) g + e
                const float* g = gbb + ((long)h * W + wc) * (2 * C) + 4 * c4;
                const f4 ga = ldf4(g), be = ldf4(g + C);
#pragma unroll
                for (int c = 0; c < 4; ++c) { pb[k][c] = fmaf(cb[c], ga[c], be[c]); pa[k][c] = ca[c] * ga[c]; }
            }
            xp[k] = xb + ((long)(h / us) * Wl + wc / us) * C + 4 * c4;
        }
        float* vo = V + ((((long)b * T) * H + h) * J + j) * C + 4 * c4;
        const long vstride_t = (long)H * J * C;
        f4 d[6];
        for (int t = 0; t < T; ++t) {
            if (t % ut == 0) {
                const long toff = (long)(t / ut) * xstride;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const f4 v = ldf4(xp[k] + toff);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float r = fmaf(v[c], pa[k][c], pb[k][c]);
                        if (lrelu) r = r >= 0.f ? r : 0.2f * r;
                        d[k][c] = in[k] ? r : 0.f;   // the conv's zero padding applies to the modulated activation
                    }
                }
            }
            f4 v[6];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
                v[0][c] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                v[1][c] = fmaf(-4.f, d1 + d2, d3 + d4);
                v[2][c] = fmaf(4.f, d1 - d2, d4 - d3);
                v[3][c] = fmaf(2.f, d3 - d1, d4 - d2);
                v[4][c] = fmaf(2.f, d1 - d3, d4 - d2);
                v[5][c] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
            }
#pragma unroll
            for (int xq = 0; xq < 6; ++xq) stf4(vo + xq * plane + t * vstride_t, v[xq]);
        }
    }
}

// y = A^T M + bias + residual [+ lrelu].  One thread = one (b, t, h, tile j, 4 output channels).
// M: [6][B][T][H][J][Cout]; res: channels-last [B][T >> rt][H >> rs][W >> rs][Cout] or null; out: [B][T][H][W][Cout].
__global__ __launch_bounds__(256) void wino4_out_f32_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ out, long total, int T,
                                                            int H, int J, int Cout, int rt_shift, int rs_shift, int lrelu, long plane) {
    const int N4 = Cout >> 2;
    const int W = 4 * J;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n4 = (int)(i % N4);
        long q = i / N4;
        const int j = (int)(q % J); q /= J;
        const int h = (int)(q % H); q /= H;
        const int t = (int)(q % T);
        const long b = q / T;
        const float* mp = M + (((b * T + t) * H + h) * J + j) * (long)Cout + 4 * n4;
        const f4 m0 = ldf4(mp), m1 = ldf4(mp + plane), m2 = ldf4(mp + 2 * plane), m3 = ldf4(mp + 3 * plane), m4 = ldf4(mp + 4 * plane),
                 m5 = ldf4(mp + 5 * plane);
        f4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = ldf4(bias + 4 * n4);
        const f4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        f4 y[4];
        y[0] = m0 + s12 + s34;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            y[1][c] = fmaf(2.f, d34[c], d12[c]);
            y[2][c] = fmaf(4.f, s34[c], s12[c]);
            y[3][c] = fmaf(8.f, d34[c], d12[c]) + m5[c];
        }
        const long rbase = res ? ((b * (T >> rt_shift) + (t >> rt_shift)) * (H >> rs_shift) + (h >> rs_shift)) * (long)(W >> rs_shift) : 0;
        float* op = out + (((b * T + t) * H + h) * (long)W + 4 * j) * Cout + 4 * n4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f4 v = y[p] + bv;
            if (res) v += ldf4(res + (rbase + ((4 * j + p) >> rs_shift)) * Cout + 4 * n4);
            if (lrelu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = v[c] >= 0.f ? v[c] : 0.2f * v[c];
            }
            stf4(op + (long)p * Cout, v);
        }
    }
}

int shift_of(int f) { return f == 4 ? 2 : f == 2 ? 1 : 0; }

}  // namespace

bool wino4f32_supported(int cout, int cin, int T, int H, int W) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    // (the plane convs run conv_mfma_f32_kernel on [T][H][W/4] maps: its bricks need power-of-two dims and span samples where a
    //  sample's map is smaller than a brick; 4x4 maps -- one tile per row, two of its six positions padding -- stay on the 27-tap kernel)
    return cout % 4 == 0 && cin % 4 == 0 && W % 4 == 0 && W >= 8 && pow2(T) && pow2(H) && pow2(W);
}

int Wino4F32Weights::pack(const float* w_src, const float* bias_src, int cout, int cin, double scale) {
    Cin = cin; Cout = cout;
    std::vector<float> tmp((size_t)cout * cin * 9);
    for (int x = 0; x < 6; ++x) {
        for (size_t nc = 0; nc < (size_t)cout * cin; ++nc)
            for (int th = 0; th < 9; ++th) {   // (kt, kh)
                const double g0 = (double)w_src[nc * 27 + th * 3] * scale, g1 = (double)w_src[nc * 27 + th * 3 + 1] * scale,
                             g2 = (double)w_src[nc * 27 + th * 3 + 2] * scale;
                double u;
                switch (x) {
                    case 0: u = g0 / 4.0; break;
                    case 1: u = -(g0 + g1 + g2) / 6.0; break;
                    case 2: u = -(g0 - g1 + g2) / 6.0; break;
                    case 3: u = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                    case 4: u = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                    default: u = g2; break;
                }
                tmp[nc * 9 + th] = (float)u;
            }
        if (int rc = u[x].pack(tmp.data(), nullptr, cout, cin, 3, 3, 1, 1.0)) return rc;
    }
    if (bias_src) return bias.upload(bias_src, (size_t)cout * 4);
    bias.release();
    return I2V_OK;
}

int modulate_wino4_f32(const float* x, const float* coef, const float* gb, float* V, int B, int T, int H, int W, int C, int ut, int us,
                       int lrelu, hipStream_t st) {
    I2V_REQUIRE(C % 4 == 0 && W % 4 == 0, I2V_E_INVALID, "modulate (fp32 F(4,3) operand): channels %d / width %d", C, W);
    const long per = (long)H * (W / 4) * (C / 4);
    I2V_REQUIRE(per < (1L << 31), I2V_E_INVALID, "modulate (fp32 F(4,3) operand): tensor too large");
    const unsigned gx = (unsigned)std::min<long>((per + 255) / 256, 8192);
    const long plane = (long)B * T * H * (W / 4) * C;
    hipLaunchKernelGGL(modulate_wino4_f32_kernel, dim3(gx, B), dim3(256), 0, st, x, reinterpret_cast<const float2*>(coef), gb, V, T, H, W, C,
                       ut, us, lrelu, plane);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

int wino4f32_forward(const Wino4F32Weights& wts, const float* V, float* M, float* out, const float* res, int rt, int rs, int B, int T, int H,
                     int W, int epi, hipStream_t st) {
    I2V_REQUIRE(wts.u[0].w.p, I2V_E_STATE, "wino4 (fp32): weights not packed");
    I2V_REQUIRE((epi & ~EPI_LRELU) == 0, I2V_E_INVALID, "wino4 (fp32): unsupported epilogue %d", epi);
    I2V_REQUIRE(wino4f32_supported(wts.Cout, wts.Cin, T, H, W), I2V_E_INVALID, "wino4 (fp32): unsupported shape [%d,%d,%d] %d -> %d", T, H, W,
                wts.Cin, wts.Cout);
    if (!res) { rt = 1; rs = 1; }
    I2V_REQUIRE((rt == 1 || rt == 2 || rt == 4) && (rs == 1 || rs == 2 || rs == 4), I2V_E_INVALID, "wino4 (fp32): residual factors %d / %d", rt, rs);
    const int J = W / 4;
    const long vplane = (long)B * T * H * J * wts.Cin, mplane = (long)B * T * H * J * wts.Cout;
    for (int x = 0; x < 6; ++x)
        if (int rc = conv_forward(wts.u[x], V + x * vplane, wts.Cin, M + x * mplane, nullptr, 1, 1, B, T, H, J, EPI_NONE, st)) return rc;
    const long total = (long)B * T * H * J * (wts.Cout / 4);
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 1L << 20);
    hipLaunchKernelGGL(wino4_out_f32_kernel, dim3(grid), dim3(256), 0, st, M, wts.bias.as<float>(), res, out, total, T, H, J, wts.Cout,
                       shift_of(rt), shift_of(rs), (epi & EPI_LRELU) ? 1 : 0, mplane);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace i2v

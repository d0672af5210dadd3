// Skinny-M linear layer kernel shared by the flow chain and the stand-alone MLP / Linear entry points.
#pragma once
#include "i2v_common.h"

namespace i2v {

struct LinArgs {
    const float* W;        // rows [N][ldw]
    int ldw;
    int K;
    const float* in;       // element (k, b) of row-group g at in[g*in_group_stride + k*in_sk + b*in_sb]
    long in_sk, in_sb, in_group_stride;
    int group_rows;        // rows per input group (H for the hidden layers; N when all rows share one input)
    const float* bias_vec; // [N] or null
    const float* bias_mat; // [N][B] or null
    float* out;            // element (n, b) at out[n*out_sn + b*out_sb]
    long out_sn, out_sb;
    int N, B;
    float slope;           // LeakyReLU slope; 1.0f = identity
};

// out[n][b] = act(bias + sum_k W[n][k] * in[k][b]) for NT rows per workgroup, lanes = samples.
// The [NT][K] weight tile is fetched with coalesced 16-byte loads by all threads (one HBM round trip for the whole
// tile) and broadcast from LDS (all lanes of a wave read the same address); the K range is split over the KW waves and
// reduced through LDS.  Activation rows are coalesced 256-byte segments served by L2.
constexpr int LIN_MAXK = 512;
constexpr int LIN_KPW = 64;  // activations per lane held in registers (K split over KW waves: K <= KW * LIN_KPW)

// Latency-oriented structure (these launches sit on a 160-deep dependent chain, each is ~one memory round trip):
// every thread first REQUESTS everything it will need -- its pieces of the [NT][K] weight tile, its up to 64
// activation rows, the bias -- and only then starts consuming, so the HBM/L2 latencies overlap instead of adding up.
template <int NT, int KW>
__global__ __launch_bounds__(64 * KW) void flow_linear_kernel(LinArgs a) {
    __shared__ __attribute__((aligned(16))) float wt[NT][LIN_MAXK + 4];
    __shared__ float red[KW][NT][64];
    constexpr int NTHR = 64 * KW;
    constexpr int WPT = (NT * LIN_MAXK + NTHR - 1) / NTHR;  // weight elements per thread
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * NT;
    const int b = blockIdx.y * 64 + lane;
    const int bl = b < a.B ? b : a.B - 1;
    const int g = n0 / a.group_rows;
    const float* inp = a.in + (long)g * a.in_group_stride + (long)bl * a.in_sb;
    const float* Wr = a.W + (long)n0 * a.ldw;
    const int K = a.K;
    // 1. weight tile requests (coalesced along k)
    float wreg[WPT];
    const int nw = NT * K;
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + u * NTHR;
        const int ic = i < nw ? i : 0;
        const int j = ic / K, k = ic - j * K;
        wreg[u] = (nw > 0 && n0 + j < a.N) ? Wr[(long)j * a.ldw + k] : 0.f;
    }
    // 2. activation requests: this wave's K slice
    const int kc = (K + KW - 1) / KW;
    const int k0 = w * kc;
    const int kn = max(0, min(K, k0 + kc) - k0);
    float v[LIN_KPW];
#pragma unroll
    for (int u = 0; u < LIN_KPW; ++u) v[u] = u < kn ? inp[(long)(k0 + u) * a.in_sk] : 0.f;
    // 3. bias requests for the rows this wave finalises
    float bias[(NT + KW - 1) / KW];
#pragma unroll
    for (int jj = 0; jj < (NT + KW - 1) / KW; ++jj) {
        const int j = w + jj * KW;
        const int n = n0 + j;
        float s = 0.f;
        if (j < NT && n < a.N && b < a.B) {
            if (a.bias_vec) s += a.bias_vec[n];
            if (a.bias_mat) s += a.bias_mat[(long)n * a.B + b];
        }
        bias[jj] = s;
    }
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + u * NTHR;
        if (i < nw) { const int j = i / K; wt[j][i - j * K] = wreg[u]; }
    }
    __syncthreads();
    float acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = 0.f;
#pragma unroll
    for (int u = 0; u < LIN_KPW; u += 4) {
        if (u < kn) {  // wave-uniform
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 w4 = *reinterpret_cast<const float4*>(&wt[j][(k0 + u) & ~3]);  // k0 is a multiple of 4 or K tiny
                if ((k0 & 3) == 0) {  // (pad columns of wt are never written: guard every product, 0 * stale-NaN = NaN)
                    acc[j] = fmaf(w4.x, v[u], acc[j]);
                    if (u + 1 < kn) acc[j] = fmaf(w4.y, v[u + 1], acc[j]);
                    if (u + 2 < kn) acc[j] = fmaf(w4.z, v[u + 2], acc[j]);
                    if (u + 3 < kn) acc[j] = fmaf(w4.w, v[u + 3], acc[j]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (u + q < kn) acc[j] = fmaf(wt[j][k0 + u + q], v[u + q], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) red[w][j][lane] = acc[j];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < (NT + KW - 1) / KW; ++jj) {
        const int j = w + jj * KW;
        if (j < NT) {
            float s = bias[jj];
#pragma unroll
            for (int q = 0; q < KW; ++q) s += red[q][j][lane];
            const int n = n0 + j;
            if (n < a.N && b < a.B) {
                s = s >= 0.f ? s : s * a.slope;
                a.out[(long)n * a.out_sn + (long)b * a.out_sb] = s;
            }
        }
    }
}

template <int NT, int KW>
inline int launch_linear(const LinArgs& a, hipStream_t st) {
    I2V_REQUIRE(a.K <= LIN_MAXK && a.K <= KW * LIN_KPW, I2V_E_INVALID, "linear: K = %d exceeds %d", a.K, KW * LIN_KPW < LIN_MAXK ? KW * LIN_KPW : LIN_MAXK);
    dim3 grid((a.N + NT - 1) / NT, (a.B + 63) / 64);
    hipLaunchKernelGGL((flow_linear_kernel<NT, KW>), grid, dim3(64 * KW), 0, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}


}  // namespace i2v

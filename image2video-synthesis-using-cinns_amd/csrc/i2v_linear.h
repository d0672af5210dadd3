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

template <int NT, int KW>
__global__ __launch_bounds__(64 * KW) void flow_linear_kernel(LinArgs a) {
    __shared__ __attribute__((aligned(16))) float wt[NT][LIN_MAXK + 4];
    __shared__ float red[KW][NT][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * NT;
    const int b = blockIdx.y * 64 + lane;
    const int bl = b < a.B ? b : a.B - 1;
    const int g = n0 / a.group_rows;
    const float* inp = a.in + (long)g * a.in_group_stride + (long)bl * a.in_sb;
    const float* Wr = a.W + (long)n0 * a.ldw;
    // weight tile -> LDS (rows are only 4-byte aligned in general: scalar dword loads, still fully coalesced)
    for (int i = tid; i < NT * a.K; i += 64 * KW) {
        const int j = i / a.K, k = i - j * a.K;
        wt[j][k] = (n0 + j < a.N) ? Wr[(long)j * a.ldw + k] : 0.f;
    }
    const int kc = (a.K + KW - 1) / KW;
    const int k0 = w * kc;
    const int k1 = min(a.K, k0 + kc);
    float acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = 0.f;
    __syncthreads();
    int k = k0;
    for (; k + 16 <= k1; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = inp[(long)(k + u) * a.in_sk];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[j] = fmaf(wt[j][k + u], v[u], acc[j]);
        }
    }
    for (; k < k1; ++k) {
        const float v = inp[(long)k * a.in_sk];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = fmaf(wt[j][k], v, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) red[w][j][lane] = acc[j];
    __syncthreads();
    for (int j = w; j < NT; j += KW) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < KW; ++q) s += red[q][j][lane];
        const int n = n0 + j;
        if (n < a.N && b < a.B) {
            if (a.bias_vec) s += a.bias_vec[n];
            if (a.bias_mat) s += a.bias_mat[(long)n * a.B + b];
            s = s >= 0.f ? s : s * a.slope;
            a.out[(long)n * a.out_sn + (long)b * a.out_sb] = s;
        }
    }
}

template <int NT, int KW>
inline int launch_linear(const LinArgs& a, hipStream_t st) {
    I2V_REQUIRE(a.K <= LIN_MAXK, I2V_E_INVALID, "linear: K = %d exceeds %d", a.K, LIN_MAXK);
    dim3 grid((a.N + NT - 1) / NT, (a.B + 63) / 64);
    hipLaunchKernelGGL((flow_linear_kernel<NT, KW>), grid, dim3(64 * KW), 0, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}


}  // namespace i2v

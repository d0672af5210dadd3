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

// Hidden layers of the chain (in and out both [rows][Bp] with Bp = batch rounded up to 64): the cost of such a launch
// is set by the NUMBER of memory requests, not by bytes (tools/flow_bench.hip: 13.5 us at 64 dword loads per lane,
// independent of the batch).  So every lane loads float4 = 4 consecutive samples of a row, 16 lanes cover a 64-sample
// row, one wave instruction fetches 4 rows: 16 requests per wave instead of 64.  Lane = (kr = lane >> 4, bq = lane & 15);
// the four kr groups of a wave are reduced with two wavefront shuffles, the KW waves through LDS.
struct HidArgs {
    const float* W;   // rows [N][ldw], row-group g = n / group_rows reads input group g
    int ldw, K;
    const float* in;  // [groups][K][Bp]
    long in_group_stride;
    int group_rows;
    const float* bias;  // [N]
    float* out;         // [N][Bp]
    int N, Bp;
    float slope;
};

template <int KW>
__global__ __launch_bounds__(64 * KW) void flow_hidden_kernel(HidArgs a) {
    constexpr int NT = 4;
    __shared__ __attribute__((aligned(16))) float wtT[LIN_MAXK][NT];  // transposed tile: one ds_read_b128 = 4 rows' weights
    __shared__ __attribute__((aligned(16))) float4 red[KW][NT][16];
    constexpr int NTHR = 64 * KW;
    constexpr int WPT = (NT * LIN_MAXK + NTHR - 1) / NTHR;
    constexpr int NI = LIN_MAXK / (4 * KW);  // float4 row loads per lane
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kr = lane >> 4, bq = lane & 15;
    const int n0 = blockIdx.x * NT;
    const int b0 = blockIdx.y * 64;
    const int g = n0 / a.group_rows;
    const int K = a.K;
    const float* Wr = a.W + (long)n0 * a.ldw;
    float wreg[WPT];
    const int nw = NT * K;
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + u * NTHR;
        const int ic = i < nw ? i : 0;
        const int j = ic / K, k = ic - j * K;
        wreg[u] = (n0 + j < a.N) ? Wr[(long)j * a.ldw + k] : 0.f;
    }
    const int kc = K / KW;  // multiple of 4 (K is a multiple of 64)
    const int k0 = w * kc;
    const int ni = kc >> 2;
    const float* inp = a.in + (long)g * a.in_group_stride + (long)(k0 + kr) * a.Bp + b0 + 4 * bq;
    float4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = i < ni ? *reinterpret_cast<const float4*>(inp + (long)(4 * i) * a.Bp) : make_float4(0.f, 0.f, 0.f, 0.f);
    float bias = 0.f;
    if (tid < NT * 16 && n0 + (tid >> 4) < a.N) bias = a.bias[n0 + (tid >> 4)];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + u * NTHR;
        if (i < nw) { const int j = i / K; wtT[i - j * K][j] = wreg[u]; }
    }
    __syncthreads();
    float4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i < ni) {
            const float4 w4 = *reinterpret_cast<const float4*>(&wtT[k0 + 4 * i + kr][0]);
            const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j].x = fmaf(wj[j], v[i].x, acc[j].x);
                acc[j].y = fmaf(wj[j], v[i].y, acc[j].y);
                acc[j].z = fmaf(wj[j], v[i].z, acc[j].z);
                acc[j].w = fmaf(wj[j], v[i].w, acc[j].w);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            acc[j].x += __shfl_xor(acc[j].x, off);
            acc[j].y += __shfl_xor(acc[j].y, off);
            acc[j].z += __shfl_xor(acc[j].z, off);
            acc[j].w += __shfl_xor(acc[j].w, off);
        }
        if (kr == 0) red[w][j][bq] = acc[j];
    }
    __syncthreads();
    if (tid < NT * 16) {
        const int j = tid >> 4, q = tid & 15;
        float4 s = make_float4(bias, bias, bias, bias);
#pragma unroll
        for (int x = 0; x < KW; ++x) {
            const float4 r = red[x][j][q];
            s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        }
        s.x = s.x >= 0.f ? s.x : s.x * a.slope; s.y = s.y >= 0.f ? s.y : s.y * a.slope;
        s.z = s.z >= 0.f ? s.z : s.z * a.slope; s.w = s.w >= 0.f ? s.w : s.w * a.slope;
        if (n0 + j < a.N) *reinterpret_cast<float4*>(a.out + (long)(n0 + j) * a.Bp + b0 + 4 * q) = s;
    }
}

inline int launch_hidden(const HidArgs& a, hipStream_t st) {
    I2V_REQUIRE(a.K <= LIN_MAXK && a.K % 64 == 0 && a.Bp % 64 == 0 && a.N % 4 == 0, I2V_E_INVALID,
                "hidden layer: K = %d / Bp = %d / N = %d unsupported", a.K, a.Bp, a.N);
    hipLaunchKernelGGL((flow_hidden_kernel<8>), dim3(a.N / 4, a.Bp / 64), dim3(512), 0, st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
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

// Does the fp16 matrix-core rate under power management depend on how many mantissa bits of the operands are live?
// Runs the conv kernel's MFMA skeleton (12 x v_mfma_f32_32x32x16_f16 per k-step, 4 accumulators, 2 waves/SIMD) with
//   mode 0: all operands zero          mode 1: all operands full random mantissas
//   mode 2: "hi" operands random, "lo" operands (2 of the 3 MFMAs' second factor) truncated to `bits` mantissa bits
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_power_probe.hip -o tools/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ _Float16 rnd(unsigned tid, unsigned k, int bits, float scale) {
    unsigned h = (tid * 2654435761u + k * 40503u) ^ ((tid + k) * 2246822519u);
    h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
    const unsigned mant = (h >> 7) & 1023u;                    // 10 explicit mantissa bits
    const unsigned keep = bits >= 10 ? 1023u : (~((1u << (10 - bits)) - 1u) & 1023u);
    const unsigned short u = (unsigned short)(((h >> 31) << 15) | (15u << 10) | (mant & keep));  // +-[1,2)
    _Float16 v; __builtin_memcpy(&v, &u, 2);
    return (_Float16)((float)v * scale);
}

__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, int mode, int bits) {
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    half8 ah[2], al[2], bh[2], bl[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) {
            ah[i][j] = mode == 0 ? (_Float16)0.f : rnd(tid, i * 8 + j, 10, 1.f);
            bh[i][j] = mode == 0 ? (_Float16)0.f : rnd(tid, 100 + i * 8 + j, 10, 1.f);
            al[i][j] = mode == 0 ? (_Float16)0.f : rnd(tid, 200 + i * 8 + j, mode == 2 ? bits : 10, 4.8e-4f);
            bl[i][j] = mode == 0 ? (_Float16)0.f : rnd(tid, 300 + i * 8 + j, mode == 2 ? bits : 10, 4.8e-4f);
        }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int wm = 0; wm < 2; ++wm)
#pragma unroll
            for (int wn = 0; wn < 2; ++wn) acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[wm], bh[wn], acc[wm][wn], 0, 0, 0);
#pragma unroll
        for (int wm = 0; wm < 2; ++wm)
#pragma unroll
            for (int wn = 0; wn < 2; ++wn) acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[wm], bl[wn], acc[wm][wn], 0, 0, 0);
#pragma unroll
        for (int wm = 0; wm < 2; ++wm)
#pragma unroll
            for (int wn = 0; wn < 2; ++wn) acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[wm], bh[wn], acc[wm][wn], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int wgs = 2048, iters = argc > 1 ? atoi(argv[1]) : 4096;
    float* out;
    hipMalloc(&out, (size_t)wgs * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfgs[][2] = {{0, 0}, {1, 10}, {2, 8}, {2, 6}, {2, 4}, {2, 2}, {2, 0}, {1, 10}, {0, 0}};
    for (auto& c : cfgs) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(wgs), dim3(512), 0, 0, out, iters, c[0], c[1]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) best = ms < best ? ms : best;
        }
        const double flops = (double)wgs * 8 * iters * 12.0 * 2 * 32 * 32 * 16;
        printf("mode %d lo-bits %2d: %7.3f ms  %7.1f TFLOP/s fp16 MFMA\n", c[0], c[1], best, flops / best / 1e9);
    }
    return 0;
}

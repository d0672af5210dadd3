// Micro-benchmark of the cINN hidden-layer launch (flow_linear_kernel) in a dependent chain, with ablations.
// hipcc -O3 --offload-arch=gfx950 -I<csrc> tools/flow_bench.hip <csrc>/i2v_common.hip -o tools/flow_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "i2v_linear.h"

using namespace i2v;

template <int NT, int KW>
float run(const float* W, const float* bias, float* hA, float* hB, int B, int K, int N, int iters, int mode) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float* cur = hA; float* nxt = hB;
    auto go = [&]() {
        LinArgs m{};
        m.W = W; m.ldw = K; m.K = mode == 1 ? 32 : K; m.in = cur; m.in_sk = B; m.in_sb = 1; m.in_group_stride = (long)(N / 2) * B;
        m.group_rows = N / 2; m.bias_vec = bias; m.bias_mat = nullptr; m.out = nxt; m.out_sn = B; m.out_sb = 1; m.N = N; m.B = B; m.slope = 0.01f;
        launch_linear<NT, KW>(m, nullptr);
        std::swap(cur, nxt);
    };
    for (int i = 0; i < 10; ++i) go();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) go();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

__global__ void empty_kernel(float* p) { if (p == nullptr) return; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, K = 512, N = 1024;
    std::vector<float> w((size_t)80 * N * K, 0.001f), b(N, 0.f), h((size_t)N * B, 0.5f);
    float *dW, *db, *hA, *hB;
    (void)hipMalloc(&dW, w.size() * 4); (void)hipMalloc(&db, N * 4); (void)hipMalloc(&hA, h.size() * 4); (void)hipMalloc(&hB, h.size() * 4);
    (void)hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(hA, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    {   // launch floor: chain of empty kernels
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, nullptr, hA);
        (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, nullptr, hA);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel chain (256 x 512 threads): %.2f us per launch\n", ms * 1e3f / 200);
    }
    printf("B=%d  K=512 N=1024 (same weights every launch -> L2/MALL resident)\n", B);
    printf("  NT=4 KW=8 (256 WG x 512 thr): %.2f us\n", run<4, 8>(dW, db, hA, hB, B, K, N, 200, 0));
    printf("  NT=8 KW=8 (128 WG x 512 thr): %.2f us\n", run<8, 8>(dW, db, hA, hB, B, K, N, 200, 0));
    printf("  NT=2 KW=8 (512 WG x 512 thr): %.2f us\n", run<2, 8>(dW, db, hA, hB, B, K, N, 200, 0));
    printf("  NT=4 KW=8, K=32 (layer-0 shape)  : %.2f us\n", run<4, 8>(dW, db, hA, hB, B, K, N, 200, 1));
    printf("  NT=1 KW=8 (1024 WG x 512 thr): %.2f us\n", run<1, 8>(dW, db, hA, hB, B, K, N, 200, 0));
    return 0;
}

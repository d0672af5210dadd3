// Stand-alone benchmark of the split-fp16 conv kernel on one layer shape (default: g_3.conv_0 of the BAIR
// decoder: [B,16,64,64,256] -> 128).  Build: hipcc -O3 --offload-arch=gfx950 -I<csrc> tools/conv16_bench.hip
//   <csrc>/i2v_conv16.hip <csrc>/i2v_common.hip -o conv16_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "i2v_conv.h"

using namespace i2v;

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 8, T = 16, H = 64, W = 64, Cin = argc > 2 ? atoi(argv[2]) : 256,
        Cout = argc > 3 ? atoi(argv[3]) : 128;
    std::vector<float> w((size_t)Cout * Cin * 27), bias(Cout, 0.1f);
    srand(1);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    Conv16Weights cw;
    if (cw.pack(w.data(), bias.data(), Cout, Cin, 3, 3, 3, 1.0)) { printf("pack: %s\n", i2v_last_error()); return 1; }
    const size_t npos = (size_t)B * T * H * W;
    std::vector<_Float16> in(npos * Cin * 2);
    for (size_t i = 0; i < in.size(); ++i) in[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * (((i >> 3) & 1) ? 4.8e-4f : 1.0f));  // hi | lo groups
    void* din; float* dout;
    hipMalloc(&din, in.size() * 2);
    hipMalloc(&dout, npos * Cout * 4);
    hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice);
    const double flops = 2.0 * npos * Cin * Cout * 27.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it)
        if (conv16_forward(cw, din, dout, nullptr, 1, 1, B, T, H, W, 0, nullptr)) { printf("err %s\n", i2v_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int n = 5;
    for (int it = 0; it < n; ++it) conv16_forward(cw, din, dout, nullptr, 1, 1, B, T, H, W, 0, nullptr);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= n;
    printf("[B=%d,16,64,64,%d] -> %d: %8.3f ms  %7.1f TFLOP/s algorithmic\n", B, Cin, Cout, ms, flops / ms / 1e9);
    return 0;
}

// Stand-alone benchmark of the direct split-fp16 conv kernel on one layer shape.
//   conv16_bench B T H W Cin Cout [tdup=0] [splitk=1] [n=20]
// (defaults: g_0.conv_1 of the BAIR decoder at B = 64: [64,2,8,8,1024] -> 1024; tdup=1: T is the OUTPUT frame count, the input has T/2)
// Prints ms per launch, TFLOP/s (27-tap algorithmic) and an exact checksum of the output (sum of the float bit patterns), so that two
// builds / switch settings can be compared for bit-equality.  Build (tools/build_measurement_libs.sh does it):
//   hipcc -O3 --offload-arch=gfx950 -DC16_TUNE -I<csrc> -I<include> tools/conv16_bench.hip <csrc>/i2v_conv16.hip <csrc>/i2v_common.hip -o tools/conv16_bench
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "i2v_conv.h"

using namespace i2v;

int main(int argc, char** argv) {
    auto arg = [&](int i, int d) { return argc > i ? atoi(argv[i]) : d; };
    const int B = arg(1, 64), T = arg(2, 2), H = arg(3, 8), W = arg(4, 8), Cin = arg(5, 1024), Cout = arg(6, 1024), tdup = arg(7, 0),
              use_split = arg(8, 1), n = arg(9, 20);
    std::vector<float> w((size_t)Cout * Cin * 27), bias(Cout, 0.1f);
    // (an own generator: the HIP runtime's start-up consumes rand() values, which made two processes' inputs differ)
    uint64_t lcg = 88172645463325252ull;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (float)((lcg >> 40) & 0xFFFFFF) / 16777216.f; };
    for (auto& v : w) v = (rnd() - 0.5f) * 0.05f;
    Conv16Weights cw;
    if (tdup ? cw.pack_tdup(w.data(), bias.data(), Cout, Cin, 1.0) : cw.pack(w.data(), bias.data(), Cout, Cin, 3, 3, 3, 1.0)) {
        printf("pack: %s\n", i2v_last_error());
        return 1;
    }
    const int Tin = tdup ? T / 2 : T;
    const size_t npos_in = (size_t)B * Tin * H * W, npos = (size_t)B * T * H * W;
    std::vector<_Float16> in(npos_in * Cin * 2);
    for (size_t i = 0; i < in.size(); ++i) in[i] = (_Float16)((rnd() - 0.5f) * (((i >> 3) & 1) ? 4.8e-4f : 1.0f));  // hi | lo groups
    void* din; float* dout; float* ws = nullptr;
    (void)hipMalloc(&din, in.size() * 2);
    (void)hipMalloc(&dout, npos * Cout * 4);
    (void)hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice);
    const size_t ws_floats = use_split ? (size_t)8 * npos * Cout : 0;
    if (ws_floats) (void)hipMalloc(&ws, ws_floats * 4);
    const double flops = 2.0 * npos * Cin * Cout * 27.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it)
        if (conv16_forward(cw, din, dout, nullptr, 1, 1, B, T, H, W, 0, nullptr, nullptr, nullptr, ws, ws_floats)) { printf("err %s\n", i2v_last_error()); return 1; }
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int it = 0; it < n; ++it) conv16_forward(cw, din, dout, nullptr, 1, 1, B, T, H, W, 0, nullptr, nullptr, nullptr, ws, ws_floats);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= n;
    std::vector<uint32_t> out(npos * Cout);
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    uint64_t sum = 0;
    for (uint32_t v : out) sum += v;
    const char* clip = getenv("I2V_C16_CLIP");
    const char* perm = getenv("I2V_C16_PERM");
    printf("[B=%d,T=%d,%d,%d,%d] -> %d tdup=%d splitk=%d clip=%s perm=%s: %8.4f ms  %7.1f TFLOP/s algorithmic  checksum %016llx\n", B, T, H, W, Cin, Cout, tdup,
           use_split ? conv16_splitk_factor((long)T * H * W, cw.nchunk) : 1, clip ? clip : "1", perm ? perm : "1", ms, flops / ms / 1e9, (unsigned long long)sum);
    return 0;
}

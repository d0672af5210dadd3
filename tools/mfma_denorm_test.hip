// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950 with hipcc's default mode register?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval, float bval) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)aval; b[j] = (_Float16)bval; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    const float tests[][2] = {{1.0f, 1.0f}, {5.96e-8f, 1024.f}, {3e-6f, 1024.f}, {6.0e-5f, 1.0f}, {3e-6f, 3e-6f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g (as fp16 %g) b=%g: mfma sum over K=16 -> %g (expected %g)\n", t[0], h[1], t[1], h[0], 16.0 * (double)h[1] * (double)(float)(_Float16)t[1]);
    }
    return 0;
}

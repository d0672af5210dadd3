// Checks the LDS-free xor reductions used by the fused statistics of the Winograd conv kernels (csrc/i2v_common.h:
// wave_xor_add_f64<8|16|32>: DPP row_ror:8, v_permlane16_swap, v_permlane32_swap) against a host sum, lane by lane.
//   hipcc -O3 --offload-arch=gfx950 tools/permlane_test.hip -o tools/permlane_test && tools/permlane_test
#include <hip/hip_runtime.h>
__device__ inline double xadd8(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);
    int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ inline double xadd16(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ inline double xadd32(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__global__ void k(const double* in, double* out) {
    double v = in[threadIdx.x];
    v = xadd8(v); v = xadd16(v); v = xadd32(v);
    out[threadIdx.x] = v;
}
int main() {
    double h[64], *d, *o; for (int i = 0; i < 64; ++i) h[i] = (double)(1 << (i % 8)) * (1 + (i / 8) * 0.001);
    hipMalloc(&d, 512); hipMalloc(&o, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); double r[64]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { double e = 0; for (int j = i % 8; j < 64; j += 8) e += h[j]; if (r[i] != e) { ++bad; printf("lane %d got %.6f want %.6f\n", i, r[i], e); } }
    printf("permlane reduction: %d bad lanes\n", bad); return bad != 0;
}

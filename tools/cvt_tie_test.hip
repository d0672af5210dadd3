// How do the three f32 -> f16 conversions the compiler picks for the split-fp16 operand writers round?  (round 6: the operand-generating
// F(4,3) kernel and modulate_wino4_kernel differ in the last bit of ~0.5 % of the conv outputs although their source expressions are
// the same.)  v_cvt_f16_f32, v_cvt_pk_f16_f32 and v_fma_mixlo_f16 (x * 1 + 0) on ties, near-ties, fp16 subnormals and large values,
// compared with the host's round-to-nearest-even.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned short* a, unsigned short* b, unsigned short* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    unsigned r0, r1, r2 = 0;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r0) : "v"(v));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r1) : "v"(v));
    asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, 0" : "+v"(r2) : "v"(v));
    a[i] = (unsigned short)r0; b[i] = (unsigned short)r1; c[i] = (unsigned short)r2;
}
int main() {
    std::vector<float> x;
    for (int e = -26; e <= 16; ++e)
        for (int m = 0; m < 64; ++m) {
            // mantissa patterns around the fp16 rounding boundary: 10 kept bits, then 1000..0 (tie), 0111..1, 1000..01
            for (unsigned tail : {0x1000u, 0x0fffu, 0x1001u, 0x0800u, 0x1800u}) {
                unsigned bits = ((unsigned)(e + 127) << 23) | ((unsigned)(m * 131 % 1024) << 13) | tail;
                float f; memcpy(&f, &bits, 4); x.push_back(f); x.push_back(-f);
            }
        }
    const int n = (int)x.size();
    float* dx; unsigned short *da, *db, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 2); hipMalloc(&db, n * 2); hipMalloc(&dc, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, dc, n);
    std::vector<unsigned short> a(n), b(n), c(n);
    hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 2, hipMemcpyDeviceToHost);
    long da_ = 0, db_ = 0, dc_ = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        const _Float16 h = (_Float16)x[i];   // host: round to nearest even
        unsigned short hb; memcpy(&hb, &h, 2);
        da_ += a[i] != hb; db_ += b[i] != hb; dc_ += c[i] != hb;
        if ((a[i] != hb || b[i] != hb || c[i] != hb) && shown++ < 12)
            printf("  x = %a  host %04x  v_cvt_f16_f32 %04x  v_cvt_pk_f16_f32 %04x  v_fma_mixlo_f16 %04x\n", x[i], hb, a[i], b[i], c[i]);
    }
    printf("%d values: differ from the host's RNE: v_cvt_f16_f32 %ld, v_cvt_pk_f16_f32 %ld, v_fma_mixlo_f16 %ld\n", n, da_, db_, dc_);
    return 0;
}

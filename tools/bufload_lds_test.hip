// Does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a raw buffer descriptor) WRITE ZEROS into LDS for lanes whose offset is out
// of range?  (The F(4,3) kernel's zero padding could then be an out-of-range offset instead of a select against a zero page.)
//   hipcc -O3 --offload-arch=gfx950 tools/bufload_lds_test.hip -o tools/bufload_lds_test && tools/bufload_lds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* p, unsigned nbytes, unsigned soff, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned sm[4 * 256];
    for (int i = threadIdx.x; i < 4 * 256; i += 256) sm[i] = 0xAAAAAAAAu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    unsigned lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sm + (threadIdx.x >> 6) * 1024;
    lds = __builtin_amdgcn_readfirstlane(lds);
    const int lane = threadIdx.x & 63;
    // lanes with (lane & 4): padding row (row index -1); the others: row (thread >> 2), 16-byte piece (thread & 3)
    const int row = (lane & 4) ? (lane & 8 ? -1 : (1 << 25)) : (int)(threadIdx.x >> 2);   // two padding markers: -1 (0xFFFFFFC0 | piece) and 2^25 (2^31 | piece)
    const unsigned off = ((unsigned)row << 6) | ((threadIdx.x & 3) * 16);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(r), "s"(lds), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += 256) out[i] = sm[i];
}
int main() {
    const unsigned n = 2 * 4096 + 4096;   // two "chunks" of 4 KB + slack
    std::vector<unsigned> h(n / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    char* d; unsigned* o;
    hipMalloc(&d, n); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    int bad_pad = 0, bad_data0 = 0, all_zero_soff = 0, data_soff = 0;
    for (unsigned soff : {0u, 4096u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, 4096u, soff, o);   // num_records = ONE 4 KB chunk
        std::vector<unsigned> r(1024);
        hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; ++t)
            for (int c = 0; c < 4; ++c) {
                const bool pad = (t & 63) & 4;
                const unsigned want = 0x1000000u + soff / 4 + (t >> 2) * 16 + (t & 3) * 4 + c;
                const unsigned got = r[t * 4 + c];
                if (pad) bad_pad += got != 0u;                       // padding lanes (both markers) must have written zeros over the 0xAA fill
                else if (soff == 0) bad_data0 += got != want;
                else { all_zero_soff += got == 0u; data_soff += got == want; }
            }
    }
    printf("buffer_load_dwordx4 ... offen lds on gfx950:\n");
    printf("  out-of-range lanes (offset 0xFFFFFFC0 | piece and 2^31 | piece) WRITE ZEROS into LDS: %s (%d wrong dwords)\n", bad_pad ? "NO" : "yes", bad_pad);
    printf("  in-range lanes deliver their 16 bytes: %s (%d wrong dwords)\n", bad_data0 ? "NO" : "yes", bad_data0);
    printf("  with soffset = num_records the in-range lanes returned: %d dwords of data, %d zeros -> the scalar offset %s part of the range check\n",
           data_soff, all_zero_soff, all_zero_soff > data_soff ? "IS" : "is NOT");
    printf("  (the F(4,3) kernel therefore uses num_records = the whole sample, soffset = the chunk, padding = 2^31: out of range either way, no 32-bit wrap)\n");
    const int bad = bad_pad + bad_data0;
    return bad != 0;
}

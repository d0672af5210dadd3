// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this repository (MI355X_MICROARCH.md: "calibrate
// on a known byte count in your own access pattern").  Each kernel reads a KNOWN number of bytes exactly once from a 2 GiB
// buffer (far beyond the 256 MB Infinity Cache):
//   calib_wide     every lane 16 B, a wave 1 KB contiguous                      (the guide's reference pattern: FETCH = 1/2)
//   calib_rows256  a wave instruction = 4 groups of 256 contiguous bytes (4 rows of 64 B, like the V rows of one (t, h)
//                  line of a Winograd brick: TJ = 4 tiles x 64 B), groups 1 KB apart
//   calib_rows64   a wave instruction = 16 isolated 64-byte rows, 256 B apart
// Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o pmc -- tools/fetch_calib
// tools/pmc_hbm_traffic.py --calibrate does both and prints bytes / (FETCH_SIZE * 1024) per pattern.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void calib_wide(const char* p, float* sink, size_t n16) {
    v4f acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        acc += *reinterpret_cast<const v4f*>(p + i * 16);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) *sink = acc[0];
}
// lane l of wave-instruction i: group g = l >> 4 (16 lanes = 256 B), address = (i * 4 + g) * 1024 + (l & 15) * 16
__global__ void calib_rows256(const char* p, float* sink, size_t ninstr) {
    v4f acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < ninstr; i += nwave) acc += *reinterpret_cast<const v4f*>(p + (i * 4 + (lane >> 4)) * 1024 + (lane & 15) * 16);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) *sink = acc[0];
}
// lane l of wave-instruction i: row r = l >> 2 (4 lanes = 64 B), address = (i * 16 + r) * 256 + (l & 3) * 16
__global__ void calib_rows64(const char* p, float* sink, size_t ninstr) {
    v4f acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < ninstr; i += nwave) acc += *reinterpret_cast<const v4f*>(p + (i * 16 + (lane >> 2)) * 256 + (lane & 3) * 16);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) *sink = acc[0];
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char* buf; float* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(calib_wide, dim3(4096), dim3(256), 0, nullptr, buf, sink, bytes / 16);
    hipLaunchKernelGGL(calib_rows256, dim3(4096), dim3(256), 0, nullptr, buf, sink, bytes / 4096);
    hipLaunchKernelGGL(calib_rows64, dim3(4096), dim3(256), 0, nullptr, buf, sink, bytes / 4096);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    // bytes actually requested by the lanes
    printf("calib_wide %zu\ncalib_rows256 %zu\ncalib_rows64 %zu\n", bytes, bytes / 4, bytes / 4);
    return 0;
}

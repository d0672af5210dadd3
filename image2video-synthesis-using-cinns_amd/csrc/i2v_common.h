// Shared host-side helpers of libi2v_hip.so (error plumbing, state_dict lookup, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/i2v_hip.h"

namespace i2v {

void set_error(const char* fmt, ...);

#define I2V_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            i2v::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return I2V_E_HIP;                                                                \
        }                                                                                    \
    } while (0)

#define I2V_REQUIRE(cond, code, ...)  \
    do {                              \
        if (!(cond)) {                \
            i2v::set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

// state_dict view: name -> host tensor
struct StateDict {
    std::unordered_map<std::string, const i2v_tensor*> map;
    StateDict(const i2v_tensor* t, int n) {
        for (int i = 0; i < n; ++i) map.emplace(t[i].name, &t[i]);
    }
    // returns nullptr (and sets the error) when missing / wrong size / wrong dtype
    const float* f32(const std::string& key, int64_t numel) const {
        auto it = map.find(key);
        if (it == map.end()) { set_error("state_dict key '%s' missing", key.c_str()); return nullptr; }
        if (it->second->dtype != I2V_F32 || it->second->numel != numel || !it->second->data) {
            set_error("state_dict key '%s': expected %lld float32 elements, got %lld (dtype %d)", key.c_str(),
                      (long long)numel, (long long)it->second->numel, it->second->dtype);
            return nullptr;
        }
        return static_cast<const float*>(it->second->data);
    }
    const int64_t* i64(const std::string& key, int64_t numel) const {
        auto it = map.find(key);
        if (it == map.end()) { set_error("state_dict key '%s' missing", key.c_str()); return nullptr; }
        if (it->second->dtype != I2V_I64 || it->second->numel != numel || !it->second->data) {
            set_error("state_dict key '%s': expected %lld int64 elements", key.c_str(), (long long)numel);
            return nullptr;
        }
        return static_cast<const int64_t*>(it->second->data);
    }
    bool has(const std::string& key) const { return map.find(key) != map.end(); }
};

// Owning device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;             // owning: movable only (containers of weight structs may reallocate)
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int upload(const void* host, size_t n) {
        release();
        if (n == 0) return I2V_OK;
        I2V_HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
        I2V_HIP_CHECK(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return I2V_OK;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Is `st` capturing a graph right now?  (Errors of the query -- a stream of another context, an old runtime -- count as "no".)
inline bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

// Cross-stream serialisation of the calls on ONE handle (one workspace, one set of device-side pointer blocks): an event is
// recorded behind every call, and a call that arrives on another stream than the previous one first waits for it.
// Graph capture (round 5 advisor finding): an event recorded OUTSIDE a capture must not be waited on inside it ("dependency
// created on uncaptured work in another stream" -- the usual torch.cuda.graph recipe warms up on stream A and captures on stream
// B), and an event recorded INSIDE a capture must not be waited on by later eager work.  While `st` captures, the helper
// therefore neither waits nor records and forgets the previous call: ordering a graph against earlier eager work on other
// streams is the capturing caller's job (torch.cuda.graph synchronises before it begins the capture).
struct StreamOrder {
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr;
    bool have_last = false;
    StreamOrder() = default;
    StreamOrder(const StreamOrder&) = delete;
    StreamOrder& operator=(const StreamOrder&) = delete;
    ~StreamOrder() { if (last_done) (void)hipEventDestroy(last_done); }
    int entry(hipStream_t st) {
        if (stream_is_capturing(st)) { have_last = false; return I2V_OK; }
        if (have_last && last_stream != st) I2V_HIP_CHECK(hipStreamWaitEvent(st, last_done, 0));
        return I2V_OK;
    }
    void exit(hipStream_t st) {
        if (stream_is_capturing(st)) { have_last = false; return; }
        if (!last_done && hipEventCreateWithFlags(&last_done, hipEventDisableTiming) != hipSuccess) { last_done = nullptr; return; }
        if (hipEventRecord(last_done, st) == hipSuccess) { last_stream = st; have_last = true; }
    }
};
// records the end of a call on its stream when the scope is left (also on the error paths: whatever was enqueued is ordered)
struct StreamOrderMark {
    StreamOrder* o; hipStream_t st;
    ~StreamOrderMark() { o->exit(st); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize has to be raised once PER DEVICE (every device has its own copy of the code
// object); `done` = the call site's static bool[I2V_MAX_DEV].
// A handle owns device memory (packed weights) on the device that was current when it was created; every call that
// enqueues work must run with that device current (one handle per GPU / rank).
#define I2V_REQUIRE_DEVICE(bound, what)                                                                              \
    do {                                                                                                             \
        int cur_ = -1;                                                                                               \
        I2V_HIP_CHECK(hipGetDevice(&cur_));                                                                          \
        I2V_REQUIRE(cur_ == (bound), I2V_E_INVALID,                                                                  \
                    "%s: the handle lives on HIP device %d but the current device is %d (one handle per GPU; make its "  \
                    "device current before the call)", what, (bound), cur_);                                         \
    } while (0)

constexpr int I2V_MAX_DEV = 64;
// 4 KiB of zeros on the current device (allocated once per device, never freed): conv kernels point the loads of padding
// rows at it instead of selecting zeros AFTER the load -- a select right behind a prefetch load makes the wave wait for it.
int zero_page(const char** out);
inline int ensure_dynamic_lds(const void* kernel, int bytes, bool* done) {
    int dev = 0;
    I2V_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < I2V_MAX_DEV && done[dev]) return I2V_OK;
    I2V_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < I2V_MAX_DEV) done[dev] = true;
    return I2V_OK;
}

#ifdef __HIPCC__
// v + (v of lane ^ OFF) for OFF = 8, 16, 32 without the LDS crossbar (a 64-bit __shfl_xor is two ds_bpermute_b32 and their
// round trip): a DPP rotation inside the 16-lane row, v_permlane16_swap / v_permlane32_swap (gfx950) across rows and halves.
// Both operands of the add are the same two values in every lane pair, so the result has the bits of the shuffle version.
template <int OFF>
__device__ __forceinline__ double wave_xor_add_f64(double v) {
    static_assert(OFF == 8 || OFF == 16 || OFF == 32, "lane distance");
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    if constexpr (OFF == 8) {
        const int lo2 = __builtin_amdgcn_update_dpp(0, (int)lo, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
        const int hi2 = __builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, false);
        return v + __hiloint2double(hi2, lo2);
    } else if constexpr (OFF == 16) {
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    } else {
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
}
#endif

}  // namespace i2v

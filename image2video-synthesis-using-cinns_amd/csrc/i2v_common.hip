// Error plumbing and library-level entry points of libi2v_hip.so.
#include "i2v_common.h"

namespace i2v {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int zero_page(const char** out) {
    static char* pages[I2V_MAX_DEV] = {};
    int dev = 0;
    I2V_HIP_CHECK(hipGetDevice(&dev));
    I2V_REQUIRE(dev >= 0 && dev < I2V_MAX_DEV, I2V_E_HIP, "zero_page: device index %d", dev);
    if (!pages[dev]) {
        void* p = nullptr;
        I2V_HIP_CHECK(hipMalloc(&p, 4096));
        I2V_HIP_CHECK(hipMemset(p, 0, 4096));
        pages[dev] = static_cast<char*>(p);
    }
    *out = pages[dev];
    return I2V_OK;
}

}  // namespace i2v

extern "C" {

const char* i2v_last_error(void) { return i2v::g_err; }

int i2v_version(void) { return 1; }

int i2v_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        i2v::set_error("hipGetDeviceCount failed");
        return I2V_E_HIP;
    }
    return n;
}

}  // extern "C"

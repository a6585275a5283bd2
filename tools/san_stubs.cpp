// san_stubs.cpp -- what tools/san_host.cpp links instead of the library's .hip files (tests/test_sanitizers.py): the sanitizer
// builds hold no device code, so the library's OWN GPU-side classes (the gzip reader and encoder on the GPU, the device
// allocator) are stand-ins that say "no device" -- the host paths under test never reach them (device = -1 everywhere).
// HIP itself is the real libamdhip64.  Nothing here is product code.
#include <stdarg.h>
#include <stdio.h>

#include <string>

#include "nh_codec.h"
#include "nh_gunzip.h"
#include "nohuman_engine.h"

namespace nh {
thread_local std::string g_last_error;
int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
hipError_t dev_malloc(void **p, size_t) {
    *p = nullptr;
    return hipErrorNoDevice;
}
hipError_t dev_set(int) { return hipErrorNoDevice; }
bool dev_gunzip_wants(const char *) { return false; }
StreamEncoder *make_gpu_gzip_encoder(int, int, const char *) {
    set_error(NH_EDEVICE, "no device in the sanitizer build");
    return nullptr;
}
static const std::string g_no_device = "no device in the sanitizer build";
DevGunzip::DevGunzip() : impl_(nullptr) {}
DevGunzip::~DevGunzip() {}
int DevGunzip::open(const char *, int, size_t, size_t, std::string &err) {
    err = g_no_device;
    return -1;
}
long DevGunzip::next(void *, size_t, hipStream_t) { return -1; }
const std::string &DevGunzip::error() const { return g_no_device; }
void DevGunzip::close() {}
}  // namespace nh

extern "C" const char *nh_last_error(void) { return nh::g_last_error.c_str(); }

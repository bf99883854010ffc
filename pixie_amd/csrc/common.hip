// pixie_amd/csrc/common.hip -- error string + build probe for libpixie_hip.so.
#include "common.h"

#include <mutex>
#include <set>
#include <utility>

#include "../../include/pixie_hip.h"

namespace pixie {
hipError_t allow_max_dynamic_lds(const void* kern) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kern, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({kern, dev});
    return e;
}

std::string& last_error_ref() {
    static thread_local std::string msg;
    return msg;
}
int set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return 1;
}
}  // namespace pixie

extern "C" const char* pixie_last_error(void) { return pixie::last_error_ref().c_str(); }
extern "C" const char* pixie_build_arch(void) { return "gfx950"; }

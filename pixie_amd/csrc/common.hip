// pixie_amd/csrc/common.hip -- error string + build probe for libpixie_hip.so.
#include "common.h"

#include "../../include/pixie_hip.h"

namespace pixie {
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

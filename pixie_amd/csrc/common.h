// pixie_amd/csrc/common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

namespace pixie {

// thread-local message returned by pixie_last_error()
std::string& last_error_ref();
int set_error(const char* fmt, ...);

#define PX_CHECK_HIP(expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return ::pixie::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define PX_REQUIRE(cond, ...)                                     \
    do {                                                          \
        if (!(cond)) return ::pixie::set_error(__VA_ARGS__);      \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KB for `kern` on the CURRENT device, once per (kernel, device), thread-safe
// (a process may drive several GPUs; a function-local `static bool` would cover only the first).  Defined in common.hip.
hipError_t allow_max_dynamic_lds(const void* kern);
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace pixie

// b4d_common.hpp -- error plumbing shared by the translation units of libb4d.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/b4d.h"

namespace b4d {

// thread-local message behind b4d_last_error(); defined in b4d_kernels.hip
std::string& last_error();
inline int fail(int code, const std::string& msg) {
    last_error() = msg;
    return code;
}
// lazily grown device scratch for second-stage reductions (b4d_stats.hip)
int get_scratch(size_t bytes, void** out);

#define B4D_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ::b4d::fail(B4D_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

}  // namespace b4d

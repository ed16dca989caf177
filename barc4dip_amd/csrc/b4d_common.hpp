// b4d_common.hpp -- error plumbing shared by the translation units of libb4d.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>

#include "../../include/b4d.h"

namespace b4d {

// thread-local message behind b4d_last_error(); defined in b4d_kernels.hip
std::string& last_error();
inline int fail(int code, const std::string& msg) {
    last_error() = msg;
    return code;
}
// lazily grown device scratch for second-stage reductions (b4d_stats.hip), one buffer per caller stream
int get_scratch(size_t bytes, void** out, hipStream_t stream);
// serialises the host side of every entry point that works in a scratch buffer (re-entrant calls from several host
// threads, e.g. joblib workers, on one stream); device-side ordering comes from the stream, and different streams get
// different buffers
std::recursive_mutex& scratch_mutex();
#define B4D_SCRATCH_LOCK() std::lock_guard<std::recursive_mutex> b4d_scratch_lk__(::b4d::scratch_mutex())

// b4d_set_option switches (defined next to it, b4d_kernels.hip): routes only, never results; read once per entry-point call
extern std::atomic<int> g_opt_track_predict;   // "track_predict_bin" 0 / 1 / 2
extern std::atomic<int> g_opt_lanes;           // "lanes": two-lane launch groups (Lanes, b4d_fft2d.hpp); 0 = the caller's stream only
extern std::atomic<int> g_opt_exp;             // "exp": development A/B switch (kernel variants under test; 0 = shipped)

// hipFuncAttributeMaxDynamicSharedMemorySize >= bytes for `kernel` on the CURRENT device, set once per (kernel address, device):
// a flag per launcher template would be per process (a second GPU of the process never gets the attribute), one per
// function-pointer type would be shared by kernels of one signature.  Defined in b4d_kernels.hip.
int ensure_dynamic_lds(const void* kernel, size_t bytes);
// The library's own streams (b4d_kernels.hip): two non-blocking streams per device, created on first use and shared by every
// plan -- a process has few hardware queues (4 by default), and a second lane that lands on the queue of the caller's stream
// serialises behind it (measured: a two-lane Wiener pass 5.8 k frames/s against 7.3 k on one lane, 8.0 k on a queue of its own)
int lane_stream(int idx, hipStream_t* out);

#define B4D_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ::b4d::fail(B4D_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

}  // namespace b4d

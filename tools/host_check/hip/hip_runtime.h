// Host stand-in for <hip/hip_runtime.h>: lets tools/host_check/*.cpp run the engine's __device__ code on the CPU
// (one lane at a time; the phases between barriers are looped over `tid`).  Dev tool only; never part of libb4d.so.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __restrict__
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline void __syncthreads() {}
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int min(int a, int b) { return a < b ? a : b; }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)

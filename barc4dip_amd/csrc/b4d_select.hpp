// b4d_select.hpp -- exact order statistics of float32 data by 3-pass radix select (11 + 11 + 10 bits)
// inside one workgroup: LDS histogram, wave-level scan, no sorting.  Keys are the order-preserving
// unsigned image of the float bits; NaNs are skipped (np.nanpercentile / np.median on finite maps).
#pragma once
#include <hip/hip_runtime.h>

namespace b4d {

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// k-th smallest (0-based) among the non-NaN elements of x[0..n).  Whole workgroup participates and gets
// the key; n_less / n_equal = number of elements strictly below / equal to it.  hist: 2048 LDS words,
// sh: 4 LDS words.
__device__ inline unsigned radix_select(const float* __restrict__ x, unsigned n, unsigned k, unsigned* hist,
                                        unsigned* sh, unsigned& n_less, unsigned& n_equal) {
    unsigned prefix = 0, mask = 0, below = 0;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sft = shifts[pass], nb = 1 << widths[pass];
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
            const float f = x[i];
            if (f != f) continue;
            const unsigned key = f2key(f);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> sft) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // one wave: 32 bins per lane, then a wave scan
            const int per = 2048 / 64;
            unsigned s = 0;
            for (int i = 0; i < per; ++i) s += hist[threadIdx.x * per + i];
            unsigned incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o, 64);
                if ((int)threadIdx.x >= o) incl += t;
            }
            const unsigned excl = incl - s, kk = k - below;
            if (kk >= excl && kk < incl) {  // the bin is in this lane's range
                unsigned run = excl;
                for (int i = 0; i < per; ++i) {
                    const unsigned c = hist[threadIdx.x * per + i];
                    if (kk < run + c) {
                        sh[0] = threadIdx.x * per + i;
                        sh[1] = run;
                        sh[2] = c;
                        break;
                    }
                    run += c;
                }
            }
        }
        __syncthreads();
        prefix |= sh[0] << sft;
        mask |= (unsigned)(nb - 1) << sft;
        below += sh[1];
        n_equal = sh[2];
        __syncthreads();
    }
    n_less = below;
    return prefix;
}

// smallest key strictly greater than `ka` among the non-NaN elements (0xffffffff if none); whole workgroup.
__device__ inline unsigned next_larger_key(const float* __restrict__ x, unsigned n, unsigned ka, unsigned* scratch16) {
    unsigned best = 0xffffffffu;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const float f = x[i];
        if (f != f) continue;
        const unsigned key = f2key(f);
        if (key > ka && key < best) best = key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_down(best, o, 64);
        best = t < best ? t : best;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch16[threadIdx.x >> 6] = best;
    __syncthreads();
    best = scratch16[0];
    for (int i = 1; i < (int)((blockDim.x + 63) >> 6); ++i) best = scratch16[i] < best ? scratch16[i] : best;
    __syncthreads();
    return best;
}

}  // namespace b4d

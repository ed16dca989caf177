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
// REP > 1: `hist` holds REP copies of the 2048-bin histogram, lanes spread over them (lane % REP): maps whose values
// crowd a few exponent bins (correlation magnitude maps) otherwise serialise on same-address LDS atomics.
// compact (optional, >= n floats of global scratch owned by this workgroup): after the first pass the elements of the
// selected 11-bit bin are gathered there and the two remaining passes (and the caller's next_larger_key) stream that
// short array instead of the whole map; *cx / *cn return the array / length to continue on.
// pass_begin = 1 resumes after a first pass done elsewhere (k_row_c2r counts and gathers the expected median bin): x then
// holds only the elements of the selected top-11-bit bin `prefix0 >> 21`, below0 = number of elements under that bin.
template <int REP = 1>
__device__ inline unsigned radix_select(const float* __restrict__ x, unsigned n, unsigned k, unsigned* hist,
                                        unsigned* sh, unsigned& n_less, unsigned& n_equal, float* __restrict__ compact = nullptr,
                                        const float** cx = nullptr, unsigned* cn = nullptr, int pass_begin = 0, unsigned prefix0 = 0,
                                        unsigned below0 = 0) {
    unsigned prefix = prefix0, mask = pass_begin ? 0xffe00000u : 0u, below = below0;
    unsigned* myhist = hist + (threadIdx.x % REP) * 2048;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = pass_begin; pass < 3; ++pass) {
        const int sft = shifts[pass], nb = 1 << widths[pass];
        for (int i = threadIdx.x; i < 2048 * REP; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        auto visit = [&](float f) {
            if (f != f) return;
            const unsigned key = f2key(f);
            if ((key & mask) == prefix) atomicAdd(&myhist[(key >> sft) & (nb - 1)], 1u);
        };
        // 16-byte loads, two per lane in flight: the pass is a pure stream over the map
        const unsigned n4 = ((reinterpret_cast<size_t>(x) & 15) == 0) ? n / 4 : 0;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        unsigned i = threadIdx.x;
        for (; i + blockDim.x < n4; i += 2 * blockDim.x) {
            const float4 a = x4[i], b = x4[i + blockDim.x];
            visit(a.x); visit(a.y); visit(a.z); visit(a.w);
            visit(b.x); visit(b.y); visit(b.z); visit(b.w);
        }
        for (; i < n4; i += blockDim.x) {
            const float4 a = x4[i];
            visit(a.x); visit(a.y); visit(a.z); visit(a.w);
        }
        for (unsigned j = 4 * n4 + threadIdx.x; j < n; j += blockDim.x) visit(x[j]);
        __syncthreads();
        if (REP > 1) {  // fold the copies into the first
            for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
                unsigned t = hist[i];
#pragma unroll
                for (int r = 1; r < REP; ++r) t += hist[r * 2048 + i];
                hist[i] = t;
            }
            __syncthreads();
        }
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
        if (pass == 0 && compact) {  // gather the selected bin (wave-aggregated append: one LDS atomic per wave and element slot)
            if (threadIdx.x == 0) sh[3] = 0;
            __syncthreads();
            auto put = [&](float f) {
                const bool m = (f == f) && ((f2key(f) & mask) == prefix);
                const unsigned long long bal = __ballot(m);
                if (bal == 0) return;
                const int lane = threadIdx.x & 63, leader = __ffsll((long long)bal) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&sh[3], (unsigned)__popcll(bal));
                base = __shfl(base, leader, 64);
                if (m) compact[base + __popcll(bal & ((1ull << lane) - 1ull))] = f;
            };
            const unsigned m4 = ((reinterpret_cast<size_t>(x) & 15) == 0) ? n / 4 : 0;
            const float4* y4 = reinterpret_cast<const float4*>(x);
            const unsigned m4r = (m4 + blockDim.x - 1) / blockDim.x * blockDim.x;   // whole waves take part in every ballot
            for (unsigned i = threadIdx.x; i < m4r; i += blockDim.x) {
                const float4 a = i < m4 ? y4[i] : make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                put(a.x); put(a.y); put(a.z); put(a.w);
            }
            const unsigned tail = n - 4 * m4, tr = (tail + blockDim.x - 1) / blockDim.x * blockDim.x;
            for (unsigned j = threadIdx.x; j < tr; j += blockDim.x) put(j < tail ? x[4 * m4 + j] : __builtin_nanf(""));
            __threadfence_block();
            __syncthreads();
            x = compact;
            n = n_equal;
        }
    }
    n_less = below;
    if (cx) *cx = x;
    if (cn) *cn = n;
    return prefix;
}

// smallest key strictly greater than `ka` among the non-NaN elements (0xffffffff if none); whole workgroup.
__device__ inline unsigned next_larger_key(const float* __restrict__ x, unsigned n, unsigned ka, unsigned* scratch16) {
    unsigned best = 0xffffffffu;
    auto visit = [&](float f) {
        if (f != f) return;
        const unsigned key = f2key(f);
        if (key > ka && key < best) best = key;
    };
    const unsigned n4 = ((reinterpret_cast<size_t>(x) & 15) == 0) ? n / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (unsigned i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 a = x4[i];
        visit(a.x); visit(a.y); visit(a.z); visit(a.w);
    }
    for (unsigned j = 4 * n4 + threadIdx.x; j < n; j += blockDim.x) visit(x[j]);
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

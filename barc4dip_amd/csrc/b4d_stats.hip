// b4d_stats.hip -- streaming reductions of the barc4dip metrics layer on gfx950:
//   temporal per-pixel sums, finite-only distribution moments
//   (metrics/statistics.py:17-125), Sobel / Laplace statistics with scipy "reflect"
//   borders (metrics/sharpness.py:405-530).
// All of them are HBM-bound single-read passes; float64 accumulation (the reference computes
// these in float64), wave64 shuffles + one LDS hop per workgroup, and a fixed-order second
// stage so results are bitwise reproducible (no float atomics).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/b4d.h"
#include "b4d_common.hpp"
#include "b4d_select.hpp"

namespace b4d {

// ------------------------------------------------------------------------------------ scratch
// Lazily grown per-process device scratch for second-stage reductions (never reallocated on the
// hot path once it has reached its high-water mark).
static std::mutex g_scratch_mu;
struct StreamScratch {
    void* p = nullptr;
    size_t bytes = 0;
};
static std::unordered_map<hipStream_t, StreamScratch> g_stream_scratch;   // one lazily grown buffer per caller stream
std::recursive_mutex& scratch_mutex() {
    static std::recursive_mutex m;
    return m;
}
// One buffer per stream: the kernels of these entry points are asynchronous, so two streams sharing one scratch would
// race on the device however well the host side is locked (the plans and the Wiener plan cache are per stream too).
int get_scratch(size_t bytes, void** out, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    StreamScratch& sc = g_stream_scratch[stream];
    if (bytes > sc.bytes) {
        if (sc.p) (void)hipFree(sc.p);   // hipFree waits for the work that still uses the old buffer
        sc.p = nullptr;
        sc.bytes = 0;
        const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        hipError_t e = hipMalloc(&sc.p, want);
        if (e != hipSuccess) return fail(B4D_ENOMEM, std::string("scratch allocation: ") + hipGetErrorString(e));
        sc.bytes = want;
    }
    *out = sc.p;
    return B4D_OK;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum of K doubles per lane; result valid in thread 0.  blockDim.x <= 1024.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* sh /* [16*K] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) sh[w * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int i = 0; i < nw; ++i) s += sh[i * K + k];
            v[k] = s;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------ temporal
// One lane owns 4 adjacent pixels (one 16-B load per frame); frames are walked with TU loads in flight.
#ifndef B4D_TACC_UNROLL
#define B4D_TACC_UNROLL 4
#endif
constexpr int TU = B4D_TACC_UNROLL;
// Every frame word is read exactly once: streaming (non-temporal) loads keep the stack out of L2 -- 5.43 -> 5.94 TB/s on the
// cfg4 shard (interleaved A/B, tools/dev_ab_temporal.py).
typedef float tacc_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 tacc_load4(const float* p) {
    const tacc_f4 q = __builtin_nontemporal_load(reinterpret_cast<const tacc_f4*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}
__global__ void __launch_bounds__(256) k_temporal_acc(const float* __restrict__ frames, int nframes, size_t npix,
                                                      double* __restrict__ sx, double* __restrict__ sxx) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= npix) return;
    double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (i4 + 3 < npix) {
        const float* p = frames + i4;
        int t = 0;
        for (; t + TU <= nframes; t += TU) {
            float4 v[TU];
#pragma unroll
            for (int k = 0; k < TU; ++k) v[k] = tacc_load4(p + (size_t)(t + k) * npix);
#pragma unroll
            for (int k = 0; k < TU; ++k) {
                const double x0 = v[k].x, x1 = v[k].y, x2 = v[k].z, x3 = v[k].w;
                a[0] += x0; a[1] += x1; a[2] += x2; a[3] += x3;
                q[0] = fma(x0, x0, q[0]); q[1] = fma(x1, x1, q[1]); q[2] = fma(x2, x2, q[2]); q[3] = fma(x3, x3, q[3]);
            }
        }
        for (; t < nframes; ++t) {
            const float4 v = tacc_load4(p + (size_t)t * npix);
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            a[0] += x0; a[1] += x1; a[2] += x2; a[3] += x3;
            q[0] = fma(x0, x0, q[0]); q[1] = fma(x1, x1, q[1]); q[2] = fma(x2, x2, q[2]); q[3] = fma(x3, x3, q[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sx[i4 + k] += a[k];
            sxx[i4 + k] += q[k];
        }
    } else {  // ragged tail
        for (size_t i = i4; i < npix; ++i) {
            double s = 0, ss = 0;
            for (int t = 0; t < nframes; ++t) {
                const double x = frames[(size_t)t * npix + i];
                s += x;
                ss = fma(x, x, ss);
            }
            sx[i] += s;
            sxx[i] += ss;
        }
    }
}

// Any pixel count / alignment (1023 x 1023 frames: odd rows start on odd dwords): one lane per pixel, dword loads
// (still whole 256-byte lines per wave), four frames in flight.  `stride` = pixels from one frame to the next.
__global__ void __launch_bounds__(256) k_temporal_acc1(const float* __restrict__ frames, int nframes, size_t stride, size_t npix,
                                                       double* __restrict__ sx, double* __restrict__ sxx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float* p = frames + i;
    double a = 0, q = 0;
    int t = 0;
    for (; t + 4 <= nframes; t += 4) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(p + (size_t)(t + k) * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = v[k];
            a += x;
            q = fma(x, x, q);
        }
    }
    for (; t < nframes; ++t) {
        const double x = p[(size_t)t * stride];
        a += x;
        q = fma(x, x, q);
    }
    sx[i] += a;
    sxx[i] += q;
}

// 16-byte variant on a pixel range of strided frames (npix and every frame start a multiple of 4 pixels)
__global__ void __launch_bounds__(256) k_temporal_acc4(const float* __restrict__ frames, int nframes, size_t stride, size_t npix,
                                                       double* __restrict__ sx, double* __restrict__ sxx) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= npix) return;
    double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const float* p = frames + i4;
    auto take = [&](const float4& v) {
        const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
        a[0] += x0; a[1] += x1; a[2] += x2; a[3] += x3;
        q[0] = fma(x0, x0, q[0]); q[1] = fma(x1, x1, q[1]); q[2] = fma(x2, x2, q[2]); q[3] = fma(x3, x3, q[3]);
    };
    int t = 0;
    for (; t + 4 <= nframes; t += 4) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = tacc_load4(p + (size_t)(t + k) * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) take(v[k]);
    }
    for (; t < nframes; ++t) take(tacc_load4(p + (size_t)t * stride));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sx[i4 + k] += a[k];
        sxx[i4 + k] += q[k];
    }
}

// count by value (count_dev == nullptr) or from device memory (the all-reduced frame count: no host round trip)
__global__ void __launch_bounds__(256) k_temporal_fin(const double* __restrict__ sx, const double* __restrict__ sxx,
                                                      double count_val, const double* __restrict__ count_dev, size_t npix,
                                                      float* __restrict__ mean, float* __restrict__ var, float* __restrict__ contrast) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const double count = count_dev ? count_dev[0] : count_val;
    const double m = sx[i] / count;
    double v = sxx[i] / count - m * m;
    v = v > 0.0 ? v : 0.0;
    if (mean) mean[i] = (float)m;
    if (var) var[i] = (float)v;
    if (contrast) contrast[i] = (float)(sqrt(v) / m);
}

// ------------------------------------------------------------------------------------ moments
// pass 1: {n_finite, sum, n_zero, n_sat}; pass 2 (mean known): {sum d^2, sum d^3, sum d^4}
// grid (split, batch).  partials: [batch][split][4] doubles.
__global__ void __launch_bounds__(1024) k_moments1(const float* __restrict__ frames, size_t npix, double eps,
                                                   double sat, double* __restrict__ part) {
    __shared__ double sh[16 * 4];
    const float* f = frames + (size_t)blockIdx.y * npix;
    const size_t n4 = npix / 4;
    double v[4] = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 q = reinterpret_cast<const float4*>(f)[i];
        const float xs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = xs[k];
            if (isfinite(x)) {
                v[0] += 1.0;
                v[1] += x;
                v[2] += (fabs(x) <= eps) ? 1.0 : 0.0;
                v[3] += (x >= sat) ? 1.0 : 0.0;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
        const double x = f[n4 * 4 + threadIdx.x];
        if (isfinite(x)) {
            v[0] += 1.0;
            v[1] += x;
            v[2] += (fabs(x) <= eps) ? 1.0 : 0.0;
            v[3] += (x >= sat) ? 1.0 : 0.0;
        }
    }
    block_sum<4>(v, sh);
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k];
    }
}

__global__ void __launch_bounds__(1024) k_moments2(const float* __restrict__ frames, size_t npix,
                                                   const double* __restrict__ part1, int split1,
                                                   double* __restrict__ part) {
    __shared__ double sh[16 * 3];
    __shared__ double s_mean;
    if (threadIdx.x == 0) {
        double n = 0, s = 0;
        for (int i = 0; i < split1; ++i) {
            n += part1[((size_t)blockIdx.y * split1 + i) * 4 + 0];
            s += part1[((size_t)blockIdx.y * split1 + i) * 4 + 1];
        }
        s_mean = n > 0 ? s / n : 0.0;
    }
    __syncthreads();
    const double mu = s_mean;
    const float* f = frames + (size_t)blockIdx.y * npix;
    const size_t n4 = npix / 4;
    double v[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 q = reinterpret_cast<const float4*>(f)[i];
        const float xs[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = xs[k];
            if (isfinite(x)) {
                const double d = x - mu, d2 = d * d;
                v[0] += d2;
                v[1] = fma(d2, d, v[1]);
                v[2] = fma(d2, d2, v[2]);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
        const double x = f[n4 * 4 + threadIdx.x];
        if (isfinite(x)) {
            const double d = x - mu, d2 = d * d;
            v[0] += d2;
            v[1] = fma(d2, d, v[1]);
            v[2] = fma(d2, d2, v[2]);
        }
    }
    block_sum<3>(v, sh);
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        o[0] = v[0];
        o[1] = v[1];
        o[2] = v[2];
    }
}

// out[b] = {n_finite, mean, sum d^2, sum d^3, sum d^4, n_zero, n_sat, 0}
__global__ void k_moments_fin(const double* __restrict__ p1, const double* __restrict__ p2, int split,
                              double* __restrict__ out) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    double n = 0, s = 0, nz = 0, ns = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int i = 0; i < split; ++i) {
        const double* a = p1 + ((size_t)b * split + i) * 4;
        const double* c = p2 + ((size_t)b * split + i) * 3;
        n += a[0]; s += a[1]; nz += a[2]; ns += a[3];
        m2 += c[0]; m3 += c[1]; m4 += c[2];
    }
    double* o = out + (size_t)b * 8;
    o[0] = n; o[1] = n > 0 ? s / n : 0.0; o[2] = m2; o[3] = m3; o[4] = m4; o[5] = nz; o[6] = ns; o[7] = 0.0;
}

// ------------------------------------------------------------------------------------ Sobel / Laplace
// scipy.ndimage "reflect" = half-sample symmetric: index -1 -> 0, n -> n-1.
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }

// grid (ceil(nx/64), ceil(ny/16), batch), block (64, 4): each thread walks 4 rows of one column.
// partials [batch][gridDim.y*gridDim.x][5] = {n_finite, sum gx^2, sum gy^2, sum lap, sum lap^2}
__global__ void __launch_bounds__(256) k_sobel_lap(const float* __restrict__ frames, int ny, int nx,
                                                   double* __restrict__ part) {
    __shared__ double sh[16 * 5];
    const float* f = frames + (size_t)blockIdx.z * ny * nx;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    double v[5] = {0, 0, 0, 0, 0};
    if (x < nx) {
        const int xm = refl(x - 1, nx), xp = refl(x + 1, nx);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = blockIdx.y * 16 + threadIdx.y * 4 + r;
            if (y >= ny) break;
            const int ym = refl(y - 1, ny), yp = refl(y + 1, ny);
            const float* rm = f + (size_t)ym * nx;
            const float* r0 = f + (size_t)y * nx;
            const float* rp = f + (size_t)yp * nx;
            const double a = rm[xm], b = rm[x], c = rm[xp];
            const double d = r0[xm], e = r0[x], g = r0[xp];
            const double h = rp[xm], i = rp[x], j = rp[xp];
            if (isfinite(e)) {
                // scipy sobel(axis=1): derivative [-1,0,1] along x, smoothing [1,2,1] along y
                const double gx = (c - a) + 2.0 * (g - d) + (j - h);
                const double gy = (h - a) + 2.0 * (i - b) + (j - c);
                const double lap = (d + g - 2.0 * e) + (b + i - 2.0 * e);
                v[0] += 1.0;
                v[1] = fma(gx, gx, v[1]);
                v[2] = fma(gy, gy, v[2]);
                v[3] += lap;
                v[4] = fma(lap, lap, v[4]);
            }
        }
    }
    // block_sum indexes by threadIdx.x; flatten the 2-D block
    {
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            v[k] = wave_sum(v[k]);
            if (lane == 0) sh[w * 5 + k] = v[k];
        }
        __syncthreads();
        if (tid == 0) {
            double* o = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) o[k] = sh[k] + sh[5 + k] + sh[10 + k] + sh[15 + k];
        }
    }
}

// out[b] = {mean gx^2, mean gy^2, mean lap, mean lap^2}; fixed-order tree over the partials.
__global__ void __launch_bounds__(256) k_sobel_fin(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double sh[16 * 5];
    const double* p = part + (size_t)blockIdx.x * nblk * 5;
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += p[(size_t)i * 5 + k];
    block_sum<5>(v, sh);
    if (threadIdx.x == 0) {
        double* o = out + (size_t)blockIdx.x * 4;
        const double n = v[0];
        o[0] = v[1] / n; o[1] = v[2] / n; o[2] = v[3] / n; o[3] = v[4] / n;
    }
}


// ------------------------------------------------------------------------------------ percentiles
// np.nanpercentile(x, q) with the default linear interpolation (utils/range.py:44-54): for each q the two
// bracketing order statistics are selected exactly; out = {lo value, hi value, fraction, n_valid} and the
// caller finishes lo + (hi - lo) * t in float64 exactly like NumPy's _lerp.   grid (batch), block 1024.
__global__ void __launch_bounds__(1024) k_percentiles(const float* __restrict__ frames, size_t npix, const double* __restrict__ q,
                                                      int nq, double* __restrict__ out) {
#ifndef B4D_PCT_REP
#define B4D_PCT_REP 4
#endif
    constexpr int REP = B4D_PCT_REP;   // replicated histograms: intensity data crowd a few exponent bins (b4d_select.hpp)
    __shared__ unsigned hist[2048 * REP];
    __shared__ unsigned sh[4];
    __shared__ unsigned s_cnt[16];
    const float* x = frames + (size_t)blockIdx.x * npix;
    const unsigned n_all = (unsigned)npix;
    unsigned c = 0;
    for (unsigned i = threadIdx.x; i < n_all; i += blockDim.x) c += (x[i] == x[i]) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    unsigned n = 0;
    for (int i = 0; i < 16; ++i) n += s_cnt[i];
    __syncthreads();
    for (int iq = 0; iq < nq; ++iq) {
        double* o = out + ((size_t)blockIdx.x * nq + iq) * 4;
        if (n == 0) {
            if (threadIdx.x == 0) o[0] = o[1] = nan(""), o[2] = 0.0, o[3] = 0.0;
            continue;
        }
        const double pos = q[iq] / 100.0 * (double)(n - 1);
        unsigned lo = (unsigned)floor(pos);
        if (lo > n - 1) lo = n - 1;
        const unsigned hi = lo + 1 < n ? lo + 1 : n - 1;
        unsigned nl, ne;
        const unsigned ka = radix_select<REP>(x, n_all, lo, hist, sh, nl, ne);
        unsigned kb = ka;
        if (hi != lo && nl + ne <= hi) kb = next_larger_key(x, n_all, ka, hist);
        if (threadIdx.x == 0) {
            o[0] = (double)key2f(ka);
            o[1] = (double)key2f(kb);
            o[2] = pos - (double)lo;
            o[3] = (double)n;
        }
        __syncthreads();
    }
}

// ---- the same selection spread over many workgroups (large frames: one workgroup would stream the map ~9 times alone)
// State per (frame, query): st[0] = prefix, st[1] = elements below the current bin, st[2] = target rank, st[3] = count
// in the chosen bin, st[4] = n_valid, st[5] = smallest key above the selected one (next-larger pass).
struct PctMulti {
    const float* frames;
    size_t npix;
    unsigned* hist;    // (batch, nq, 2048), zero on entry of every pass
    unsigned* state;   // (batch, nq, 8)
    const double* q;
    double* out;
    int nq;
};

// pass: 0, 1, 2 (bits 31..21, 20..10, 9..0).  grid (NB, batch, pass == 0 ? 1 : nq), block 256
template <int PASS>
__global__ void __launch_bounds__(256) k_pctm_hist(PctMulti p) {
    constexpr int REP = 4;
    __shared__ unsigned hist[2048 * REP];
    const int f = blockIdx.y, j = blockIdx.z;
    const float* x = p.frames + (size_t)f * p.npix;
    const unsigned n = (unsigned)p.npix;
    constexpr int sft = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0), nb = PASS == 2 ? 1024 : 2048;
    constexpr unsigned mask = PASS == 0 ? 0u : (PASS == 1 ? 0xffe00000u : 0xfffffc00u);
    const unsigned prefix = PASS == 0 ? 0u : p.state[((size_t)f * p.nq + j) * 8];
    for (int i = threadIdx.x; i < 2048 * REP; i += 256) hist[i] = 0;
    __syncthreads();
    unsigned* my = hist + (threadIdx.x % REP) * 2048;
    auto visit = [&](float v) {
        if (v != v) return;
        const unsigned key = f2key(v);
        if ((key & mask) == prefix) atomicAdd(&my[(key >> sft) & (nb - 1)], 1u);
    };
    const unsigned n4 = ((reinterpret_cast<size_t>(x) & 15) == 0) ? n / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 a = x4[i];
        visit(a.x); visit(a.y); visit(a.z); visit(a.w);
    }
    if (blockIdx.x == 0)
        for (unsigned i = 4 * n4 + threadIdx.x; i < n; i += 256) visit(x[i]);
    __syncthreads();
    unsigned* g = p.hist + ((size_t)f * p.nq + j) * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256) {
        unsigned t = 0;
#pragma unroll
        for (int r = 0; r < REP; ++r) t += hist[r * 2048 + i];
        if (t) atomicAdd(&g[i], t);
    }
}

// choose the bin of the target rank, update the state, clear the histogram for the next pass.  grid (batch, nq), block 64.
// PASS 0 reads the shared pass-0 histogram of query 0 and derives n_valid and the rank from q.
template <int PASS>
__global__ void __launch_bounds__(64) k_pctm_pick(PctMulti p) {
    const int f = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    unsigned* st = p.state + ((size_t)f * p.nq + j) * 8;
    unsigned* g = p.hist + ((size_t)f * p.nq + (PASS == 0 ? 0 : j)) * 2048;
    constexpr int sft = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    const int per = 2048 / 64;
    unsigned s = 0;
    for (int i = 0; i < per; ++i) s += g[lane * per + i];
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const unsigned total = __shfl(incl, 63, 64);
    unsigned kk;
    if (PASS == 0) {
        const unsigned n = total;
        unsigned lo = 0;
        if (n > 0) {
            const double pos = p.q[j] / 100.0 * (double)(n - 1);
            lo = (unsigned)floor(pos);
            if (lo > n - 1) lo = n - 1;
        }
        kk = lo;
        if (lane == 0) {
            st[0] = 0;
            st[1] = 0;
            st[2] = lo;
            st[4] = n;
            st[5] = 0xffffffffu;
        }
    } else {
        kk = st[2] - st[1];
    }
    __syncthreads();
    const unsigned excl = incl - s;
    if (total > 0 && kk >= excl && kk < incl) {
        unsigned run = excl;
        for (int i = 0; i < per; ++i) {
            const unsigned c = g[lane * per + i];
            if (kk < run + c) {
                st[0] |= (unsigned)(lane * per + i) << sft;
                st[1] += run;
                st[3] = c;
                break;
            }
            run += c;
        }
    }
    __syncthreads();
    if (PASS != 0 || j == gridDim.y - 1 || true) {   // every query's own histogram slot is cleared for the next pass
        unsigned* mine = p.hist + ((size_t)f * p.nq + j) * 2048;
        if (PASS != 0 || j != 0)
            for (int i = lane; i < 2048; i += 64) mine[i] = 0;
    }
}

// pass-0 histogram lives in slot 0 and is read by every query's pick: clear it afterwards.  grid (batch), block 256
__global__ void __launch_bounds__(256) k_pctm_clear0(PctMulti p) {
    unsigned* g = p.hist + (size_t)blockIdx.x * p.nq * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256) g[i] = 0;
}

// smallest key above the selected one.  grid (NB, batch, nq), block 256
__global__ void __launch_bounds__(256) k_pctm_next(PctMulti p) {
    __shared__ unsigned sh[4];
    const int f = blockIdx.y, j = blockIdx.z;
    const float* x = p.frames + (size_t)f * p.npix;
    const unsigned n = (unsigned)p.npix;
    unsigned* st = p.state + ((size_t)f * p.nq + j) * 8;
    const unsigned ka = st[0];
    unsigned best = 0xffffffffu;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = x[i];
        if (v != v) continue;
        const unsigned key = f2key(v);
        if (key > ka && key < best) best = key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_down(best, o, 64);
        best = t < best ? t : best;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = sh[i] < best ? sh[i] : best;
        if (best != 0xffffffffu) atomicMin(&st[5], best);
    }
}

// out[f][j] = {lo value, hi value, fraction, n_valid}.  grid (batch), block 64 (lane = query)
__global__ void __launch_bounds__(64) k_pctm_out(PctMulti p) {
    const int f = blockIdx.x, j = threadIdx.x;
    if (j >= p.nq) return;
    const unsigned* st = p.state + ((size_t)f * p.nq + j) * 8;
    double* o = p.out + ((size_t)f * p.nq + j) * 4;
    const unsigned n = st[4];
    if (n == 0) {
        o[0] = o[1] = nan("");
        o[2] = 0.0;
        o[3] = 0.0;
        return;
    }
    const double pos = p.q[j] / 100.0 * (double)(n - 1);
    const unsigned lo = st[2], hi = lo + 1 < n ? lo + 1 : n - 1;
    unsigned kb = st[0];
    if (hi != lo && st[1] + st[3] <= hi && st[5] != 0xffffffffu) kb = st[5];
    o[0] = (double)key2f(st[0]);
    o[1] = (double)key2f(kb);
    o[2] = pos - (double)lo;
    o[3] = (double)n;
}

// ------------------------------------------------------------------------------------ radial profile
// maths/radial.py:101-169 radial_mean_interpolated: r = linspace(0, r_max, nr), theta = linspace(0, 2pi, ntheta,
// endpoint=False), bilinear RegularGridInterpolator on the pixel-centre grid (x = j - nx//2), fill 0 outside,
// mean over theta.  grid (nr, batch), block 256.
__global__ void __launch_bounds__(256) k_radial_profile(const float* __restrict__ maps, int ny, int nx, int nr, int ntheta,
                                                        double r_max, double* __restrict__ out) {
    __shared__ double sh[4];
    const float* z = maps + (size_t)blockIdx.y * ny * nx;
    const int ir = blockIdx.x;
    const double r = nr > 1 ? (ir == nr - 1 ? r_max : (double)ir * (r_max / (double)(nr - 1))) : 0.0;
    const double x0 = -(double)(nx / 2), x1 = (double)(nx - 1 - nx / 2), y0 = -(double)(ny / 2), y1 = (double)(ny - 1 - ny / 2);
    const double two_pi = 6.283185307179586476925286766559;
    double acc = 0.0;
    for (int it = threadIdx.x; it < ntheta; it += blockDim.x) {
        const double th = (double)it * (two_pi / (double)ntheta);
        const double px = r * cos(th), py = r * sin(th);
        double val = 0.0;
        if (px >= x0 && px <= x1 && py >= y0 && py <= y1) {
            const double fx = px - x0, fy = py - y0;  // index space
            int jx = (int)floor(fx), jy = (int)floor(fy);
            jx = jx > nx - 2 ? nx - 2 : jx;
            jy = jy > ny - 2 ? ny - 2 : jy;
            const double tx = fx - (double)jx, ty = fy - (double)jy;
            const float* p = z + (size_t)jy * nx + jx;
            const double v00 = p[0], v01 = p[1], v10 = p[nx], v11 = p[nx + 1];
            val = (1.0 - ty) * ((1.0 - tx) * v00 + tx * v01) + ty * ((1.0 - tx) * v10 + tx * v11);
        }
        acc += val;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * nr + ir] = (sh[0] + sh[1] + sh[2] + sh[3]) / (double)ntheta;
}

// ------------------------------------------------------------------------------------ PSD statistics
// metrics/speckles.py:669-817 (bandwidth) and metrics/sharpness.py:536-629 (spectral entropy) from ONE shifted
// PSD map whose DC bin is treated as zero (both functions remove the mean and zero the DC bin):
//   sums over the inscribed frequency disc FR <= min(max|fx|, max|fy|): S, sum FR^2 P, sum FX^2 P, sum FY^2 P, sum P^2
//   sum over ALL bins: S_all, and sum P ln P (entropy follows as ln S_all - sum(P ln P)/S_all)
//   f95: radius where the radius-ordered cumulative power reaches 0.95 S (two histogram passes: coarse integer
//        radius, then the exact r^2 values inside that ring) -- square maps only.
struct PsdStatArgs {
    const float* psd;
    int ny, nx, nblk;
    double* part;     // [batch][nblk][8]
    double* coarse;   // [batch][ncoarse]
    double* fine;     // [batch][nfine]
    const int* ring;  // [batch] coarse ring index (pass 2)
    int ncoarse, nfine;
};

__device__ __forceinline__ void psd_coords(int i, int j, int ny, int nx, double& fx, double& fy, long long& r2) {
    const int dy = i - ny / 2, dx = j - nx / 2;
    fx = (double)dx / (double)nx;
    fy = (double)dy / (double)ny;
    r2 = (long long)dx * dx + (long long)dy * dy;
}

// pass 1: partial sums + coarse radial histogram (bin = floor(sqrt(r2)), square maps).  grid (nblk, batch), block 256
__global__ void __launch_bounds__(256) k_psd_stats1(PsdStatArgs a, double fmax) {
    __shared__ double sh[16 * 7];
    extern __shared__ double lhist[];  // ncoarse
    const float* P = a.psd + (size_t)blockIdx.y * a.ny * a.nx;
    for (int i = threadIdx.x; i < a.ncoarse; i += blockDim.x) lhist[i] = 0.0;
    __syncthreads();
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    const size_t n = (size_t)a.ny * a.nx;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / a.nx), j = (int)(e % a.nx);
        double p = (double)P[e];
        if (i == a.ny / 2 && j == a.nx / 2) p = 0.0;
        if (!isfinite(p)) p = 0.0;  // np.nan_to_num(..., 0)
        double fx, fy;
        long long r2;
        psd_coords(i, j, a.ny, a.nx, fx, fy, r2);
        const double fr2 = __dadd_rn(__dmul_rn(fx, fx), __dmul_rn(fy, fy));
        const double fr = sqrt(fr2);
        v[5] += p;
        if (p > 0.0) v[6] = fma(p, log(p), v[6]);
        if (fr <= fmax) {
            v[0] += p;
            v[1] = fma(__dmul_rn(fr, fr), p, v[1]);
            v[2] = fma(__dmul_rn(fx, fx), p, v[2]);
            v[3] = fma(__dmul_rn(fy, fy), p, v[3]);
            v[4] = fma(p, p, v[4]);
            if (a.ncoarse > 0) {
                int b = (int)floor(sqrt((double)r2));
                b = b < a.ncoarse ? b : a.ncoarse - 1;
                atomicAdd(&lhist[b], p);
            }
        }
    }
    block_sum<7>(v, sh);
    if (threadIdx.x == 0) {
        double* o = a.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = v[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.ncoarse; i += blockDim.x)
        if (lhist[i] != 0.0) atomicAdd(&a.coarse[(size_t)blockIdx.y * a.ncoarse + i], lhist[i]);
}

// reduce partials -> out[b][0..6]; pick the coarse ring where the cumulative power crosses 0.95 S.  grid (batch), block 64
__global__ void k_psd_stats_mid(PsdStatArgs a, double* __restrict__ out, int* __restrict__ ring, double* __restrict__ below) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < a.nblk; ++i)
        for (int k = 0; k < 7; ++k) v[k] += a.part[((size_t)b * a.nblk + i) * 8 + k];
    for (int k = 0; k < 7; ++k) out[(size_t)b * 8 + k] = v[k];
    int r = a.ncoarse - 1;
    double cum = 0.0;
    if (a.ncoarse > 0) {
        double tot = 0.0;
        for (int i = 0; i < a.ncoarse; ++i) tot += a.coarse[(size_t)b * a.ncoarse + i];
        for (int i = 0; i < a.ncoarse; ++i) {
            const double c = a.coarse[(size_t)b * a.ncoarse + i];
            if ((cum + c) / tot >= 0.95) {
                r = i;
                break;
            }
            cum += c;
        }
        out[(size_t)b * 8 + 7] = tot;  // masked total as accumulated by the histogram (diagnostic)
    }
    ring[b] = r;
    below[b] = cum;
}

// pass 2: exact r^2 histogram inside the selected ring [R^2, (R+1)^2).  grid (nblk, batch), block 256
__global__ void __launch_bounds__(256) k_psd_stats2(PsdStatArgs a, double fmax) {
    extern __shared__ double lhist[];  // nfine
    const float* P = a.psd + (size_t)blockIdx.y * a.ny * a.nx;
    const int R = a.ring[blockIdx.y];
    const long long lo = (long long)R * R;
    for (int i = threadIdx.x; i < a.nfine; i += blockDim.x) lhist[i] = 0.0;
    __syncthreads();
    const size_t n = (size_t)a.ny * a.nx;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / a.nx), j = (int)(e % a.nx);
        double fx, fy;
        long long r2;
        psd_coords(i, j, a.ny, a.nx, fx, fy, r2);
        const long long d = r2 - lo;
        if (d < 0 || d >= a.nfine) continue;
        if ((int)floor(sqrt((double)r2)) != R) continue;
        const double fr = sqrt(__dadd_rn(__dmul_rn(fx, fx), __dmul_rn(fy, fy)));
        if (!(fr <= fmax)) continue;
        double p = (double)P[e];
        if (i == a.ny / 2 && j == a.nx / 2) p = 0.0;
        if (!isfinite(p)) p = 0.0;
        if (p != 0.0) atomicAdd(&lhist[(int)d], p);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.nfine; i += blockDim.x)
        if (lhist[i] != 0.0) atomicAdd(&a.fine[(size_t)blockIdx.y * a.nfine + i], lhist[i]);
}

// f95 = sqrt(r2*)/n of the first exact radius whose cumulative power reaches 0.95 S.  grid (batch), block 64
__global__ void k_psd_stats_fin(PsdStatArgs a, const int* __restrict__ ring, const double* __restrict__ below,
                                double* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    const double tot = out[(size_t)b * 8 + 7];
    const long long lo = (long long)ring[b] * ring[b];
    double cum = below[b];
    long long r2 = lo;
    bool found = false;
    for (int i = 0; i < a.nfine; ++i) {
        const double c = a.fine[(size_t)b * a.nfine + i];
        if (c == 0.0) continue;
        r2 = lo + i;
        if ((cum + c) / tot >= 0.95) {
            found = true;
            break;
        }
        cum += c;
    }
    (void)found;
    out[(size_t)b * 8 + 7] = sqrt((double)r2) / (double)a.nx;
}

}  // namespace b4d

using namespace b4d;

extern "C" {

int b4d_temporal_accumulate(const float* frames, int nframes, size_t npix, double* sum_x, double* sum_xx,
                            void* stream) {
    if (!frames || !sum_x || !sum_xx) return fail(B4D_EINVAL, "null argument");
    if (nframes < 1 || npix < 1) return fail(B4D_EINVAL, "nframes and npix must be >= 1");
    return b4d_temporal_accumulate_range(frames, nframes, npix, 0, npix, sum_x, sum_xx, stream);
}

int b4d_temporal_accumulate_range(const float* frames, int nframes, size_t frame_stride, size_t pix0, size_t npix, double* sum_x,
                                  double* sum_xx, void* stream) {
    if (!frames || !sum_x || !sum_xx) return fail(B4D_EINVAL, "null argument");
    if (nframes < 1 || npix < 1 || pix0 + npix > frame_stride) return fail(B4D_EINVAL, "bad frame count or pixel range");
    const float* f0 = frames + pix0;
    if ((reinterpret_cast<uintptr_t>(f0) & 15) == 0 && (npix & 3) == 0 && (frame_stride & 3) == 0) {
        if (pix0 == 0 && npix == frame_stride) {   // whole contiguous frames: the original kernel (ragged tail handled inside)
            hipLaunchKernelGGL(k_temporal_acc, dim3((unsigned)((npix / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames, nframes,
                               npix, sum_x, sum_xx);
        } else {
            hipLaunchKernelGGL(k_temporal_acc4, dim3((unsigned)((npix / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f0, nframes,
                               frame_stride, npix, sum_x, sum_xx);
        }
    } else {   // odd pixel counts / unaligned views (the reference's data.mean(axis=0) takes any shape, io/rw.py:129-132)
        hipLaunchKernelGGL(k_temporal_acc1, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f0, nframes,
                           frame_stride, npix, sum_x, sum_xx);
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int b4d_temporal_finalize(const double* sum_x, const double* sum_xx, double count, size_t npix, float* mean,
                          float* var, float* contrast, void* stream) {
    if (!sum_x || !sum_xx) return fail(B4D_EINVAL, "null argument");
    if (!(count > 0)) return fail(B4D_EINVAL, "count must be > 0");
    hipLaunchKernelGGL(k_temporal_fin, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sum_x,
                       sum_xx, count, (const double*)nullptr, npix, mean, var, contrast);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int b4d_temporal_finalize_dev(const double* sum_x, const double* sum_xx, const double* count_dev, size_t npix, float* mean,
                              float* var, float* contrast, void* stream) {
    if (!sum_x || !sum_xx || !count_dev) return fail(B4D_EINVAL, "null argument");
    hipLaunchKernelGGL(k_temporal_fin, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sum_x,
                       sum_xx, 0.0, count_dev, npix, mean, var, contrast);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int b4d_moments(const float* frames, int batch, size_t npix, double eps, double saturation, double* out,
                void* stream) {
    B4D_SCRATCH_LOCK();
    if (!frames || !out) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || npix < 1) return fail(B4D_EINVAL, "batch and npix must be >= 1");
    if ((reinterpret_cast<uintptr_t>(frames) & 15) || (npix & 3))
        return fail(B4D_EINVAL, "frames must be 16-byte aligned and npix a multiple of 4");
    int split = 2048 / batch;
    split = split < 1 ? 1 : (split > 256 ? 256 : split);
    const size_t work = (npix / 4 + 1023) / 1024;
    if ((size_t)split > work) split = (int)(work ? work : 1);
    void* ws = nullptr;
    int rc = get_scratch(sizeof(double) * 7 * (size_t)batch * split, &ws, (hipStream_t)stream);
    if (rc) return rc;
    double* p1 = static_cast<double*>(ws);
    double* p2 = p1 + (size_t)4 * batch * split;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_moments1, dim3(split, batch), dim3(1024), 0, st, frames, npix, eps, saturation, p1);
    hipLaunchKernelGGL(k_moments2, dim3(split, batch), dim3(1024), 0, st, frames, npix, p1, split, p2);
    hipLaunchKernelGGL(k_moments_fin, dim3(batch), dim3(64), 0, st, p1, p2, split, out);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int b4d_sobel_laplace_stats(const float* frames, int batch, int ny, int nx, double* out, void* stream) {
    B4D_SCRATCH_LOCK();
    if (!frames || !out) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || ny < 1 || nx < 1) return fail(B4D_EINVAL, "batch, ny, nx must be >= 1");
    const dim3 grid((nx + 63) / 64, (ny + 15) / 16, batch);
    const int nblk = grid.x * grid.y;
    void* ws = nullptr;
    int rc = get_scratch(sizeof(double) * 5 * (size_t)batch * nblk, &ws, (hipStream_t)stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sobel_lap, grid, dim3(64, 4), 0, st, frames, ny, nx, static_cast<double*>(ws));
    hipLaunchKernelGGL(k_sobel_fin, dim3(batch), dim3(256), 0, st, static_cast<const double*>(ws), nblk, out);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}


int b4d_percentiles(const float* frames, int batch, size_t npix, const double* q_host, int nq, double* out,
                    void* stream) {
    B4D_SCRATCH_LOCK();
    if (!frames || !q_host || !out) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || npix < 1 || nq < 1 || nq > 16) return fail(B4D_EINVAL, "batch, npix >= 1 and 1 <= nq <= 16 required");
    if (npix > 0xffffffffull) return fail(B4D_EINVAL, "frame too large");
    hipStream_t st = (hipStream_t)stream;
    const bool multi = npix >= ((size_t)1 << 17);
    const size_t hist_words = multi ? (size_t)batch * nq * 2048 : 0, state_words = multi ? (size_t)batch * nq * 8 : 0;
    void* ws = nullptr;
    int rc = get_scratch(sizeof(double) * 16 + sizeof(unsigned) * (hist_words + state_words), &ws, (hipStream_t)stream);
    if (rc) return rc;
    B4D_HIP(hipMemcpyAsync(ws, q_host, sizeof(double) * nq, hipMemcpyHostToDevice, st));
    B4D_HIP(hipStreamSynchronize(st));
    if (!multi) {
        hipLaunchKernelGGL(k_percentiles, dim3(batch), dim3(1024), 0, st, frames, npix, static_cast<const double*>(ws), nq, out);
    } else {  // large frames: every pass of the radix select runs on the whole chip
        PctMulti pm{};
        pm.frames = frames;
        pm.npix = npix;
        pm.q = static_cast<const double*>(ws);
        pm.hist = reinterpret_cast<unsigned*>(static_cast<double*>(ws) + 16);
        pm.state = pm.hist + hist_words;
        pm.out = out;
        pm.nq = nq;
        const int nbw = (int)std::min<size_t>(256, (npix / 4 + 255) / 256);
        B4D_HIP(hipMemsetAsync(pm.hist, 0, sizeof(unsigned) * (hist_words + state_words), st));
        hipLaunchKernelGGL((k_pctm_hist<0>), dim3(nbw, batch, 1), dim3(256), 0, st, pm);
        hipLaunchKernelGGL((k_pctm_pick<0>), dim3(batch, nq), dim3(64), 0, st, pm);
        hipLaunchKernelGGL(k_pctm_clear0, dim3(batch), dim3(256), 0, st, pm);
        hipLaunchKernelGGL((k_pctm_hist<1>), dim3(nbw, batch, nq), dim3(256), 0, st, pm);
        hipLaunchKernelGGL((k_pctm_pick<1>), dim3(batch, nq), dim3(64), 0, st, pm);
        hipLaunchKernelGGL((k_pctm_hist<2>), dim3(nbw, batch, nq), dim3(256), 0, st, pm);
        hipLaunchKernelGGL((k_pctm_pick<2>), dim3(batch, nq), dim3(64), 0, st, pm);
        hipLaunchKernelGGL(k_pctm_next, dim3(nbw, batch, nq), dim3(256), 0, st, pm);
        hipLaunchKernelGGL(k_pctm_out, dim3(batch), dim3(64), 0, st, pm);
    }
    B4D_HIP(hipGetLastError());
    B4D_HIP(hipStreamSynchronize(st));  // the shared scratch holds q (and the selection state) until the kernels have run
    return B4D_OK;
}

int b4d_radial_profile(const float* maps, int batch, int ny, int nx, int nr, int ntheta, double r_max, double* out,
                       void* stream) {
    B4D_SCRATCH_LOCK();
    if (!maps || !out) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || ny < 2 || nx < 2 || nr < 2 || ntheta < 4 || !(r_max > 0)) return fail(B4D_EINVAL, "bad shape / sampling");
    hipLaunchKernelGGL(k_radial_profile, dim3(nr, batch), dim3(256), 0, (hipStream_t)stream, maps, ny, nx, nr, ntheta, r_max, out);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int b4d_psd_stats(const float* psd, int batch, int ny, int nx, double* out, void* stream) {
    B4D_SCRATCH_LOCK();
    if (!psd || !out) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || ny < 2 || nx < 2) return fail(B4D_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool square = ny == nx;
    PsdStatArgs a{};
    a.psd = psd;
    a.ny = ny;
    a.nx = nx;
    a.nblk = 256;
    a.ncoarse = square ? nx / 2 + 2 : 0;
    a.nfine = square ? 2 * (nx / 2 + 2) + 1 : 0;
    const size_t nd = (size_t)batch * (8 * a.nblk + a.ncoarse + a.nfine + 2) + 64;
    void* ws = nullptr;
    int rc = get_scratch(sizeof(double) * nd + sizeof(int) * batch, &ws, (hipStream_t)stream);
    if (rc) return rc;
    double* base = static_cast<double*>(ws);
    a.part = base;
    a.coarse = a.part + (size_t)batch * 8 * a.nblk;
    a.fine = a.coarse + (size_t)batch * a.ncoarse;
    double* below = a.fine + (size_t)batch * a.nfine;
    int* ring = reinterpret_cast<int*>(below + batch + 2);
    a.ring = ring;
    B4D_HIP(hipMemsetAsync(a.coarse, 0, sizeof(double) * (size_t)batch * (a.ncoarse + a.nfine), st));
    // f_max = min(max|fx|, max|fy|) of the shifted fftfreq axes
    auto amax = [](int n) { return (double)(n / 2) / (double)n; };
    const double fmax = std::fmin(amax(nx), amax(ny));
    hipLaunchKernelGGL(k_psd_stats1, dim3(a.nblk, batch), dim3(256), sizeof(double) * a.ncoarse, st, a, fmax);
    hipLaunchKernelGGL(k_psd_stats_mid, dim3(batch), dim3(64), 0, st, a, out, ring, below);
    if (square) {
        hipLaunchKernelGGL(k_psd_stats2, dim3(a.nblk, batch), dim3(256), sizeof(double) * a.nfine, st, a, fmax);
        hipLaunchKernelGGL(k_psd_stats_fin, dim3(batch), dim3(64), 0, st, a, ring, below, out);
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

}  // extern "C"

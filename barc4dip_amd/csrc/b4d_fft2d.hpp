// b4d_fft2d.hpp -- batched 2-D real FFT pipeline kernels (row R2C, fused column, row C2R), the plan
// object and the launch dispatchers.  Included by the translation units that launch them
// (b4d_kernels.hip: fft2d / psd2d / autocorr2d; b4d_track.hip: xcorr2d / phase correlation).
//
// Data flow for a batch of real (ny, nx) frames ("rows first"):
//
//   K1 row_r2c   two real rows are packed as one complex row (z = a + i b), one FFT of
//                length nx, Hermitian split -> half spectra of both rows.  The half
//                spectrum keeps kx = 0..nx/2-1; the (real) Nyquist bin rides in the imaginary
//                part of the (real) DC bin, so a row is exactly nx/2 complex values.
//                Written in a column-tile-major layout: tile ct holds CT adjacent kx for
//                all ny rows contiguously ([ct][y][c]), so that K2 streams whole tiles.
//   K2 col       one workgroup owns a tile (CT = 16 columns x ny rows, 256 KiB at 2048^2)
//                entirely in registers: forward FFT along y, |F|^2 (PSD written shifted,
//                with its Hermitian mirror), inverse FFT along y of the power spectrum,
//                written back in place.  Fusing forward and inverse column passes removes
//                one full read+write of the spectrum (SURVEY.md §8d counts 4 passes).
//   K3 row_c2r   rebuilds the two-row packing from the half spectra, one inverse FFT of
//                length nx, shift + normalise -> two autocorrelation rows.
//
// All three are HBM-bandwidth bound; see DESIGN.md for the byte accounting.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/b4d.h"
#include "b4d_common.hpp"
#include "b4d_fft.hpp"

namespace b4d {

constexpr int E16 = 16;

// radix plans: N = 16 * R2 * R3 with 16 points per lane
constexpr int radix2(int n) { return n / 16 < 16 ? n / 16 : 16; }
constexpr int radix3(int n) { return n / (16 * radix2(n)); }
template <int N>
using RowGeom = FftGeom<N, E16, 16, radix2(N), radix3(N), 1>;
template <int N, int CP>
using ColGeom = FftGeom<N, E16, 16, radix2(N), radix3(N), CP>;
// row transforms (two image rows each) per workgroup: 256 lanes, fewer where LDS (64 KiB static) binds
constexpr int row_seq(int nx) { return nx == 4096 ? 1 : (nx == 64 ? 32 : 256 / (nx / 16)); }

// spectrum element index in the tile-major layout
__device__ __forceinline__ size_t spec_index(size_t frame, int nt, int ny, int ct_w, int y, int kx) {
    return ((frame * nt + (kx / ct_w)) * (size_t)ny + y) * ct_w + (kx % ct_w);
}

// ------------------------------------------------------------------------------------ K1
// Optional per-item source descriptor (tracking): item b reads frame `frame` of the input, only
// inside the ROI [y0,y1) x [x0,x1) (zero elsewhere), z-scored as (x - mean) / denom
// (signal/tracking.py:308-311 + geometry/roi.py:175-222 embed_roi with fill 0).
struct RowSrc {
    int frame, y0, y1, x0, x1;
    float mean, denom;
    int pad;
};

// grid (ny/2/SEQ, batch); block T*SEQ.  ct_w = tile width (complex columns) of the spectrum layout.
template <int NX, int SEQ, bool SRC>
__global__ void __launch_bounds__((NX / E16) * SEQ)
k_row_r2c(const float* __restrict__ in, float2* __restrict__ spec, const float2* __restrict__ tw, int ny, int ct_w,
          const RowSrc* __restrict__ srcs) {
    using G = RowGeom<NX>;
    constexpr int T = G::T, E = E16;
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const int pair = blockIdx.x * SEQ + seq;
    const bool live = 2 * pair < ny;  // ragged last workgroup: idle transforms still take part in the barriers
    const size_t frame = blockIdx.y;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    float2 v[E];
    if (!live) {
#pragma unroll
        for (int j = 0; j < E; ++j) v[j] = make_float2(0.f, 0.f);
    } else if (SRC) {
        const RowSrc sd = srcs[frame];
        const int ya = 2 * pair, yb = ya + 1;
        const bool ina = ya >= sd.y0 && ya < sd.y1, inb = yb >= sd.y0 && yb < sd.y1;
        const float* r0 = in + ((size_t)sd.frame * ny + ya) * NX;
        const float* r1 = r0 + NX;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int x = u + T * j;
            const bool inx = x >= sd.x0 && x < sd.x1;
            v[j].x = (ina && inx) ? (r0[x] - sd.mean) / sd.denom : 0.f;
            v[j].y = (inb && inx) ? (r1[x] - sd.mean) / sd.denom : 0.f;
        }
    } else {
        const float* r0 = in + (frame * ny + 2 * (size_t)pair) * NX;
        const float* r1 = r0 + NX;
#pragma unroll
        for (int j = 0; j < E; ++j) v[j] = make_float2(r0[u + T * j], r1[u + T * j]);
    }
    Fft3<G, 1>::run(v, v, u, 0, lds, tw);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; ++j) lds[u + T * j] = v[j];
    __syncthreads();
    const int nt = (NX / 2) / ct_w;
    if (!live) return;
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        const int k = u + T * j;
        const float2 z = v[j], zr = lds[(NX - k) & (NX - 1)];
        float2 a = make_float2(0.5f * (z.x + zr.x), 0.5f * (z.y - zr.y));
        float2 b = make_float2(0.5f * (z.y + zr.y), 0.5f * (zr.x - z.x));
        if (k == 0) {  // DC and Nyquist are both real: pack them
            const float2 zn = lds[NX / 2];
            a = make_float2(z.x, zn.x);
            b = make_float2(z.y, zn.y);
        }
        const size_t o = spec_index(frame, nt, ny, ct_w, 2 * pair, k);
        spec[o] = a;
        spec[o + ct_w] = b;  // next row of the same tile
    }
}

// ------------------------------------------------------------------------------------ K2
enum ColMode { COL_PSD_AC = 0, COL_SPECTRUM = 1, COL_FORWARD = 2 };

struct ColArgs {
    float2* spec;     // tile-major half spectra, in/out
    float* psd;       // (batch, ny, nx) or null
    float2* full;     // (batch, ny, nx) complex, COL_SPECTRUM only
    const float2* tw;
    float psd_scale;
    int nx, nt;       // nt = number of column tiles = (nx/2)/CT
    unsigned flags;
};

// block CP*NY/16; CT = 2*CP columns per tile.  Tile 0 holds the packed DC/Nyquist column and is
// handled by its own instantiation (TILE0, grid (1, batch)); the others run with grid (nt-1, batch).
template <int NY, int CP, int MODE, bool TILE0>
__global__ void __launch_bounds__(CP * (NY / E16)) k_col(ColArgs p) {
    using G = ColGeom<NY, CP>;
    constexpr int T = G::T, E = E16, CT = 2 * CP;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int cp = threadIdx.x % CP, u = threadIdx.x / CP;
    const int ct = TILE0 ? 0 : blockIdx.x + 1, nt = p.nt;
    const size_t frame = blockIdx.y;
    const int nx = p.nx;
    float2* tile = p.spec + ((frame * nt + ct) * (size_t)NY) * CT;
    const unsigned toff = (unsigned)u * CT + 2 * cp;  // element offset of (row u, column pair cp) in the tile
    float2 va[E], vb[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(tile + (size_t)(T * j * CT) + toff);
        va[j] = make_float2(q.x, q.y);
        vb[j] = make_float2(q.z, q.w);
    }
    Fft3<G, 2>::run(va, vb, u, cp, lds, p.tw);
    // va[j] = F[ky = u + T j][kx0], vb[j] = F[ky][kx0 + 1]
    const int kx0 = ct * CT + 2 * cp;
    const bool packed = TILE0 && cp == 0;  // column 0 carries the DC (re) and Nyquist (im) rows' transforms
    if (TILE0) {                           // publish column 0 in natural order for the Hermitian split
        __syncthreads();
        if (cp == 0) {
#pragma unroll
            for (int j = 0; j < E; ++j) lds[u + T * j] = va[j];
        }
        __syncthreads();
    }
    // F[ky][0] and F[ky][nx/2] from Z[ky], Z[-ky] of the packed column
    auto split = [&](int j, float2& f0, float2& fn) {
        const int ky = u + T * j;
        const float2 z = va[j], zr = lds[(NY - ky) & (NY - 1)];
        f0 = make_float2(0.5f * (z.x + zr.x), 0.5f * (z.y - zr.y));
        fn = make_float2(0.5f * (z.y + zr.y), 0.5f * (zr.x - z.x));
    };

    if (MODE == COL_FORWARD) {
        // keep the 2-D half spectrum in the tile; the Nyquist column goes to the side buffer p.full (batch, NY)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float2 f0 = va[j], fn;
            if (packed) {
                split(j, f0, fn);
                p.full[frame * NY + u + T * j] = fn;
            }
            *reinterpret_cast<float4*>(tile + (size_t)(T * j * CT) + toff) = make_float4(f0.x, f0.y, vb[j].x, vb[j].y);
        }
        return;
    }

    if (MODE == COL_SPECTRUM) {
        float2* out = p.full + frame * (size_t)NY * nx;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int ky = u + T * j;
            const unsigned rd = (unsigned)((ky + NY / 2) & (NY - 1)) * nx, rm = (unsigned)((NY / 2 - ky) & (NY - 1)) * nx;
            float2 f0 = va[j], fn;
            if (packed) {
                split(j, f0, fn);
                out[rd] = fn;
            }
            *reinterpret_cast<float4*>(&out[rd + nx / 2 + kx0]) = make_float4(f0.x, f0.y, vb[j].x, vb[j].y);
            if (kx0 >= 1) out[rm + nx / 2 - kx0] = make_float2(f0.x, -f0.y);
            out[rm + nx / 2 - kx0 - 1] = make_float2(vb[j].x, -vb[j].y);
        }
        return;
    }

    // ---- COL_PSD_AC: power spectrum, optional PSD store, inverse transform along y
    const float s = p.psd_scale;
    float* psd = p.psd ? p.psd + frame * (size_t)NY * nx : nullptr;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int ky = u + T * j;
        float2 f0 = va[j], fn = make_float2(0.f, 0.f);
        if (packed) split(j, f0, fn);
        const float pa = f0.x * f0.x + f0.y * f0.y;
        const float pb = vb[j].x * vb[j].x + vb[j].y * vb[j].y;
        const float pn = fn.x * fn.x + fn.y * fn.y;
        if (psd) {
            const unsigned rd = (unsigned)((ky + NY / 2) & (NY - 1)) * nx, rm = (unsigned)((NY / 2 - ky) & (NY - 1)) * nx;
            *reinterpret_cast<float2*>(&psd[rd + nx / 2 + kx0]) = make_float2(pa * s, pb * s);
            if (kx0 >= 1) psd[rm + nx / 2 - kx0] = pa * s;
            psd[rm + nx / 2 - kx0 - 1] = pb * s;
            if (packed) psd[rd] = pn * s;
        }
        // inverse input, already (im, re)-swapped.
        //  tile 0: column a -> (Pnyq, Pdc) on the packed lanes else (0, Pa); column b -> (0, Pb)
        //  others: the two REAL power columns ride one complex transform: Pa + i Pb -> (Pb, Pa)
        if (TILE0) {
            va[j] = make_float2(pn, pa);
            vb[j] = make_float2(0.f, pb);
        } else {
            va[j] = make_float2(pb, pa);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // Launder the twiddle pointer: otherwise the compiler keeps the first transform's 30 twiddles
    // alive across the whole kernel (they are provably the same loads) and spills.
    const float2* tw2 = p.tw;
    asm volatile("" : "+s"(tw2));
    unsigned toff2 = toff;  // same for the store addresses (16 x 64-bit pairs would stay live from the loads)
    asm volatile("" : "+v"(toff2));
    __syncthreads();
    if (TILE0) {
        if (packed && u == 0 && (p.flags & B4D_REMOVE_MEAN)) va[0].y = 0.f;  // DC bin: ky = 0 <-> u = 0, j = 0
        Fft3<G, 2>::run(va, vb, u, cp, lds, tw2);
#pragma unroll
        for (int j = 0; j < E; ++j)
            *reinterpret_cast<float4*>(tile + (size_t)(T * j * CT) + toff2) =
                make_float4(va[j].y, va[j].x, vb[j].y, vb[j].x);
    } else {
        Fft3<G, 1>::run(va, va, u, cp, lds, tw2);
        // V[y] = Ga[y] + i Gb[y] with Ga, Gb Hermitian in y: split with V[-y]
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) lds[(u + T * j) * CP + cp] = make_float2(va[j].y, va[j].x);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const float2 z = make_float2(va[j].y, va[j].x), zr = lds[((NY - (u + T * j)) & (NY - 1)) * CP + cp];
            *reinterpret_cast<float4*>(tile + (size_t)(T * j * CT) + toff2) =
                make_float4(0.5f * (z.x + zr.x), 0.5f * (z.y - zr.y), 0.5f * (z.y + zr.y), 0.5f * (zr.x - z.x));
        }
    }
}

// ------------------------------------------------------------------------------------ K3
struct RowOutArgs {
    const float2* g;   // tile-major inverse-column output
    float* out;        // (batch, ny, nx) float32 shifted
    float* peak;       // (batch) zero-lag values
    const float2* tw;
    float scale;       // used when !NORM_PEAK
    int ny, ct_w;
    unsigned flags;
    float* part_val;   // C2R_MAG: per-workgroup arg-max partials (batch, gridDim.x)
    int* part_idx;
};

enum RowOutMode { C2R_OUT = 0, C2R_PEAK = 1, C2R_MAG = 2 };

// (value, flat index) arg-max with NumPy's first-occurrence rule: larger value wins, ties go to the lower index
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// grid (ny/2/SEQ, batch) -- or (1, batch) for C2R_PEAK; block T*SEQ.
//   C2R_OUT   shifted real output, scaled (flags & NORM_PEAK: by 1/peak[frame], zero lag forced to 1)
//   C2R_PEAK  only the zero-lag value of each frame -> peak[frame]
//   C2R_MAG   |value| * scale (signal/tracking.py:283-285) + per-workgroup arg-max partials
template <int NX, int SEQ, int MODE>
__global__ void __launch_bounds__((NX / E16) * SEQ) k_row_c2r(RowOutArgs p) {
    using G = RowGeom<NX>;
    constexpr int T = G::T, E = E16;
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const int pair = MODE == C2R_PEAK ? 0 : blockIdx.x * SEQ + seq;
    const size_t frame = blockIdx.y;
    const int ny = p.ny, ct_w = p.ct_w, nt = (NX / 2) / ct_w;
    const bool live = 2 * pair < ny;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    float2 v[E];
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        const int k = u + T * j;
        const size_t o = spec_index(frame, nt, ny, ct_w, live ? 2 * pair : 0, k);
        const float2 a = p.g[o], b = p.g[o + ct_w];
        if (k == 0) {  // packed: a = (A_dc, A_nyq), b = (B_dc, B_nyq), all real
            v[j] = make_float2(b.x, a.x);            // swap(A_dc + i B_dc)
            lds[NX / 2] = make_float2(b.y, a.y);     // swap(A_nyq + i B_nyq)
        } else {
            v[j] = make_float2(a.y + b.x, a.x - b.y);        // swap(A + iB)
            lds[NX - k] = make_float2(b.x - a.y, a.x + b.y);  // swap(conj A + i conj B)
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = E / 2; j < E; ++j) v[j] = lds[u + T * j];
    __syncthreads();
    Fft3<G, 1>::run(v, v, u, 0, lds, p.tw);
    // v[j] = swap(z[x]), x = u + T j: row 2*pair = Re z = v.y, row 2*pair+1 = Im z = v.x
    if (MODE == C2R_PEAK) {
        if (threadIdx.x == 0) p.peak[frame] = v[0].y;
        return;
    }
    const int y0 = 2 * pair;
    const int ra = (y0 + ny / 2) & (ny - 1), rb = (y0 + 1 + ny / 2) & (ny - 1);
    float* o0 = p.out + (frame * ny + ra) * (size_t)NX;
    float* o1 = p.out + (frame * ny + rb) * (size_t)NX;
    if (MODE == C2R_OUT) {
        if (!live) return;
        const bool norm = (p.flags & B4D_NORM_PEAK) != 0;
        float s = p.scale;
        if (norm) {
            const float pk = p.peak[frame];
            s = pk != 0.f ? 1.0f / pk : 1.0f;
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int x = u + T * j, c = (x + NX / 2) & (NX - 1);
            float r0 = v[j].y * s;
            if (norm && pair == 0 && x == 0) r0 = 1.0f;  // peak normalisation: zero lag is 1 by definition
            o0[c] = r0;
            o1[c] = v[j].x * s;
        }
        return;
    }
    // ---- C2R_MAG
    float bv = -1.0f;
    int bi = 0x7fffffff;
    if (live) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int x = u + T * j, c = (x + NX / 2) & (NX - 1);
            const float m0 = fabsf(v[j].y * p.scale), m1 = fabsf(v[j].x * p.scale);
            o0[c] = m0;
            o1[c] = m1;
            argmax_merge(bv, bi, m0, ra * NX + c);
            argmax_merge(bv, bi, m1, rb * NX + c);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        argmax_merge(bv, bi, ov, oi);
    }
    __syncthreads();  // everyone is done with lds_all (FFT exchange reads)
    float* sv = reinterpret_cast<float*>(lds_all);
    int* si = reinterpret_cast<int*>(lds_all) + 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (T * SEQ + 63) / 64;
    if (lane == 0) {
        sv[w] = bv;
        si[w] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) argmax_merge(bv, bi, sv[i], si[i]);
        p.part_val[frame * gridDim.x + blockIdx.x] = bv;
        p.part_idx[frame * gridDim.x + blockIdx.x] = bi;
    }
}

}  // namespace b4d

// ===================================================================================== host
using namespace b4d;

struct b4d_plan {
    int ny, nx, chunk, ct_w, cp;
    float2* tw_x = nullptr;   // nx-point twiddles
    float2* tw_y = nullptr;   // ny-point twiddles
    float2* spec = nullptr;   // chunk * ny * nx/2
    float* peak = nullptr;    // chunk
    size_t ws_bytes = 0;
    void* track_ws = nullptr; // lazily grown arena of the xcorr / tracking entry points
    size_t track_bytes = 0;
};

static inline bool pow2_ok(int n) { return n >= 64 && n <= 4096 && (n & (n - 1)) == 0; }

static inline int make_twiddles(int n, float2** out) {
    std::vector<float2> h(n);
    for (int k = 0; k < n; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)n;
        h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    B4D_HIP(hipMalloc((void**)out, sizeof(float2) * n));
    B4D_HIP(hipMemcpy(*out, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    return B4D_OK;
}

template <int NY, int CP, int MODE, bool TILE0>
static int launch_col1(const ColArgs& a, int gx, int batch, hipStream_t st) {
    using G = ColGeom<NY, CP>;
    const size_t lds = sizeof(float2) * (size_t)G::LDS_ELEMS * CP;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_col<NY, CP, MODE, TILE0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    B4D_HIP(attr_err);
    if (gx < 1) return B4D_OK;
    hipLaunchKernelGGL((k_col<NY, CP, MODE, TILE0>), dim3(gx, batch), dim3(CP * (NY / E16)), lds, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
template <int NY, int CP, int MODE>
static int launch_col(const ColArgs& a, int ntiles, int batch, hipStream_t st) {
    int rc = launch_col1<NY, CP, MODE, false>(a, ntiles - 1, batch, st);
    if (rc) return rc;
    return launch_col1<NY, CP, MODE, true>(a, 1, batch, st);
}

template <int MODE>
static int dispatch_col(const b4d_plan* pl, const ColArgs& a_in, int batch, hipStream_t st) {
    const int ntiles = (pl->nx / 2) / pl->ct_w;
    ColArgs a = a_in;
    a.nt = ntiles;
    switch (pl->ny) {
        case 64: return launch_col<64, 8, MODE>(a, ntiles, batch, st);
        case 128: return launch_col<128, 8, MODE>(a, ntiles, batch, st);
        case 256: return launch_col<256, 8, MODE>(a, ntiles, batch, st);
        case 512: return launch_col<512, 8, MODE>(a, ntiles, batch, st);
        case 1024: return launch_col<1024, 8, MODE>(a, ntiles, batch, st);
        case 2048: return launch_col<2048, 8, MODE>(a, ntiles, batch, st);
        case 4096: return launch_col<4096, 4, MODE>(a, ntiles, batch, st);
    }
    return fail(B4D_ESIZE, "unsupported ny");
}

template <int NX>
static int launch_r2c(const b4d_plan* pl, const float* in, float2* spec, const RowSrc* srcs, int batch, hipStream_t st) {
    constexpr int SEQ = row_seq(NX);
    const dim3 grid((pl->ny / 2 + SEQ - 1) / SEQ, batch), block((NX / E16) * SEQ);
    if (srcs)
        hipLaunchKernelGGL((k_row_r2c<NX, SEQ, true>), grid, block, 0, st, in, spec, pl->tw_x, pl->ny, pl->ct_w, srcs);
    else
        hipLaunchKernelGGL((k_row_r2c<NX, SEQ, false>), grid, block, 0, st, in, spec, pl->tw_x, pl->ny, pl->ct_w, srcs);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
// rows -> half spectra into `spec` (default: the plan's chunk workspace); srcs != null selects ROI / z-score sources
static int dispatch_r2c(const b4d_plan* pl, const float* in, int batch, hipStream_t st, float2* spec = nullptr,
                        const RowSrc* srcs = nullptr) {
    if (!spec) spec = pl->spec;
    switch (pl->nx) {
        case 64: return launch_r2c<64>(pl, in, spec, srcs, batch, st);
        case 128: return launch_r2c<128>(pl, in, spec, srcs, batch, st);
        case 256: return launch_r2c<256>(pl, in, spec, srcs, batch, st);
        case 512: return launch_r2c<512>(pl, in, spec, srcs, batch, st);
        case 1024: return launch_r2c<1024>(pl, in, spec, srcs, batch, st);
        case 2048: return launch_r2c<2048>(pl, in, spec, srcs, batch, st);
        case 4096: return launch_r2c<4096>(pl, in, spec, srcs, batch, st);
    }
    return fail(B4D_ESIZE, "unsupported nx");
}

template <int NX>
static int launch_c2r(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st, std::vector<hipEvent_t>* ev) {
    constexpr int SEQ = row_seq(NX);
    if (a.flags & B4D_NORM_PEAK) {
        hipLaunchKernelGGL((k_row_c2r<NX, SEQ, C2R_PEAK>), dim3(1, batch), dim3((NX / E16) * SEQ), 0, st, a);
        B4D_HIP(hipGetLastError());
        if (ev) {
            hipEvent_t e;
            B4D_HIP(hipEventCreate(&e));
            ev->push_back(e);
            B4D_HIP(hipEventRecord(e, st));
        }
    }
    hipLaunchKernelGGL((k_row_c2r<NX, SEQ, C2R_OUT>), dim3((pl->ny / 2 + SEQ - 1) / SEQ, batch), dim3((NX / E16) * SEQ), 0, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
static int dispatch_c2r(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st,
                        std::vector<hipEvent_t>* ev = nullptr) {
    switch (pl->nx) {
        case 64: return launch_c2r<64>(pl, a, batch, st, ev);
        case 128: return launch_c2r<128>(pl, a, batch, st, ev);
        case 256: return launch_c2r<256>(pl, a, batch, st, ev);
        case 512: return launch_c2r<512>(pl, a, batch, st, ev);
        case 1024: return launch_c2r<1024>(pl, a, batch, st, ev);
        case 2048: return launch_c2r<2048>(pl, a, batch, st, ev);
        case 4096: return launch_c2r<4096>(pl, a, batch, st, ev);
    }
    return fail(B4D_ESIZE, "unsupported nx");
}


// b4d_kernels.hip -- hand-written gfx950 kernels + C ABI (include/b4d.h) for the barc4dip
// FFT -> PSD -> autocorrelation hot path.
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
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/b4d.h"
#include "b4d_fft.hpp"

namespace b4d {

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define B4D_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return fail(B4D_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));     \
    } while (0)

constexpr int E16 = 16;

// spectrum element index in the tile-major layout
__device__ __forceinline__ size_t spec_index(size_t frame, int nt, int ny, int ct_w, int y, int kx) {
    return ((frame * nt + (kx / ct_w)) * (size_t)ny + y) * ct_w + (kx % ct_w);
}

// ------------------------------------------------------------------------------------ K1
// grid (ny/2/SEQ, batch); block T*SEQ.  ct_w = tile width (complex columns) of the spectrum layout.
template <int NX, int SEQ>
__global__ void __launch_bounds__((NX / E16) * SEQ)
k_row_r2c(const float* __restrict__ in, float2* __restrict__ spec, const float2* __restrict__ tw, int ny, int ct_w) {
    using G = FftGeom<NX, E16, 16, 16, NX / 256, 1>;
    constexpr int T = G::T, E = E16;
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const int pair = blockIdx.x * SEQ + seq;
    const size_t frame = blockIdx.y;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    const float* r0 = in + (frame * ny + 2 * (size_t)pair) * NX;
    const float* r1 = r0 + NX;
    float2 v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = make_float2(r0[u + T * j], r1[u + T * j]);
    Fft3<G, 1>::run(v, v, u, 0, lds, tw);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < E; ++j) lds[u + T * j] = v[j];
    __syncthreads();
    const int nt = (NX / 2) / ct_w;
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
    using G = FftGeom<NY, E16, 16, 16, NY / 256, CP>;
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
};

// grid (ny/2/SEQ, batch) -- or (1, batch) when PEAK_ONLY; block T*SEQ.
template <int NX, int SEQ, bool PEAK_ONLY>
__global__ void __launch_bounds__((NX / E16) * SEQ) k_row_c2r(RowOutArgs p) {
    using G = FftGeom<NX, E16, 16, 16, NX / 256, 1>;
    constexpr int T = G::T, E = E16;
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const int pair = PEAK_ONLY ? 0 : blockIdx.x * SEQ + seq;
    const size_t frame = blockIdx.y;
    const int ny = p.ny, ct_w = p.ct_w, nt = (NX / 2) / ct_w;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    float2 v[E];
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        const int k = u + T * j;
        const size_t o = spec_index(frame, nt, ny, ct_w, 2 * pair, k);
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
    if (PEAK_ONLY) {
        if (threadIdx.x == 0) p.peak[frame] = v[0].y;
        return;
    }
    const bool norm = (p.flags & B4D_NORM_PEAK) != 0;
    float s = p.scale;
    if (norm) {
        const float pk = p.peak[frame];
        s = pk != 0.f ? 1.0f / pk : 1.0f;
    }
    const int y0 = 2 * pair;
    float* o0 = p.out + (frame * ny + ((y0 + ny / 2) & (ny - 1))) * (size_t)NX;
    float* o1 = p.out + (frame * ny + ((y0 + 1 + ny / 2) & (ny - 1))) * (size_t)NX;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int x = u + T * j, c = (x + NX / 2) & (NX - 1);
        float r0 = v[j].y * s;
        if (norm && pair == 0 && x == 0) r0 = 1.0f;  // peak normalisation: zero lag is 1 by definition
        o0[c] = r0;
        o1[c] = v[j].x * s;
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
};

static bool pow2_ok(int n) { return n == 512 || n == 1024 || n == 2048 || n == 4096; }

static int make_twiddles(int n, float2** out) {
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
    using G = FftGeom<NY, E16, 16, 16, NY / 256, CP>;
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
        case 512: return launch_col<512, 8, MODE>(a, ntiles, batch, st);
        case 1024: return launch_col<1024, 8, MODE>(a, ntiles, batch, st);
        case 2048: return launch_col<2048, 8, MODE>(a, ntiles, batch, st);
        case 4096: return launch_col<4096, 4, MODE>(a, ntiles, batch, st);
    }
    return fail(B4D_ESIZE, "unsupported ny");
}

template <int NX>
static int launch_r2c(const b4d_plan* pl, const float* in, int batch, hipStream_t st) {
    constexpr int SEQ = (NX == 4096) ? 1 : (NX == 2048 ? 2 : (NX == 1024 ? 4 : 8));
    hipLaunchKernelGGL((k_row_r2c<NX, SEQ>), dim3(pl->ny / 2 / SEQ, batch), dim3((NX / E16) * SEQ), 0, st, in,
                       pl->spec, pl->tw_x, pl->ny, pl->ct_w);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
static int dispatch_r2c(const b4d_plan* pl, const float* in, int batch, hipStream_t st) {
    switch (pl->nx) {
        case 512: return launch_r2c<512>(pl, in, batch, st);
        case 1024: return launch_r2c<1024>(pl, in, batch, st);
        case 2048: return launch_r2c<2048>(pl, in, batch, st);
        case 4096: return launch_r2c<4096>(pl, in, batch, st);
    }
    return fail(B4D_ESIZE, "unsupported nx");
}

template <int NX>
static int launch_c2r(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st) {
    constexpr int SEQ = (NX == 4096) ? 1 : (NX == 2048 ? 2 : (NX == 1024 ? 4 : 8));
    if (a.flags & B4D_NORM_PEAK) {
        hipLaunchKernelGGL((k_row_c2r<NX, SEQ, true>), dim3(1, batch), dim3((NX / E16) * SEQ), 0, st, a);
        B4D_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((k_row_c2r<NX, SEQ, false>), dim3(pl->ny / 2 / SEQ, batch), dim3((NX / E16) * SEQ), 0, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
static int dispatch_c2r(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st) {
    switch (pl->nx) {
        case 512: return launch_c2r<512>(pl, a, batch, st);
        case 1024: return launch_c2r<1024>(pl, a, batch, st);
        case 2048: return launch_c2r<2048>(pl, a, batch, st);
        case 4096: return launch_c2r<4096>(pl, a, batch, st);
    }
    return fail(B4D_ESIZE, "unsupported nx");
}

extern "C" {

const char* b4d_version(void) { return "b4d 0.1.0 (gfx950)"; }
const char* b4d_last_error(void) { return g_err.c_str(); }
int b4d_size_supported(int ny, int nx) { return pow2_ok(ny) && pow2_ok(nx); }

int b4d_plan_create(int ny, int nx, int chunk, b4d_plan** out) {
    if (!out) return fail(B4D_EINVAL, "out is null");
    *out = nullptr;
    if (!b4d_size_supported(ny, nx))
        return fail(B4D_ESIZE, "native plans need ny, nx in {512, 1024, 2048, 4096}; got " + std::to_string(ny) + "x" +
                                   std::to_string(nx));
    if (chunk < 1) return fail(B4D_EINVAL, "chunk must be >= 1");
    b4d_plan* p = new b4d_plan();
    p->ny = ny;
    p->nx = nx;
    p->chunk = chunk;
    p->cp = (ny == 4096) ? 4 : 8;
    p->ct_w = 2 * p->cp;
    int rc = make_twiddles(nx, &p->tw_x);
    if (rc == B4D_OK) rc = make_twiddles(ny, &p->tw_y);
    if (rc != B4D_OK) {
        b4d_plan_destroy(p);
        return rc;
    }
    p->ws_bytes = sizeof(float2) * (size_t)chunk * ny * (nx / 2);
    hipError_t e = hipMalloc((void**)&p->spec, p->ws_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&p->peak, sizeof(float) * chunk);
    if (e != hipSuccess) {
        b4d_plan_destroy(p);
        return fail(B4D_ENOMEM, std::string("workspace allocation failed: ") + hipGetErrorString(e));
    }
    *out = p;
    return B4D_OK;
}

int b4d_plan_destroy(b4d_plan* p) {
    if (!p) return B4D_OK;
    if (p->tw_x) (void)hipFree(p->tw_x);
    if (p->tw_y) (void)hipFree(p->tw_y);
    if (p->spec) (void)hipFree(p->spec);
    if (p->peak) (void)hipFree(p->peak);
    delete p;
    return B4D_OK;
}

size_t b4d_plan_workspace_bytes(const b4d_plan* p) { return p ? p->ws_bytes : 0; }

int b4d_psd_autocorr2d(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale, float* autocorr,
                       unsigned flags, void* stream) {
    if (!pl || !frames) return fail(B4D_EINVAL, "null plan or input");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    if (!psd && !autocorr) return fail(B4D_EINVAL, "both outputs are null");
    hipStream_t st = (hipStream_t)stream;
    const size_t fpix = (size_t)pl->ny * pl->nx;
    for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
        const int nb = std::min(pl->chunk, batch - b0);
        int rc = dispatch_r2c(pl, frames + b0 * fpix, nb, st);
        if (rc) return rc;
        ColArgs ca{};
        ca.spec = pl->spec;
        ca.psd = psd ? psd + b0 * fpix : nullptr;
        ca.tw = pl->tw_y;
        ca.psd_scale = psd_scale;
        ca.nx = pl->nx;
        ca.flags = flags;
        rc = dispatch_col<COL_PSD_AC>(pl, ca, nb, st);
        if (rc) return rc;
        if (autocorr) {
            RowOutArgs ra{};
            ra.g = pl->spec;
            ra.out = autocorr + b0 * fpix;
            ra.peak = pl->peak;
            ra.tw = pl->tw_x;
            ra.scale = 1.0f / ((float)pl->nx * (float)pl->ny);
            ra.ny = pl->ny;
            ra.ct_w = pl->ct_w;
            ra.flags = flags;
            rc = dispatch_c2r(pl, ra, nb, st);
            if (rc) return rc;
        }
    }
    return B4D_OK;
}

int b4d_psd2d(b4d_plan* pl, const float* frames, int batch, float* psd, float scale, void* stream) {
    if (!psd) return fail(B4D_EINVAL, "psd is null");
    return b4d_psd_autocorr2d(pl, frames, batch, psd, scale, nullptr, 0u, stream);
}

int b4d_autocorr2d(b4d_plan* pl, const float* frames, int batch, float* autocorr, unsigned flags, void* stream) {
    if (!autocorr) return fail(B4D_EINVAL, "autocorr is null");
    return b4d_psd_autocorr2d(pl, frames, batch, nullptr, 1.0f, autocorr, flags, stream);
}

int b4d_fft2d(b4d_plan* pl, const float* frames, int batch, float* out_c64, void* stream) {
    if (!pl || !frames || !out_c64) return fail(B4D_EINVAL, "null argument");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const size_t fpix = (size_t)pl->ny * pl->nx;
    for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
        const int nb = std::min(pl->chunk, batch - b0);
        int rc = dispatch_r2c(pl, frames + b0 * fpix, nb, st);
        if (rc) return rc;
        ColArgs ca{};
        ca.spec = pl->spec;
        ca.full = reinterpret_cast<float2*>(out_c64) + b0 * fpix;
        ca.tw = pl->tw_y;
        ca.nx = pl->nx;
        rc = dispatch_col<COL_SPECTRUM>(pl, ca, nb, st);
        if (rc) return rc;
    }
    return B4D_OK;
}

}  // extern "C"

// b4d_general.hip -- general-length 2-D transforms: any 2 <= ny, nx <= 512 as dense DFT-matrix products; larger frames
// whose sides split as P * A * B through the fused in-LDS mixed-radix row transform of b4d_wiener.hip (plan->large).
//
// The aggregators of barc4dip evaluate every metric on 3x3 tiles or 9x9 sub-tiles (metrics/common.py:75-106,
// 278-378): 170/171-pixel tiles at 512^2, 227/228 at 2048^2 -- sizes with large prime factors (19, 227).  For these
// small, awkward lengths the 2-D DFT is evaluated as F = Wy * X * Wx with precomputed DFT matrices: a genuinely
// dense contraction (the one place on this path where that appears), regular, any length, float32-exact FMA
// chains.  O(n^3) per transform keeps it to n <= 512; power-of-two frames never come here (b4d_fft2d.hpp).
// The product kernel is a plain LDS-tiled complex GEMM on the vector ALUs (64x64 tile, 4x4 register block);
// moving it to v_mfma_f32_32x32x2_f32 is listed in DESIGN.md §8.
#include "b4d_fft2d.hpp"
#include "b4d_wiener_mr.hpp"

namespace b4d {

struct GemmArgs {
    const void* A;   // (M, K) row-major, float or float2
    const void* B;   // (K, N) row-major, float or float2
    float2* C;       // (M, N) row-major
    int M, N, K;
    long long sA, sB, sC;  // batch strides in elements (0 = shared operand)
    int conj_a, conj_b;
};

template <bool REAL>
__device__ __forceinline__ float2 ld_elem(const void* p, long long i, int conj) {
    if (REAL) return make_float2(static_cast<const float*>(p)[i], 0.f);
    const float2 v = static_cast<const float2*>(p)[i];
    return conj ? make_float2(v.x, -v.y) : v;
}

// grid (ceil(N/64), ceil(M/64), batch), block 256 = 16 x 16 lanes, 4 x 4 complex outputs per lane
template <bool AREAL, bool BREAL>
__global__ void __launch_bounds__(256) k_cgemm(GemmArgs g) {
    __shared__ float2 As[16][65];  // [k][m]
    __shared__ float2 Bs[16][65];  // [k][n]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const long long oa = (long long)blockIdx.z * g.sA, ob = (long long)blockIdx.z * g.sB, oc = (long long)blockIdx.z * g.sC;
    float2 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = make_float2(0.f, 0.f);
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = threadIdx.x + 256 * r;  // 0..1023
            {   // A tile: 64 rows x 16 k, k fastest in memory
                const int kk = e & 15, mm = e >> 4;
                const int m = m0 + mm, k = k0 + kk;
                As[kk][mm] = (m < g.M && k < g.K) ? ld_elem<AREAL>(g.A, oa + (long long)m * g.K + k, g.conj_a) : make_float2(0.f, 0.f);
            }
            {   // B tile: 16 k x 64 cols, n fastest in memory
                const int nn = e & 63, kk = e >> 6;
                const int n = n0 + nn, k = k0 + kk;
                Bs[kk][nn] = (n < g.N && k < g.K) ? ld_elem<BREAL>(g.B, ob + (long long)k * g.N + n, g.conj_b) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float2 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j].x = fmaf(a[i].x, b[j].x, fmaf(-a[i].y, b[j].y, acc[i][j].x));
                    acc[i][j].y = fmaf(a[i].x, b[j].y, fmaf(a[i].y, b[j].x, acc[i][j].y));
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < g.N) g.C[oc + (long long)m * g.N + n] = acc[i][j];
        }
    }
}

// The same product on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-for-bit an fmaf chain).
// 64 x 64 complex tile per workgroup, 4 waves in 2 x 2, each wave a 32 x 32 complex tile = two f32x16 accumulators
// (re, im); per k-step of 2 one ds_read_b64 per operand feeds four MFMAs:
//   Cr += Ar*Br - Ai*Bi,  Ci += Ar*Bi + Ai*Br.
// Fragment maps (MI355X guide §3): A: lane l holds A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31];
// C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool AREAL, bool BREAL>
__global__ void __launch_bounds__(256) k_cgemm_mfma(GemmArgs g) {
    __shared__ float2 As[16][65];  // [k][m]
    __shared__ float2 Bs[16][65];  // [k][n]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const long long oa = (long long)blockIdx.z * g.sA, ob = (long long)blockIdx.z * g.sB, oc = (long long)blockIdx.z * g.sC;
    f32x16 cr, ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) cr[r] = ci[r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = threadIdx.x + 256 * r;
            {
                const int kk = e & 15, mm = e >> 4;
                const int m = m0 + mm, k = k0 + kk;
                As[kk][mm] = (m < g.M && k < g.K) ? ld_elem<AREAL>(g.A, oa + (long long)m * g.K + k, g.conj_a) : make_float2(0.f, 0.f);
            }
            {
                const int nn = e & 63, kk = e >> 6;
                const int n = n0 + nn, k = k0 + kk;
                Bs[kk][nn] = (n < g.N && k < g.K) ? ld_elem<BREAL>(g.B, ob + (long long)k * g.N + n, g.conj_b) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float2 a = As[kk + lk][32 * wr + li];
            const float2 b = Bs[kk + lk][32 * wc + li];
            cr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, cr, 0, 0, 0);
            cr = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.y, b.y, cr, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, ci, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, ci, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + 32 * wc + li;
    if (n < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < g.M) g.C[oc + (long long)m * g.N + n] = make_float2(cr[r], ci[r]);
        }
    }
}

// ---- elementwise epilogues (one lane per pixel; grid (ceil(npix/256), batch))
__global__ void __launch_bounds__(256) k_gen_power(const float2* __restrict__ F, int ny, int nx, float* __restrict__ psd,
                                                   float scale, float* __restrict__ P, unsigned flags) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const size_t fo = (size_t)blockIdx.y * ny * nx;
    const int ky = e / nx, kx = e % nx;
    const float2 f = F[fo + e];
    const float p = f.x * f.x + f.y * f.y;
    if (psd) psd[fo + (size_t)((ky + ny / 2) % ny) * nx + (kx + nx / 2) % nx] = p * scale;
    if (P) P[fo + e] = (e == 0 && (flags & B4D_REMOVE_MEAN)) ? 0.f : p;
}

__global__ void __launch_bounds__(256) k_gen_shift_c(const float2* __restrict__ F, int ny, int nx, float2* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const size_t fo = (size_t)blockIdx.y * ny * nx;
    const int ky = e / nx, kx = e % nx;
    out[fo + (size_t)((ky + ny / 2) % ny) * nx + (kx + nx / 2) % nx] = F[fo + e];
}

// natural[k] = shifted[(k + n/2) % n] (np.fft.ifftshift of an fftshift-ed array), optionally scaled
__global__ void __launch_bounds__(256) k_gen_unshift_c(const float2* __restrict__ F, int ny, int nx, float scale, float2* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const size_t fo = (size_t)blockIdx.y * ny * nx;
    const int ky = e / nx, kx = e % nx;
    const float2 v = F[fo + (size_t)((ky + ny / 2) % ny) * nx + (kx + nx / 2) % nx];
    out[fo + e] = make_float2(v.x * scale, v.y * scale);
}

__global__ void __launch_bounds__(256) k_gen_scale_c(const float2* __restrict__ F, size_t n, float scale, float2* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    out[e] = make_float2(F[e].x * scale, F[e].y * scale);
}

__global__ void __launch_bounds__(256) k_gen_cross(const float2* __restrict__ Fa, const float2* __restrict__ Fb, int npix,
                                                   float2* __restrict__ C, unsigned flags) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npix) return;
    const size_t fo = (size_t)blockIdx.y * npix;
    float2 c = cross_power<false>(Fa[fo + e], Fb[fo + e], 0.f);
    if (e == 0 && (flags & B4D_REMOVE_MEAN)) c = make_float2(0.f, 0.f);
    C[fo + e] = c;
}

// real part of the inverse transform, shifted; NORM_PEAK: divided by the zero-lag value (exactly 1 there)
__global__ void __launch_bounds__(256) k_gen_real_out(const float2* __restrict__ R, int ny, int nx, float* __restrict__ out,
                                                      float scale, unsigned flags) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const size_t fo = (size_t)blockIdx.y * ny * nx;
    const int y = e / nx, x = e % nx;
    float s = scale;
    bool unit_peak = false;   // only a positive peak is normalised (signal/corr.py:247-250)
    if ((flags & B4D_NORM_PEAK) != 0) {
        const float pk = R[fo].x;
        unit_peak = pk > 0.f;
        if (unit_peak) s = 1.0f / pk;
    }
    float v = R[fo + e].x * s;
    if (unit_peak && e == 0) v = 1.0f;
    out[fo + (size_t)((y + ny / 2) % ny) * nx + (x + nx / 2) % nx] = v;
}

// ---- half-spectrum path of the fused mixed-radix plans (real rows ride in pairs, only nx/2 + 1 columns are transformed)
// Power spectrum from the TRANSPOSED half spectrum F_T (batch, Wh, ny): P_T (real, same layout; DC zeroed for mean
// removal) for the inverse transform, and the full fftshift-ed PSD (direct half + Hermitian mirror) through a 32 x 32 LDS
// transpose so that both sides are coalesced.  grid (ceil(ny/32), ceil(Wh/32), batch), block (32, 8)
__global__ void __launch_bounds__(256) k_half_power(const float2* __restrict__ FT, int ny, int nx, float* __restrict__ PT,
                                                    float* __restrict__ psd, float psd_scale, unsigned flags) {
    __shared__ float t[32][33];
    const int Wh = nx / 2 + 1;
    const int ky0 = blockIdx.x * 32, kx0 = blockIdx.y * 32;
    const size_t fh = (size_t)blockIdx.z * Wh * ny, ff = (size_t)blockIdx.z * ny * nx;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int kx = kx0 + i, ky = ky0 + threadIdx.x;
        float p = 0.f;
        if (kx < Wh && ky < ny) {
            const float2 f = FT[fh + (size_t)kx * ny + ky];
            p = f.x * f.x + f.y * f.y;
            if (PT) PT[fh + (size_t)kx * ny + ky] = (kx == 0 && ky == 0 && (flags & B4D_REMOVE_MEAN)) ? 0.f : p;
        }
        t[i][threadIdx.x] = p;
    }
    if (!psd) return;
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int ky = ky0 + i, kx = kx0 + threadIdx.x;
        if (ky >= ny || kx >= Wh) continue;
        const float v = t[threadIdx.x][i] * psd_scale;
        psd[ff + (size_t)((ky + ny / 2) % ny) * nx + (kx + nx / 2) % nx] = v;
        if (kx > 0 && 2 * kx != nx)   // Hermitian mirror (-ky, -kx); the Nyquist column of an even nx is its own mirror
            psd[ff + (size_t)(((ny - ky) % ny + ny / 2) % ny) * nx + ((nx - kx) + nx / 2) % nx] = v;
    }
}

// unscaled zero-lag value of every frame from the column-inverted half spectrum G_T (batch, Wh, ny):
// sum over the full kx range of G[kx][y = 0] = G[0] + 2 sum Re G[kx] (+ G[nx/2] for even nx).  grid (batch), block 256
__global__ void __launch_bounds__(256) k_half_peak(const float2* __restrict__ GT, int ny, int nx, float* __restrict__ peak) {
    __shared__ double sh[4];
    const int Wh = nx / 2 + 1;
    const float2* g = GT + (size_t)blockIdx.x * Wh * ny;
    double acc = 0.0;
    for (int kx = threadIdx.x; kx < Wh; kx += 256) {
        const double v = (double)g[(size_t)kx * ny].x;
        acc += (kx == 0 || 2 * kx == nx) ? v : 2.0 * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) peak[blockIdx.x] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

}  // namespace b4d

using namespace b4d;

int b4d_cgemm(const void* A, bool a_real, long long sA, int conj_a, const void* B, bool b_real, long long sB, int conj_b,
              float2* C, long long sC, int M, int N, int K, int batch, hipStream_t st) {
    GemmArgs g{A, B, C, M, N, K, sA, sB, sC, conj_a, conj_b};
    const dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
#ifdef B4D_CGEMM_VALU   // reference build: the vector-ALU product (tools/dev_ab.py compares both)
    if (a_real && !b_real)
        hipLaunchKernelGGL((k_cgemm<true, false>), grid, dim3(256), 0, st, g);
    else if (!a_real && !b_real)
        hipLaunchKernelGGL((k_cgemm<false, false>), grid, dim3(256), 0, st, g);
#else
    if (a_real && !b_real)
        hipLaunchKernelGGL((k_cgemm_mfma<true, false>), grid, dim3(256), 0, st, g);
    else if (!a_real && !b_real)
        hipLaunchKernelGGL((k_cgemm_mfma<false, false>), grid, dim3(256), 0, st, g);
#endif
    else
        return fail(B4D_EINVAL, "cgemm: unsupported operand types");
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// DFT matrix W[j][k] = exp(-2 pi i (j k mod n) / n), float64 on the host
int make_dft_matrix(int n, float2** out) {
    std::vector<float2> h((size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) {
            const double a = -2.0 * M_PI * (double)(((long long)j * k) % n) / (double)n;
            h[(size_t)j * n + k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    B4D_HIP(hipMalloc((void**)out, sizeof(float2) * h.size()));
    B4D_HIP(hipMemcpy(*out, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    return B4D_OK;
}

// F[b] = Wy * (X[b] * Wx)   (inverse: conjugated matrices, no 1/(nx ny))
static int dft2(const b4d_plan* pl, const void* X, bool x_real, int batch, int conj, float2* tmp, float2* F, hipStream_t st) {
    const int ny = pl->ny, nx = pl->nx;
    const long long fp = (long long)ny * nx;
    if (pl->large) {  // rows, transpose, columns, transpose back (sizes beyond the O(n^3) DFT-matrix range)
        int rc = pm_rows(X, x_real, F, batch * ny, nx, pl->tw_x, conj != 0, 1.f, st);
        if (rc == B4D_OK) rc = transpose_batch(F, tmp, ny, nx, batch, st);
        if (rc == B4D_OK) rc = pm_rows(tmp, false, tmp, batch * nx, ny, pl->tw_y, conj != 0, 1.f, st);
        if (rc == B4D_OK) rc = transpose_batch(tmp, F, nx, ny, batch, st);
        return rc;
    }
    int rc = b4d_cgemm(X, x_real, fp, 0, pl->wx, false, 0, conj, tmp, fp, ny, nx, nx, batch, st);
    if (rc) return rc;
    return b4d_cgemm(pl->wy, false, 0, conj, tmp, false, fp, 0, F, fp, ny, nx, ny, batch, st);
}

// exported: the 2-D transform of a general plan (F natural order; conj: conj(DFT(conj X)), no 1/(nx ny)); batch <= chunk
int general_dft2(const b4d_plan* pl, const void* X, bool x_real, int batch, int conj, float2* tmp, float2* F, hipStream_t st) {
    return dft2(pl, X, x_real, batch, conj, tmp, F, st);
}

int general_psd_autocorr(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale, float* autocorr,
                         unsigned flags, hipStream_t st) {
    const int ny = pl->ny, nx = pl->nx, npix = ny * nx;
    const dim3 eg((npix + 255) / 256, 1);
    if (pl->wmr) {
        // both sides have in-register three-radix kernels (b4d_wiener_mr.hip): forward row pairs -> transposed half spectra,
        // one pass over the columns (forward, |F|^2, inverse; the column stays in LDS), inverse row pairs + PSD rows:
        // 3 passes and ~32 bytes per pixel where the route below makes 7 passes
        // two-lane launch groups (Lanes, b4d_fft2d.hpp): lane l works in slot l (sub frames of ny nx complex words) of each buffer
        Lanes ln;
        int rc = ln.open(pl, st, batch, wmr_spectrum_elems(ny, nx) * sizeof(float2), (size_t)npix * sizeof(float),
                         wmr_spectrum_elems(ny, nx) <= (size_t)npix, (size_t)8 << 20);   // a slot holds the transposed half spectra of its frames
        if (rc) return rc;
        int g = 0;
        for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += ln.sub, ++g) {
            const int nb = std::min(ln.sub, batch - b0);
            const size_t off = (size_t)b0 * npix, so = (size_t)ln.slot(g) * ln.sub * npix;
            rc = wmr_psd_autocorr(frames + off, nb, ny, nx, pl->tw_x, pl->tw_y, pl->gbuf1 + so, reinterpret_cast<float*>(pl->gbuf2 + so),
                                  reinterpret_cast<float*>(pl->gbuf3 + so), psd ? psd + off : nullptr, psd_scale,
                                  autocorr ? autocorr + off : nullptr, flags, ln.stream(g));
        }
        const int rj = ln.close();
        return rc ? rc : rj;
    }
    if (pl->large && pm_fusable(nx) && pm_fusable(ny)) {
        // half-spectrum path: pair rows -> transpose -> columns on nx/2 + 1 sequences -> |F|^2 (+ PSD) -> inverse columns ->
        // transpose -> Hermitian pair rows written fftshift-ed and normalised: ~56 instead of ~148 bytes per pixel
        const int Wh = nx / 2 + 1;
        for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
            const int nb = std::min(pl->chunk, batch - b0);
            const size_t off = (size_t)b0 * npix;
            int rc = pm_rows_pair_fwd(frames + off, pl->gbuf1, nb, ny, nx, pl->tw_x, st);
            if (rc == B4D_OK) rc = transpose_batch(pl->gbuf1, pl->gbuf2, ny, Wh, nb, st);
            if (rc == B4D_OK) rc = pm_rows(pl->gbuf2, false, pl->gbuf2, nb * Wh, ny, pl->tw_y, false, 1.f, st);
            if (rc) return rc;
            float* PT = reinterpret_cast<float*>(pl->gbuf1);
            hipLaunchKernelGGL(k_half_power, dim3((ny + 31) / 32, (Wh + 31) / 32, nb), dim3(32, 8), 0, st, pl->gbuf2, ny, nx,
                               autocorr ? PT : nullptr, psd ? psd + off : nullptr, psd_scale, flags);
            B4D_HIP(hipGetLastError());
            if (!autocorr) continue;
            if ((rc = pm_rows(PT, true, pl->gbuf2, nb * Wh, ny, pl->tw_y, true, 1.f, st))) return rc;
            float* peak = reinterpret_cast<float*>(pl->gbuf3);
            const bool norm = (flags & B4D_NORM_PEAK) != 0;
            if (norm) hipLaunchKernelGGL(k_half_peak, dim3(nb), dim3(256), 0, st, pl->gbuf2, ny, nx, peak);
            if ((rc = transpose_batch(pl->gbuf2, pl->gbuf1, Wh, ny, nb, st))) return rc;
            if ((rc = pm_rows_pair_inv(pl->gbuf1, autocorr + off, nb, ny, nx, pl->tw_x, 1.0f / ((float)nx * (float)ny), norm ? peak : nullptr, st)))
                return rc;
        }
        return B4D_OK;
    }
    for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
        const int nb = std::min(pl->chunk, batch - b0);
        const size_t off = (size_t)b0 * npix;
        int rc = dft2(pl, frames + off, true, nb, 0, pl->gbuf1, pl->gbuf2, st);
        if (rc) return rc;
        float* P = reinterpret_cast<float*>(pl->gbuf1);  // gbuf1 is free again: power spectrum (real)
        hipLaunchKernelGGL(k_gen_power, dim3(eg.x, nb), dim3(256), 0, st, pl->gbuf2, ny, nx, psd ? psd + off : nullptr, psd_scale,
                           autocorr ? P : nullptr, flags);
        B4D_HIP(hipGetLastError());
        if (!autocorr) continue;
        // inverse of the REAL power spectrum: tmp = P * conj(Wx) lands in gbuf2, result in gbuf3
        if ((rc = dft2(pl, P, true, nb, 1, pl->gbuf2, pl->gbuf3, st))) return rc;
        hipLaunchKernelGGL(k_gen_real_out, dim3(eg.x, nb), dim3(256), 0, st, pl->gbuf3, ny, nx, autocorr + off,
                           1.0f / ((float)nx * (float)ny), flags);
        B4D_HIP(hipGetLastError());
    }
    return B4D_OK;
}

int general_fft2d(b4d_plan* pl, const float* frames, int batch, float2* out, hipStream_t st) {
    const int ny = pl->ny, nx = pl->nx, npix = ny * nx;
    if (pl->wmr) {   // mixed-radix kernels on both sides (b4d_wiener_mr.hip)
        // two-lane launch groups (Lanes, b4d_fft2d.hpp): lane l works in gbuf(1 + l)
        const size_t selems = wmr_spectrum_elems(ny, nx);
        Lanes ln;
        int rc = ln.open(pl, st, batch, selems * sizeof(float2), (size_t)npix * sizeof(float), true, (size_t)8 << 20);   // (lane 1 has a buffer of its own)
        if (rc) return rc;
        int g = 0;
        for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += ln.sub, ++g) {
            const int nb = std::min(ln.sub, batch - b0);
            float2* T = ln.slot(g) ? pl->gbuf2 : pl->gbuf1;
            rc = wmr_fft2d(frames + (size_t)b0 * npix, nb, ny, nx, pl->tw_x, pl->tw_y, T, reinterpret_cast<float*>(T + selems * nb),
                           out + (size_t)b0 * npix, ln.stream(g));
        }
        const int rj = ln.close();
        return rc ? rc : rj;
    }
    for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
        const int nb = std::min(pl->chunk, batch - b0);
        int rc = dft2(pl, frames + (size_t)b0 * npix, true, nb, 0, pl->gbuf1, pl->gbuf2, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gen_shift_c, dim3((npix + 255) / 256, nb), dim3(256), 0, st, pl->gbuf2, ny, nx, out + (size_t)b0 * npix);
        B4D_HIP(hipGetLastError());
    }
    return B4D_OK;
}

// complex input: forward = fftshift(fft2(x)) (signal/fft.py:198-237 for complex frames); inverse = ifft2(ifftshift(F))
// (signal/fft.py:240-258), natural order out, scaled by 1 / (nx ny)
int general_fft2d_c2c(b4d_plan* pl, const float2* in, int batch, int inverse, float2* out, hipStream_t st) {
    const int ny = pl->ny, nx = pl->nx, npix = ny * nx;
    for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
        const int nb = std::min(pl->chunk, batch - b0);
        const size_t off = (size_t)b0 * npix;
        int rc;
        if (!inverse) {
            if ((rc = dft2(pl, in + off, false, nb, 0, pl->gbuf1, pl->gbuf2, st))) return rc;
            hipLaunchKernelGGL(k_gen_shift_c, dim3((npix + 255) / 256, nb), dim3(256), 0, st, pl->gbuf2, ny, nx, out + off);
        } else {
            hipLaunchKernelGGL(k_gen_unshift_c, dim3((npix + 255) / 256, nb), dim3(256), 0, st, in + off, ny, nx, 1.0f, pl->gbuf3);
            if ((rc = dft2(pl, pl->gbuf3, false, nb, 1, pl->gbuf1, pl->gbuf2, st))) return rc;
            const size_t n = (size_t)nb * npix;
            hipLaunchKernelGGL(k_gen_scale_c, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pl->gbuf2, n, 1.0f / ((float)nx * (float)ny),
                               out + off);
        }
        B4D_HIP(hipGetLastError());
    }
    return B4D_OK;
}

int general_xcorr(b4d_plan* pl, const float* a, const float* b, int batch, float* corr, unsigned flags, hipStream_t st) {
    const int ny = pl->ny, nx = pl->nx, npix = ny * nx;
    if (pl->wmr) {
        // mixed-radix kernels on both sides: two forward half-spectrum passes per operand, product + inverse columns, inverse
        // row pairs (b4d_wiener_mr.hip); the chunk buffers hold complex (chunk, ny, nx), twice a half spectrum
        // two-lane launch groups (Lanes, b4d_fft2d.hpp; 2160 x 2560: 10.4 -> 11.5 k pairs/s): lane l works in slot l (sub frames of
        // ny nx complex words) of each buffer; a slot of gbuf3 holds G of its frames and the row-maxima scratch behind it
        const size_t selems = wmr_spectrum_elems(ny, nx);
        Lanes ln;
        int rc = ln.open(pl, st, batch, selems * sizeof(float2), 2 * (size_t)npix * sizeof(float), selems + (size_t)ny <= (size_t)npix,
                         (size_t)8 << 20);   // (groups of 2-4 frames at 2160 x 2560: measured best)
        if (rc) return rc;
        int g = 0;
        for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += ln.sub, ++g) {
            const int nb = std::min(ln.sub, batch - b0);
            const size_t off = (size_t)b0 * npix, so = (size_t)ln.slot(g) * ln.sub * npix;
            hipStream_t ls = ln.stream(g);
            float2 *A = pl->gbuf1 + so, *B = pl->gbuf2 + so, *G = pl->gbuf3 + so;
            float* scratch = reinterpret_cast<float*>(G + selems * nb);
            rc = wmr_forward_spectra(a + off, nb, ny, nx, pl->tw_x, pl->tw_y, A, scratch, ls);
            if (rc == B4D_OK) rc = wmr_forward_spectra(b + off, nb, ny, nx, pl->tw_x, pl->tw_y, B, scratch, ls);
            if (rc == B4D_OK) rc = wmr_product_inverse(A, B, nullptr, nullptr, nb, ny, nx, pl->tw_y, G, 0, 0.f, flags & B4D_REMOVE_MEAN, ls);
            if (rc == B4D_OK) rc = wmr_rows_real_out(G, nb, ny, nx, pl->tw_x, corr + off, ls);
            if (rc == B4D_OK && (flags & B4D_NORM_PEAK))
                rc = normalise_by_absmax(corr + off, (size_t)npix, nb, reinterpret_cast<float*>(A), ls);
        }
        const int rj = ln.close();
        return rc ? rc : rj;
    }
    for (int b0 = 0; b0 < batch; b0 += pl->chunk) {
        const int nb = std::min(pl->chunk, batch - b0);
        const size_t off = (size_t)b0 * npix;
        int rc = dft2(pl, a + off, true, nb, 0, pl->gbuf1, pl->gbuf2, st);
        if (rc) return rc;
        if ((rc = dft2(pl, b + off, true, nb, 0, pl->gbuf1, pl->gbuf3, st))) return rc;
        hipLaunchKernelGGL(k_gen_cross, dim3((npix + 255) / 256, nb), dim3(256), 0, st, pl->gbuf2, pl->gbuf3, npix, pl->gbuf2, flags);
        B4D_HIP(hipGetLastError());
        if ((rc = dft2(pl, pl->gbuf2, false, nb, 1, pl->gbuf1, pl->gbuf3, st))) return rc;
        hipLaunchKernelGGL(k_gen_real_out, dim3((npix + 255) / 256, nb), dim3(256), 0, st, pl->gbuf3, ny, nx, corr + off,
                           1.0f / ((float)nx * (float)ny), 0u);
        B4D_HIP(hipGetLastError());
        if (flags & B4D_NORM_PEAK)
            if ((rc = normalise_by_absmax(corr + off, (size_t)npix, nb, reinterpret_cast<float*>(pl->gbuf1), st))) return rc;
    }
    return B4D_OK;
}

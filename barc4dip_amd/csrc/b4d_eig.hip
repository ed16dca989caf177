// b4d_eig.hip -- STA2 sharpness eigenvalues (metrics/sharpness.py:752-861) without a dense SVD.
//
// The reference takes ALL singular values of the energy-normalised, mean-removed image J (M x N) with LAPACK and
// then uses only the first k = 5: eig_i = s_i^2 / (M N - 1).  Here:
//   1. J = (x - mean(x)) / ||x||_2 in one pass (float64 reductions, float32 J);
//   2. the Gram matrix G = J J^T (or J^T J, whichever is smaller) on the matrix cores
//      (v_mfma_f32_32x32x2_f32: the dense real contraction SURVEY.md §7 step 8 anticipates);
//   3. block subspace iteration on G with 32 vectors: W = G V (MFMA), Cholesky-QR with a float64 Gram matrix,
//      Rayleigh-Ritz every few sweeps (32 x 32 symmetric eigenproblem, cyclic Jacobi on the host) until the leading
//      Ritz values stop moving.
// Everything is batched over frames / tiles of equal shape.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/b4d.h"
#include "b4d_common.hpp"

namespace b4d {

constexpr int NB = 32;  // subspace width

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T>
__device__ __forceinline__ T wsum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---- 1. normalisation: part[b][blk] = {sum x, sum x^2, n_nonfinite}; grid (nblk, batch), block 1024
__global__ void __launch_bounds__(1024) k_sta2_sums(const float* __restrict__ x, size_t npix, double* __restrict__ part) {
    __shared__ double sh[16 * 3];
    const float* f = x + (size_t)blockIdx.y * npix;
    double s = 0, q = 0, bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const double v = f[i];
        if (isfinite(v)) {
            s += v;
            q = fma(v, v, q);
        } else {
            bad += 1;
        }
    }
    s = wsum(s);
    q = wsum(q);
    bad = wsum(bad);
    if ((threadIdx.x & 63) == 0) {
        sh[(threadIdx.x >> 6) * 3] = s;
        sh[(threadIdx.x >> 6) * 3 + 1] = q;
        sh[(threadIdx.x >> 6) * 3 + 2] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < 16; ++i) {
            a += sh[i * 3];
            b += sh[i * 3 + 1];
            c += sh[i * 3 + 2];
        }
        double* o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        o[0] = a;
        o[1] = b;
        o[2] = c;
    }
}

// J = (x - mean) / norm; stat[b] = {mean, norm, n_nonfinite}.  grid (ceil(npix/256), batch)
__global__ void __launch_bounds__(256) k_sta2_norm(const float* __restrict__ x, size_t npix, const double* __restrict__ part, int nblk,
                                                   float* __restrict__ J, double* __restrict__ stat) {
    __shared__ double s_mean, s_norm;
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < nblk; ++i) {
            const double* p = part + ((size_t)blockIdx.y * nblk + i) * 3;
            a += p[0];
            b += p[1];
            c += p[2];
        }
        s_mean = a / (double)npix;
        s_norm = sqrt(b);
        if (blockIdx.x == 0) {
            stat[blockIdx.y * 3] = s_mean;
            stat[blockIdx.y * 3 + 1] = s_norm;
            stat[blockIdx.y * 3 + 2] = c;
        }
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const double n = s_norm;
    J[(size_t)blockIdx.y * npix + i] = n > 0 ? (float)(((double)x[(size_t)blockIdx.y * npix + i] - s_mean) / n) : 0.f;
}

// ---- 2. Gram matrix on the matrix cores: G[i][j] = sum_k P(i, k) P(j, k), P(i, k) = J[i * si + k * sk]
// (si = nx, sk = 1 for J J^T; si = 1, sk = nx for J^T J).  Only block pairs bx >= by are computed, the mirror is
// written from the same registers, so G is exactly symmetric.
struct GramArgs {
    const float* J;
    float* G;
    int m, K;
    long long si, sk;
    long long bJ, bG;  // batch strides (elements)
};

// grid (ceil(m/128), ceil(m/128), batch), block 256 = 4 waves in 2 x 2, each a 64 x 64 tile (4 f32x16 accumulators);
// K staged 32 deep through LDS
__global__ void __launch_bounds__(256) k_gram_mfma(GramArgs g) {
    if (blockIdx.x < blockIdx.y) return;
    __shared__ float As[32][129];
    __shared__ float Bs[32][129];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const float* J = g.J + (long long)blockIdx.z * g.bJ;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool kc = g.sk == 1;  // k contiguous in memory
    for (int k0 = 0; k0 < g.K; k0 += 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = threadIdx.x + 256 * r;
            const int kk = kc ? (e & 31) : (e >> 7);
            const int xx = kc ? (e >> 5) : (e & 127);
            const int k = k0 + kk;
            const int ia = m0 + xx, ib = n0 + xx;
            As[kk][xx] = (ia < g.m && k < g.K) ? J[ia * g.si + k * g.sk] : 0.f;
            Bs[kk][xx] = (ib < g.m && k < g.K) ? J[ib * g.si + k * g.sk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float a0 = As[kk + lk][64 * wr + li], a1 = As[kk + lk][64 * wr + 32 + li];
            const float b0 = Bs[kk + lk][64 * wc + li], b1 = Bs[kk + lk][64 * wc + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    float* G = g.G + (long long)blockIdx.z * g.bG;
    const bool mirror = blockIdx.x != blockIdx.y;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + 64 * wc + 32 * b + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + 64 * wr + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (mm < g.m && n < g.m) {
                    G[(long long)mm * g.m + n] = acc[a][b][r];
                    if (mirror) G[(long long)n * g.m + mm] = acc[a][b][r];
                }
            }
        }
}

// ---- 3. W = G V for symmetric G (m x m) and a block V (m x NB): every wave streams its K slice of a 32-row slab of G
// straight from memory into the MFMA operands (symmetry makes the slab's columns contiguous rows), partial tiles are
// summed through LDS in a fixed order.  grid (ceil(m/32), batch), block 1024 = 16 waves
constexpr int SYMM_WAVES = 16;
__global__ void __launch_bounds__(SYMM_WAVES * 64) k_symm_block(const float* __restrict__ G, const float* __restrict__ V, int m,
                                                                float* __restrict__ W) {
    __shared__ float red[SYMM_WAVES / 2][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int row0 = blockIdx.x * 32;
    const float* g = G + (size_t)blockIdx.y * m * m + min(row0 + li, m - 1);
    const float* v = V + (size_t)blockIdx.y * m * NB + li;
    int kc = (m + SYMM_WAVES - 1) / SYMM_WAVES;
    kc = (kc + 1) & ~1;
    const int kbeg = min(m, wave * kc), kend = min(m, kbeg + kc);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int k = kbeg;
    for (; k + 16 <= kend; k += 16) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = g[(size_t)(k + 2 * u + lk) * m];
            b[u] = v[(size_t)(k + 2 * u + lk) * NB];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k < kend; k += 2) {
        const int kk = k + lk;
        const float a = kk < kend ? g[(size_t)kk * m] : 0.f;
        const float b = kk < kend ? v[(size_t)kk * NB] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // waves 8..15 hand their tiles to waves 0..7, then the 8 survivors are summed by all lanes
    if (wave >= SYMM_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - SYMM_WAVES / 2][(r & 3) + 8 * (r >> 2) + 4 * lk][li] = acc[r];
    }
    __syncthreads();
    if (wave < SYMM_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[wave][(r & 3) + 8 * (r >> 2) + 4 * lk][li];
    }
    __syncthreads();
    if (wave < SYMM_WAVES / 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lk][li] = acc[r];
    }
    __syncthreads();
    const int row = threadIdx.x / NB, col = threadIdx.x % NB;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SYMM_WAVES / 2; ++w) sum += red[w][row][col];
    if (row0 + row < m) W[(size_t)blockIdx.y * m * NB + (size_t)(row0 + row) * NB + col] = sum;
}

// ---- float64 Gram matrices of tall-skinny blocks: part[b][s][p][q] = sum_{i in slice s} X[i][p] * Y[i][q]
// X, Y: (batch, m, NB) float.  grid (nsplit, batch), block 1024 = NB x NB (p = tid / NB, q = tid % NB)
__global__ void __launch_bounds__(1024) k_gram64(const float* __restrict__ X, const float* __restrict__ Y, int m,
                                                 double* __restrict__ part) {
    __shared__ float xs[32][NB + 1];
    __shared__ float ys[32][NB + 1];
    const int p = threadIdx.x / NB, q = threadIdx.x % NB;
    const size_t base = (size_t)blockIdx.y * m * NB;
    const int rows_per = (m + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per, r1 = min(m, r0 + rows_per);
    double acc = 0.0;
    for (int c0 = r0; c0 < r1; c0 += 32) {
        const int i = c0 + p;  // thread (p, q) loads row p of the chunk, column q
        xs[p][q] = i < r1 ? X[base + (size_t)i * NB + q] : 0.f;
        ys[p][q] = i < r1 ? Y[base + (size_t)i * NB + q] : 0.f;
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < 32; ++t) acc = fma((double)xs[t][p], (double)ys[t][q], acc);
        __syncthreads();
    }
    part[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NB + p) * NB + q] = acc;
}

// reduce partials -> C; optionally Cholesky C = L L^T (ridge-guarded) and Rinv = (L^T)^-1 as float (NB x NB, row-major).
// grid (batch), block 1024
__global__ void __launch_bounds__(1024) k_gram_fin(const double* __restrict__ part, int nsplit, double* __restrict__ Cout,
                                                   float* __restrict__ Rinv, int do_chol) {
    __shared__ double C[NB][NB + 1];
    __shared__ double L[NB][NB + 1];
    __shared__ double Ri[NB][NB + 1];
    const int p = threadIdx.x / NB, q = threadIdx.x % NB;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += part[(((size_t)blockIdx.x * nsplit + k) * NB + p) * NB + q];
    C[p][q] = s;
    if (Cout) Cout[((size_t)blockIdx.x * NB + p) * NB + q] = s;
    if (!do_chol) return;
    __syncthreads();
    if (threadIdx.x < 64) {  // one wave: lane i owns row i of L
        const int i = threadIdx.x;
        double tr = 0.0;
        for (int t = 0; t < NB; ++t) tr += C[t][t];
        const double floor_piv = tr * 1e-14 + 1e-300;
        for (int j = 0; j < NB; ++j) {
            // column j: L[j][j] first (by lane j), then the rows below
            double v = 0.0;
            if (i < NB && i >= j) {
                v = C[i][j];
                for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            }
            double piv = __shfl(v, j, 64);
            if (!(piv > floor_piv)) piv = floor_piv;  // rank-deficient block: keep going with a tiny pivot
            const double d = sqrt(piv);
            if (i < NB && i >= j) L[i][j] = (i == j) ? d : v / d;
            if (i < NB && i < j) L[i][j] = 0.0;
            __threadfence_block();
        }
        // Rinv = (L^T)^-1: column c of Rinv solves L^T x = e_c (back substitution), lane c owns column c
        if (i < NB) {
            const int c = i;
            for (int r = NB - 1; r >= 0; --r) {
                double v = (r == c) ? 1.0 : 0.0;
                for (int k = r + 1; k < NB; ++k) v -= L[k][r] * Ri[k][c];
                Ri[r][c] = v / L[r][r];
            }
        }
    }
    __syncthreads();
    double tr = 0.0;
    for (int t = 0; t < NB; ++t) tr += C[t][t];
    Rinv[((size_t)blockIdx.x * NB + p) * NB + q] = tr > 1e-250 ? (float)Ri[p][q] : 0.f;  // all-zero block stays zero
}

// V = W * Rinv (m x NB times NB x NB).  grid (ceil(m/32), batch), block 1024: thread (row r, col q)
__global__ void __launch_bounds__(1024) k_apply_rinv(const float* __restrict__ W, const float* __restrict__ Rinv, int m,
                                                     float* __restrict__ V) {
    __shared__ float ws[32][NB + 1];
    __shared__ float rs[NB][NB + 1];
    const int r = threadIdx.x / NB, q = threadIdx.x % NB;
    const int row = blockIdx.x * 32 + r;
    const size_t base = (size_t)blockIdx.y * m * NB;
    ws[r][q] = row < m ? W[base + (size_t)row * NB + q] : 0.f;
    rs[r][q] = Rinv[((size_t)blockIdx.y * NB + r) * NB + q];
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t) acc = fmaf(ws[r][t], rs[t][q], acc);
    if (row < m) V[base + (size_t)row * NB + q] = acc;
}

// deterministic start block (integer hash -> [-1, 1))
__global__ void k_init_block(float* __restrict__ V, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)(i * 2654435761u) ^ 0x9E3779B9u;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    V[i] = (float)h * (2.0f / 4294967296.0f) - 1.0f;
}

// ---- Rayleigh-Ritz on the device: T = V^T W, S = V^T V (float64 partials from k_gram64), eigenvalues of the pencil
// (T, S): S = L L^T, A = L^-1 T L^-T, parallel cyclic Jacobi (round-robin pairing: 16 disjoint rotations per round,
// row phase then column phase), leading 8 eigenvalues written descending.  grid (batch), block 1024 = NB x NB
constexpr int RITZ_SWEEPS = 10;
__global__ void __launch_bounds__(1024) k_ritz(const double* __restrict__ partT, const double* __restrict__ partS, int nsplit,
                                               double* __restrict__ ev) {
    __shared__ double A0[NB][NB + 1];
    __shared__ double A1[NB][NB + 1];
    __shared__ double L[NB][NB + 1];
    __shared__ double cs[NB], sg[NB];
    __shared__ int partner[NB];
    const int p = threadIdx.x / NB, q = threadIdx.x % NB;
    double t = 0.0, sv = 0.0;
    for (int k = 0; k < nsplit; ++k) {
        const size_t o = ((size_t)blockIdx.x * nsplit + k) * NB * NB;
        t += 0.5 * (partT[o + p * NB + q] + partT[o + q * NB + p]);
        sv += 0.5 * (partS[o + p * NB + q] + partS[o + q * NB + p]);
    }
    A0[p][q] = t;
    A1[p][q] = sv;
    __syncthreads();
    // Cholesky of S (A1) into L and the two triangular solves, right-looking on the whole workgroup: once column j (row i) is final,
    // every element of the trailing part takes its one update.  Each element sees the same products subtracted in the same
    // (ascending) order as in the row-by-row form -- bit-identical results -- in 32 short steps per phase instead of a
    // 500-step dependent chain on one wavefront (the check sits on the critical path of small batches: tile grids).
    {
        double tr = 0.0;
        for (int c = 0; c < NB; ++c) tr += A1[c][c];
        const double floor_piv = tr * 1e-14 + 1e-300;
        for (int j = 0; j < NB; ++j) {
            double piv = A1[j][j];
            if (!(piv > floor_piv)) piv = floor_piv;
            const double d = sqrt(piv);
            if (q == j && p >= j) L[p][j] = p > j ? A1[p][j] / d : d;
            __syncthreads();
            if (p > j && q > j && q <= p) A1[p][q] -= L[p][j] * L[q][j];
            __syncthreads();
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < NB; ++i) {  // forward substitution L X = A0 in place: row i final, then its share of every later row
            if (p == i) A0[i][q] = A0[i][q] / L[i][i];
            __syncthreads();
            if (p > i) A0[p][q] -= L[p][i] * A0[i][q];
            __syncthreads();
        }
        const double x = A0[q][p];
        __syncthreads();
        A0[p][q] = x;  // transpose: second pass applies L^-1 from the other side
        __syncthreads();
    }
    {
        const double x = 0.5 * (A0[p][q] + A0[q][p]);
        __syncthreads();
        A0[p][q] = x;
        __syncthreads();
    }
    // A sweep in which no pair needed a rotation leaves the matrix as it is, and so would every later one: stop there (the
    // Ritz matrix of a nearly converged subspace is nearly diagonal: 2-3 sweeps instead of the fixed 10, same bits out).
    __shared__ int rotated;
    for (int sweep = 0; sweep < RITZ_SWEEPS; ++sweep) {
        if (threadIdx.x == 0) rotated = 0;
        __syncthreads();
        for (int r = 0; r < NB - 1; ++r) {
            if (threadIdx.x < NB / 2) {
                const int i = threadIdx.x;
                const int a = i == 0 ? r : (r + i) % (NB - 1);
                const int b = i == 0 ? NB - 1 : (r - i + NB - 1) % (NB - 1);
                const int pp = min(a, b), qq = max(a, b);
                const double app = A0[pp][pp], aqq = A0[qq][qq], apq = A0[pp][qq];
                double c = 1.0, s_ = 0.0;
                if (fabs(apq) > 1e-18 * (fabs(app) + fabs(aqq)) && fabs(apq) > 1e-300) {
                    const double theta = (aqq - app) / (2.0 * apq);
                    const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(tt * tt + 1.0);
                    s_ = tt * c;
                    rotated = 1;
                }
                cs[pp] = c;
                cs[qq] = c;
                sg[pp] = -s_;
                sg[qq] = s_;
                partner[pp] = qq;
                partner[qq] = pp;
            }
            __syncthreads();
            A1[p][q] = cs[p] * A0[p][q] + sg[p] * A0[partner[p]][q];  // rows
            __syncthreads();
            A0[p][q] = cs[q] * A1[p][q] + sg[q] * A1[p][partner[q]];  // columns
            __syncthreads();
        }
        const int any = rotated;
        __syncthreads();
        if (!any) break;
    }
    if (q == 0) {
        const double d = A0[p][p];
        int rank = 0;
        for (int j = 0; j < NB; ++j) {
            const double e = A0[j][j];
            rank += (e > d || (e == d && j < p)) ? 1 : 0;
        }
        if (rank < 8) ev[(size_t)blockIdx.x * 8 + rank] = d;
    }
}

// ---- small frames (min side < 64: the 32-vector subspace would be most of the matrix): every eigenvalue of the
// m x m Gram matrix by parallel cyclic Jacobi in float64 (64 x 64 padded with zeros; 32 disjoint rotations per round,
// row phase then column phase), leading 8 written descending.  grid (batch), block 1024, dynamic LDS 2 * 64 * 65 doubles.
constexpr int JS = 64;
__global__ void __launch_bounds__(1024) k_jacobi_small(const float* __restrict__ G, int m, double* __restrict__ ev) {
    extern __shared__ double jsm[];
    double(*A0)[JS + 1] = reinterpret_cast<double(*)[JS + 1]>(jsm);
    double(*A1)[JS + 1] = reinterpret_cast<double(*)[JS + 1]>(jsm + JS * (JS + 1));
    __shared__ double cs[JS], sg[JS];
    __shared__ int partner[JS];
    const float* g = G + (size_t)blockIdx.x * m * m;
    for (int e = threadIdx.x; e < JS * JS; e += 1024) {
        const int p = e / JS, q = e % JS;
        A0[p][q] = (p < m && q < m) ? 0.5 * ((double)g[p * m + q] + (double)g[q * m + p]) : 0.0;
    }
    __syncthreads();
    __shared__ int rotated;   // as in k_ritz: a sweep without a rotation ends the iteration
    for (int sweep = 0; sweep < 12; ++sweep) {
        if (threadIdx.x == 0) rotated = 0;
        __syncthreads();
        for (int r = 0; r < JS - 1; ++r) {
            if (threadIdx.x < JS / 2) {
                const int i = threadIdx.x;
                const int a = i == 0 ? r : (r + i) % (JS - 1);
                const int b = i == 0 ? JS - 1 : (r - i + JS - 1) % (JS - 1);
                const int pp = min(a, b), qq = max(a, b);
                const double app = A0[pp][pp], aqq = A0[qq][qq], apq = A0[pp][qq];
                double c = 1.0, s_ = 0.0;
                if (fabs(apq) > 1e-18 * (fabs(app) + fabs(aqq)) && fabs(apq) > 1e-300) {
                    const double theta = (aqq - app) / (2.0 * apq);
                    const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(tt * tt + 1.0);
                    s_ = tt * c;
                    rotated = 1;
                }
                cs[pp] = c;
                cs[qq] = c;
                sg[pp] = -s_;
                sg[qq] = s_;
                partner[pp] = qq;
                partner[qq] = pp;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < JS * JS; e += 1024) {
                const int p = e / JS, q = e % JS;
                A1[p][q] = cs[p] * A0[p][q] + sg[p] * A0[partner[p]][q];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < JS * JS; e += 1024) {
                const int p = e / JS, q = e % JS;
                A0[p][q] = cs[q] * A1[p][q] + sg[q] * A1[p][partner[q]];
            }
            __syncthreads();
        }
        const int any = rotated;
        __syncthreads();
        if (!any) break;
    }
    if (threadIdx.x < JS) {
        const int p = threadIdx.x;
        const double d = p < m ? A0[p][p] : -1e300;   // padding never ranks among the eigenvalues
        int rank = 0;
        for (int j = 0; j < JS; ++j) {
            const double e = j < m ? A0[j][j] : -1e300;
            rank += (e > d || (e == d && j < p)) ? 1 : 0;
        }
        if (rank < 8) ev[(size_t)blockIdx.x * 8 + rank] = p < m ? d : 0.0;
    }
}

}  // namespace b4d

using namespace b4d;

extern "C" int b4d_sta2_eigenvalues(const float* frames, int batch, int ny, int nx, double* out_host, int nout, void* stream) {
    B4D_SCRATCH_LOCK();
    if (!frames || !out_host) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || ny < 1 || nx < 1 || nout < 1 || nout > 8) return fail(B4D_EINVAL, "batch, ny, nx >= 1 and 1 <= nout <= 8 required");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)ny * nx;
    const int m = std::min(ny, nx);
    const int nblk = (int)std::min<size_t>(64, (npix + 1023) / 1024);
    if (m < 2 * NB) {   // small frames: all eigenvalues of the m x m Gram matrix by Jacobi
        size_t sb = 0;
        auto tk = [&](size_t b) {
            const size_t o = sb;
            sb += (b + 255) & ~(size_t)255;
            return o;
        };
        const size_t oJ = tk(sizeof(float) * npix * batch), oG = tk(sizeof(float) * (size_t)m * m * batch);
        const size_t oP = tk(sizeof(double) * (size_t)nblk * 3 * batch), oE = tk(sizeof(double) * 8 * batch), oS = tk(sizeof(double) * 3 * batch);
        void* ws = nullptr;
        int rc = get_scratch(sb, &ws, (hipStream_t)stream);
        if (rc) return rc;
        char* base = static_cast<char*>(ws);
        float* J = reinterpret_cast<float*>(base + oJ);
        float* G = reinterpret_cast<float*>(base + oG);
        double* part = reinterpret_cast<double*>(base + oP);
        double* evd = reinterpret_cast<double*>(base + oE);
        double* stat = reinterpret_cast<double*>(base + oS);
        hipLaunchKernelGGL(k_sta2_sums, dim3(nblk, batch), dim3(1024), 0, st, frames, npix, part);
        hipLaunchKernelGGL(k_sta2_norm, dim3((unsigned)((npix + 255) / 256), batch), dim3(256), 0, st, frames, npix, part, nblk, J, stat);
        GramArgs g{J, G, m, ny <= nx ? nx : ny, ny <= nx ? (long long)nx : 1LL, ny <= nx ? 1LL : (long long)nx, (long long)npix, (long long)m * m};
        hipLaunchKernelGGL(k_gram_mfma, dim3(1, 1, batch), dim3(256), 0, st, g);
        B4D_HIP(hipMemsetAsync(evd, 0, sizeof(double) * 8 * batch, st));
        const size_t lds = sizeof(double) * 2 * JS * (JS + 1);
        if (int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_jacobi_small), lds)) return rc_lds;
        hipLaunchKernelGGL(k_jacobi_small, dim3(batch), dim3(1024), lds, st, G, m, evd);
        B4D_HIP(hipGetLastError());
        std::vector<double> cur((size_t)batch * 8), hstat((size_t)3 * batch);
        B4D_HIP(hipMemcpyAsync(cur.data(), evd, sizeof(double) * cur.size(), hipMemcpyDeviceToHost, st));
        B4D_HIP(hipMemcpyAsync(hstat.data(), stat, sizeof(double) * hstat.size(), hipMemcpyDeviceToHost, st));
        B4D_HIP(hipStreamSynchronize(st));
        const double denom = (double)npix - 1.0;
        for (int b = 0; b < batch; ++b)
            for (int k = 0; k < nout; ++k) {
                const bool ok = hstat[(size_t)b * 3 + 2] == 0.0 && hstat[(size_t)b * 3 + 1] > 0.0 && denom > 0.0;
                out_host[(size_t)b * nout + k] = ok ? std::max(cur[(size_t)b * 8 + k], 0.0) / denom : std::nan("");
            }
        return B4D_OK;
    }
    const int nsplit = std::max(1, std::min(16, m / 128));
    size_t bytes = 0;
    auto take = [&](size_t b) {
        const size_t o = bytes;
        bytes += (b + 255) & ~(size_t)255;
        return o;
    };
    const size_t blk = sizeof(float) * (size_t)m * NB * batch;
    const size_t oJ = take(sizeof(float) * npix * batch), oG = take(sizeof(float) * (size_t)m * m * batch);
    const size_t oV = take(blk), oW = take(blk), oX = take(blk);
    const size_t part_elems = std::max<size_t>((size_t)nblk * 3, (size_t)nsplit * NB * NB) * batch;
    const size_t oP = take(sizeof(double) * part_elems), oP2 = take(sizeof(double) * part_elems);
    const size_t oPc = take(sizeof(double) * part_elems), oPc2 = take(sizeof(double) * part_elems);   // partials of the convergence check
    const size_t oR = take(sizeof(float) * NB * NB * batch), oE = take(sizeof(double) * 8 * batch), oS = take(sizeof(double) * 3 * batch);
    void* ws = nullptr;
    int rc = get_scratch(bytes, &ws, (hipStream_t)stream);
    if (rc) return rc;
    char* base = static_cast<char*>(ws);
    float* J = reinterpret_cast<float*>(base + oJ);
    float* G = reinterpret_cast<float*>(base + oG);
    float* V = reinterpret_cast<float*>(base + oV);
    float* W = reinterpret_cast<float*>(base + oW);
    float* X = reinterpret_cast<float*>(base + oX);
    double* part = reinterpret_cast<double*>(base + oP);
    double* part2 = reinterpret_cast<double*>(base + oP2);
    double* partc = reinterpret_cast<double*>(base + oPc);
    double* partc2 = reinterpret_cast<double*>(base + oPc2);
    float* Rinv = reinterpret_cast<float*>(base + oR);
    double* evd = reinterpret_cast<double*>(base + oE);
    double* stat = reinterpret_cast<double*>(base + oS);

    hipLaunchKernelGGL(k_sta2_sums, dim3(nblk, batch), dim3(1024), 0, st, frames, npix, part);
    hipLaunchKernelGGL(k_sta2_norm, dim3((unsigned)((npix + 255) / 256), batch), dim3(256), 0, st, frames, npix, part, nblk, J, stat);
    {
        GramArgs g{J, G, m, ny <= nx ? nx : ny, ny <= nx ? (long long)nx : 1LL, ny <= nx ? 1LL : (long long)nx, (long long)npix,
                   (long long)m * m};
        const int nb = (m + 127) / 128;
        hipLaunchKernelGGL(k_gram_mfma, dim3(nb, nb, batch), dim3(256), 0, st, g);
    }
    hipLaunchKernelGGL(k_init_block, dim3((unsigned)(((size_t)m * NB * batch + 255) / 256)), dim3(256), 0, st, W, (size_t)m * NB * batch);
    B4D_HIP(hipGetLastError());
    auto symm = [&](const float* in, float* out) {
        hipLaunchKernelGGL(k_symm_block, dim3((m + 31) / 32, batch), dim3(SYMM_WAVES * 64), 0, st, G, in, m, out);
    };
    auto orthonormalise = [&]() {  // W -> V (Cholesky-QR, float64 Gram)
        hipLaunchKernelGGL(k_gram64, dim3(nsplit, batch), dim3(1024), 0, st, W, W, m, part);
        hipLaunchKernelGGL(k_gram_fin, dim3(batch), dim3(1024), 0, st, part, nsplit, (double*)nullptr, Rinv, 1);
        hipLaunchKernelGGL(k_apply_rinv, dim3((m + 31) / 32, batch), dim3(1024), 0, st, W, Rinv, m, V);
    };
    orthonormalise();
    // second pass: the hash block is far from orthogonal, one more Cholesky-QR brings V to float32 orthonormality
    hipLaunchKernelGGL(k_gram64, dim3(nsplit, batch), dim3(1024), 0, st, V, V, m, part);
    hipLaunchKernelGGL(k_gram_fin, dim3(batch), dim3(1024), 0, st, part, nsplit, (double*)nullptr, Rinv, 1);
    hipLaunchKernelGGL(k_apply_rinv, dim3((m + 31) / 32, batch), dim3(1024), 0, st, V, Rinv, m, W);
    std::swap(V, W);
    B4D_HIP(hipGetLastError());
    std::vector<double> prev((size_t)batch * 8, 0.0), cur((size_t)batch * 8), hstat((size_t)3 * batch);
    const int max_cycles = 120, check_every = 2;   // 3 multiplications by G per cycle
    // The convergence check (two float64 Gram matrices, the 32 x 32 Ritz problem -- one workgroup per item, ~220 us of latency --
    // and a copy to the host) runs on the library's second stream while the caller's stream goes on with the next two cycles; the
    // host reads a check when the next one is due.  The values checked, the cycles at which they are taken and the result are
    // those of the serial loop; the last two cycles queued before the verdict arrives are the price (they touch scratch only).
    hipStream_t aux = nullptr;
    hipEvent_t ev_part = nullptr;
    if (!g_opt_lanes.load() || lane_stream(1, &aux) != B4D_OK || hipEventCreateWithFlags(&ev_part, hipEventDisableTiming) != hipSuccess)
        aux = nullptr;   // option "lanes" 0 (or no second stream): the serial loop
    struct EvGuard {
        hipEvent_t& a;
        ~EvGuard() {
            if (a) (void)hipEventDestroy(a);
        }
    } ev_guard{ev_part};
    bool done = false, pending = false, first_check = true;
    int ncyc = 0;
    auto verdict = [&]() {   // cur holds a finished check
        bool ok = true;
        for (int b = 0; b < batch; ++b) {
            const double ref = std::max(std::fabs(cur[(size_t)b * 8]), 1e-300);
            for (int k = 0; k < 8; ++k)
                if (!(std::fabs(cur[(size_t)b * 8 + k] - prev[(size_t)b * 8 + k]) <= 2e-7 * ref)) ok = false;
        }
        prev = cur;
        return ok;
    };
    auto collect = [&]() -> int {   // wait for the check in flight on the second stream
        B4D_HIP(hipMemcpyAsync(cur.data(), evd, sizeof(double) * cur.size(), hipMemcpyDeviceToHost, aux));
        if (first_check) B4D_HIP(hipMemcpyAsync(hstat.data(), stat, sizeof(double) * hstat.size(), hipMemcpyDeviceToHost, aux));
        B4D_HIP(hipStreamSynchronize(aux));
        first_check = false;
        pending = false;
        return B4D_OK;
    };
    for (int cyc = 0; cyc < max_cycles && !done; ++cyc) {
        ncyc = cyc + 1;
        symm(V, W);
        if ((cyc + 1) % check_every == 0 || cyc == max_cycles - 1) {
            if (aux) {
                if (pending) {
                    if ((rc = collect())) return rc;
                    if (verdict()) {
                        done = true;
                        ncyc = cyc + 1 - check_every;
                        break;
                    }
                }
                // the Gram partials of a check have buffers of their own: the next cycle's orthonormalisation reuses `part`
                hipLaunchKernelGGL(k_gram64, dim3(nsplit, batch), dim3(1024), 0, st, V, W, m, partc);
                hipLaunchKernelGGL(k_gram64, dim3(nsplit, batch), dim3(1024), 0, st, V, V, m, partc2);
                B4D_HIP(hipGetLastError());
                B4D_HIP(hipEventRecord(ev_part, st));
                B4D_HIP(hipStreamWaitEvent(aux, ev_part, 0));
                hipLaunchKernelGGL(k_ritz, dim3(batch), dim3(1024), 0, aux, partc, partc2, nsplit, evd);
                B4D_HIP(hipGetLastError());
                pending = true;
                if (cyc == max_cycles - 1) {
                    if ((rc = collect())) return rc;
                    done = verdict();
                    break;
                }
                // (the next check collects this one on the host before its partials overwrite partc / partc2)
            } else {
                hipLaunchKernelGGL(k_gram64, dim3(nsplit, batch), dim3(1024), 0, st, V, W, m, part);
                hipLaunchKernelGGL(k_gram64, dim3(nsplit, batch), dim3(1024), 0, st, V, V, m, part2);
                hipLaunchKernelGGL(k_ritz, dim3(batch), dim3(1024), 0, st, part, part2, nsplit, evd);
                B4D_HIP(hipGetLastError());
                B4D_HIP(hipMemcpyAsync(cur.data(), evd, sizeof(double) * cur.size(), hipMemcpyDeviceToHost, st));
                B4D_HIP(hipStreamSynchronize(st));
                done = verdict();
                if (done) break;
            }
        }
        symm(W, X);
        symm(X, W);
        orthonormalise();
        B4D_HIP(hipGetLastError());
    }
    if (pending) {   // (not reached: the last cycle collects its own check)
        if ((rc = collect())) return rc;
        done = verdict();
    }
    if (getenv("B4D_DEBUG_EIG")) fprintf(stderr, "[b4d_sta2] batch %d m %d cycles %d converged %d\n", batch, m, ncyc, (int)done);
    if (!aux) {
        B4D_HIP(hipMemcpyAsync(hstat.data(), stat, sizeof(double) * hstat.size(), hipMemcpyDeviceToHost, st));
        B4D_HIP(hipStreamSynchronize(st));
    }
    const double denom = (double)npix - 1.0;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < nout; ++k) {
            const bool ok = hstat[(size_t)b * 3 + 2] == 0.0 && hstat[(size_t)b * 3 + 1] > 0.0 && denom > 0.0;
            out_host[(size_t)b * nout + k] = ok ? std::max(prev[(size_t)b * 8 + k], 0.0) / denom : std::nan("");
        }
    return B4D_OK;
}

// b4d_wiener.hip -- Gaussian-PSF Wiener deconvolution of barc4dip.preprocessing.deconvolve_psf
// (preprocessing/filters.py:17-289, method="wiener"; BASELINE.json config 5) on gfx950.
//
// The reference pads every frame by psf//2 with "reflect", normalises by max|.|, applies the Wiener-Hunt
// filter of skimage.restoration.wiener in the Fourier domain of the PADDED size, clips to [-1, 1], rescales
// and crops.  The padded sizes are awkward by construction (4096 + 2*4 = 4104 = 2^3 * 3^3 * 19), so the
// transforms here are mixed: N = P * M with P <= 16 a power of two done as an in-register radix-P butterfly
// plus twiddles, and the length-M part as a dense DFT-matrix product (the complex GEMM of b4d_general.hip):
//
//   X[k1 + P k2] = sum_{n2 < M} W_M^{n2 k2} [ W_N^{n2 k1} sum_{n1 < P} x[M n1 + n2] W_P^{n1 k1} ]
//
// 2-D: row pass, transpose, row pass (the spectrum stays transposed: the filter is stored transposed as well),
// multiply, and the same two passes back with conjugated inputs/outputs.
// Parity: UNPINNED (scikit-image is not installable here); oracle/wiener_np.py restates the published algorithm.
#include "b4d_fft2d.hpp"
#include "b4d_wiener_mr.hpp"

// complex GEMM of b4d_general.hip
int b4d_cgemm(const void* A, bool a_real, long long sA, int conj_a, const void* B, bool b_real, long long sB, int conj_b,
              float2* C, long long sC, int M, int N, int K, int batch, hipStream_t st);

namespace b4d {

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// step A: y[(s*P + k1)*M + n2] = W_N^{n2 k1} * sum_{n1} x[s*N + M n1 + n2] * W_P^{n1 k1}     (conj_in: x -> conj x)
// one lane per (s, n2); twN: N-point twiddles exp(-2 pi i k / N).  grid (ceil(M/256), S)
template <int P, bool REAL_IN>
__global__ void __launch_bounds__(256) k_pm_pre(const void* __restrict__ xin, float2* __restrict__ y, const float2* __restrict__ twN,
                                                int M, int conj_in) {
    const int n2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (n2 >= M) return;
    const size_t s = blockIdx.y;
    const int N = P * M;
    float2 v[P];
#pragma unroll
    for (int n1 = 0; n1 < P; ++n1) {
        const size_t i = s * (size_t)N + (size_t)M * n1 + n2;
        if (REAL_IN) {
            v[n1] = make_float2(static_cast<const float*>(xin)[i], 0.f);
        } else {
            const float2 q = static_cast<const float2*>(xin)[i];
            v[n1] = conj_in ? make_float2(q.x, -q.y) : q;
        }
    }
    Dft<P>::run(v);
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
        const float2 w = k1 == 0 ? make_float2(1.f, 0.f) : twN[(size_t)((long long)n2 * k1 % N)];
        y[(s * P + k1) * (size_t)M + n2] = cmulf(v[k1], w);
    }
}

// Fused length-N transform of one row per workgroup when M = A * B splits into two small factors: the radix-P
// butterfly + twiddle of step A, then DFT_M as DFT_A (over a, n2 = B a + b), the twiddle W_M^{b c} and DFT_B (over b,
// k2 = c + A d), all in LDS; the two small DFTs are dense sums from LDS-resident tables (M (A + B) complex MACs per
// sequence instead of the M^2 of the DFT-matrix product).  Output k = k1 + P (c + A d), optional pointwise filter,
// conjugation and scale fused into the coalesced copy-out.  Safe in place.
//   IN  0 complex rows | 1 real rows | 2 rows of the reflect-padded, max-normalised frame taken straight from the
//       (h, w) frame (np.pad(..., "reflect") / max|frame|, filters.py:252-261: no padded copy in memory)
//       3 as 2 for the row PAIR (2 s, 2 s + 1) packed as real + i imaginary part of one transform
//       4 the Hermitian pair: half rows 2 s, 2 s + 1 (io.half values each) extended to Ga + i Gb (inverse pass)
//       5 two plain real rows of a (frames, io.rows, N) float stack packed as real + i imaginary part
//   OUT 0 complex rows | 1 clip(Re, -1, 1) * max|frame| cropped back to (h, w) (filters.py:266, 287-289)
//       2 the pair's half spectra Fa, Fb (k = 0 .. io.half - 1) unpacked to half rows 2 s, 2 s + 1
//       3 as 1 for the pair: real part -> row 2 s, imaginary part -> row 2 s + 1
//       4 the pair's two real rows written fftshift-ed into a (frames, io.rows, N) float stack, scaled or divided by
//         the frame's zero-lag value io.amax[frame] (autocorrelation peak normalisation)
//   Pair modes index sequences as s = frame * ceil(io.rows / 2) + pair: pairs never straddle two frames.
// grid (S), block FT, dynamic LDS (2 N + A + B) complex values.
constexpr int FT_MAX = 1024, FT_ONEBUF = 512;
// acc += x * w (complex) in two packed FMAs
__device__ __forceinline__ v2f cmac(v2f acc, v2f x, v2f w) {
    v2f t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(x), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
    return r;
}
// Dense small DFT on the matrix cores (the one dense contraction of this path; v_mfma_f32_16x16x4_f32 is exact f32 at
// the packed-FP32 flop rate, but one ds_read_b64 per operand feeds 4 MFMAs = 1024 complex MACs: ~12 x less LDS traffic
// than the 4 x 2 register blocks below).  Y(r, n) = sum_{k < R} T[k][r] * X(k, n) for r < R <= 32: one wave per strip
// of 16 columns n, two 16 x 16 complex tiles (rows 0..15, 16..31), k in steps of 4.
// Fragment maps (MI355X guide): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D: col = l & 15, row = 4 (l >> 4) + reg.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct DftTiles {
    f32x4 r0, i0, r1, i1;
};
// tab: T[k * Tp + r]; x: this lane's column (nullptr = padding column), element k at x[k * xstride]
__device__ __forceinline__ DftTiles small_dft_mfma(const float2* __restrict__ tab, int R, int Tp, const float2* __restrict__ x, int xstride,
                                                   int lane) {
    const int j = lane & 15, kq = lane >> 4;
    DftTiles t;
    t.r0 = t.i0 = t.r1 = t.i1 = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool two = R > 16, row0 = j < R, row1 = 16 + j < R;
    for (int kk = 0; kk < R; kk += 4) {
        const int k = kk + kq;
        const bool vk = k < R;
        const float2 xv = (vk && x) ? x[k * xstride] : make_float2(0.f, 0.f);
        const float2 w0 = (vk && row0) ? tab[k * Tp + j] : make_float2(0.f, 0.f);
        t.r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, xv.x, t.r0, 0, 0, 0);
        t.i0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, xv.y, t.i0, 0, 0, 0);
        if (two) {
            const float2 w1 = (vk && row1) ? tab[k * Tp + 16 + j] : make_float2(0.f, 0.f);
            t.r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, xv.x, t.r1, 0, 0, 0);
            t.i1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, xv.y, t.i1, 0, 0, 0);
            t.r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-w1.y, xv.y, t.r1, 0, 0, 0);
            t.i1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, xv.x, t.i1, 0, 0, 0);
        }
        t.r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-w0.y, xv.y, t.r0, 0, 0, 0);
        t.i0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, xv.x, t.i0, 0, 0, 0);
    }
    return t;
}
// complex LDS words of one row transform: two row buffers + the small-DFT tables (full A x A / B x B matrices, padded to
// multiples of 4 columns, when both factors are <= 32; otherwise the A + B roots of unity)
__host__ __device__ inline size_t pm_lds_elems(int P, int A, int B, bool onebuf = false) {
    const size_t tabs = (A <= 32 && B <= 32) ? (size_t)A * ((A + 3) & ~3) + (size_t)B * ((B + 3) & ~3) : (size_t)A + B;
    return (onebuf ? 1 : 2) * (size_t)P * A * B + tabs;
}
// one-buffer mode: blocked small DFTs whose item counts fit one round of FT_ONEBUF lanes (measured: 2560 = 16 * 16 * 10
// gains 1.5x from three workgroups per CU; 4104 = 8 * 27 * 19 with 560 items is faster on two 1024-lane workgroups)
inline bool pm_onebuf(int P, int A, int B) {
    if (!(A <= 32 && B <= 32)) return false;
    const int Ap = (A + 3) & ~3, Bp = (B + 3) & ~3;
    return P * (Ap / 4) * ((B + 1) / 2) <= FT_ONEBUF && P * ((A + 1) / 2) * (Bp / 4) <= FT_ONEBUF;
}
struct FusedIO {
    const float* frame;   // IN 2 / OUT 1: the (h, w) frame read / written
    float* crop;
    const float* amax;    // max|frame| (device scalar)
    int h, w, py, px, clip;
    int half, rows;       // pair modes: half-row length N/2 + 1 and the number of (padded) rows
    int filt_bcast;       // the pointwise multiplier is ONE row shared by every sequence (Bluestein's chirp spectrum)
    int norm_peak;        // OUT 4: divide by io.amax[frame] when it is > 0 and force the zero lag to exactly 1
};

#ifdef B4D_DIAG
// Diagnostic build (never shipped, never timed as a whole): wall-clock stamps (100 MHz) of lane 0 at the phase boundaries.
__device__ unsigned long long* g_pm_diag = nullptr;
#define B4D_PM_STAMP(i)                                                                       \
    do {                                                                                      \
        if (g_pm_diag && threadIdx.x == 0 && DIAG_SEL) g_pm_diag[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define B4D_PM_STAMP(i) do { } while (0)
#endif
template <int P, int IN, int OUT, bool ONEBUF>
// two 1024-lane workgroups per CU need <= 64 VGPRs: asked for explicitly where the radix-P stage leaves room (P <= 8)
__global__ void __launch_bounds__(ONEBUF ? FT_ONEBUF : FT_MAX, (!ONEBUF && P <= 8) ? 8 : 1) k_pm_fused(const void* __restrict__ xin, float2* __restrict__ out, const float2* __restrict__ twN,
                                                 int A, int B, const float2* __restrict__ filt, int conj_io, float scale, FusedIO io) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr bool onebuf = ONEBUF;
    constexpr int FT = ONEBUF ? FT_ONEBUF : FT_MAX;
    const int M = A * B, N = P * M;
    // onebuf (small item counts): every lane owns at most one item of the two small-DFT phases, keeps its outputs in
    // registers across a barrier and writes them back into the SAME row buffer: half the LDS, three workgroups per CU
    float2* buf0 = sm;
    float2* buf1 = onebuf ? sm : sm + N;
    float2* tabA = sm + (onebuf ? N : 2 * N);
    const bool blocked = A <= 32 && B <= 32;   // full small-DFT matrices in LDS, 4 x 2 register blocks, packed FMAs
    const int Ap = (A + 3) & ~3, Bp = (B + 3) & ~3;
    float2* tabB = tabA + (blocked ? A * Ap : A);
    const size_t s = blockIdx.x;
#ifdef B4D_DIAG
#ifndef B4D_DIAG_PM_IN
#define B4D_DIAG_PM_IN 3
#endif
    constexpr bool DIAG_SEL = IN == B4D_DIAG_PM_IN;
#endif
    B4D_PM_STAMP(0);
#ifndef B4D_EXP_PM_NOTAB   // timing-only switch: what the per-row table build costs
    if (blocked) {   // tabA[a][c] = W_A^{a c} (c < A, else 0), tabB[b][d] = W_B^{b d}
        for (int i = threadIdx.x; i < A * Ap; i += FT) {
            const int a = i / Ap, c = i % Ap;
            tabA[i] = c < A ? twN[(size_t)(N / A) * ((a * c) % A)] : make_float2(0.f, 0.f);
        }
        for (int i = threadIdx.x; i < B * Bp; i += FT) {
            const int b = i / Bp, d = i % Bp;
            tabB[i] = d < B ? twN[(size_t)(N / B) * ((b * d) % B)] : make_float2(0.f, 0.f);
        }
    } else {
        for (int i = threadIdx.x; i < A; i += FT) tabA[i] = twN[(size_t)(N / A) * i];
        for (int i = threadIdx.x; i < B; i += FT) tabB[i] = twN[(size_t)(N / B) * i];
    }
#endif
    float fsc = 1.f;
    bool fok = true;
    if (IN == 2 || IN == 3 || OUT == 1 || OUT == 3) {
        fsc = io.amax[0];
        fok = isfinite(fsc) && fsc != 0.f;
    }
    // pair modes: sequence s = frame * hp + pr covers rows 2 pr, 2 pr + 1 of that frame
    const int hp = (io.rows + 1) / 2 > 0 ? (io.rows + 1) / 2 : 1;
    const int pfr = (int)(s / hp), ppr = (int)(s % hp);
    const size_t prow0 = (size_t)pfr * io.rows + 2 * ppr;     // global index of the pair's first row
    const bool phas_b = 2 * ppr + 1 < io.rows;
    B4D_PM_STAMP(1);
    // ---- radix-P butterflies over n1 (stride M) and the twiddle W_N^{n2 k1}
    for (int n2 = threadIdx.x; n2 < M; n2 += FT) {
        float2 v[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
            const size_t i = s * (size_t)N + (size_t)M * n1 + n2;
            if (IN == 3) {
                int x = M * n1 + n2 - io.px;
                x = x < 0 ? -x : (x >= io.w ? 2 * io.w - 2 - x : x);
                // loads and divisions are unconditional (clamped row) and masked afterwards: a load under a branch would wait
                // for its data before the next one is issued (measured: 16 serial round trips, 7.5 us per row)
                float q[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int row = min(2 * (int)s + e, io.rows - 1);
                    int y = row - io.py;
                    y = y < 0 ? -y : (y >= io.h ? 2 * io.h - 2 - y : y);
                    const float d = io.frame[(size_t)y * io.w + x] / fsc;
                    q[e] = (fok && 2 * (int)s + e < io.rows) ? d : 0.f;
                }
                v[n1] = make_float2(q[0], q[1]);
            } else if (IN == 5) {
                const int idx = M * n1 + n2;
                const float* pa = static_cast<const float*>(xin) + prow0 * N;
                v[n1] = make_float2(pa[idx], phas_b ? pa[N + idx] : 0.f);
            } else if (IN == 4) {
                const int idx = M * n1 + n2, j = idx <= N / 2 ? idx : N - idx;
                const float2* pa = static_cast<const float2*>(xin) + prow0 * io.half;
                const float2 fa = pa[j];
                const float2 fb = phas_b ? pa[io.half + j] : make_float2(0.f, 0.f);
                // Ga + i Gb, Hermitian-extended beyond N/2; then the inverse's input conjugation
                const float2 z = idx <= N / 2 ? make_float2(fa.x - fb.y, fa.y + fb.x) : make_float2(fa.x + fb.y, fb.x - fa.y);
                v[n1] = make_float2(z.x, -z.y);
            } else if (IN == 2) {
                int y = (int)s - io.py, x = M * n1 + n2 - io.px;
                y = y < 0 ? -y : (y >= io.h ? 2 * io.h - 2 - y : y);
                x = x < 0 ? -x : (x >= io.w ? 2 * io.w - 2 - x : x);
                const float d = io.frame[(size_t)y * io.w + x] / fsc;
                v[n1] = make_float2(fok ? d : 0.f, 0.f);
            } else if (IN == 1) {
                v[n1] = make_float2(static_cast<const float*>(xin)[i], 0.f);
            } else {
                const float2 q = static_cast<const float2*>(xin)[i];
                v[n1] = conj_io ? make_float2(q.x, -q.y) : q;
            }
        }
#ifdef B4D_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        B4D_PM_STAMP(6);
#endif
        Dft<P>::run(v);
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) buf0[k1 * M + n2] = k1 == 0 ? v[0] : cmulf(v[k1], twN[n2 * k1]);
        B4D_PM_STAMP(7);
    }
    __syncthreads();
    B4D_PM_STAMP(2);
    // ---- DFT_A over a (n2 = B a + b), then the twiddle W_M^{b c} = W_N^{P b c}
#ifndef B4D_EXP_PM_SKIP23
    if (onebuf) {    // one item per lane: compute into registers, barrier, write back into the same buffer
        const int nCB = Ap / 4, nBB = (B + 1) / 2;
        const bool act = (int)threadIdx.x < P * nCB * nBB;
        const int it = act ? threadIdx.x : 0;
        const int bb = it % nBB, r = it / nBB, c0 = (r % nCB) * 4, k1 = r / nCB;
        const int b0 = 2 * bb, b1 = min(b0 + 1, B - 1);
        v2f acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = v2f{0.f, 0.f};
        if (act) {
            const float2* src = buf0 + k1 * M;
            const float4* wrow = reinterpret_cast<const float4*>(tabA + c0);
            for (int a = 0; a < A; ++a) {
                const v2f x0 = to_v(src[B * a + b0]), x1 = to_v(src[B * a + b1]);
                const float4 wa = wrow[a * (Ap / 2)], wb = wrow[a * (Ap / 2) + 1];
                const v2f w[4] = {v2f{wa.x, wa.y}, v2f{wa.z, wa.w}, v2f{wb.x, wb.y}, v2f{wb.z, wb.w}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = cmac(acc[j][0], x0, w[j]);
                    acc[j][1] = cmac(acc[j][1], x1, w[j]);
                }
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                if (c >= A) continue;
                buf0[k1 * M + c * B + b0] = cmulf(to_f(acc[j][0]), twN[(size_t)P * b0 * c]);
                if (b0 + 1 < B) buf0[k1 * M + c * B + b0 + 1] = cmulf(to_f(acc[j][1]), twN[(size_t)P * (b0 + 1) * c]);
            }
        }
    } else
#ifndef B4D_EXP_PM_PACKED
    if (blocked) {   // column n = (k1, b) = k1 * B + b; rows c
        const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
        const int ncols = P * B;
        for (int strip = threadIdx.x >> 6; strip * 16 < ncols; strip += FT / 64) {
            const int n = strip * 16 + j;
            const bool vn = n < ncols;
            const int k1 = vn ? n / B : 0, b = vn ? n % B : 0;
            float2 tw0[4], tw1[4];   // W_M^{b c} of this lane's outputs: in flight under the matrix products
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * kq + i;
                tw0[i] = twN[(size_t)P * b * min(c, A - 1)];          // unconditional (clamped): loads under a branch serialise
                tw1[i] = twN[(size_t)P * b * min(c + 16, A - 1)];
            }
            const DftTiles t = small_dft_mfma(tabA, A, Ap, vn ? buf0 + k1 * M + b : nullptr, B, lane);
            if (vn) {
                float2* dst = buf1 + k1 * M + b;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * kq + i;
                    if (c < A) dst[c * B] = cmulf(make_float2(t.r0[i], t.i0[i]), tw0[i]);
                    if (c + 16 < A) dst[(c + 16) * B] = cmulf(make_float2(t.r1[i], t.i1[i]), tw1[i]);
                }
            }
        }
    } else
#endif
    if (blocked) {   // item = (k1, 4 outputs c, 2 columns b): per a two x reads and one 4-wide table row feed 8 complex MACs
        const int nCB = Ap / 4, nBB = (B + 1) / 2;
        for (int it = threadIdx.x; it < P * nCB * nBB; it += FT) {
            const int bb = it % nBB, r = it / nBB, c0 = (r % nCB) * 4, k1 = r / nCB;
            const int b0 = 2 * bb, b1 = min(b0 + 1, B - 1);
            v2f acc[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = v2f{0.f, 0.f};
            const float2* src = buf0 + k1 * M;
            const float4* wrow = reinterpret_cast<const float4*>(tabA + c0);
            auto step = [&](int a) {
                const v2f x0 = to_v(src[B * a + b0]), x1 = to_v(src[B * a + b1]);
                const float4 wa = wrow[a * (Ap / 2)], wb = wrow[a * (Ap / 2) + 1];
                const v2f w[4] = {v2f{wa.x, wa.y}, v2f{wa.z, wa.w}, v2f{wb.x, wb.y}, v2f{wb.z, wb.w}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = cmac(acc[j][0], x0, w[j]);
                    acc[j][1] = cmac(acc[j][1], x1, w[j]);
                }
            };
            int a = 0;
            for (; a + 3 < A; a += 4) {   // unrolled by hand: the loads of four steps are in flight together
                step(a);
                step(a + 1);
                step(a + 2);
                step(a + 3);
            }
            for (; a < A; ++a) step(a);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                if (c >= A) continue;
                buf1[k1 * M + c * B + b0] = cmulf(to_f(acc[j][0]), twN[(size_t)P * b0 * c]);
                if (b0 + 1 < B) buf1[k1 * M + c * B + b0 + 1] = cmulf(to_f(acc[j][1]), twN[(size_t)P * (b0 + 1) * c]);
            }
        }
    } else {
        // ---- DFT_A over a for every (k1, b), four outputs c per item, then the twiddle W_M^{b c} = W_N^{P b c}
        const int nCB = (A + 3) / 4;
        for (int it = threadIdx.x; it < P * B * nCB; it += FT) {
            const int b = it % B, r = it / B, c0 = (r % nCB) * 4, k1 = r / nCB;
            float2 acc[4];
            int idx[4], cj[4];
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = make_float2(0.f, 0.f);
                idx[j] = 0;
                cj[j] = (c0 + j) % A;
            }
            const float2* src = buf0 + k1 * M + b;
    #pragma unroll 4
            for (int a = 0; a < A; ++a) {
                const float2 x = src[B * a];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 w = tabA[idx[j]];
                    acc[j].x = fmaf(x.x, w.x, fmaf(-x.y, w.y, acc[j].x));
                    acc[j].y = fmaf(x.x, w.y, fmaf(x.y, w.x, acc[j].y));
                    idx[j] += cj[j];
                    idx[j] -= idx[j] >= A ? A : 0;
                }
            }
    #pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < A) buf1[k1 * M + (c0 + j) * B + b] = cmulf(acc[j], twN[(size_t)P * b * (c0 + j)]);
        }
        }
    __syncthreads();
    B4D_PM_STAMP(3);
    // ---- DFT_B over b (k2 = c + A d) -> natural order k = k1 + P (c + A d) in buf0
    if (onebuf) {
        const int nCP = (A + 1) / 2, nDB = Bp / 4;
        const bool act = (int)threadIdx.x < P * nCP * nDB;
        const int it = act ? threadIdx.x : 0;
        const int cp = it % nCP, r = it / nCP, d0 = (r % nDB) * 4, k1 = r / nDB;
        const int c0 = 2 * cp, c1 = min(c0 + 1, A - 1);
        v2f acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = v2f{0.f, 0.f};
        if (act) {
            const float2* s0 = buf0 + k1 * M + c0 * B;
            const float2* s1 = buf0 + k1 * M + c1 * B;
            const float4* wrow = reinterpret_cast<const float4*>(tabB + d0);
            for (int b = 0; b < B; ++b) {
                const v2f x0 = to_v(s0[b]), x1 = to_v(s1[b]);
                const float4 wa = wrow[b * (Bp / 2)], wb = wrow[b * (Bp / 2) + 1];
                const v2f w[4] = {v2f{wa.x, wa.y}, v2f{wa.z, wa.w}, v2f{wb.x, wb.y}, v2f{wb.z, wb.w}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = cmac(acc[j][0], x0, w[j]);
                    acc[j][1] = cmac(acc[j][1], x1, w[j]);
                }
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = d0 + j;
                if (d >= B) continue;
                buf0[k1 + P * (c0 + A * d)] = to_f(acc[j][0]);
                if (c0 + 1 < A) buf0[k1 + P * (c0 + 1 + A * d)] = to_f(acc[j][1]);
            }
        }
    } else
#ifndef B4D_EXP_PM_PACKED
    if (blocked) {   // column n = (c, k1) = c * P + k1 (natural output order k1 + P (c + A d) = n + P A d); rows d
        const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
        const int ncols = P * A;
        for (int strip = threadIdx.x >> 6; strip * 16 < ncols; strip += FT / 64) {
            const int n = strip * 16 + j;
            const bool vn = n < ncols;
            const int c = vn ? n / P : 0, k1 = vn ? n % P : 0;
            const DftTiles t = small_dft_mfma(tabB, B, Bp, vn ? buf1 + k1 * M + c * B : nullptr, 1, lane);
            if (vn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = 4 * kq + i;
                    if (d < B) buf0[n + P * A * d] = make_float2(t.r0[i], t.i0[i]);
                    if (d + 16 < B) buf0[n + P * A * (d + 16)] = make_float2(t.r1[i], t.i1[i]);
                }
            }
        }
    } else
#endif
    if (blocked) {   // item = (k1, 2 rows c, 4 outputs d)
        const int nCP = (A + 1) / 2, nDB = Bp / 4;
        for (int it = threadIdx.x; it < P * nCP * nDB; it += FT) {
            const int cp = it % nCP, r = it / nCP, d0 = (r % nDB) * 4, k1 = r / nDB;
            const int c0 = 2 * cp, c1 = min(c0 + 1, A - 1);
            v2f acc[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = v2f{0.f, 0.f};
            const float2* s0 = buf1 + k1 * M + c0 * B;
            const float2* s1 = buf1 + k1 * M + c1 * B;
            const float4* wrow = reinterpret_cast<const float4*>(tabB + d0);
            auto step = [&](int b) {
                const v2f x0 = to_v(s0[b]), x1 = to_v(s1[b]);
                const float4 wa = wrow[b * (Bp / 2)], wb = wrow[b * (Bp / 2) + 1];
                const v2f w[4] = {v2f{wa.x, wa.y}, v2f{wa.z, wa.w}, v2f{wb.x, wb.y}, v2f{wb.z, wb.w}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = cmac(acc[j][0], x0, w[j]);
                    acc[j][1] = cmac(acc[j][1], x1, w[j]);
                }
            };
            int b = 0;
            for (; b + 3 < B; b += 4) {
                step(b);
                step(b + 1);
                step(b + 2);
                step(b + 3);
            }
            for (; b < B; ++b) step(b);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = d0 + j;
                if (d >= B) continue;
                buf0[k1 + P * (c0 + A * d)] = to_f(acc[j][0]);
                if (c0 + 1 < A) buf0[k1 + P * (c0 + 1 + A * d)] = to_f(acc[j][1]);
            }
        }
    } else {
        // ---- DFT_B over b for every (k1, c), four outputs d per item -> natural order k = k1 + P (c + A d) in buf0
        const int nDB = (B + 3) / 4;
        for (int it = threadIdx.x; it < P * A * nDB; it += FT) {
            const int c = it % A, r = it / A, d0 = (r % nDB) * 4, k1 = r / nDB;
            float2 acc[4];
            int idx[4], dj[4];
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = make_float2(0.f, 0.f);
                idx[j] = 0;
                dj[j] = (d0 + j) % B;
            }
            const float2* src = buf1 + k1 * M + c * B;
    #pragma unroll 4
            for (int b = 0; b < B; ++b) {
                const float2 x = src[b];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 w = tabB[idx[j]];
                    acc[j].x = fmaf(x.x, w.x, fmaf(-x.y, w.y, acc[j].x));
                    acc[j].y = fmaf(x.x, w.y, fmaf(x.y, w.x, acc[j].y));
                    idx[j] += dj[j];
                    idx[j] -= idx[j] >= B ? B : 0;
                }
            }
    #pragma unroll
            for (int j = 0; j < 4; ++j)
                if (d0 + j < B) buf0[k1 + P * (c + A * (d0 + j))] = acc[j];
        }
        }
    __syncthreads();
#endif
    B4D_PM_STAMP(4);
    if (OUT == 4) {
        const float pk = io.norm_peak ? io.amax[pfr] : 0.f;
        const bool unit = io.norm_peak && pk > 0.f;
        const float se = unit ? 1.0f / pk : scale;
        float* fo = io.crop + (size_t)pfr * io.rows * N;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int y = 2 * ppr + e;
            if (y >= io.rows) continue;
            float* orow = fo + (size_t)((y + io.rows / 2) % io.rows) * N;
            for (int x = threadIdx.x; x < N; x += FT) {
                const float2 z = buf0[x];
                float v = (e == 0 ? z.x : -z.y) * se;     // conj(buf0): real part row a, imaginary part row b
                if (unit && y == 0 && x == 0) v = 1.0f;
                orow[(x + N / 2) % N] = v;
            }
        }
        B4D_PM_STAMP(5);
        return;
    }
    if (OUT == 2) {
        float2* oa = out + prow0 * io.half;
        const bool has_b = phas_b;
        for (int k = threadIdx.x; k < io.half; k += FT) {
            const float2 z = buf0[k], w = buf0[k == 0 ? 0 : N - k];
            oa[k] = make_float2(0.5f * (z.x + w.x), 0.5f * (z.y - w.y));
            if (has_b) oa[io.half + k] = make_float2(0.5f * (z.y + w.y), 0.5f * (w.x - z.x));
        }
        B4D_PM_STAMP(5);
        return;
    }
    if (OUT == 3) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int y = 2 * (int)s + e - io.py;
            if (y < 0 || y >= io.h || 2 * (int)s + e >= io.rows) continue;
            for (int x = threadIdx.x; x < io.w; x += FT) {
                const float2 z = buf0[x + io.px];
                float v = (e == 0 ? z.x : -z.y) * scale;   // conj(buf0): real part row a, imaginary part row b
                if (io.clip) v = (v > 1.f ? 1.f : (v < -1.f ? -1.f : v))   /* np.clip: NaN stays NaN */;
                io.crop[(size_t)y * io.w + x] = fok ? v * fsc : 0.f;
            }
        }
        B4D_PM_STAMP(5);
        return;
    }
    if (OUT == 1) {
        const int y = (int)s - io.py;
        if (y < 0 || y >= io.h) return;
        for (int x = threadIdx.x; x < io.w; x += FT) {
            float v = buf0[x + io.px].x * scale;   // conj_io only flips the imaginary part
            if (io.clip) v = (v > 1.f ? 1.f : (v < -1.f ? -1.f : v))   /* np.clip: NaN stays NaN */;
            io.crop[(size_t)y * io.w + x] = fok ? v * fsc : 0.f;
        }
        B4D_PM_STAMP(5);
        return;
    }
    for (int k = threadIdx.x; k < N; k += FT) {
        float2 v = buf0[k];
        if (filt) v = cmulf(v, filt[(io.filt_bcast ? 0 : s * (size_t)N) + k]);
        if (conj_io) v.y = -v.y;
        out[s * (size_t)N + k] = make_float2(v.x * scale, v.y * scale);
    }
    B4D_PM_STAMP(5);
}

// amax[0] = max of the `nparts` partial maxima (one wave)
__global__ void __launch_bounds__(64) k_absmax_final(float* __restrict__ amax, int nparts) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) m = fmaxf(m, amax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if (threadIdx.x == 0) amax[0] = m;
}

// step C: out[s*N + k1 + P k2] = z[(s*P + k1)*M + k2]  (* filt[same index], conj_out, * scale).  grid (ceil(N/256), S)
__global__ void __launch_bounds__(256) k_pm_post(const float2* __restrict__ z, float2* __restrict__ out, int P, int M,
                                                 const float2* __restrict__ filt, int conj_out, float scale) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = P * M;
    if (k >= N) return;
    const size_t s = blockIdx.y;
    const int k1 = k % P, k2 = k / P;
    float2 v = z[(s * P + k1) * (size_t)M + k2];
    if (filt) v = cmulf(v, filt[s * (size_t)N + k]);
    if (conj_out) v.y = -v.y;
    out[s * (size_t)N + k] = make_float2(v.x * scale, v.y * scale);
}

// 32 x 32 LDS-tiled transpose of a (rows, cols) complex array.  grid (ceil(cols/32), ceil(rows/32)), block (32, 8)
__global__ void __launch_bounds__(256) k_transpose_c(const float2* __restrict__ in, float2* __restrict__ out, int rows, int cols) {
    __shared__ float2 t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    in += (size_t)blockIdx.z * rows * cols;   // grid.z = batch of equally shaped matrices
    out += (size_t)blockIdx.z * rows * cols;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < rows && c < cols) t[i][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = t[threadIdx.x][i];
    }
}

// np.pad(frame, ((py,py),(px,px)), mode="reflect") / scale  (filters.py:252-261); scale = max|frame| read from `amax`
__global__ void __launch_bounds__(256) k_pad_reflect(const float* __restrict__ frame, int h, int w, int py, int px,
                                                     const float* __restrict__ amax, int nparts, float* __restrict__ out) {
    const float s_scale = amax[0];   // reduced by k_absmax_final
    const int H = h + 2 * py, W = w + 2 * px;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)H * W) return;
    int y = (int)(e / W) - py, x = (int)(e % W) - px;
    y = y < 0 ? -y : (y >= h ? 2 * h - 2 - y : y);
    x = x < 0 ? -x : (x >= w ? 2 * w - 2 - x : x);
    const float sc = s_scale;
    out[e] = (isfinite(sc) && sc != 0.f) ? frame[(size_t)y * w + x] / sc : 0.f;
}

// restored = clip(Re(z), -1, 1) * scale, cropped back to (h, w)  (filters.py:266, 287-289)
__global__ void __launch_bounds__(256) k_crop_out(const float2* __restrict__ z, int h, int w, int py, int px,
                                                  const float* __restrict__ amax, int nparts, int clip, float* __restrict__ out) {
    const float s_scale = amax[0];
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)h * w) return;
    const int y = (int)(e / w), x = (int)(e % w), W = w + 2 * px;
    float v = z[(size_t)(y + py) * W + x + px].x;
    if (clip) v = (v > 1.f ? 1.f : (v < -1.f ? -1.f : v))   /* np.clip: NaN stays NaN */;
    const float sc = s_scale;
    out[e] = (isfinite(sc) && sc != 0.f) ? v * sc : 0.f;
}

// max|x| partials, NaN-ignoring (np.nanmax(np.abs(padded)))
__global__ void __launch_bounds__(1024) k_nanabsmax(const float* __restrict__ x, size_t n, float* __restrict__ part) {
    __shared__ float sh[16];
    float m = 0.f;
    auto take = [&](float v) {
        const float a = fabsf(v);
        if (a == a) m = fmaxf(m, a);
    };
    // 16-byte loads, four per lane in flight (a frame is one stream: the scalar grid-stride loop was latency-bound)
    const size_t n4 = ((reinterpret_cast<size_t>(x) & 15) == 0) ? n / 4 : 0, stride = (size_t)gridDim.x * blockDim.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        take(a.x); take(a.y); take(a.z); take(a.w);
        take(b.x); take(b.y); take(b.z); take(b.w);
        take(c.x); take(c.y); take(c.z); take(c.w);
        take(d.x); take(d.y); take(d.z); take(d.w);
    }
    for (; i < n4; i += stride) {
        const float4 a = x4[i];
        take(a.x); take(a.y); take(a.z); take(a.w);
    }
    for (size_t j = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) take(x[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, sh[i]);
        part[blockIdx.x] = m;
    }
}

// W^T = conj(H) / (|H|^2 + balance |L|^2) from the (transposed) transfer functions of the PSF and the Laplacian
__global__ void __launch_bounds__(256) k_wiener_filter(const float2* __restrict__ Hf, const float2* __restrict__ Lf, size_t n,
                                                       float balance, float2* __restrict__ Wf) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float2 h = Hf[e], l = Lf[e];
    const float den = (h.x * h.x + h.y * h.y) + balance * (l.x * l.x + l.y * l.y);
    Wf[e] = make_float2(h.x / den, -h.y / den);
}

// ---- Richardson-Lucy (skimage.restoration.richardson_lucy as published; filters.py:270-277)
// One step of the iteration on a (H, W) float32 image with a (ky, kx) kernel, "same" convolution, zero boundary:
//   MODE 0: dst = image / (conv(src, psf) + 1e-12)          (0 where conv < feps when feps > 0)
//   MODE 1: dst = dst * conv(src, flip(psf))
// 64 x 32 output tile per workgroup, source tile + halo and the kernel staged in LDS.  grid (ceil(W/64), ceil(H/32))
constexpr int RL_TX = 64, RL_TY = 32, RL_MAXK = 33;
template <int MODE>
__global__ void __launch_bounds__(256) k_rl_step(const float* __restrict__ src, const float* __restrict__ image, float* __restrict__ dst,
                                                 const float* __restrict__ psf, int H, int W, int ky, int kx, float feps) {
    extern __shared__ float rl_sm[];
    const int hy = ky / 2, hx = kx / 2, tw = RL_TX + 2 * hx, th = RL_TY + 2 * hy;
    float* tile = rl_sm;              // th x tw
    float* kk = rl_sm + th * tw;      // ky x kx, already oriented for a correlation-style inner loop
    const int x0 = blockIdx.x * RL_TX, y0 = blockIdx.y * RL_TY;
    for (int i = threadIdx.x; i < th * tw; i += 256) {
        const int y = y0 - hy + i / tw, x = x0 - hx + i % tw;
        tile[i] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f;
    }
    // conv(f, k)[y, x] = sum_{i, j} f[y + hy - i, x + hx - j] k[i, j] = sum_{p, q} tile[ty + p, tx + q] k[ky-1-p, kx-1-q];
    // MODE 1 convolves with the flipped kernel: the two flips cancel
    for (int i = threadIdx.x; i < ky * kx; i += 256) kk[i] = MODE == 0 ? psf[ky * kx - 1 - i] : psf[i];
    __syncthreads();
    const int tx = threadIdx.x % RL_TX, tyb = threadIdx.x / RL_TX;   // 4 rows of 64 lanes; each lane 8 output rows
#pragma unroll 1
    for (int r = 0; r < RL_TY / 4; ++r) {
        const int ty = tyb + 4 * r, y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        float acc = 0.f;
        for (int p = 0; p < ky; ++p) {
            const float* row = tile + (ty + p) * tw + tx;
            const float* krow = kk + p * kx;
            for (int q = 0; q < kx; ++q) acc = fmaf(row[q], krow[q], acc);
        }
        const size_t o = (size_t)y * W + x;
        if (MODE == 0) {
            const float conv = acc + 1e-12f;
            dst[o] = (feps > 0.f && conv < feps) ? 0.f : image[o] / conv;
        } else {
            dst[o] = dst[o] * acc;
        }
    }
}

__global__ void __launch_bounds__(256) k_fill(float* __restrict__ x, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// restored = clip(est, -1, 1) * scale cropped back to (h, w)
__global__ void __launch_bounds__(256) k_rl_crop(const float* __restrict__ est, int h, int w, int py, int px, const float* __restrict__ amax,
                                                 int clip, float* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)h * w) return;
    const int y = (int)(e / w), x = (int)(e % w), W = w + 2 * px;
    float v = est[(size_t)(y + py) * W + x + px];
    if (clip) v = (v > 1.f ? 1.f : (v < -1.f ? -1.f : v))   /* np.clip: NaN stays NaN */;
    const float sc = amax[0];
    out[e] = (isfinite(sc) && sc != 0.f) ? v * sc : 0.f;
}

}  // namespace b4d

using namespace b4d;

#ifndef B4D_WIENER_LANES
#define B4D_WIENER_LANES 2
#endif
struct b4d_wiener {
    std::recursive_mutex mu;     // host-side re-entrancy (several host threads, one plan)
    int h, w, py, px, H, W;      // frame, half kernel, padded sizes
    int Px, Mx, Py, My;          // H = Py*My, W = Px*Mx
    int Ax = 0, Bx = 0, Ay = 0, By = 0;  // Mx = Ax*Bx, My = Ay*By when the fused LDS transform applies (0: DFT-matrix product)
    float2* twx = nullptr;       // W-point twiddles
    float2* twy = nullptr;
    float2* dmx = nullptr;       // Mx x Mx DFT matrix
    float2* dmy = nullptr;
    float2* filt = nullptr;      // transposed Wiener filter (W, H)
    float2* a = nullptr;         // (H*W) work buffers
    float2* b = nullptr;
    float2* c = nullptr;
    float* padded = nullptr;     // (H, W) real
    float* amax = nullptr;       // 256 partial maxima
    // further sets of work buffers + internal streams (lazily created by the first multi-frame call): consecutive frames
    // run on alternate streams, so one frame's row transforms fill the chip while another's last partial round of
    // workgroups (2052 rows on 512 resident workgroups: 4 full rounds + 4 stragglers) and its HBM bursts drain
    struct Lane {
        float2* a = nullptr;
        float2* b = nullptr;
        float2* c = nullptr;
        float* padded = nullptr;
        float* amax = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;
    };
    Lane lane[B4D_WIENER_LANES];   // lane[0] aliases a, b, c, padded, amax
    hipEvent_t fork = nullptr;
    // mixed-radix route (b4d_wiener_mr.hip): both padded sides have a compiled three-radix kernel
    bool mr = false;
    WmrGeom geom{};
    float2* mr_filt = nullptr;   // (Wh, Hp) transposed half filter (general PSF)
    float2* mr_sepx = nullptr;   // separable, point-symmetric PSF (the Gaussian of deconvolve_psf): {hx, lx}[Wh] and {hy, ly}[H] instead,
    float2* mr_sepy = nullptr;   // the filter is rebuilt per element in the column pass (b4d_wiener_mr.hip: WmrSep)
    float mr_balance = 0.f;
    float2* mr_T = nullptr;      // (mr_cap, Wh, Hp) transposed half spectra of the frames of one launch
    float* mr_amax = nullptr;    // (mr_cap * (hp + 1)): max|frame| per frame, then the pair maxima
    int mr_cap = 0, mr_slots = 0;   // frames per slot, slots (one per lane of wiener_mr_apply)
    static constexpr int MR_LANES = 2;
    hipStream_t mr_aux[MR_LANES] = {};      // lane 1: the library's shared lane_stream(0) (lane 0 is the caller's stream)
    hipEvent_t mr_fork = nullptr, mr_join[MR_LANES] = {};
};

// frames per launch of the mixed-radix route on ONE lane: the persistent row kernels hand 513 quads per 4k frame to 256 workgroups
// (2.004 rounds), so a launch needs several frames to amortise its last, nearly empty round (measured on MI355X,
// 4096^2: 1 frame per launch 5.6 k frames/s, 4: 6.8 k, 8: 7.15 k, 16: 7.2 k); 68 MB of workspace per frame.  On two lanes
// (wiener_mr_apply) the other lane's kernels fill that round and ONE frame per group (64 MiB of workspace) is best: it stays in the
// memory-side cache between the passes (same box, 64 x 4096^2: one lane x 8 frames 7.65-7.78 k frames/s; two lanes x 8: 7.9-8.1 k,
// x 4: 8.1 k, x 2: 7.8-8.0 k, x 1: 8.1-8.4 k; three lanes x 1: 7.4-7.6 k, four: 6.9 k)
#ifndef B4D_WIENER_FPL
#define B4D_WIENER_FPL 8
#endif
static int wiener_fpl(bool lanes, size_t t_bytes) {   // t_bytes: transposed half spectrum of one frame
    static const int v = [] {
        const char* e = getenv("B4D_WIENER_FPL");   // tuning aid (tools/dev_cfg5.py); results do not depend on it
        const int n = e ? atoi(e) : 0;
        return n >= 1 && n <= 64 ? n : 0;
    }();
    return v ? v : lanes ? (int)std::min<size_t>(64, std::max<size_t>(1, ((size_t)64 << 20) / t_bytes)) : B4D_WIENER_FPL;
}

static void split_pm(int n, int* P, int* M) {
    int p = 1;
    while (p < 16 && n % (2 * p) == 0) p *= 2;
    *P = p;
    *M = n / p;
}

// M = A * B with the smallest A + B; fused LDS path when the two small DFTs are cheap and the row fits in LDS
static void split_ab(int P, int M, int* A, int* B) {
    int best = 1;
    for (int f = 1; (long long)f * f <= M; ++f)
        if (M % f == 0) best = f;
    const int a = M / best, b = best;
    const size_t lds = sizeof(float2) * pm_lds_elems(P, a, b);
    if (a + b <= 320 && lds <= 150 * 1024) {   // beyond ~300 complex MACs per element the chirp-z / DFT-matrix routes win
        *A = a;
        *B = b;
    } else {
        *A = *B = 0;
    }
}

template <int P, int IN, int OUT>
static int pm_fused_launch2(const void* x, float2* out, const float2* tw, int A, int B, int S, const float2* filt, int conj_io, float scale,
                            const FusedIO& io, hipStream_t st) {
    const bool onebuf = pm_onebuf(P, A, B);
    const size_t lds = sizeof(float2) * pm_lds_elems(P, A, B, onebuf);
    if (int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(onebuf ? &k_pm_fused<P, IN, OUT, true> : &k_pm_fused<P, IN, OUT, false>),
                                        150 * 1024))
        return rc_lds;
    if (onebuf)
        hipLaunchKernelGGL((k_pm_fused<P, IN, OUT, true>), dim3(S), dim3(FT_ONEBUF), lds, st, x, out, tw, A, B, filt, conj_io, scale, io);
    else
        hipLaunchKernelGGL((k_pm_fused<P, IN, OUT, false>), dim3(S), dim3(FT_MAX), lds, st, x, out, tw, A, B, filt, conj_io, scale, io);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// in_mode: 0 complex, 1 real, 2 reflect-padded frame (io.frame); out_mode: 0 complex, 1 cropped real (io.crop)
template <int P>
static int pm_fused_launch(const void* x, int in_mode, int out_mode, float2* out, const float2* tw, int A, int B, int S, const float2* filt,
                           int conj_io, float scale, const FusedIO& io, hipStream_t st) {
    if (in_mode == 3) return pm_fused_launch2<P, 3, 2>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    if (in_mode == 5) return pm_fused_launch2<P, 5, 2>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    if (in_mode == 4 && out_mode == 4) return pm_fused_launch2<P, 4, 4>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    if (in_mode == 4) return pm_fused_launch2<P, 4, 3>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    if (out_mode == 1) return pm_fused_launch2<P, 0, 1>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    if (in_mode == 2) return pm_fused_launch2<P, 2, 0>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    if (in_mode == 1) return pm_fused_launch2<P, 1, 0>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
    return pm_fused_launch2<P, 0, 0>(x, out, tw, A, B, S, filt, conj_io, scale, io, st);
}

template <int P>
static int pm_pre_launch(const void* x, bool real_in, float2* y, const float2* tw, int M, int S, int conj_in, hipStream_t st) {
    const dim3 grid((M + 255) / 256, S);
    if (real_in)
        hipLaunchKernelGGL((k_pm_pre<P, true>), grid, dim3(256), 0, st, x, y, tw, M, conj_in);
    else
        hipLaunchKernelGGL((k_pm_pre<P, false>), grid, dim3(256), 0, st, x, y, tw, M, conj_in);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// S sequences of length N = P*M, contiguous: out = DFT(in) (forward) or conj(DFT(conj(in))) * scale (inverse);
// tmp holds S*N complex values; optional pointwise multiplier applied to the forward output.
static int dft_rows(const void* in, bool real_in, float2* tmp, float2* tmp2, float2* out, int S, int P, int M, const float2* tw,
                    const float2* dm, bool inverse, const float2* filt, float scale, hipStream_t st, int A = 0, int B = 0,
                    const FusedIO* fio = nullptr, int in_mode = -1, int out_mode = 0) {
    int rc;
    if (A > 0) {  // fused LDS transform (M = A * B)
        const FusedIO io = fio ? *fio : FusedIO{};
        const int im = in_mode >= 0 ? in_mode : (real_in ? 1 : 0);
        switch (P) {
            case 1: return pm_fused_launch<1>(in, im, out_mode, out, tw, A, B, S, filt, inverse, scale, io, st);
            case 2: return pm_fused_launch<2>(in, im, out_mode, out, tw, A, B, S, filt, inverse, scale, io, st);
            case 4: return pm_fused_launch<4>(in, im, out_mode, out, tw, A, B, S, filt, inverse, scale, io, st);
            case 8: return pm_fused_launch<8>(in, im, out_mode, out, tw, A, B, S, filt, inverse, scale, io, st);
            case 16: return pm_fused_launch<16>(in, im, out_mode, out, tw, A, B, S, filt, inverse, scale, io, st);
            default: return fail(B4D_ESIZE, "unsupported radix");
        }
    }
    switch (P) {
        case 1: rc = pm_pre_launch<1>(in, real_in, tmp, tw, M, S, inverse, st); break;
        case 2: rc = pm_pre_launch<2>(in, real_in, tmp, tw, M, S, inverse, st); break;
        case 4: rc = pm_pre_launch<4>(in, real_in, tmp, tw, M, S, inverse, st); break;
        case 8: rc = pm_pre_launch<8>(in, real_in, tmp, tw, M, S, inverse, st); break;
        case 16: rc = pm_pre_launch<16>(in, real_in, tmp, tw, M, S, inverse, st); break;
        default: return fail(B4D_ESIZE, "unsupported radix");
    }
    if (rc) return rc;
    // (S*P, M) x (M, M); rows are independent, so slice the batch to keep the launch grid.y within limits
    const long long rows = (long long)S * P;
    if ((rc = b4d_cgemm(tmp, false, 0, 0, dm, false, 0, 0, tmp2, 0, (int)rows, M, M, 1, st))) return rc;
    hipLaunchKernelGGL(k_pm_post, dim3((P * M + 255) / 256, S), dim3(256), 0, st, tmp2, out, P, M, filt, inverse ? 1 : 0, scale);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

static int transpose_c(const float2* in, float2* out, int rows, int cols, hipStream_t st, int batch = 1) {
    hipLaunchKernelGGL(k_transpose_c, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(32, 8), 0, st, in, out, rows, cols);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// ---- exported to b4d_general.hip: the fused row transform as a general-length 1-D engine
namespace b4d {
bool pm_fusable(int n) {
    int P, M, A, B;
    split_pm(n, &P, &M);
    split_ab(P, M, &A, &B);
    return A > 0;
}

// ---- Bluestein (chirp-z) for lengths without a small-factor split (e.g. 2056 = 8 * 257): with c[n] = exp(-i pi n^2 / N),
//   X[k] = c[k] * sum_n (x[n] c[n]) conj(c[k - n]),
// a convolution carried by two fused power-of-two transforms of length L >= 2 N - 1 and a pointwise product with the
// precomputed spectrum of the chirp.  Per-length tables are cached for the life of the process.
__global__ void __launch_bounds__(256) k_blue_pre(const void* __restrict__ xin, int real_in, int conj_in, int N, int L,
                                                  const float2* __restrict__ chirp, float2* __restrict__ a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= L) return;
    const size_t s = blockIdx.y;
    float2 v = make_float2(0.f, 0.f);
    if (n < N) {
        float2 x;
        if (real_in) {
            x = make_float2(static_cast<const float*>(xin)[s * N + n], 0.f);
        } else {
            x = static_cast<const float2*>(xin)[s * N + n];
            if (conj_in) x.y = -x.y;
        }
        v = cmulf(x, chirp[n]);
    }
    a[s * (size_t)L + n] = v;
}

__global__ void __launch_bounds__(256) k_blue_post(const float2* __restrict__ c, int N, int L, const float2* __restrict__ chirp,
                                                   int conj_out, float scale, float2* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const size_t s = blockIdx.y;
    float2 v = cmulf(c[s * (size_t)L + k], chirp[k]);
    if (conj_out) v.y = -v.y;
    out[s * (size_t)N + k] = make_float2(v.x * scale, v.y * scale);
}

namespace {
struct BluePlan {
    int n = 0, L = 0;
    float2* chirp = nullptr;   // c[n] = exp(-i pi n^2 / N), n < N
    float2* bspec = nullptr;   // FFT_L of b[m] = conj(c[|m|]) wrapped to length L
    float2* twL = nullptr;     // L-point twiddles
};
std::mutex g_blue_mu;
std::vector<BluePlan> g_blue_plans;
void* g_blue_ws = nullptr;
size_t g_blue_ws_bytes = 0;

int blue_len(int n) {
    int L = 64;
    while (L < 2 * n - 1) L *= 2;
    return L;
}

int blue_plan(int n, hipStream_t st, BluePlan* out) {
    for (const BluePlan& p : g_blue_plans)
        if (p.n == n) {
            *out = p;
            return B4D_OK;
        }
    BluePlan p;
    p.n = n;
    p.L = blue_len(n);
    std::vector<float2> c(n), b(p.L, make_float2(0.f, 0.f));
    for (int k = 0; k < n; ++k) {
        const long long q = ((long long)k * k) % (2LL * n);        // k^2 mod 2N keeps the phase exact
        const double a = -M_PI * (double)q / (double)n;
        c[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        const float2 bk = make_float2(c[k].x, -c[k].y);
        b[k] = bk;
        if (k) b[p.L - k] = bk;
    }
    int rc = make_twiddles(p.L, &p.twL);
    if (rc) return rc;
    B4D_HIP(hipMalloc((void**)&p.chirp, sizeof(float2) * n));
    B4D_HIP(hipMalloc((void**)&p.bspec, sizeof(float2) * p.L));
    B4D_HIP(hipMemcpy(p.chirp, c.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    B4D_HIP(hipMemcpy(p.bspec, b.data(), sizeof(float2) * p.L, hipMemcpyHostToDevice));
    int P, M, A, B;
    split_pm(p.L, &P, &M);
    split_ab(P, M, &A, &B);
    if (A <= 0) return fail(B4D_ESIZE, "Bluestein length has no fused split");
    if ((rc = dft_rows(p.bspec, false, nullptr, nullptr, p.bspec, 1, P, M, p.twL, nullptr, false, nullptr, 1.f, st, A, B))) return rc;
    B4D_HIP(hipStreamSynchronize(st));
    g_blue_plans.push_back(p);
    *out = p;
    return B4D_OK;
}

int blue_rows(const void* in, bool real_in, float2* out, int S, int n, bool inverse, float scale, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_blue_mu);
    BluePlan bp;
    int rc = blue_plan(n, st, &bp);
    if (rc) return rc;
    const size_t need = sizeof(float2) * (size_t)S * bp.L;
    if (need > g_blue_ws_bytes) {
        if (g_blue_ws) (void)hipFree(g_blue_ws);
        g_blue_ws = nullptr;
        g_blue_ws_bytes = 0;
        hipError_t e = hipMalloc(&g_blue_ws, need);
        if (e != hipSuccess) return fail(B4D_ENOMEM, std::string("Bluestein workspace: ") + hipGetErrorString(e));
        g_blue_ws_bytes = need;
    }
    float2* a = static_cast<float2*>(g_blue_ws);
    int P, M, A, B;
    split_pm(bp.L, &P, &M);
    split_ab(P, M, &A, &B);
    hipLaunchKernelGGL(k_blue_pre, dim3((bp.L + 255) / 256, S), dim3(256), 0, st, in, real_in ? 1 : 0, inverse ? 1 : 0, n, bp.L, bp.chirp, a);
    B4D_HIP(hipGetLastError());
    FusedIO io{};
    io.filt_bcast = 1;
    if ((rc = dft_rows(a, false, nullptr, nullptr, a, S, P, M, bp.twL, nullptr, false, bp.bspec, 1.f, st, A, B, &io))) return rc;
    if ((rc = dft_rows(a, false, nullptr, nullptr, a, S, P, M, bp.twL, nullptr, true, nullptr, 1.0f / (float)bp.L, st, A, B))) return rc;
    hipLaunchKernelGGL(k_blue_post, dim3((n + 255) / 256, S), dim3(256), 0, st, a, n, bp.L, bp.chirp, inverse ? 1 : 0, scale, out);
    B4D_HIP(hipGetLastError());
    B4D_HIP(hipStreamSynchronize(st));   // the shared workspace is reused by the next call
    return B4D_OK;
}
}  // namespace

bool pm_supported(int n) { return n >= 2 && (pm_fusable(n) || n <= 4096); }

// S contiguous sequences of length n: out = DFT(in), or conj(DFT(conj(in))) * scale when inverse; tw: n-point twiddles
int pm_rows(const void* in, bool real_in, float2* out, int S, int n, const float2* tw, bool inverse, float scale, hipStream_t st) {
    int P, M, A, B;
    split_pm(n, &P, &M);
    split_ab(P, M, &A, &B);
    if (A <= 0) {
        if (n <= 4096) return blue_rows(in, real_in, out, S, n, inverse, scale, st);
        return fail(B4D_ESIZE, "length " + std::to_string(n) + " has no P * A * B split that fits the fused transform");
    }
    return dft_rows(in, real_in, nullptr, nullptr, out, S, P, M, tw, nullptr, inverse, nullptr, scale, st, A, B);
}
// Real rows in pairs (SURVEY's R2C / C2R passes for general lengths; n must have a fused split):
//   forward: (frames, rows, n) float -> (frames, rows, n/2 + 1) half spectra
//   inverse: half rows -> (frames, rows, n) float, fftshift-ed in both axes, scaled by `scale` or, with `peak`
//            (device, one unscaled zero-lag value per frame), divided by it with the zero lag forced to 1
int pm_rows_pair_fwd(const float* in, float2* half_out, int frames, int rows, int n, const float2* tw, hipStream_t st) {
    int P, M, A, B;
    split_pm(n, &P, &M);
    split_ab(P, M, &A, &B);
    if (A <= 0) return fail(B4D_ESIZE, "pair transform needs a fused split");
    FusedIO io{};
    io.half = n / 2 + 1;
    io.rows = rows;
    return dft_rows(in, true, nullptr, nullptr, half_out, frames * ((rows + 1) / 2), P, M, tw, nullptr, false, nullptr, 1.f, st, A, B, &io, 5, 0);
}
int pm_rows_pair_inv(const float2* half_in, float* real_out, int frames, int rows, int n, const float2* tw, float scale, const float* peak,
                     hipStream_t st) {
    int P, M, A, B;
    split_pm(n, &P, &M);
    split_ab(P, M, &A, &B);
    if (A <= 0) return fail(B4D_ESIZE, "pair transform needs a fused split");
    FusedIO io{};
    io.half = n / 2 + 1;
    io.rows = rows;
    io.crop = real_out;
    io.amax = peak;
    io.norm_peak = peak ? 1 : 0;
    return dft_rows(half_in, false, nullptr, nullptr, nullptr, frames * ((rows + 1) / 2), P, M, tw, nullptr, true, nullptr, scale, st, A, B, &io, 4, 4);
}
int transpose_batch(const float2* in, float2* out, int rows, int cols, int batch, hipStream_t st) {
    return transpose_c(in, out, rows, cols, st, batch);
}
}  // namespace b4d

// forward 2-D DFT of a real (H, W) array -> TRANSPOSED spectrum (W, H), left in pl->a (b, c are scratch).
// dft_rows needs in != tmp != tmp2 != out (its first and last steps are permutations).
static int fft2_real_T(b4d_wiener* pl, const float* x, hipStream_t st) {
    int rc = dft_rows(x, true, pl->a, pl->b, pl->c, pl->H, pl->Px, pl->Mx, pl->twx, pl->dmx, false, nullptr, 1.f, st, pl->Ax, pl->Bx);
    if (rc) return rc;
    if ((rc = transpose_c(pl->c, pl->a, pl->H, pl->W, st))) return rc;
    return dft_rows(pl->a, false, pl->b, pl->c, pl->a, pl->W, pl->Py, pl->My, pl->twy, pl->dmy, false, nullptr, 1.f, st, pl->Ay, pl->By);
}

extern "C" {

int b4d_wiener_destroy(b4d_wiener* p) {
    if (!p) return B4D_OK;
    for (void* q : {(void*)p->twx, (void*)p->twy, (void*)p->dmx, (void*)p->dmy, (void*)p->filt, (void*)p->a, (void*)p->b, (void*)p->c,
                    (void*)p->padded, (void*)p->amax, (void*)p->mr_filt, (void*)p->mr_T, (void*)p->mr_amax, (void*)p->mr_sepx, (void*)p->mr_sepy})
        if (q) (void)hipFree(q);
    for (int l = 0; l < B4D_WIENER_LANES; ++l) {
        b4d_wiener::Lane& L = p->lane[l];
        if (l > 0)
            for (void* q : {(void*)L.a, (void*)L.b, (void*)L.c, (void*)L.padded, (void*)L.amax})
                if (q) (void)hipFree(q);
        if (L.st) (void)hipStreamSynchronize(L.st);   // shared lane_stream(l): not destroyed with the plan
        if (L.done) (void)hipEventDestroy(L.done);
    }
    if (p->fork) (void)hipEventDestroy(p->fork);
    for (int l = 1; l < b4d_wiener::MR_LANES; ++l) {
        if (p->mr_aux[l]) (void)hipStreamSynchronize(p->mr_aux[l]);
        if (p->mr_join[l]) (void)hipEventDestroy(p->mr_join[l]);
    }
    if (p->mr_fork) (void)hipEventDestroy(p->mr_fork);
    delete p;
    return B4D_OK;
}

int b4d_wiener_create(int h, int w, const float* psf_host, int ky, int kx, float balance, b4d_wiener** out) {
    if (!out || !psf_host) return fail(B4D_EINVAL, "null argument");
    *out = nullptr;
    if (h < 2 || w < 2 || ky < 1 || kx < 1 || !(ky & 1) || !(kx & 1)) return fail(B4D_EINVAL, "bad frame or kernel shape");
    if (ky / 2 >= h || kx / 2 >= w) return fail(B4D_EINVAL, "kernel larger than the frame");
    b4d_wiener* p = new b4d_wiener();
    p->h = h;
    p->w = w;
    p->py = ky / 2;
    p->px = kx / 2;
    p->H = h + 2 * p->py;
    p->W = w + 2 * p->px;
    split_pm(p->W, &p->Px, &p->Mx);
    split_pm(p->H, &p->Py, &p->My);
    split_ab(p->Px, p->Mx, &p->Ax, &p->Bx);
    split_ab(p->Py, p->My, &p->Ay, &p->By);
    if (p->Mx > 4200 || p->My > 4200) {
        b4d_wiener_destroy(p);
        return fail(B4D_ESIZE, "padded size " + std::to_string(p->H) + "x" + std::to_string(p->W) + " has an odd factor > 4200");
    }
    int rc = make_twiddles(p->W, &p->twx);
    if (rc == B4D_OK) rc = make_twiddles(p->H, &p->twy);
    if (rc == B4D_OK && !p->Ax) rc = make_dft_matrix(p->Mx, &p->dmx);
    if (rc == B4D_OK && !p->Ay) rc = make_dft_matrix(p->My, &p->dmy);
    const size_t n = (size_t)p->H * p->W;
    hipError_t e = hipSuccess;
    if (rc == B4D_OK) {
        e = hipMalloc((void**)&p->filt, sizeof(float2) * n);
        if (e == hipSuccess) e = hipMalloc((void**)&p->a, sizeof(float2) * n);
        if (e == hipSuccess) e = hipMalloc((void**)&p->b, sizeof(float2) * n);
        if (e == hipSuccess) e = hipMalloc((void**)&p->c, sizeof(float2) * n);
        if (e == hipSuccess) e = hipMalloc((void**)&p->padded, sizeof(float) * n);
        if (e == hipSuccess) e = hipMalloc((void**)&p->amax, sizeof(float) * 256);
        if (e != hipSuccess) rc = fail(B4D_ENOMEM, std::string("wiener workspace: ") + hipGetErrorString(e));
    }
    if (rc != B4D_OK) {
        b4d_wiener_destroy(p);
        return rc;
    }
    // transfer functions (published skimage.restoration.uft.ir2tf): kernel in the top-left corner of a zero (H, W)
    // array, every axis rolled by -floor(size/2), 2-D DFT.  Built on the host, transformed on the device.
    auto impulse = [&](const float* k, int kh, int kw, std::vector<float>& img) {
        img.assign(n, 0.f);
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
                const int y = ((i - kh / 2) % p->H + p->H) % p->H, x = ((j - kw / 2) % p->W + p->W) % p->W;
                img[(size_t)y * p->W + x] = k[i * kw + j];
            }
    };
    std::vector<float> img;
    float2* Hf = nullptr;
    e = hipMalloc((void**)&Hf, sizeof(float2) * n);
    if (e != hipSuccess) {
        b4d_wiener_destroy(p);
        return fail(B4D_ENOMEM, "wiener setup allocation");
    }
    hipStream_t st = nullptr;
    impulse(psf_host, ky, kx, img);
    rc = (hipMemcpy(p->padded, img.data(), sizeof(float) * n, hipMemcpyHostToDevice) == hipSuccess) ? B4D_OK : fail(B4D_EHIP, "memcpy");
    if (rc == B4D_OK) rc = fft2_real_T(p, p->padded, st);
    if (rc == B4D_OK && hipMemcpyAsync(Hf, p->a, sizeof(float2) * n, hipMemcpyDeviceToDevice, st) != hipSuccess)
        rc = fail(B4D_EHIP, "memcpy");
    const float lap[9] = {0.f, -1.f, 0.f, -1.f, 4.f, -1.f, 0.f, -1.f, 0.f};
    if (rc == B4D_OK) {
        impulse(lap, 3, 3, img);
        rc = (hipMemcpy(p->padded, img.data(), sizeof(float) * n, hipMemcpyHostToDevice) == hipSuccess) ? B4D_OK : fail(B4D_EHIP, "memcpy");
    }
    if (rc == B4D_OK) rc = fft2_real_T(p, p->padded, st);  // Laplacian transfer function in p->a
    if (rc == B4D_OK) {
        hipLaunchKernelGGL(k_wiener_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Hf, p->a, n, balance, p->filt);
        if (hipDeviceSynchronize() != hipSuccess) rc = fail(B4D_EHIP, "wiener filter setup failed");
    }
    (void)hipFree(Hf);
    if (rc == B4D_OK && wmr_supported(p->H) && wmr_supported(p->W)) {
        // mixed-radix route: keep the Wh = W/2 + 1 independent filter columns at the workspace pitch, drop the work
        // buffers of the general route (only the set-up transforms above needed them)
        WmrGeom& g = p->geom;
        g.h = h, g.w = w, g.py = p->py, g.px = p->px, g.H = p->H, g.W = p->W;
        g.Wh = p->W / 2 + 1;
        g.Hp = wmr_pitch(p->H);
        g.hp = (p->H + 1) / 2;
        g.clip = 0;
        g.inv = 1.0f / ((float)p->H * (float)p->W);
        // Separable (rank one) and point-symmetric kernel?  psf = u v^T / s with u = row sums, v = column sums, s = total; then the
        // transfer function is the real product hx[k] hy[ky] of two cosine sums and the column pass needs no filter table at all.
        std::vector<double> u(ky, 0.0), v(kx, 0.0);
        double tot = 0.0, amaxk = 0.0;
        for (int i = 0; i < ky; ++i)
            for (int j = 0; j < kx; ++j) {
                const double q = psf_host[i * kx + j];
                u[i] += q;
                v[j] += q;
                tot += q;
                amaxk = std::max(amaxk, std::fabs(q));
            }
        bool sep = tot != 0.0 && std::getenv("B4D_WIENER_TABLE") == nullptr;
        for (int i = 0; i < ky && sep; ++i) {
            if (std::fabs(u[i] - u[ky - 1 - i]) > 1e-7 * std::fabs(tot)) sep = false;
            for (int j = 0; j < kx && sep; ++j)
                if (std::fabs(psf_host[i * kx + j] * tot - u[i] * v[j]) > 2e-6 * amaxk * std::fabs(tot)) sep = false;
        }
        for (int j = 0; j < kx && sep; ++j)
            if (std::fabs(v[j] - v[kx - 1 - j]) > 1e-7 * std::fabs(tot)) sep = false;
        if (sep) {
            std::vector<float2> sx(g.Wh), sy(p->H);
            for (int n = 0; n < p->H; ++n) {
                double hy = 0.0;
                for (int i = 0; i < ky; ++i) hy += u[i] * std::cos(2.0 * M_PI * (double)n * (double)(i - ky / 2) / (double)p->H);
                sy[n] = make_float2((float)hy, (float)(2.0 - 2.0 * std::cos(2.0 * M_PI * (double)n / (double)p->H)));
            }
            for (int k = 0; k < g.Wh; ++k) {
                double hx = 0.0;
                for (int j = 0; j < kx; ++j) hx += v[j] / tot * std::cos(2.0 * M_PI * (double)k * (double)(j - kx / 2) / (double)p->W);
                sx[k] = make_float2((float)hx, (float)(2.0 - 2.0 * std::cos(2.0 * M_PI * (double)k / (double)p->W)));
            }
            e = hipMalloc((void**)&p->mr_sepx, sizeof(float2) * sx.size());
            if (e == hipSuccess) e = hipMalloc((void**)&p->mr_sepy, sizeof(float2) * sy.size());
            if (e == hipSuccess) e = hipMemcpy(p->mr_sepx, sx.data(), sizeof(float2) * sx.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->mr_sepy, sy.data(), sizeof(float2) * sy.size(), hipMemcpyHostToDevice);
            p->mr_balance = balance;
        } else {
            e = hipMalloc((void**)&p->mr_filt, sizeof(float2) * (size_t)g.Wh * g.Hp);
            if (e == hipSuccess) e = hipMemset(p->mr_filt, 0, sizeof(float2) * (size_t)g.Wh * g.Hp);
            if (e == hipSuccess)
                e = hipMemcpy2D(p->mr_filt, sizeof(float2) * g.Hp, p->filt, sizeof(float2) * p->H, sizeof(float2) * p->H, g.Wh,
                                hipMemcpyDeviceToDevice);
        }
        if (e != hipSuccess) rc = fail(B4D_ENOMEM, std::string("wiener filter (mixed-radix route): ") + hipGetErrorString(e));
        if (rc == B4D_OK) {
            for (float2** q : {&p->filt, &p->a, &p->b, &p->c}) {
                (void)hipFree(*q);
                *q = nullptr;
            }
            (void)hipFree(p->padded);
            p->padded = nullptr;
            p->mr = true;
        }
    }
    if (rc != B4D_OK) {
        b4d_wiener_destroy(p);
        return rc;
    }
    *out = p;
    return B4D_OK;
}

// frames [0, batch) through the three kernels of b4d_wiener_mr.hip, B4D_WIENER_FPL frames per launch; the launch groups are
// dealt alternately to `st` and a second stream, each with its own slot of the workspace (two lanes: the passes of one group run
// under those of the other)
static int wiener_mr_apply(b4d_wiener* p, const float* frames, int batch, float* out, int clip, hipStream_t st) {
    WmrGeom g = p->geom;
    g.clip = clip;
    static const int lanes_env = [] {
        const char* e = getenv("B4D_WIENER_LANES2");   // tuning aid (tools/dev_cfg5.py); results do not depend on it
        const int n = e ? atoi(e) : 2;
        return n >= 1 && n <= b4d_wiener::MR_LANES ? n : 2;
    }();
    const int nl = g_opt_lanes.load() ? std::min(lanes_env, batch) : 1;
    int fpl = std::min(batch, wiener_fpl(nl > 1, sizeof(float2) * (size_t)g.Wh * g.Hp));
    if (nl > 1) fpl = std::min(fpl, (batch + nl - 1) / nl);
    if (fpl > p->mr_cap || nl > p->mr_slots) {   // work queued earlier on other streams may still use the old buffers: drain before freeing
        const int cap = std::max(fpl, p->mr_cap), slots = std::max(nl, p->mr_slots);
        B4D_HIP(hipDeviceSynchronize());
        for (void* q : {(void*)p->mr_T, (void*)p->mr_amax})
            if (q) (void)hipFree(q);
        p->mr_T = nullptr;
        p->mr_amax = nullptr;
        p->mr_cap = p->mr_slots = 0;
        hipError_t e = hipMalloc((void**)&p->mr_T, sizeof(float2) * (size_t)slots * cap * g.Wh * g.Hp);
        if (e == hipSuccess) e = hipMalloc((void**)&p->mr_amax, sizeof(float) * (size_t)slots * cap * (g.hp + 1));
        if (e != hipSuccess) return fail(B4D_ENOMEM, std::string("wiener workspace: ") + hipGetErrorString(e));
        p->mr_cap = cap;
        p->mr_slots = slots;
    }
    if (nl > 1) {
        if (!p->mr_fork) B4D_HIP(hipEventCreateWithFlags(&p->mr_fork, hipEventDisableTiming));
        B4D_HIP(hipEventRecord(p->mr_fork, st));
        for (int l = 1; l < nl; ++l) {
            if (!p->mr_aux[l]) {
                const int rs = lane_stream(l - 1, &p->mr_aux[l]);
                if (rs) return rs;
            }
            if (!p->mr_join[l]) B4D_HIP(hipEventCreateWithFlags(&p->mr_join[l], hipEventDisableTiming));
            B4D_HIP(hipStreamWaitEvent(p->mr_aux[l], p->mr_fork, 0));
        }
    }
    const size_t fp = (size_t)g.h * g.w;
    int rc = B4D_OK, grp = 0;
    for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += fpl, ++grp) {
        const int nf = std::min(fpl, batch - b0), lane = grp % nl;
        hipStream_t ls = lane ? p->mr_aux[lane] : st;
        float2* T = p->mr_T + (size_t)lane * p->mr_cap * g.Wh * g.Hp;
        float* amax = p->mr_amax + (size_t)lane * p->mr_cap * (g.hp + 1);
        float* pmax = amax + p->mr_cap;
        rc = wmr_rows_fwd(frames + b0 * fp, T, p->twx, pmax, g, nf, ls);
        if (rc == B4D_OK) rc = wmr_cols(T, p->mr_filt, p->twy, pmax, amax, g, nf, ls, p->mr_sepx, p->mr_sepy, p->mr_balance);
        if (rc == B4D_OK) rc = wmr_rows_inv(T, out + b0 * fp, p->twx, amax, g, nf, ls);
    }
    for (int l = 1; l < nl; ++l) {
        B4D_HIP(hipEventRecord(p->mr_join[l], p->mr_aux[l]));
        B4D_HIP(hipStreamWaitEvent(st, p->mr_join[l], 0));
    }
    return rc;
}

// one frame through pad/normalise -> rows -> columns x filter -> inverse, on stream st with the work buffers of `lane`
static int wiener_frame(b4d_wiener* p, int lane, const float* f, float* o, int clip, hipStream_t st) {
    const b4d_wiener::Lane& L = p->lane[lane];
    float2* const wa = lane ? L.a : p->a;
    float2* const wb = lane ? L.b : p->b;
    float2* const wc = lane ? L.c : p->c;
    float* const wpad = lane ? L.padded : p->padded;
    float* const wmax = lane ? L.amax : p->amax;
    const size_t fp = (size_t)p->h * p->w, n = (size_t)p->H * p->W;
    const float inv = 1.0f / (float)n;
    hipLaunchKernelGGL(k_nanabsmax, dim3(256), dim3(1024), 0, st, f, fp, wmax);
    hipLaunchKernelGGL(k_absmax_final, dim3(1), dim3(64), 0, st, wmax, 256);
    B4D_HIP(hipGetLastError());
    FusedIO io{f, o, wmax, p->h, p->w, p->py, p->px, clip, p->W / 2 + 1, p->H};
    int rc;
    if (p->Ax && p->Ay) {
        // real input: rows ride in pairs (a + i b) through one complex transform, only the W/2 + 1 independent
        // columns go through the column passes (the filter of a real PSF is Hermitian), the inverse row pass
        // rebuilds each pair from its two half rows
        const int Wh = p->W / 2 + 1, Hp = (p->H + 1) / 2;
        if ((rc = dft_rows(nullptr, true, wa, wb, wc, Hp, p->Px, p->Mx, p->twx, p->dmx, false, nullptr, 1.f, st, p->Ax, p->Bx, &io, 3, 0))) return rc;
        if ((rc = transpose_c(wc, wa, p->H, Wh, st))) return rc;
        if ((rc = dft_rows(wa, false, wb, wc, wa, Wh, p->Py, p->My, p->twy, p->dmy, false, p->filt, 1.f, st, p->Ay, p->By))) return rc;
        if ((rc = dft_rows(wa, false, wb, wc, wa, Wh, p->Py, p->My, p->twy, p->dmy, true, nullptr, 1.f, st, p->Ay, p->By))) return rc;
        if ((rc = transpose_c(wa, wb, Wh, p->H, st))) return rc;
        return dft_rows(wb, false, wc, wa, wc, Hp, p->Px, p->Mx, p->twx, p->dmx, true, nullptr, inv, st, p->Ax, p->Bx, &io, 4, 0);
    }
    // forward: rows (frame -> c), transpose (c -> a), columns + filter (a -> a), all in the transposed domain after that
    if (p->Ax) {  // reflect padding and normalisation folded into the row pass's loads
        rc = dft_rows(nullptr, true, wa, wb, wc, p->H, p->Px, p->Mx, p->twx, p->dmx, false, nullptr, 1.f, st, p->Ax, p->Bx, &io, 2, 0);
    } else {
        hipLaunchKernelGGL(k_pad_reflect, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, p->h, p->w, p->py, p->px, wmax, 1, wpad);
        B4D_HIP(hipGetLastError());
        rc = dft_rows(wpad, true, wa, wb, wc, p->H, p->Px, p->Mx, p->twx, p->dmx, false, nullptr, 1.f, st);
    }
    if (rc) return rc;
    if ((rc = transpose_c(wc, wa, p->H, p->W, st))) return rc;
    if ((rc = dft_rows(wa, false, wb, wc, wa, p->W, p->Py, p->My, p->twy, p->dmy, false, p->filt, 1.f, st, p->Ay, p->By))) return rc;
    // inverse: columns (a -> a), transpose (a -> b), rows (b -> b) with the 1/(H W) factor
    if ((rc = dft_rows(wa, false, wb, wc, wa, p->W, p->Py, p->My, p->twy, p->dmy, true, nullptr, 1.f, st, p->Ay, p->By))) return rc;
    if ((rc = transpose_c(wa, wb, p->W, p->H, st))) return rc;
    if (p->Ax)   // clip, rescale and crop folded into the last pass's stores
        return dft_rows(wb, false, wc, wa, wb, p->H, p->Px, p->Mx, p->twx, p->dmx, true, nullptr, inv, st, p->Ax, p->Bx, &io, 0, 1);
    if ((rc = dft_rows(wb, false, wc, wa, wb, p->H, p->Px, p->Mx, p->twx, p->dmx, true, nullptr, inv, st))) return rc;
    hipLaunchKernelGGL(k_crop_out, dim3((unsigned)((fp + 255) / 256)), dim3(256), 0, st, wb, p->h, p->w, p->py, p->px, wmax, 1, clip, o);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// lanes 1..: work buffers; every lane: a non-blocking stream and its join event
static int wiener_lanes(b4d_wiener* p) {
    if (p->fork) return B4D_OK;
    const size_t n = (size_t)p->H * p->W;
    for (int l = 0; l < B4D_WIENER_LANES; ++l) {
        b4d_wiener::Lane& L = p->lane[l];
        if (l > 0) {   // each step only if still missing: a call that failed half-way (out of memory) can be repeated
            if (!L.a) B4D_HIP(hipMalloc((void**)&L.a, sizeof(float2) * n));
            if (!L.b) B4D_HIP(hipMalloc((void**)&L.b, sizeof(float2) * n));
            if (!L.c) B4D_HIP(hipMalloc((void**)&L.c, sizeof(float2) * n));
            if (!L.padded) B4D_HIP(hipMalloc((void**)&L.padded, sizeof(float) * n));
            if (!L.amax) B4D_HIP(hipMalloc((void**)&L.amax, sizeof(float) * 256));
        }
        if (!L.st) {
            const int rs = lane_stream(l, &L.st);
            if (rs) return rs;
        }
        if (!L.done) B4D_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    }
    B4D_HIP(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
    return B4D_OK;
}

int b4d_wiener_apply(b4d_wiener* p, const float* frames, int batch, float* out, int clip, void* stream) {
    if (!p) return fail(B4D_EINVAL, "null argument");
    std::lock_guard<std::recursive_mutex> lk(p->mu);
    if (!p || !frames || !out) return fail(B4D_EINVAL, "null argument");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const size_t fp = (size_t)p->h * p->w;
    if (p->mr) return wiener_mr_apply(p, frames, batch, out, clip, st);
    if (batch == 1) return wiener_frame(p, 0, frames, out, clip, st);
    if (!g_opt_lanes.load()) {   // option "lanes" 0: the caller's stream only
        int r1 = B4D_OK;
        for (int b = 0; b < batch && r1 == B4D_OK; ++b) r1 = wiener_frame(p, 0, frames + b * fp, out + b * fp, clip, st);
        return r1;
    }
    int rc = wiener_lanes(p);
    if (rc) return rc;
    // fork: the lanes start after everything already queued on the caller's stream; join: the caller's stream waits for all
    const int nl = std::min(batch, B4D_WIENER_LANES);
    B4D_HIP(hipEventRecord(p->fork, st));
    for (int l = 0; l < nl; ++l) B4D_HIP(hipStreamWaitEvent(p->lane[l].st, p->fork, 0));
    for (int b = 0; b < batch && rc == B4D_OK; ++b) rc = wiener_frame(p, b % nl, frames + b * fp, out + b * fp, clip, p->lane[b % nl].st);
    for (int l = 0; l < nl; ++l) {
        B4D_HIP(hipEventRecord(p->lane[l].done, p->lane[l].st));
        B4D_HIP(hipStreamWaitEvent(st, p->lane[l].done, 0));
    }
    return rc;
}

int b4d_richardson_lucy(const float* frames, int batch, int h, int w, const float* psf_host, int ky, int kx, int num_iter,
                        float filter_epsilon, int clip, float* out, void* stream) {
    B4D_SCRATCH_LOCK();
    if (!frames || !out || !psf_host) return fail(B4D_EINVAL, "null argument");
    if (batch < 1 || h < 2 || w < 2 || num_iter < 1) return fail(B4D_EINVAL, "batch, num_iter >= 1 and h, w >= 2 required");
    if (ky < 1 || kx < 1 || !(ky & 1) || !(kx & 1) || ky > RL_MAXK || kx > RL_MAXK) return fail(B4D_EINVAL, "kernel sides must be odd and <= 33");
    if (ky / 2 >= h || kx / 2 >= w) return fail(B4D_EINVAL, "kernel larger than the frame");
    hipStream_t st = (hipStream_t)stream;
    const int py = ky / 2, px = kx / 2, H = h + 2 * py, W = w + 2 * px;
    const size_t n = (size_t)H * W, fp = (size_t)h * w;
    void* ws = nullptr;
    int rc = get_scratch(sizeof(float) * (3 * n + 256 + (size_t)ky * kx) + 1024, &ws, (hipStream_t)stream);
    if (rc) return rc;
    float* work = static_cast<float*>(ws);
    float* est = work + n;
    float* rel = est + n;
    float* amax = rel + n;
    float* psf = amax + 256;
    B4D_HIP(hipMemcpyAsync(psf, psf_host, sizeof(float) * ky * kx, hipMemcpyHostToDevice, st));
    B4D_HIP(hipStreamSynchronize(st));   // psf_host is caller-owned
    const size_t lds = sizeof(float) * ((size_t)(RL_TX + 2 * px) * (RL_TY + 2 * py) + (size_t)ky * kx);
    const dim3 grid((W + RL_TX - 1) / RL_TX, (H + RL_TY - 1) / RL_TY);
    for (int b = 0; b < batch; ++b) {
        const float* f = frames + b * fp;
        hipLaunchKernelGGL(k_nanabsmax, dim3(256), dim3(1024), 0, st, f, fp, amax);
        hipLaunchKernelGGL(k_absmax_final, dim3(1), dim3(64), 0, st, amax, 256);
        hipLaunchKernelGGL(k_pad_reflect, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, h, w, py, px, amax, 1, work);
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, est, n, 0.5f);
        for (int it = 0; it < num_iter; ++it) {
            hipLaunchKernelGGL((k_rl_step<0>), grid, dim3(256), lds, st, est, work, rel, psf, H, W, ky, kx, filter_epsilon);
            hipLaunchKernelGGL((k_rl_step<1>), grid, dim3(256), lds, st, rel, work, est, psf, H, W, ky, kx, 0.f);
        }
        hipLaunchKernelGGL(k_rl_crop, dim3((unsigned)((fp + 255) / 256)), dim3(256), 0, st, est, h, w, py, px, amax, clip, out + b * fp);
        B4D_HIP(hipGetLastError());
    }
    B4D_HIP(hipStreamSynchronize(st));   // the shared scratch must outlive the kernels
    return B4D_OK;
}

}  // extern "C"

#ifdef B4D_DIAG
extern "C" int b4d_debug_set_pm_diag(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(b4d::g_pm_diag), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

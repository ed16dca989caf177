// b4d_track.hip -- cross-correlation and phase-correlation translation tracking on gfx950
// (reference signal/corr.py:169-253, signal/tracking.py:191-375).
//
// One 2-D forward transform per distinct image and per distinct template (K1 with ROI/z-score
// sources + forward column pass), then per (image, template) pair:
//   k_col_prod   Fi * conj(Ft) [/(|.| + eps) for phase correlation] fused into the inverse column FFT
//   k_row_c2r    inverse row FFT -> real map (xcorr) or |map| + arg-max partials (tracking)
//   k_track_fin  first-occurrence arg-max, exact median by 3-pass radix select, 3x3 Taylor step
// Image spectra are computed once and reused by every template (the reference re-transforms
// the same frame 18 times per time step, metrics/speckles.py:347-415).
#include <cstring>

#include "b4d_fft2d.hpp"
#include "b4d_select.hpp"
#include "b4d_wiener_mr.hpp"

namespace b4d {

// ------------------------------------------------------------------------------------ ROI statistics
// Population mean / std of every ROI in float64, stored as float like NumPy's float32 arithmetic does:
// z = (x - f32(mean)) / f32(std + eps).  Two kernels so that a full-frame "ROI" is spread over the chip: shifted power
// sums sum (x - x0), sum (x - x0)^2 (x0 = first ROI pixel: no cancellation) per slice, then one wave per item.
constexpr int ROI_SPLIT = 32;
// grid (ROI_SPLIT, items), block 256; part[item][slice] = {S1, S2}
__global__ void __launch_bounds__(256) k_roi_part(const float* __restrict__ frames, int ny, int nx, const RowSrc* __restrict__ srcs,
                                                  double* __restrict__ part) {
    __shared__ double sh[8];
    const RowSrc sd = srcs[blockIdx.y];
    const int h = sd.y1 - sd.y0, w = sd.x1 - sd.x0, n = h * w;
    const float* f = frames + (size_t)sd.frame * ny * nx;
    const double x0 = (double)f[(size_t)sd.y0 * nx + sd.x0];
    // a slice = a range of ROI rows, walked row by row: no division per pixel (the flat-index form cost a full-frame source,
    // 5.5 M pixels at 2160 x 2560, 160 us: a quarter of a small general-size tracking call), four rows in flight per lane
    (void)n;
    const int per = (h + ROI_SPLIT - 1) / ROI_SPLIT, r0 = blockIdx.x * per, r1 = min(h, r0 + per);
    double a1 = 0.0, a2 = 0.0;
    const float* base = f + (size_t)sd.y0 * nx + sd.x0;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        const float* q = base + (size_t)r * nx;
        for (int x = threadIdx.x; x < w; x += 256) {
            const double d0 = (double)q[x] - x0, d1 = (double)q[x + (size_t)nx] - x0, d2 = (double)q[x + 2 * (size_t)nx] - x0,
                         d3 = (double)q[x + 3 * (size_t)nx] - x0;
            a1 += (d0 + d1) + (d2 + d3);
            a2 = fma(d0, d0, a2);
            a2 = fma(d1, d1, a2);
            a2 = fma(d2, d2, a2);
            a2 = fma(d3, d3, a2);
        }
    }
    for (; r < r1; ++r) {
        const float* q = base + (size_t)r * nx;
        for (int x = threadIdx.x; x < w; x += 256) {
            const double d = (double)q[x] - x0;
            a1 += d;
            a2 = fma(d, d, a2);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_down(a1, o, 64);
        a2 += __shfl_down(a2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = a1;
        sh[4 + (threadIdx.x >> 6)] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)blockIdx.y * ROI_SPLIT + blockIdx.x) * 2;
        o[0] = sh[0] + sh[1] + sh[2] + sh[3];
        o[1] = sh[4] + sh[5] + sh[6] + sh[7];
    }
}

// grid (ceil(items / 64)), block 64: one lane per item
__global__ void __launch_bounds__(64) k_roi_fin(const float* __restrict__ frames, int ny, int nx, double eps, int items,
                                                const double* __restrict__ part, RowSrc* __restrict__ srcs) {
    const int it = blockIdx.x * 64 + threadIdx.x;
    if (it >= items) return;
    const RowSrc sd = srcs[it];
    const double n = (double)(sd.y1 - sd.y0) * (sd.x1 - sd.x0);
    const double x0 = (double)frames[((size_t)sd.frame * ny + sd.y0) * nx + sd.x0];
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < ROI_SPLIT; ++k) {
        s1 += part[((size_t)it * ROI_SPLIT + k) * 2];
        s2 += part[((size_t)it * ROI_SPLIT + k) * 2 + 1];
    }
    const double m = s1 / n, var = fmax(s2 / n - m * m, 0.0);
    srcs[it].mean = (float)(x0 + m);
    srcs[it].denom = (float)(sqrt(var) + eps);
}

static int roi_stats(const float* frames, int ny, int nx, double eps, RowSrc* srcs, int items, double* part, hipStream_t st) {
    hipLaunchKernelGGL(k_roi_part, dim3(ROI_SPLIT, items), dim3(256), 0, st, frames, ny, nx, srcs, part);
    hipLaunchKernelGGL(k_roi_fin, dim3((items + 63) / 64), dim3(64), 0, st, frames, ny, nx, eps, items, part, srcs);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// ------------------------------------------------------------------------------------ product + inverse column pass
struct ProdArgs {
    const float2* spec_a;   // image half spectra (tile-major), item-indexed
    const float2* spec_b;   // template half spectra
    const int* idx_a;       // per pair; null = identity
    const int* idx_b;
    float2* g;              // (pairs) tile-major output of the inverse column pass
    const float2* tw;
    const float2* tw2;      // the same table through a second pointer (TPL: keeps the first transform's twiddles from staying live)
    float eps;
    int nt;
    unsigned flags;         // B4D_REMOVE_MEAN: zero the DC bin of the product (both means removed)
    const RowSrc* srcs_b;   // TPL: item ib of spec_b holds ROW-transformed data in rows [y0, y1) only (k_row_r2c of a zero-embedded
                            // ROI); its column transform happens here
};

// grid (nt, pairs), block ColCfg<NY>::THREADS (same tile geometry as k_col).
// TPL (phase-correlation tracking): the template operand arrives row-transformed only, in its ROI rows (121 of 1024 in cfg3), and its
// forward column transform runs here, in registers, ahead of the product -- the same Fft3 code on the same values as
// k_col<COL_FORWARD>, i.e. bit-identical spectra, without writing a 4-MB half spectrum per template and reading it back per pair
// (585 of the 649 spectra of a cfg3 call are templates, 576 of them used once).
template <int NY, bool WHITEN, bool TPL = false>
__global__ void __launch_bounds__(ColCfg<NY>::THREADS, ColCfg<NY>::WAVES_PER_EU) k_col_prod(ProdArgs p) {
    using Cfg = ColCfg<NY>;
    using G = typename Cfg::G;
    constexpr int T = G::T, E = E16, NC = Cfg::NC, CPT = Cfg::CPT, CT = Cfg::CT;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int cp = threadIdx.x % CPT, u = threadIdx.x / CPT;
    const int ct = blockIdx.x, nt = gridDim.x;
    const size_t pair = blockIdx.y;
    const size_t ia = p.idx_a ? p.idx_a[pair] : pair, ib = p.idx_b ? p.idx_b[pair] : pair;
    const float2* ta = p.spec_a + ((ia * nt + ct) * (size_t)NY) * CT;
    const float2* tb = p.spec_b + ((ib * nt + ct) * (size_t)NY) * CT;
    float2* tg = p.g + ((pair * nt + ct) * (size_t)NY) * CT;
    const unsigned toff = (unsigned)u * CT + NC * cp;
    float2 v[NC][E];
    if (TPL) {
        const int ry0 = p.srcs_b[ib].y0, ry1 = p.srcs_b[ib].y1;
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int h = 0; h < NC / 2; ++h) {   // clamped row + select: unconditional loads (as k_col<COL_FORWARD>)
                const int ky = u + T * j, kc = min(max(ky, ry0), ry1 - 1);
                float4 q = *reinterpret_cast<const float4*>(tb + (size_t)kc * CT + NC * cp + 2 * h);
                if (ky < ry0 || ky >= ry1) q = make_float4(0.f, 0.f, 0.f, 0.f);
                v[2 * h][j] = make_float2(q.x, q.y);
                v[2 * h + 1][j] = make_float2(q.z, q.w);
            }
        }
        Fft3<G, 1>::template run_sets<NC, Cfg::SERIAL, true, Cfg::SB>(v, u, cp, lds, p.tw);
        // v[c][j] = Ft[ky = u + T j][column c]: the product with the image tile replaces it in place
        int tid1 = threadIdx.x;   // lane position derived again: the first transform's index set need not stay live (cf. k_col)
        asm volatile("" : "+v"(tid1));
        const unsigned toff1 = (unsigned)(tid1 / CPT) * CT + NC * (tid1 % CPT);
        __syncthreads();          // every lane is done reading the exchange buffer
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int h = 0; h < NC / 2; ++h) {
                const float4 qa = *reinterpret_cast<const float4*>(ta + (size_t)(T * j * CT) + toff1 + 2 * h);
                float2 c0 = cross_power<WHITEN>(make_float2(qa.x, qa.y), v[2 * h][j], p.eps);
                const float2 c1 = cross_power<WHITEN>(make_float2(qa.z, qa.w), v[2 * h + 1][j], p.eps);
                if (h == 0 && j == 0 && ct == 0 && tid1 == 0 && (p.flags & B4D_REMOVE_MEAN)) c0 = make_float2(0.f, 0.f);
                v[2 * h][j] = make_float2(c0.y, c0.x);
                v[2 * h + 1][j] = make_float2(c1.y, c1.x);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int h = 0; h < NC / 2; ++h) {
                const float4 qa = *reinterpret_cast<const float4*>(ta + (size_t)(T * j * CT) + toff + 2 * h);
                const float4 qb = *reinterpret_cast<const float4*>(tb + (size_t)(T * j * CT) + toff + 2 * h);
                float2 c0 = cross_power<WHITEN>(make_float2(qa.x, qa.y), make_float2(qb.x, qb.y), p.eps);
                const float2 c1 = cross_power<WHITEN>(make_float2(qa.z, qa.w), make_float2(qb.z, qb.w), p.eps);
                if (h == 0 && j == 0 && ct == 0 && cp == 0 && u == 0 && (p.flags & B4D_REMOVE_MEAN))
                    c0 = make_float2(0.f, 0.f);  // DC bin of the cross spectrum (both means removed)
                v[2 * h][j] = make_float2(c0.y, c0.x);  // (im, re)-swapped: inverse transform with the forward code
                v[2 * h + 1][j] = make_float2(c1.y, c1.x);
            }
        }
    }
    {
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        Fft3<G, 1>::template run_sets<NC, Cfg::SERIAL, true, Cfg::SB>(v, tid2 / CPT, tid2 % CPT, lds, TPL ? p.tw2 : p.tw);
    }
    unsigned toff2 = toff;   // laundered: the store addresses are not kept live across the transform (spills otherwise)
    asm volatile("" : "+v"(toff2));
#pragma unroll
    for (int j = 0; j < E; ++j) {
        float2 c[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) c[k] = make_float2(v[k][j].y, v[k][j].x);
        store_cols<NC>(tg + (size_t)(T * j * CT) + toff2, c);
    }
}

// ------------------------------------------------------------------------------------ tracking epilogue
struct FinArgs {
    const float* mag;       // (pairs, ny, nx)
    int partial_map;        // k_track_fin2: only the rows around the peak are valid in `mag` (k_row_c2r C2R_MAG nomap + C2R_ROWS)
    const float* part_val;  // (pairs, nblk)
    const int* part_idx;
    double* out;            // (pairs, 4): dy, dx, peak, snr
    int* peak_ij;           // (pairs, 2) or null
    int ny, nx, nblk, subpixel;
    double eps;
    // template matching: the map of pair i is geom[4 i] x geom[4 i + 1] values (row-major, compact) at mag + i * stride,
    // shifts are counted from (geom[4 i + 2], geom[4 i + 3]), the median runs over med_src (|map|).  Null / 0 for
    // phase correlation: ny x nx maps, origin (ny/2, nx/2), median of the map itself.
    const int* geom;
    const float* med_src;
    size_t stride;
    float* compact;         // optional (pairs, stride) scratch for the median's gathered bin (b4d_select.hpp)
    const unsigned* skip;   // optional: pair i is left alone when skip[i * skip_stride] != 0 (finished by k_track_fin2)
    int skip_stride;
};

// peak quality + Taylor step of one pair (one lane), op for op like tracking.py:314-375 (float32 scalars, no contraction)
__device__ inline void track_finish(const FinArgs& p, size_t pair, const float* __restrict__ mag, int mny, int mnx, int oy, int ox,
                                    float bv, int bi, float med) {
    const int mi = bi / mnx, mj = bi % mnx;
    const double peak = (double)bv;
    const double snr = fabs(peak) / ((double)med + p.eps);
    double dy = (double)(mi - oy), dx = (double)(mj - ox);
    if (p.subpixel && mi > 0 && mi < mny - 1 && mj > 0 && mj < mnx - 1) {
        auto c = [&](int di, int dj) { return mag[(size_t)(mi + di) * mnx + (mj + dj)]; };
        const float c00 = c(0, 0);
        const float gy = __fdiv_rn(__fsub_rn(c(1, 0), c(-1, 0)), 2.0f);
        const float hyy = __fsub_rn(__fadd_rn(c(1, 0), c(-1, 0)), __fmul_rn(2.0f, c00));
        const float gx = __fdiv_rn(__fsub_rn(c(0, 1), c(0, -1)), 2.0f);
        const float hxx = __fsub_rn(__fadd_rn(c(0, 1), c(0, -1)), __fmul_rn(2.0f, c00));
        const float hxy = __fdiv_rn(__fadd_rn(__fsub_rn(__fsub_rn(c(1, 1), c(1, -1)), c(-1, 1)), c(-1, -1)), 4.0f);
        const float det = __fsub_rn(__fmul_rn(hxx, hyy), __fmul_rn(hxy, hxy));
        if (det != 0.0f) {
            const float inv = __fdiv_rn(1.0f, det);
            // NOTE the reference's swapped corrections (tracking.py:372-373), reproduced on purpose
            const float di = __fmul_rn(-__fsub_rn(__fmul_rn(hyy, gx), __fmul_rn(hxy, gy)), inv);
            const float dj = __fmul_rn(-__fsub_rn(__fmul_rn(hxx, gy), __fmul_rn(hxy, gx)), inv);
            dy += (double)di;
            dx += (double)dj;
        }
    }
    double* o = p.out + pair * 4;
    o[0] = dy;
    o[1] = dx;
    o[2] = peak;
    o[3] = snr;
    if (p.peak_ij) {
        p.peak_ij[pair * 2] = mi;
        p.peak_ij[pair * 2 + 1] = mj;
    }
}

// grid (pairs), block 1024, dynamic LDS FIN_LDS bytes
constexpr int FIN_REP = 4;
constexpr size_t FIN_LDS = sizeof(unsigned) * 2048 * FIN_REP;
__global__ void __launch_bounds__(1024) k_track_fin(FinArgs p) {
    constexpr int REP = FIN_REP;
    extern __shared__ unsigned hist[];   // FIN_REP copies of the 2048-bin histogram
    __shared__ unsigned sh[4];
    __shared__ float sv[16];
    __shared__ int si[16];
    const size_t pair = blockIdx.x;
    if (p.skip && p.skip[pair * p.skip_stride]) return;
    int mny = p.ny, mnx = p.nx, oy = p.ny / 2, ox = p.nx / 2;
    if (p.geom) {
        mny = p.geom[4 * pair];
        mnx = p.geom[4 * pair + 1];
        oy = p.geom[4 * pair + 2];
        ox = p.geom[4 * pair + 3];
    }
    const unsigned n = (unsigned)mny * mnx;
    const size_t stride = p.stride ? p.stride : (size_t)n;
    const float* mag = p.mag + pair * stride;
    const float* msrc = p.med_src ? p.med_src + pair * stride : mag;
    // ---- arg-max over the per-workgroup partials (first occurrence in row-major order)
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < p.nblk; i += blockDim.x)
        argmax_merge(bv, bi, p.part_val[pair * p.nblk + i], p.part_idx[pair * p.nblk + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        argmax_merge(bv, bi, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    bv = sv[0];
    bi = si[0];
    for (int i = 1; i < 16; ++i) argmax_merge(bv, bi, sv[i], si[i]);
    __syncthreads();
    // ---- median of the magnitude map (np.median: mean of the two middle values for even counts,
    //      evaluated in float32 like NumPy does for a float32 array)
    unsigned nl, ne, cn = n;
    const float* cx = msrc;
    float* comp = p.compact ? p.compact + pair * stride : nullptr;
    float med;
    if (n & 1u) {
        med = key2f(radix_select<REP>(msrc, n, n / 2, hist, sh, nl, ne, comp));
    } else {
        const unsigned ka = radix_select<REP>(msrc, n, n / 2 - 1, hist, sh, nl, ne, comp, &cx, &cn);
        float a = key2f(ka), b = a;
        if (nl + ne <= n / 2) {  // upper middle value = next larger element: in the gathered bin, else (rare) anywhere above it
            unsigned kb = next_larger_key(cx, cn, ka, hist);
            if (kb == 0xffffffffu && cx != msrc) kb = next_larger_key(msrc, n, ka, hist);
            b = key2f(kb);
        }
        med = __fmul_rn(__fadd_rn(a, b), 0.5f);
    }
    if (threadIdx.x != 0) return;
    track_finish(p, pair, mag, mny, mnx, oy, ox, bv, bi, med);
}

static int launch_track_fin(const FinArgs& fa, int pairs, hipStream_t st) {
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_track_fin), FIN_LDS);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL(k_track_fin, dim3(pairs), dim3(1024), FIN_LDS, st, fa);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// ---- phase correlation: the median's first select pass comes from k_row_c2r (one global 2048-bin histogram per pair), the
// bin's elements are gathered by SEVERAL workgroups per pair (one streamed read of the map over the whole chip), and one
// workgroup per pair finishes on the gathered ~10 % of the map: passes 2-3 of the select, arg-max partials, Taylor step.
struct SelState {
    unsigned bin, below, count, fill;   // selected top-11-bit bin, elements under it, elements in it, gather cursor
    unsigned ok, pad[3];                // the median is in the expected bin: k_row_c2r has already counted and gathered it
};
constexpr int SEL_WORDS = sizeof(SelState) / sizeof(unsigned);

// Whitened correlation maps (signal/tracking.py:280-285) have sum corr^2 = 1 and corr is REAL (a Hermitian spectrum), so
// away from the peak it is Gaussian with variance 1 / N and |corr| half-normal: median 0.6745 / sqrt(N) (x sqrt(1 - peak^2),
// a per cent or so; observed on the cfg3 protocol: 6.55e-4 ... 6.58e-4 against 6.587e-4).  The top-11-bit bin of that value
// is where the median of nearly every pair falls.  k_row_c2r counts the elements below / inside that bin and gathers the
// bin while it writes the map; pairs whose counts put the median elsewhere take the full three-pass select on the map
// (k_track_fin).  Exactness never depends on the guess.
// `mode` = b4d_set_option("track_predict_bin", 0 / 1 / 2), read ONCE per entry-point call (b4d::g_opt_track_predict, an atomic in
// b4d_kernels.hip): off, on, on with a deliberately WRONG bin (every pair then takes the gated full-map pass behind the
// map-free one: tests run all three)
static unsigned predicted_median_bin(size_t n, int mode) {
    if (!mode) return 0u;
    const float med = (float)(0.6744897501960817 / std::sqrt((double)n));
    unsigned bits;
    memcpy(&bits, &med, sizeof(bits));
    return 1024u + (bits >> 21) + (mode == 2 ? 3u : 0u);
}

// grid (ceil(pairs / 64)), block 64: the expectation holds for pair i when the median's rank falls inside the expected bin
// (and the gathered count agrees with the counted one)
__global__ void __launch_bounds__(64) k_track_select(size_t n, SelState* __restrict__ sel, unsigned pred, int pairs) {
    const int pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= pairs) return;
    SelState* st = sel + pair;
    const unsigned k = (unsigned)((n & 1u) ? n / 2 : n / 2 - 1);   // rank of the (lower) middle element
    st->bin = pred;
    st->ok = (pred != 0u && k >= st->below && k < st->below + st->count && st->fill == st->count) ? 1u : 0u;
}

// grid (pairs), block 1024, dynamic LDS FIN_LDS bytes
__global__ void __launch_bounds__(1024) k_track_fin2(FinArgs p, SelState* __restrict__ sel) {
    constexpr int REP = FIN_REP;
    extern __shared__ unsigned hist[];
    __shared__ unsigned sh[4];
    __shared__ float sv[16];
    __shared__ int si[16];
    const size_t pair = blockIdx.x;
    if (!sel[pair].ok) return;    // the expectation failed: k_track_fin does the whole selection on the map
    int mny = p.ny, mnx = p.nx, oy = p.ny / 2, ox = p.nx / 2;
    if (p.geom) {   // template matching: compact (rows, columns) maps, shifts counted from the template's position
        mny = p.geom[4 * pair];
        mnx = p.geom[4 * pair + 1];
        oy = p.geom[4 * pair + 2];
        ox = p.geom[4 * pair + 3];
    }
    const unsigned n = (unsigned)mny * mnx;
    const size_t stride = p.stride ? p.stride : (size_t)n;
    const float* mag = p.mag + pair * stride;
    const float* msrc = p.med_src ? p.med_src + pair * stride : mag;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < p.nblk; i += blockDim.x)
        argmax_merge(bv, bi, p.part_val[pair * p.nblk + i], p.part_idx[pair * p.nblk + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        argmax_merge(bv, bi, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    bv = sv[0];
    bi = si[0];
    for (int i = 1; i < 16; ++i) argmax_merge(bv, bi, sv[i], si[i]);
    __syncthreads();
    // ---- median (np.median of a float32 map): passes 2-3 of the select on the gathered bin
    const SelState ss = sel[pair];
    const float* cx = p.compact + pair * stride;
    const unsigned cn = ss.count;
    unsigned nl, ne;
    float med;
    if (n & 1u) {
        med = key2f(radix_select<REP>(cx, cn, n / 2, hist, sh, nl, ne, nullptr, nullptr, nullptr, 1, ss.bin << 21, ss.below));
    } else {
        const unsigned ka = radix_select<REP>(cx, cn, n / 2 - 1, hist, sh, nl, ne, nullptr, nullptr, nullptr, 1, ss.bin << 21, ss.below);
        float a = key2f(ka), b = a;
        if (nl + ne <= n / 2) {  // upper middle value = next larger element: in the gathered bin, else (rare) anywhere above it
            unsigned kb = next_larger_key(cx, cn, ka, hist);
            if (kb == 0xffffffffu) {   // (rare) the upper middle value lies above the gathered bin: that needs the whole map
                if (p.partial_map) {   // hand the pair to the full-map route (uniform: every lane holds the same kb)
                    if (threadIdx.x == 0) sel[pair].ok = 0u;
                    return;
                }
                kb = next_larger_key(msrc, n, ka, hist);
            }
            b = key2f(kb);
        }
        med = __fmul_rn(__fadd_rn(a, b), 0.5f);
    }
    if (threadIdx.x != 0) return;
    track_finish(p, pair, mag, mny, mnx, oy, ox, bv, bi, med);
}

static int launch_track_fin2(const FinArgs& fa, SelState* sel, unsigned pred, int pairs, hipStream_t st) {
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_track_fin2), FIN_LDS);
        if (rc_lds) return rc_lds;
    }
    const size_t n = (size_t)fa.ny * fa.nx;
    hipLaunchKernelGGL(k_track_select, dim3((pairs + 63) / 64), dim3(64), 0, st, n, sel, pred, pairs);
    hipLaunchKernelGGL(k_track_fin2, dim3(pairs), dim3(1024), FIN_LDS, st, fa, sel);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
// template matching: the verdicts come from k_ncc_check
static int launch_track_fin2_checked(const FinArgs& fa, SelState* sel, int pairs, hipStream_t st) {
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_track_fin2), FIN_LDS);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL(k_track_fin2, dim3(pairs), dim3(1024), FIN_LDS, st, fa, sel);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
// the pairs whose expectation failed (all of them when it is switched off): whole selection on the (full) map
static int launch_track_fin_rest(const FinArgs& fa, SelState* sel, int pairs, hipStream_t st) {
    FinArgs fb = fa;
    fb.skip = &sel[0].ok;
    fb.skip_stride = SEL_WORDS;
    return launch_track_fin(fb, pairs, st);
}

// ------------------------------------------------------------------------------------ NCC template matching
// Summed-area tables (float64) of the image as the matcher sees it, v = (x - mean) / denom in float32 (raw image:
// mean 0, denom 1), and of v^2: sat[(y + 1) (nx + 1) + x + 1] = sum over rows <= y, columns <= x.
// Row pass: grid (ny, nimg), block 256; column pass: grid (ceil((nx + 1) / 64), nimg), block 64.
// win > 0: the row holds the WINDOW sums H(y, j) = sum_{x = j .. j + win - 1} instead of the prefix; the column pass then sums the h
// rows under each window position (k_sat_cols), so that the matcher reads ONE value per table and position instead of four
// corners (k_ncc_map waits for exactly these loads: 1.09 -> 0.97 ms per group of 192 pairs).
__global__ void __launch_bounds__(256) k_sat_rows(const float* __restrict__ frames, int ny, int nx, const RowSrc* __restrict__ srcs,
                                                  double* __restrict__ sat1, double* __restrict__ sat2, int win) {
    __shared__ double s1[256], s2[256];
    const RowSrc sd = srcs[blockIdx.y];
    const int y = blockIdx.x, per = (nx + 255) / 256;
    const float* row = frames + ((size_t)sd.frame * ny + y) * nx;
    const size_t W1 = (size_t)nx + 1, base = (size_t)blockIdx.y * (ny + 1) * W1;
    double a1 = 0.0, a2 = 0.0;
    const int x0 = threadIdx.x * per, x1 = min(nx, x0 + per);
    const float inv = 1.0f / sd.denom;   // the same float32 z-score as k_row_r2c's (one division per item)
    for (int x = x0; x < x1; ++x) {
        const double v = (double)((row[x] - sd.mean) * inv);
        a1 += v;
        a2 = fma(v, v, a2);
    }
    s1[threadIdx.x] = a1;
    s2[threadIdx.x] = a2;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan of the per-thread sums
        const double t1 = threadIdx.x >= o ? s1[threadIdx.x - o] : 0.0, t2 = threadIdx.x >= o ? s2[threadIdx.x - o] : 0.0;
        __syncthreads();
        s1[threadIdx.x] += t1;
        s2[threadIdx.x] += t2;
        __syncthreads();
    }
    double r1 = s1[threadIdx.x] - a1, r2 = s2[threadIdx.x] - a2;  // exclusive prefix of this thread's segment
    double* o1 = sat1 + base + (size_t)(y + 1) * W1;
    double* o2 = sat2 + base + (size_t)(y + 1) * W1;
    if (y == 0)
        for (int x = threadIdx.x; x <= nx; x += 256) {
            sat1[base + x] = 0.0;
            sat2[base + x] = 0.0;
        }
    if (win > 0) {   // the row's prefix P (P[x] = sum of the first x values) stays in LDS; the table receives P[j + win] - P[j]
        extern __shared__ double pre[];   // 2 (nx + 1) doubles
        double* p1 = pre;
        double* p2 = pre + (nx + 1);
        for (int x = x0; x < x1; ++x) {
            const double v = (double)((row[x] - sd.mean) * inv);
            r1 += v;
            r2 = fma(v, v, r2);
            p1[x + 1] = r1;
            p2[x + 1] = r2;
        }
        if (threadIdx.x == 0) p1[0] = p2[0] = 0.0;
        __syncthreads();
        for (int j = threadIdx.x; j <= nx; j += 256) {   // lanes along the row: whole lines per store
            const bool full = j + win <= nx;
            o1[j] = full ? p1[j + win] - p1[j] : 0.0;
            o2[j] = full ? p2[j + win] - p2[j] : 0.0;
        }
        return;
    }
    for (int x = x0; x < x1; ++x) {
        const double v = (double)((row[x] - sd.mean) * inv);
        r1 += v;
        r2 = fma(v, v, r2);
        o1[x + 1] = r1;
        o2[x + 1] = r2;
    }
    if (threadIdx.x == 0) {
        o1[0] = 0.0;
        o2[0] = 0.0;
    }
}

// Column pass: the sum of the h rows of window sums under each window position, D(i, j) = sum_{y = i .. i + h - 1} H(y, j), as a
// running window down the column (row i of the table receives D(i, .); rows beyond ny - h are not used).  One column per lane;
// the running sum is a dependent chain, the loads are not: eight rows of both tables are requested before the first is added.
__global__ void __launch_bounds__(64) k_sat_cols(int ny, int nx, double* __restrict__ sat1, double* __restrict__ sat2, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x > nx) return;
    const size_t W1 = (size_t)nx + 1, base = (size_t)blockIdx.y * (ny + 1) * W1 + x;
    double w1 = 0.0, w2 = 0.0;
    for (int y = 1; y <= h; ++y) {   // H(y - 1) sits in row y
        w1 += sat1[base + (size_t)y * W1];
        w2 += sat2[base + (size_t)y * W1];
    }
    const int last = ny - h;         // D(0 .. last)
    int i = 0;
    if (h >= 8)
        for (; i + 8 <= last; i += 8) {
            double in1[8], in2[8], out1[8], out2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {   // entering row H(i + t + h), leaving row H(i + t)
                in1[t] = sat1[base + (size_t)(i + t + h + 1) * W1];
                in2[t] = sat2[base + (size_t)(i + t + h + 1) * W1];
                out1[t] = sat1[base + (size_t)(i + t + 1) * W1];
                out2[t] = sat2[base + (size_t)(i + t + 1) * W1];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                sat1[base + (size_t)(i + t) * W1] = w1;
                sat2[base + (size_t)(i + t) * W1] = w2;
                w1 = (w1 + in1[t]) - out1[t];
                w2 = (w2 + in2[t]) - out2[t];
            }
        }
    for (; i <= last; ++i) {
        sat1[base + (size_t)i * W1] = w1;
        sat2[base + (size_t)i * W1] = w2;
        if (i < last) {
            w1 = (w1 + sat1[base + (size_t)(i + h + 1) * W1]) - sat1[base + (size_t)(i + 1) * W1];
            w2 = (w2 + sat2[base + (size_t)(i + h + 1) * W1]) - sat2[base + (size_t)(i + 1) * W1];
        }
    }
}

// template statistics of the z-scored float32 template: tstat[2 k] = mean, tstat[2 k + 1] = sum (z - mean)^2.
// grid (ntpl), block 1024
__global__ void __launch_bounds__(1024) k_tpl_stats(const float* __restrict__ frames, int ny, int nx, const RowSrc* __restrict__ srcs,
                                                    double* __restrict__ tstat) {
    __shared__ double sh[32];
    const RowSrc sd = srcs[blockIdx.x];
    const int h = sd.y1 - sd.y0, w = sd.x1 - sd.x0, n = h * w;
    const float* f = frames + (size_t)sd.frame * ny * nx;
    double a1 = 0.0, a2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double v = (double)((f[(size_t)(sd.y0 + i / w) * nx + sd.x0 + i % w] - sd.mean) / sd.denom);
        a1 += v;
        a2 = fma(v, v, a2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_down(a1, o, 64);
        a2 += __shfl_down(a2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = a1;
        sh[16 + (threadIdx.x >> 6)] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < 16; ++i) {
            t1 += sh[i];
            t2 += sh[16 + i];
        }
        const double mean = t1 / n;
        tstat[2 * blockIdx.x] = mean;
        tstat[2 * blockIdx.x + 1] = fmax(t2 - n * mean * mean, 0.0);
    }
}

struct NccArgs {
    const float* xc;      // (pairs, ny, nx) shifted circular cross-correlation sum I[p + d] T[p], d = index - (ny/2, nx/2)
    const double* sat1;   // (nshapes, nimg, ny + 1, nx + 1) window sums per template shape (k_sat_rows over the width, k_sat_cols over the height)
    const double* sat2;
    const int* tpl_widx;  // template -> shape slot
    int nimg;
    const int* pair_img;
    const int* pair_tpl;
    const RowSrc* tsrc;   // template descriptors (ROI = reference position)
    const double* tstat;
    float* ncc;           // (pairs, stride): compact (ny - h + 1, nx - w + 1) maps
    float* absncc;
    float* part_val;      // (pairs, nblk)
    int* part_idx;
    int* geom;            // (pairs, 4) map rows, columns, origin (y0, x0) for the epilogue
    int ny, nx, nblk;
    int img_h, img_w;     // extent of the images inside the (ny, nx) canvas (zero beyond)
    // the median of |map| (peak quality): pred[pair] = top-11-bit key bin expected to hold it (k_ncc_sample; 0 = none); the map
    // kernel counts the elements below / inside that bin into sel[pair] and gathers the bin into gathered + pair * stride
    const unsigned* pred;
    SelState* sel;
    float* gathered;
};

// one value of the zero-mean normalised cross-correlation map (element (i, j) of the valid region)
struct NccPair {
    const double *s1, *s2;
    const float* xc;
    double tmean, tssd, vol;
    int h, y0, x0;
    size_t W1;
};
__device__ __forceinline__ NccPair ncc_pair(const NccArgs& p, int pair, const RowSrc& ts) {
    NccPair q;
    const size_t W1 = (size_t)p.nx + 1;
    const size_t slot = (size_t)p.tpl_widx[p.pair_tpl[pair]] * p.nimg + p.pair_img[pair];
    q.s1 = p.sat1 + slot * (p.ny + 1) * W1;
    q.s2 = p.sat2 + slot * (p.ny + 1) * W1;
    q.xc = p.xc + (size_t)pair * p.ny * p.nx;
    q.tmean = p.tstat[2 * p.pair_tpl[pair]];
    q.tssd = p.tstat[2 * p.pair_tpl[pair] + 1];
    q.h = ts.y1 - ts.y0;
    q.vol = (double)q.h * (ts.x1 - ts.x0);
    q.y0 = ts.y0;
    q.x0 = ts.x0;
    q.W1 = W1;
    return q;
}
__device__ __forceinline__ float ncc_value(const NccPair& q, int ny, int nx, int i, int j) {
    const size_t a = (size_t)i * q.W1 + j;
    const double S1 = q.s1[a];   // window sums of the image and of its square under the template at (i, j): one load each
    const double S2 = q.s2[a];
    const int yy = (ny / 2 + i - q.y0) & (ny - 1), xx = (nx / 2 + j - q.x0) & (nx - 1);
    const double num = (double)q.xc[(size_t)yy * nx + xx] - S1 * q.tmean;
    const double den = sqrt(fmax((S2 - S1 * S1 / q.vol) * q.tssd, 0.0));
    // (measured: a float32 reciprocal square root instead of the float64 square root and division is worth 1.5 % -- the kernel
    // waits for its four window-sum loads, not for the arithmetic -- and moves the last bit of the map: not taken)
    return den > 1.1920928955078125e-07 ? (float)(num / den) : 0.f;
}

// The bin of the median of |map|, guessed from NCC_SAMPLES evenly spaced elements.  grid (pairs), block 256.  Exactness never
// depends on the guess: k_ncc_map counts what lies below / inside the bin, k_ncc_check accepts the pair only if the middle rank
// falls inside it, every other pair takes the whole select on its map (k_track_fin).
constexpr int NCC_SAMPLES = 4096;
__global__ void __launch_bounds__(256) k_ncc_sample(NccArgs p, unsigned* __restrict__ pred) {
    __shared__ unsigned hist[1024];
    __shared__ unsigned sh_total;
    const int pair = blockIdx.x;
    const RowSrc ts = p.tsrc[p.pair_tpl[pair]];
    const NccPair q = ncc_pair(p, pair, ts);
    const int hv = p.img_h - q.h + 1, wv = p.img_w - (ts.x1 - ts.x0) + 1;
    const unsigned n = (unsigned)hv * (unsigned)wv, step = max(1u, n / NCC_SAMPLES);
    for (int b = threadIdx.x; b < 1024; b += 256) hist[b] = 0u;
    if (threadIdx.x == 0) sh_total = 0u;
    __syncthreads();
    unsigned mine = 0;
    for (unsigned sidx = threadIdx.x; sidx < NCC_SAMPLES; sidx += 256) {
        const unsigned e = sidx * step;
        if (e >= n) break;
        const float a = fabsf(ncc_value(q, p.ny, p.nx, (int)(e / wv), (int)(e % wv)));
        if (a == a) {
            atomicAdd(&hist[__float_as_uint(a) >> 21], 1u);
            ++mine;
        }
    }
    atomicAdd(&sh_total, mine);
    __syncthreads();
    if (threadIdx.x < 64) {   // lane l scans bins 16 l .. 16 l + 15
        const unsigned total = sh_total, k = total / 2;
        unsigned sum = 0;
        for (int b = 0; b < 16; ++b) sum += hist[16 * threadIdx.x + b];
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += t;
        }
        if (total == 0u && threadIdx.x == 0) pred[pair] = 0u;
        unsigned below = incl - sum;
        if (total != 0u && k >= below && k < incl)
            for (int b = 0; b < 16; ++b) {
                const unsigned c = hist[16 * threadIdx.x + b];
                if (k < below + c) {
                    pred[pair] = 1024u + 16u * threadIdx.x + (unsigned)b;   // f2key of a non-negative float: bits | 0x80000000
                    break;
                }
                below += c;
            }
    }
}

// grid (ceil(pairs / 64)), block 64: the guess holds for pair i when the middle rank falls inside its bin and the whole bin was gathered
__global__ void __launch_bounds__(64) k_ncc_check(const int* __restrict__ geom, const unsigned* __restrict__ pred, SelState* __restrict__ sel,
                                                  int pairs) {
    const int pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= pairs) return;
    SelState* st = sel + pair;
    const unsigned n = (unsigned)geom[4 * pair] * (unsigned)geom[4 * pair + 1];
    const unsigned k = (n & 1u) ? n / 2 : n / 2 - 1;
    st->bin = pred[pair];
    st->ok = (pred[pair] != 0u && k >= st->below && k < st->below + st->count && st->fill == st->count) ? 1u : 0u;
}


// zero-mean normalised cross-correlation at every "valid" window position (signal/tracking.py:157-167: the arithmetic
// of cv2.TM_CCOEFF_NORMED / skimage.match_template).  grid (nblk, pairs), block 256; a workgroup owns at most NCC_STAGE consecutive
// elements: it writes the map and |map|, keeps a first-occurrence arg-max partial, counts the |map| keys below / inside the
// expected median bin and gathers the bin (staged in LDS: ONE returning atomic per workgroup).
constexpr int NCC_STAGE = 4096;
__global__ void __launch_bounds__(256) k_ncc_map(NccArgs p) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float stage[NCC_STAGE];
    __shared__ unsigned cursor, nbelow, gbase;
    const int pair = blockIdx.y;
    const RowSrc ts = p.tsrc[p.pair_tpl[pair]];
    const NccPair q = ncc_pair(p, pair, ts);
    const int hv = p.img_h - q.h + 1, wv = p.img_w - (ts.x1 - ts.x0) + 1, n = hv * wv;
    const size_t fpix = (size_t)p.ny * p.nx;
    const unsigned want = p.pred ? p.pred[pair] : 0u;
    if (threadIdx.x == 0) cursor = nbelow = 0u;
    __syncthreads();
    const int per = (n + gridDim.x - 1) / gridDim.x, e0 = blockIdx.x * per, e1 = min(n, e0 + per);
    const int lane = threadIdx.x & 63;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    unsigned below = 0;
    for (int eb = e0; eb < e1; eb += 256) {   // whole wavefronts stay in the loop (ballot)
        const int e = eb + threadIdx.x;
        bool hit = false;
        float a = 0.f;
        if (e < e1) {
            const float r = ncc_value(q, p.ny, p.nx, e / wv, e % wv);
            a = fabsf(r);
            p.ncc[pair * fpix + e] = r;
            p.absncc[pair * fpix + e] = a;
            argmax_merge(bv, bi, r, e);
            if (want && a == a) {   // NaNs are not ranked (b4d_select.hpp)
                const unsigned key = 1024u + (__float_as_uint(a) >> 21);
                below += key < want ? 1u : 0u;
                hit = key == want;
            }
        }
        if (want) {
            const unsigned long long m = __ballot(hit);
            if (m) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&cursor, (unsigned)__popcll(m));
                base = __shfl(base, 0, 64);
                if (hit) stage[base + __popcll(m & ((1ull << lane) - 1ull))] = a;   // per <= NCC_STAGE: never past the end
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        argmax_merge(bv, bi, ov, oi);
        below += __shfl_down(below, o, 64);
    }
    if (lane == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
        if (below) atomicAdd(&nbelow, below);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) argmax_merge(bv, bi, sv[k], si[k]);
        p.part_val[(size_t)pair * gridDim.x + blockIdx.x] = bv;
        p.part_idx[(size_t)pair * gridDim.x + blockIdx.x] = bi;
        if (blockIdx.x == 0) {
            p.geom[4 * pair] = hv;
            p.geom[4 * pair + 1] = wv;
            p.geom[4 * pair + 2] = ts.y0;
            p.geom[4 * pair + 3] = ts.x0;
        }
        if (want) {
            if (nbelow) atomicAdd(&p.sel[pair].below, nbelow);
            if (cursor) {
                atomicAdd(&p.sel[pair].count, cursor);
                gbase = atomicAdd(&p.sel[pair].fill, cursor);
            }
        }
    }
    __syncthreads();
    if (want) {
        const unsigned cnt = cursor, base = gbase;
        float* out = p.gathered + pair * fpix + base;
        for (unsigned t = threadIdx.x; t < cnt; t += 256) out[t] = stage[t];
    }
}

// ------------------------------------------------------------------------------------ general-length phase correlation
// z-scored ROI of a source frame embedded in a zero (ny, nx) canvas (geometry/roi.py:175-222).  grid (ceil(npix/256), items)
__global__ void __launch_bounds__(256) k_embed_roi(const float* __restrict__ frames, int ny, int nx, const RowSrc* __restrict__ srcs,
                                                   float* __restrict__ canvas) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ny * nx) return;
    const RowSrc sd = srcs[blockIdx.y];
    const int y = e / nx, x = e % nx;
    const bool in = y >= sd.y0 && y < sd.y1 && x >= sd.x0 && x < sd.x1;
    canvas[(size_t)blockIdx.y * ny * nx + e] = in ? (frames[(size_t)sd.frame * ny * nx + e] - sd.mean) / sd.denom : 0.f;
}

// whitened cross-power spectra of `pairs` (image, template) spectrum pairs.  grid (ceil(npix/256), pairs)
__global__ void __launch_bounds__(256) k_gen_cps(const float2* __restrict__ spec, const int* __restrict__ ia, const int* __restrict__ ib,
                                                 int npix, float eps, float2* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npix) return;
    out[(size_t)blockIdx.y * npix + e] =
        cross_power<true>(spec[(size_t)ia[blockIdx.y] * npix + e], spec[(size_t)ib[blockIdx.y] * npix + e], eps);
}

// mag = |Re(R)| * scale, fftshift-ed (signal/tracking.py:283-285; the imaginary part of the Hermitian inverse is rounding
// noise), with per-workgroup first-occurrence arg-max partials over contiguous row-major chunks of the SHIFTED map.
// grid (nblk, pairs), block 256
__global__ void __launch_bounds__(256) k_gen_mag(const float2* __restrict__ R, int ny, int nx, float scale, float* __restrict__ mag,
                                                 float* __restrict__ part_val, int* __restrict__ part_idx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int n = ny * nx, per = (n + gridDim.x - 1) / gridDim.x, e0 = blockIdx.x * per, e1 = min(n, e0 + per);
    const size_t fo = (size_t)blockIdx.y * n;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {   // e: index in the shifted map
        const int y = e / nx, x = e % nx;
        const int sy = (y + ny - ny / 2) % ny, sx = (x + nx - nx / 2) % nx;   // source (unshifted) position
        const float v = fabsf(R[fo + (size_t)sy * nx + sx].x) * scale;
        mag[fo + e] = v;
        argmax_merge(bv, bi, v, e);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        argmax_merge(bv, bi, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = bv;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) argmax_merge(bv, bi, sv[k], si[k]);
        part_val[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = bv;
        part_idx[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = bi;
    }
}

// ------------------------------------------------------------------------------------ |max| normalisation (xcorr2d, normalize="peak")
__global__ void __launch_bounds__(1024) k_absmax_part(const float* __restrict__ x, size_t n, float* __restrict__ part) {
    __shared__ float sh[16];
    const float* f = x + (size_t)blockIdx.y * n;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(f[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, sh[i]);
        part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
}
__global__ void __launch_bounds__(256) k_scale_by_max(float* __restrict__ x, size_t n, const float* __restrict__ part,
                                                      int nparts) {
    __shared__ float s_inv;
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int i = 0; i < nparts; ++i) m = fmaxf(m, part[(size_t)blockIdx.y * nparts + i]);
        s_inv = m > 0.f ? m : 1.f;
    }
    __syncthreads();
    float* f = x + (size_t)blockIdx.y * n;
    const float m = s_inv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) f[i] = f[i] / m;
}

}  // namespace b4d

using namespace b4d;

// ===================================================================================== host
template <int NY, bool WHITEN>
static int launch_prod(const ProdArgs& a, int nt, int pairs, hipStream_t st) {
    using Cfg = ColCfg<NY>;
    if (a.srcs_b) {   // tracking (phase correlation and NCC): template column transforms fused in (TPL)
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_col_prod<NY, WHITEN, true>), Cfg::LDS_BYTES);
        if (rc_lds) return rc_lds;
        hipLaunchKernelGGL((k_col_prod<NY, WHITEN, true>), dim3(nt, pairs), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, a);
        B4D_HIP(hipGetLastError());
        return B4D_OK;
    }
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_col_prod<NY, WHITEN>), Cfg::LDS_BYTES);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL((k_col_prod<NY, WHITEN>), dim3(nt, pairs), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
template <bool WHITEN>
static int dispatch_prod(const b4d_plan* pl, ProdArgs a, int pairs, hipStream_t st) {
    const int nt = (pl->nx / 2) / pl->ct_w;
    a.nt = nt;
    switch (pl->ny) {
        case 64: return launch_prod<64, WHITEN>(a, nt, pairs, st);
        case 128: return launch_prod<128, WHITEN>(a, nt, pairs, st);
        case 256: return launch_prod<256, WHITEN>(a, nt, pairs, st);
        case 512: return launch_prod<512, WHITEN>(a, nt, pairs, st);
        case 1024: return launch_prod<1024, WHITEN>(a, nt, pairs, st);
        case 2048: return launch_prod<2048, WHITEN>(a, nt, pairs, st);
        case 4096: return launch_prod<4096, WHITEN>(a, nt, pairs, st);
    }
    return fail(B4D_ESIZE, "unsupported ny");
}

// forward 2-D half spectra of `items` sources into spec (tile-major) + nyq (items, ny) complex
// columns = false: the tiles stay row-transformed (ROI rows only) -- k_col_prod<.., TPL> transforms them per pair; the Nyquist
// column is transformed either way
static int forward_spectra(const b4d_plan* pl, const float* frames, const RowSrc* srcs, int items, float2* spec,
                           float* nyq_rows, float2* nyq, hipStream_t st, bool columns = true) {
    int rc = dispatch_r2c(pl, frames, items, st, spec, nyq_rows, srcs);
    if (rc) return rc;
    ColArgs ca{};
    ca.spec = spec;
    ca.tw = pl->tw_y;
    ca.nx = pl->nx;
    ca.srcs = srcs;    // zero-embedded ROIs: only the ROI rows exist in `spec` / `nyq_rows`
    if (columns && (rc = dispatch_col<COL_FORWARD>(pl, ca, items, st))) return rc;
    NyqArgs na{};
    na.rows = nyq_rows;
    na.f_out = nyq;
    na.srcs = srcs;
    return dispatch_nyq<NYQ_FORWARD>(pl, na, items, st);
}

namespace {
// byte-carving helper over one hipMalloc'ed arena
struct Arena {
    char* base = nullptr;
    size_t off = 0, cap = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};
}  // namespace

static int track_arena(b4d_plan* pl, size_t bytes, Arena* a) {
    if (bytes > pl->track_bytes) {
        if (pl->track_ws) (void)hipFree(pl->track_ws);
        pl->track_ws = nullptr;
        pl->track_bytes = 0;
        hipError_t e = hipMalloc(&pl->track_ws, bytes);
        if (e != hipSuccess) return fail(B4D_ENOMEM, std::string("tracking workspace: ") + hipGetErrorString(e));
        pl->track_bytes = bytes;
    }
    a->base = static_cast<char*>(pl->track_ws);
    a->off = 0;
    a->cap = pl->track_bytes;
    return B4D_OK;
}

// Fi * conj(Ft) [whitened] -> inverse column pass (tiles + Nyquist column) for `pairs` pairs
template <bool WHITEN>
static int product_inverse(const b4d_plan* pl, const float2* spec, const float2* nyq, const int* idx_a, const int* idx_b,
                           const float2* spec_b, const float2* nyq_b, int pairs, float2* g, float* gnyq, float eps,
                           unsigned flags, hipStream_t st, const RowSrc* srcs_b = nullptr) {
    ProdArgs pa{};
    pa.spec_a = spec;
    pa.spec_b = spec_b;
    pa.idx_a = idx_a;
    pa.idx_b = idx_b;
    pa.g = g;
    pa.tw = pl->tw_y;
    pa.tw2 = pl->tw_y;
    pa.srcs_b = srcs_b;
    pa.eps = eps;
    pa.flags = flags;
    int rc = dispatch_prod<WHITEN>(pl, pa, pairs, st);
    if (rc) return rc;
    NyqArgs na{};
    na.fa = nyq;
    na.fb = nyq_b;
    na.idx_a = idx_a;
    na.idx_b = idx_b;
    na.g_out = gnyq;
    na.eps = eps;
    return dispatch_nyq<WHITEN ? NYQ_PROD_WHITEN : NYQ_PROD>(pl, na, pairs, st);
}

int normalise_by_absmax(float* x, size_t n, int batch, float* scratch, hipStream_t st) {
    hipLaunchKernelGGL(k_absmax_part, dim3(256, batch), dim3(1024), 0, st, x, n, scratch);
    hipLaunchKernelGGL(k_scale_by_max, dim3(1024, batch), dim3(256), 0, st, x, n, scratch, 256);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// phase correlation on a general-length plan (DFT-matrix or fused mixed-radix transforms): same steps as the
// power-of-two path with full complex spectra
// General sizes whose two sides have mixed-radix kernels (b4d_wiener_mr.hip; 2560 x 2160 detector frames ...): half spectra in
// the transposed [k][ky] layout, three passes per pair (product + inverse columns, inverse row pairs -> |corr| map + arg-max
// partials, selection) instead of full-complex DFT-matrix / fused transforms with transposes in between.
static int wmr_phase_correlation(b4d_plan* pl, const float* images, int nimg, const float* tpl_src, const int32_t* tpl_frame,
                                 const int32_t* tpl_roi, int ntpl, const int32_t* pair_img, const int32_t* pair_tpl, int npairs,
                                 int subpixel, double eps, double* out, int32_t* peak_ij, hipStream_t st) {
    const int ny = pl->ny, nx = pl->nx, nsrc = nimg + ntpl, hp = (ny + 1) / 2, qpf = wmr_quads_per_frame(ny);
    const int predict_mode = g_opt_track_predict.load();   // one route per call
    const size_t npix = (size_t)ny * nx, selems = wmr_spectrum_elems(ny, nx);
    const int sc = std::max(1, pl->chunk), pc = std::max(1, std::min(npairs, pl->chunk));
    size_t need = 0;
    auto add = [&](size_t b) { need += ((b + 255) & ~(size_t)255) + 256; };
    add(sizeof(float2) * selems * nsrc);
    add(sizeof(float) * npix * sc);
    add(sizeof(float) * (size_t)hp * sc);
    add(sizeof(RowSrc) * nsrc);
    add(sizeof(double) * 2 * ROI_SPLIT * nsrc);
    add(sizeof(int) * 2 * (size_t)npairs);
    add(sizeof(float2) * selems * pc);
    add(sizeof(float) * npix * pc);
    add(sizeof(float) * npix * pc);
    add(sizeof(float) * (size_t)qpf * pc);
    add(sizeof(int) * (size_t)qpf * pc);
    add(sizeof(unsigned) * SEL_WORDS * (size_t)pc);
    Arena ar;
    int rc = track_arena(pl, need, &ar);
    if (rc) return rc;
    float2* spec = ar.take<float2>(selems * nsrc);
    float* canvas = ar.take<float>(npix * sc);
    float* scratch = ar.take<float>((size_t)hp * sc);
    RowSrc* srcs = ar.take<RowSrc>(nsrc);
    double* roi_part = ar.take<double>((size_t)2 * ROI_SPLIT * nsrc);
    int* pidx = ar.take<int>(2 * (size_t)npairs);
    float2* G = ar.take<float2>(selems * pc);
    float* mag = ar.take<float>(npix * pc);
    float* medws = ar.take<float>(npix * pc);
    float* pval = ar.take<float>((size_t)qpf * pc);
    int* pind = ar.take<int>((size_t)qpf * pc);
    SelState* msel = reinterpret_cast<SelState*>(ar.take<unsigned>((size_t)SEL_WORDS * pc));
    std::vector<RowSrc> h(nsrc);
    for (int i = 0; i < nimg; ++i) h[i] = RowSrc{i, 0, ny, 0, nx, 0.f, 1.f, 0};
    for (int k = 0; k < ntpl; ++k)
        h[nimg + k] = RowSrc{tpl_frame[k], tpl_roi[4 * k], tpl_roi[4 * k + 1], tpl_roi[4 * k + 2], tpl_roi[4 * k + 3], 0.f, 1.f, 0};
    std::vector<int> hpi(2 * (size_t)npairs);
    for (int i = 0; i < npairs; ++i) {
        hpi[i] = pair_img[i];
        hpi[npairs + i] = nimg + pair_tpl[i];
    }
    B4D_HIP(hipMemcpyAsync(srcs, h.data(), sizeof(RowSrc) * nsrc, hipMemcpyHostToDevice, st));
    B4D_HIP(hipMemcpyAsync(pidx, hpi.data(), sizeof(int) * hpi.size(), hipMemcpyHostToDevice, st));
    B4D_HIP(hipStreamSynchronize(st));
    if ((rc = roi_stats(images, ny, nx, eps, srcs, nimg, roi_part, st))) return rc;
    if ((rc = roi_stats(tpl_src, ny, nx, eps, srcs + nimg, ntpl, roi_part + (size_t)2 * ROI_SPLIT * nimg, st))) return rc;
    const dim3 eg((unsigned)((npix + 255) / 256));
    for (int s0 = 0; s0 < nsrc;) {   // spectra, once per distinct image / template; never straddle the two frame arrays
        const int n = std::min(sc, (s0 < nimg ? nimg : nsrc) - s0);
        hipLaunchKernelGGL(k_embed_roi, dim3(eg.x, n), dim3(256), 0, st, s0 < nimg ? images : tpl_src, ny, nx, srcs + s0, canvas);
        B4D_HIP(hipGetLastError());
        if ((rc = wmr_forward_spectra(canvas, n, ny, nx, pl->tw_x, pl->tw_y, spec + selems * s0, scratch, st))) return rc;
        s0 += n;
    }
    for (int p0 = 0; p0 < npairs; p0 += pc) {
        const int np = std::min(pc, npairs - p0);
        if ((rc = wmr_product_inverse(spec, spec, pidx + p0, pidx + npairs + p0, np, ny, nx, pl->tw_y, G, 1, (float)eps, 0u, st))) return rc;
        const unsigned pred = predicted_median_bin(npix, predict_mode);
        B4D_HIP(hipMemsetAsync(msel, 0, sizeof(SelState) * (size_t)pc, st));
        if ((rc = wmr_rows_magnitude(G, np, ny, nx, pl->tw_x, mag, pval, pind, reinterpret_cast<unsigned*>(msel), SEL_WORDS, pred, medws, st)))
            return rc;
        FinArgs fa{};
        fa.mag = mag;
        fa.compact = medws;
        fa.part_val = pval;
        fa.part_idx = pind;
        fa.out = out + (size_t)p0 * 4;
        fa.peak_ij = peak_ij ? peak_ij + (size_t)p0 * 2 : nullptr;
        fa.ny = ny;
        fa.nx = nx;
        fa.nblk = qpf;
        fa.subpixel = subpixel;
        fa.eps = eps;
        if ((rc = launch_track_fin2(fa, msel, pred, np, st))) return rc;
        if ((rc = launch_track_fin_rest(fa, msel, np, st))) return rc;
    }
    return B4D_OK;
}

static int general_phase_correlation(b4d_plan* pl, const float* images, int nimg, const float* tpl_src, const int32_t* tpl_frame,
                                     const int32_t* tpl_roi, int ntpl, const int32_t* pair_img, const int32_t* pair_tpl, int npairs,
                                     int subpixel, double eps, double* out, int32_t* peak_ij, hipStream_t st) {
    if (pl->wmr)
        return wmr_phase_correlation(pl, images, nimg, tpl_src, tpl_frame, tpl_roi, ntpl, pair_img, pair_tpl, npairs, subpixel, eps, out,
                                     peak_ij, st);
    const int ny = pl->ny, nx = pl->nx, npix = ny * nx, nsrc = nimg + ntpl, nblk = 256;
    const int pc = std::max(1, std::min(npairs, pl->chunk));
    size_t need = 0;
    auto add = [&](size_t b) { need += ((b + 255) & ~(size_t)255) + 256; };
    add(sizeof(float2) * (size_t)npix * nsrc);
    add(sizeof(float) * (size_t)npix * pl->chunk);
    add(sizeof(RowSrc) * nsrc);
    add(sizeof(double) * 2 * ROI_SPLIT * nsrc);
    add(sizeof(int) * 2 * (size_t)npairs);
    add(sizeof(float) * (size_t)npix * pc);
    add(sizeof(float) * (size_t)nblk * pc);
    add(sizeof(int) * (size_t)nblk * pc);
    Arena ar;
    int rc = track_arena(pl, need, &ar);
    if (rc) return rc;
    float2* spec = ar.take<float2>((size_t)npix * nsrc);
    float* canvas = ar.take<float>((size_t)npix * pl->chunk);
    RowSrc* srcs = ar.take<RowSrc>(nsrc);
    double* roi_part = ar.take<double>((size_t)2 * ROI_SPLIT * nsrc);
    int* pidx = ar.take<int>(2 * (size_t)npairs);
    float* mag = ar.take<float>((size_t)npix * pc);
    float* pval = ar.take<float>((size_t)nblk * pc);
    int* pind = ar.take<int>((size_t)nblk * pc);
    std::vector<RowSrc> h(nsrc);
    for (int i = 0; i < nimg; ++i) h[i] = RowSrc{i, 0, ny, 0, nx, 0.f, 1.f, 0};
    for (int k = 0; k < ntpl; ++k)
        h[nimg + k] = RowSrc{tpl_frame[k], tpl_roi[4 * k], tpl_roi[4 * k + 1], tpl_roi[4 * k + 2], tpl_roi[4 * k + 3], 0.f, 1.f, 0};
    std::vector<int> hp(2 * (size_t)npairs);
    for (int i = 0; i < npairs; ++i) {
        hp[i] = pair_img[i];
        hp[npairs + i] = nimg + pair_tpl[i];
    }
    B4D_HIP(hipMemcpyAsync(srcs, h.data(), sizeof(RowSrc) * nsrc, hipMemcpyHostToDevice, st));
    B4D_HIP(hipMemcpyAsync(pidx, hp.data(), sizeof(int) * hp.size(), hipMemcpyHostToDevice, st));
    B4D_HIP(hipStreamSynchronize(st));
    if ((rc = roi_stats(images, ny, nx, eps, srcs, nimg, roi_part, st))) return rc;
    if ((rc = roi_stats(tpl_src, ny, nx, eps, srcs + nimg, ntpl, roi_part + (size_t)2 * ROI_SPLIT * nimg, st))) return rc;
    const dim3 eg((npix + 255) / 256);
    for (int s0 = 0; s0 < nsrc;) {   // spectra, once per distinct image / template; never straddle the two frame arrays
        const int n = std::min(pl->chunk, (s0 < nimg ? nimg : nsrc) - s0);
        hipLaunchKernelGGL(k_embed_roi, dim3(eg.x, n), dim3(256), 0, st, s0 < nimg ? images : tpl_src, ny, nx, srcs + s0, canvas);
        B4D_HIP(hipGetLastError());
        if ((rc = general_dft2(pl, canvas, true, n, 0, pl->gbuf1, spec + (size_t)s0 * npix, st))) return rc;
        s0 += n;
    }
    for (int p0 = 0; p0 < npairs; p0 += pc) {
        const int np = std::min(pc, npairs - p0);
        hipLaunchKernelGGL(k_gen_cps, dim3(eg.x, np), dim3(256), 0, st, spec, pidx + p0, pidx + npairs + p0, npix, (float)eps, pl->gbuf2);
        B4D_HIP(hipGetLastError());
        if ((rc = general_dft2(pl, pl->gbuf2, false, np, 1, pl->gbuf1, pl->gbuf3, st))) return rc;
        hipLaunchKernelGGL(k_gen_mag, dim3(nblk, np), dim3(256), 0, st, pl->gbuf3, ny, nx, 1.0f / ((float)nx * (float)ny), mag, pval, pind);
        FinArgs fa{};
        fa.mag = mag;
        fa.part_val = pval;
        fa.part_idx = pind;
        fa.out = out + (size_t)p0 * 4;
        fa.peak_ij = peak_ij ? peak_ij + (size_t)p0 * 2 : nullptr;
        fa.ny = ny;
        fa.nx = nx;
        fa.nblk = nblk;
        fa.subpixel = subpixel;
        fa.eps = eps;
        if ((rc = launch_track_fin(fa, np, st))) return rc;
        B4D_HIP(hipGetLastError());
    }
    return B4D_OK;
}

extern "C" {

int b4d_xcorr2d(b4d_plan* pl, const float* a, const float* b, int batch, float* corr, unsigned flags, void* stream) {
    if (!pl || !a || !b || !corr) return fail(B4D_EINVAL, "null argument");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    B4D_PLAN_LOCK(pl);
    hipStream_t st = (hipStream_t)stream;
    if (pl->general) return general_xcorr(pl, a, b, batch, corr, flags, st);
    const size_t fpix = (size_t)pl->ny * pl->nx, half = fpix / 2, ny = pl->ny;
    const int chunk = pl->chunk;
    size_t need = 0;
    auto add = [&](size_t bytes) { need += ((bytes + 255) & ~(size_t)255) + 256; };
    for (int i = 0; i < 3; ++i) add(sizeof(float2) * half * chunk);
    for (int i = 0; i < 2; ++i) add(sizeof(float2) * ny * chunk);
    for (int i = 0; i < 3; ++i) add(sizeof(float) * ny * chunk);
    add(sizeof(float) * 256 * chunk);
    Arena ar;
    int rc = track_arena(pl, need, &ar);
    if (rc) return rc;
    float2* sa = ar.take<float2>(half * chunk);
    float2* sb = ar.take<float2>(half * chunk);
    float2* g = ar.take<float2>(half * chunk);
    float2* fa = ar.take<float2>(ny * chunk);
    float2* fb = ar.take<float2>(ny * chunk);
    float* ra_rows = ar.take<float>(ny * chunk);
    float* rb_rows = ar.take<float>(ny * chunk);
    float* gnyq = ar.take<float>(ny * chunk);
    float* part0 = ar.take<float>((size_t)256 * chunk);
    // two-lane launch groups (Lanes, b4d_fft2d.hpp): 2048^2 16.3 -> 17.1 k pairs/s, 1024^2 50.5 -> 54.7 k (tools/dev_xcorr_chunk.py);
    // lane l works in slot l (sub items) of every chunk buffer
    // (groups of half the plan's chunk: the column kernels of this route want thousands of tiles per launch, 4-frame groups lose)
    Lanes ln;
    const bool two = chunk >= 2 && batch >= 2 && (size_t)batch * fpix * sizeof(float) >= ((size_t)32 << 20);
    if ((rc = ln.fork(pl, st, two))) return rc;
    ln.sub = ln.two ? std::max(1, std::min(chunk / 2, (batch + 1) / 2)) : chunk;
    int grp = 0;
    for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += ln.sub, ++grp) {
        const int nb = std::min(ln.sub, batch - b0);
        const size_t so = (size_t)ln.slot(grp) * ln.sub;
        hipStream_t ls = ln.stream(grp);
        float2 *sa_ = sa + half * so, *sb_ = sb + half * so, *g_ = g + half * so, *fa_ = fa + ny * so, *fb_ = fb + ny * so;
        float *ra_ = ra_rows + ny * so, *rb_ = rb_rows + ny * so, *gn_ = gnyq + ny * so, *part = part0 + 256 * so;
        if ((rc = forward_spectra(pl, a + b0 * fpix, nullptr, nb, sa_, ra_, fa_, ls))) break;
        if ((rc = forward_spectra(pl, b + b0 * fpix, nullptr, nb, sb_, rb_, fb_, ls))) break;
        if ((rc = product_inverse<false>(pl, sa_, fa_, nullptr, nullptr, sb_, fb_, nb, g_, gn_, 0.f, flags, ls))) break;
        RowOutArgs ra{};
        ra.g = g_;
        ra.gnyq = gn_;
        ra.out = corr + b0 * fpix;
        ra.tw = pl->tw_x;
        ra.scale = 1.0f / ((float)pl->nx * (float)pl->ny);
        ra.ny = pl->ny;
        ra.ct_w = pl->ct_w;
        ra.flags = 0;
        if ((rc = dispatch_c2r(pl, ra, nb, ls, C2R_OUT))) break;
        if (flags & B4D_NORM_PEAK)
            if ((rc = normalise_by_absmax(corr + b0 * fpix, fpix, nb, part, ls))) break;
    }
    const int rj = ln.close();
    return rc ? rc : rj;
}

int b4d_phase_correlation(b4d_plan* pl, const float* images, int nimg, const float* tpl_src, int ntplsrc,
                          const int32_t* tpl_frame, const int32_t* tpl_roi, int ntpl, const int32_t* pair_img,
                          const int32_t* pair_tpl, int npairs, int subpixel, double eps, double* out, int32_t* peak_ij,
                          void* stream) {
    if (!pl || !images || !tpl_src || !tpl_frame || !tpl_roi || !pair_img || !pair_tpl || !out)
        return fail(B4D_EINVAL, "null argument");
    if (nimg < 1 || ntplsrc < 1 || ntpl < 1 || npairs < 1) return fail(B4D_EINVAL, "counts must be >= 1");
    B4D_PLAN_LOCK(pl);
    const int ny = pl->ny, nx = pl->nx;
    const int predict_mode = g_opt_track_predict.load();   // one route per call
    for (int k = 0; k < ntpl; ++k) {
        const int32_t* r = tpl_roi + 4 * k;
        if (tpl_frame[k] < 0 || tpl_frame[k] >= ntplsrc || r[0] < 0 || r[1] > ny || r[0] >= r[1] || r[2] < 0 || r[3] > nx ||
            r[2] >= r[3])
            return fail(B4D_EINVAL, "template " + std::to_string(k) + ": frame or ROI out of range");
    }
    for (int i = 0; i < npairs; ++i)
        if (pair_img[i] < 0 || pair_img[i] >= nimg || pair_tpl[i] < 0 || pair_tpl[i] >= ntpl)
            return fail(B4D_EINVAL, "pair " + std::to_string(i) + ": index out of range");
    hipStream_t st = (hipStream_t)stream;
    if (pl->general)
        return general_phase_correlation(pl, images, nimg, tpl_src, tpl_frame, tpl_roi, ntpl, pair_img, pair_tpl, npairs, subpixel, eps,
                                         out, peak_ij, st);
    const size_t fpix = (size_t)ny * nx, half = fpix / 2;
    // pairs per launch group; a call of more than one group runs on two lanes (Lanes, b4d_fft2d.hpp: alternate groups on the
    // caller's stream and the library's second one, each with its own slot of the per-group buffers)
    const int pc_one = std::max(1, std::min(npairs, pl->chunk * 4));
    const bool two = npairs > pc_one / 2 && pc_one >= 64 && g_opt_lanes.load() != 0;   // cfg3 (1152 pairs, groups of 192): 226 -> 231 k pairs/s
    const int pc = two ? (pc_one + 1) / 2 : pc_one, nslot = two ? 2 : 1;
    const int nsrc = nimg + ntpl;
    size_t need = 0;
    auto add = [&](size_t b) { need += ((b + 255) & ~(size_t)255) + 256; };
    add(sizeof(float2) * half * nsrc);        // spectra
    add(sizeof(float2) * (size_t)ny * nsrc);  // Nyquist column spectra
    add(sizeof(float) * (size_t)ny * nsrc);   // Nyquist bins after the row pass
    add(sizeof(RowSrc) * nsrc);
    add(sizeof(double) * 2 * ROI_SPLIT * nsrc);
    add(sizeof(int) * 2 * (size_t)npairs);
    add(sizeof(float2) * half * pc * nslot);          // G
    add(sizeof(float) * (size_t)ny * pc * nslot);     // G of the Nyquist column
    add(sizeof(float) * fpix * pc * nslot);           // magnitude maps
    add(sizeof(float) * fpix * pc * nslot);           // median scratch (gathered bin)
    add(sizeof(float) * 2048 * (size_t)pc * nslot);
    add(sizeof(int) * 2048 * (size_t)pc * nslot);
    add(sizeof(unsigned) * SEL_WORDS * (size_t)pc * nslot);   // select state of the median
    Arena ar;
    int rc = track_arena(pl, need, &ar);
    if (rc) return rc;
    float2* spec = ar.take<float2>(half * nsrc);
    float2* nyq = ar.take<float2>((size_t)ny * nsrc);
    float* nyq_rows = ar.take<float>((size_t)ny * nsrc);
    RowSrc* srcs = ar.take<RowSrc>(nsrc);
    double* roi_part = ar.take<double>((size_t)2 * ROI_SPLIT * nsrc);
    int* pidx = ar.take<int>(2 * (size_t)npairs);
    float2* g0 = ar.take<float2>(half * pc * nslot);
    float* gnyq0 = ar.take<float>((size_t)ny * pc * nslot);
    float* mag0 = ar.take<float>(fpix * pc * nslot);
    float* medws0 = ar.take<float>(fpix * pc * nslot);
    float* pval0 = ar.take<float>((size_t)2048 * pc * nslot);
    int* pind0 = ar.take<int>((size_t)2048 * pc * nslot);
    SelState* msel0 = reinterpret_cast<SelState*>(ar.take<unsigned>((size_t)SEL_WORDS * pc * nslot));

    // ---- source descriptors: images (full frame, z-scored), then templates (ROI, z-scored, zero elsewhere)
    std::vector<RowSrc> h(nsrc);
    for (int i = 0; i < nimg; ++i) h[i] = RowSrc{i, 0, ny, 0, nx, 0.f, 1.f, 0};
    for (int k = 0; k < ntpl; ++k)
        h[nimg + k] = RowSrc{tpl_frame[k], tpl_roi[4 * k], tpl_roi[4 * k + 1], tpl_roi[4 * k + 2], tpl_roi[4 * k + 3], 0.f, 1.f, 0};
    std::vector<int> hp(2 * (size_t)npairs);
    for (int i = 0; i < npairs; ++i) {
        hp[i] = pair_img[i];
        hp[npairs + i] = nimg + pair_tpl[i];
    }
    B4D_HIP(hipMemcpyAsync(srcs, h.data(), sizeof(RowSrc) * nsrc, hipMemcpyHostToDevice, st));
    B4D_HIP(hipMemcpyAsync(pidx, hp.data(), sizeof(int) * hp.size(), hipMemcpyHostToDevice, st));
    B4D_HIP(hipStreamSynchronize(st));  // h / hp are stack-owned
    if ((rc = roi_stats(images, ny, nx, eps, srcs, nimg, roi_part, st))) return rc;
    if ((rc = roi_stats(tpl_src, ny, nx, eps, srcs + nimg, ntpl, roi_part + (size_t)2 * ROI_SPLIT * nimg, st))) return rc;
    // ---- spectra (once per distinct image / template)
    const int fc = std::max(1, pl->chunk * 2);
    for (int i0 = 0; i0 < nimg; i0 += fc) {
        const int n = std::min(fc, nimg - i0);
        if ((rc = forward_spectra(pl, images, srcs + i0, n, spec + half * i0, nyq_rows + (size_t)ny * i0,
                                  nyq + (size_t)ny * i0, st)))
            return rc;
    }
    for (int k0 = 0; k0 < ntpl; k0 += fc) {
        const int n = std::min(fc, ntpl - k0), o = nimg + k0;
        if ((rc = forward_spectra(pl, tpl_src, srcs + o, n, spec + half * o, nyq_rows + (size_t)ny * o,
                                  nyq + (size_t)ny * o, st, /*columns=*/false)))   // transformed per pair in k_col_prod<.., TPL>
            return rc;
    }
    // ---- pairs
    Lanes ln;
    if ((rc = ln.fork(pl, st, two))) return rc;
    int grp = 0;
    for (int p0 = 0; p0 < npairs; p0 += pc, ++grp) {
        const int np = std::min(pc, npairs - p0), slot = ln.slot(grp);
        hipStream_t ls = ln.stream(grp);
        float2* g = g0 + half * pc * slot;
        float* gnyq = gnyq0 + (size_t)ny * pc * slot;
        float* mag = mag0 + fpix * pc * slot;
        float* medws = medws0 + fpix * pc * slot;
        float* pval = pval0 + (size_t)2048 * pc * slot;
        int* pind = pind0 + (size_t)2048 * pc * slot;
        SelState* msel = msel0 + (size_t)pc * slot;
        if ((rc = product_inverse<true>(pl, spec, nyq, pidx + p0, pidx + npairs + p0, spec, nyq, np, g, gnyq, (float)eps,
                                        0u, ls, srcs)))
            return rc;
        RowOutArgs ra{};
        ra.g = g;
        ra.gnyq = gnyq;
        ra.out = mag;
        ra.tw = pl->tw_x;
        ra.scale = 1.0f / ((float)nx * (float)ny);
        ra.ny = ny;
        ra.ct_w = pl->ct_w;
        ra.part_val = pval;
        ra.part_idx = pind;
        const unsigned pred = predicted_median_bin(fpix, predict_mode);
        ra.selw = reinterpret_cast<unsigned*>(msel);
        ra.sel_stride = SEL_WORDS;
        ra.pred_bin = pred;
        ra.compact = medws;
        B4D_HIP(hipMemsetAsync(msel, 0, sizeof(SelState) * (size_t)pc, ls));   // counts, cursors, verdicts
        // With an expected median bin the common path never reads the map itself: partials, counts and the gathered bin come
        // out of the row pass, the 3 x 3 Taylor neighbourhood from three row pairs recomputed around the peak (C2R_ROWS).  The
        // 4 ny nx bytes per pair are written only for the pairs the expectation fails on (gated second pass below).
        const bool nomap = pred != 0u;
        ra.nomap = nomap ? 1 : 0;
        int nblk = 0;
        if ((rc = dispatch_c2r(pl, ra, np, ls, C2R_MAG, nullptr, &nblk))) return rc;
        if (nomap) {
            RowOutArgs rr = ra;
            rr.nblk = nblk;
            if ((rc = dispatch_c2r(pl, rr, np, ls, C2R_ROWS))) return rc;
        }
        FinArgs fa{};
        fa.mag = mag;
        fa.compact = medws;
        fa.part_val = pval;
        fa.part_idx = pind;
        fa.out = out + (size_t)p0 * 4;
        fa.peak_ij = peak_ij ? peak_ij + (size_t)p0 * 2 : nullptr;
        fa.ny = ny;
        fa.nx = nx;
        fa.nblk = nblk;
        fa.subpixel = subpixel;
        fa.eps = eps;
        fa.partial_map = nomap ? 1 : 0;
        if ((rc = launch_track_fin2(fa, msel, pred, np, ls))) return rc;
        if (nomap) {   // full maps for the pairs left over (verdict word of the select state != 0: nothing to do)
            RowOutArgs rf = ra;
            rf.nomap = 0;
            rf.selw = nullptr;
            rf.gate = &msel[0].ok;
            rf.gate_stride = SEL_WORDS;
            if ((rc = dispatch_c2r(pl, rf, np, ls, C2R_MAG))) return rc;
        }
        if ((rc = launch_track_fin_rest(fa, msel, np, ls))) return rc;
        B4D_HIP(hipGetLastError());
    }
    if ((rc = ln.close())) return rc;
    return B4D_OK;
}

int b4d_template_match(b4d_plan* pl, const float* images, int nimg, const float* tpl_src, int ntplsrc,
                       const int32_t* tpl_frame, const int32_t* tpl_roi, int ntpl, const int32_t* pair_img,
                       const int32_t* pair_tpl, int npairs, int img_h, int img_w, int zscore_image, int subpixel, double eps,
                       double* out, int32_t* peak_ij, void* stream) {
    if (!pl || !images || !tpl_src || !tpl_frame || !tpl_roi || !pair_img || !pair_tpl || !out)
        return fail(B4D_EINVAL, "null argument");
    if (nimg < 1 || ntplsrc < 1 || ntpl < 1 || npairs < 1) return fail(B4D_EINVAL, "counts must be >= 1");
    B4D_PLAN_LOCK(pl);
    if (pl->general) return fail(B4D_ESIZE, "template matching needs a power-of-two canvas: ny, nx in [64, 4096]");
    const int ny = pl->ny, nx = pl->nx;
    if (img_h <= 0) img_h = ny;
    if (img_w <= 0) img_w = nx;
    if (img_h > ny || img_w > nx) return fail(B4D_EINVAL, "image extent exceeds the canvas");
    for (int k = 0; k < ntpl; ++k) {
        const int32_t* r = tpl_roi + 4 * k;
        if (tpl_frame[k] < 0 || tpl_frame[k] >= ntplsrc || r[0] < 0 || r[1] > img_h || r[0] >= r[1] || r[2] < 0 || r[3] > img_w ||
            r[2] >= r[3])
            return fail(B4D_EINVAL, "template " + std::to_string(k) + ": frame or ROI out of range");
    }
    for (int i = 0; i < npairs; ++i)
        if (pair_img[i] < 0 || pair_img[i] >= nimg || pair_tpl[i] < 0 || pair_tpl[i] >= ntpl)
            return fail(B4D_EINVAL, "pair " + std::to_string(i) + ": index out of range");
    hipStream_t st = (hipStream_t)stream;
    const size_t fpix = (size_t)ny * nx, half = fpix / 2, satn = (size_t)(ny + 1) * (nx + 1);
    const int pc = std::max(1, std::min(npairs, pl->chunk * 2));
    // workgroups per map: at most NCC_STAGE elements each (the gathered median bin is staged in LDS)
    const int nsrc = nimg + ntpl, nblk = std::max(256, (int)(((size_t)img_h * img_w + NCC_STAGE - 1) / NCC_STAGE));
    size_t need = 0;
    auto add = [&](size_t b) { need += ((b + 255) & ~(size_t)255) + 256; };
    add(sizeof(float2) * half * nsrc);
    add(sizeof(float2) * (size_t)ny * nsrc);
    add(sizeof(float) * (size_t)ny * nsrc);
    add(sizeof(RowSrc) * nsrc);
    add(sizeof(double) * 2 * ROI_SPLIT * nsrc);
    add(sizeof(int) * 3 * (size_t)npairs);
    std::vector<int> widths, heights, widx(ntpl);   // distinct template shapes: one pair of window-sum tables per (shape, image)
    for (int k = 0; k < ntpl; ++k) {
        const int w = tpl_roi[4 * k + 3] - tpl_roi[4 * k + 2], hh = tpl_roi[4 * k + 1] - tpl_roi[4 * k];
        size_t j = 0;
        while (j < widths.size() && (widths[j] != w || heights[j] != hh)) ++j;
        if (j == widths.size()) {
            widths.push_back(w);
            heights.push_back(hh);
        }
        widx[k] = (int)j;
    }
    const size_t nw = widths.size();
    add(sizeof(double) * satn * nimg * nw);
    add(sizeof(double) * satn * nimg * nw);
    add(sizeof(int) * (size_t)ntpl);
    add(sizeof(double) * 2 * (size_t)ntpl);
    add(sizeof(float2) * half * pc);
    add(sizeof(float) * (size_t)ny * pc);
    for (int i = 0; i < 4; ++i) add(sizeof(float) * fpix * pc);
    add(sizeof(float) * (size_t)nblk * pc);
    add(sizeof(int) * (size_t)nblk * pc);
    add(sizeof(int) * 4 * (size_t)pc);
    add(sizeof(unsigned) * (size_t)pc);
    add(sizeof(unsigned) * SEL_WORDS * (size_t)pc);
    Arena ar;
    int rc = track_arena(pl, need, &ar);
    if (rc) return rc;
    float2* spec = ar.take<float2>(half * nsrc);
    float2* nyq = ar.take<float2>((size_t)ny * nsrc);
    float* nyq_rows = ar.take<float>((size_t)ny * nsrc);
    RowSrc* srcs = ar.take<RowSrc>(nsrc);
    double* roi_part = ar.take<double>((size_t)2 * ROI_SPLIT * nsrc);
    int* pidx = ar.take<int>(3 * (size_t)npairs);
    int* sidx = pidx + 2 * (size_t)npairs;
    double* sat1 = ar.take<double>(satn * nimg * nw);
    double* sat2 = ar.take<double>(satn * nimg * nw);
    int* d_widx = ar.take<int>((size_t)ntpl);
    double* tstat = ar.take<double>(2 * (size_t)ntpl);
    float2* g = ar.take<float2>(half * pc);
    float* gnyq = ar.take<float>((size_t)ny * pc);
    float* xc = ar.take<float>(fpix * pc);
    float* ncc = ar.take<float>(fpix * pc);
    float* absncc = ar.take<float>(fpix * pc);
    float* pval = ar.take<float>((size_t)nblk * pc);
    int* pind = ar.take<int>((size_t)nblk * pc);
    int* geom = ar.take<int>(4 * (size_t)pc);
    float* gathered = ar.take<float>(fpix * pc);
    unsigned* pred = ar.take<unsigned>((size_t)pc);
    SelState* msel = reinterpret_cast<SelState*>(ar.take<unsigned>((size_t)SEL_WORDS * pc));

    std::vector<RowSrc> h(nsrc);
    for (int i = 0; i < nimg; ++i) h[i] = RowSrc{i, 0, img_h, 0, img_w, 0.f, 1.f, 0};
    for (int k = 0; k < ntpl; ++k)
        h[nimg + k] = RowSrc{tpl_frame[k], tpl_roi[4 * k], tpl_roi[4 * k + 1], tpl_roi[4 * k + 2], tpl_roi[4 * k + 3], 0.f, 1.f, 0};
    std::vector<int> hp(3 * (size_t)npairs);
    for (int i = 0; i < npairs; ++i) {
        hp[i] = pair_img[i];
        hp[npairs + i] = pair_tpl[i];
        hp[2 * (size_t)npairs + i] = nimg + pair_tpl[i];  // templates sit after the images in `spec`
    }
    B4D_HIP(hipMemcpyAsync(srcs, h.data(), sizeof(RowSrc) * nsrc, hipMemcpyHostToDevice, st));
    B4D_HIP(hipMemcpyAsync(pidx, hp.data(), sizeof(int) * hp.size(), hipMemcpyHostToDevice, st));
    B4D_HIP(hipMemcpyAsync(d_widx, widx.data(), sizeof(int) * ntpl, hipMemcpyHostToDevice, st));
    B4D_HIP(hipStreamSynchronize(st));
    // "opencv": the image is z-scored as a whole (tracking.py:157); "skimage": raw float32 image (tracking.py:166)
    if (zscore_image && (rc = roi_stats(images, ny, nx, eps, srcs, nimg, roi_part, st))) return rc;
    if ((rc = roi_stats(tpl_src, ny, nx, eps, srcs + nimg, ntpl, roi_part + (size_t)2 * ROI_SPLIT * nimg, st))) return rc;
    hipLaunchKernelGGL(k_tpl_stats, dim3(ntpl), dim3(1024), 0, st, tpl_src, ny, nx, srcs + nimg, tstat);
    for (size_t j = 0; j < nw; ++j) {
        const size_t row_lds = sizeof(double) * 2 * ((size_t)nx + 1);
        if ((rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_sat_rows), row_lds))) return rc;
        hipLaunchKernelGGL(k_sat_rows, dim3(ny, nimg), dim3(256), row_lds, st, images, ny, nx, srcs, sat1 + j * satn * nimg,
                           sat2 + j * satn * nimg, widths[j]);
        hipLaunchKernelGGL(k_sat_cols, dim3((nx + 64) / 64, nimg), dim3(64), 0, st, ny, nx, sat1 + j * satn * nimg, sat2 + j * satn * nimg,
                           heights[j]);
    }
    B4D_HIP(hipGetLastError());
    const int fc = std::max(1, pl->chunk * 2);
    for (int i0 = 0; i0 < nimg; i0 += fc) {
        const int n = std::min(fc, nimg - i0);
        if ((rc = forward_spectra(pl, images, srcs + i0, n, spec + half * i0, nyq_rows + (size_t)ny * i0, nyq + (size_t)ny * i0, st)))
            return rc;
    }
    for (int k0 = 0; k0 < ntpl; k0 += fc) {
        const int n = std::min(fc, ntpl - k0), o = nimg + k0;
        if ((rc = forward_spectra(pl, tpl_src, srcs + o, n, spec + half * o, nyq_rows + (size_t)ny * o, nyq + (size_t)ny * o, st,
                                  /*columns=*/false)))   // transformed per pair in k_col_prod<.., TPL>
            return rc;
    }
    {
        for (int p0 = 0; p0 < npairs; p0 += pc) {
            const int np = std::min(pc, npairs - p0);
            int r2 = product_inverse<false>(pl, spec, nyq, pidx + p0, sidx + p0, spec, nyq, np, g, gnyq, 0.f, 0u, st, srcs);
            if (r2) return r2;
            RowOutArgs ra{};
            ra.g = g;
            ra.gnyq = gnyq;
            ra.out = xc;
            ra.tw = pl->tw_x;
            ra.scale = 1.0f / ((float)nx * (float)ny);
            ra.ny = ny;
            ra.ct_w = pl->ct_w;
            if ((r2 = dispatch_c2r(pl, ra, np, st, C2R_OUT))) return r2;
            NccArgs na{};
            na.xc = xc;
            na.sat1 = sat1;
            na.sat2 = sat2;
            na.pair_img = pidx + p0;
            na.pair_tpl = pidx + npairs + p0;
            na.tsrc = srcs + nimg;
            na.tstat = tstat;
            na.tpl_widx = d_widx;
            na.nimg = nimg;
            na.ncc = ncc;
            na.absncc = absncc;
            na.part_val = pval;
            na.part_idx = pind;
            na.geom = geom;
            na.ny = ny;
            na.nx = nx;
            na.nblk = nblk;
            na.img_h = img_h;
            na.img_w = img_w;
            const bool expect = g_opt_track_predict.load() != 0;   // "track_predict_bin" 0: the whole select on every map
            na.pred = expect ? pred : nullptr;
            na.sel = msel;
            na.gathered = gathered;
            B4D_HIP(hipMemsetAsync(msel, 0, sizeof(SelState) * (size_t)np, st));
            if (expect) hipLaunchKernelGGL(k_ncc_sample, dim3(np), dim3(256), 0, st, na, pred);
            hipLaunchKernelGGL(k_ncc_map, dim3(nblk, np), dim3(256), 0, st, na);
            if (expect) hipLaunchKernelGGL(k_ncc_check, dim3((np + 63) / 64), dim3(64), 0, st, geom, pred, msel, np);
            FinArgs fa{};
            fa.mag = ncc;
            fa.med_src = absncc;
            fa.compact = xc;          // the correlation maps are consumed: their buffer serves as the median's scratch
            fa.geom = geom;
            fa.stride = fpix;
            fa.part_val = pval;
            fa.part_idx = pind;
            fa.out = out + (size_t)p0 * 4;
            fa.peak_ij = peak_ij ? peak_ij + (size_t)p0 * 2 : nullptr;
            fa.ny = ny;
            fa.nx = nx;
            fa.nblk = nblk;
            fa.subpixel = subpixel;
            fa.eps = eps;
            if (expect) {   // passes 2-3 of the select on the gathered bin; whoever is left takes the whole select on its map
                FinArgs fg = fa;
                fg.compact = gathered;
                if ((rc = launch_track_fin2_checked(fg, msel, np, st))) return rc;
            }
            if ((rc = launch_track_fin_rest(fa, msel, np, st))) return rc;
            B4D_HIP(hipGetLastError());
        }
    }
    return B4D_OK;
}

}  // extern "C"

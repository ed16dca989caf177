// b4d_wiener_mr.hip -- the three-kernel Wiener pipeline of deconvolve_psf (preprocessing/filters.py:233-289,
// BASELINE.json config 5) for padded sizes whose sides factor into three in-register radices (4104 = 8 * 27 * 19).
//
// Per frame (several frames per launch), no padded copy, no transpose passes, no separate max pass:
//   k_wmr_rows_fwd  a workgroup owns FOUR row pairs (8 padded rows) at a time, one 256-lane group and one LDS row buffer
//                   per pair: the pair (2 s, 2 s + 1) of the reflect-padded frame is read straight from the (h, w) frame as
//                   a + i b, ONE complex transform (b4d_mixed.hpp), Hermitian split into the two half rows.  The store is
//                   the transposition: column k of the half spectrum is a contiguous run T[k][0 .. H), and four
//                   neighbouring lanes write the four pairs' pieces {Fa[k], Fb[k]} = ONE aligned 64-byte sector.
//                   (Measured with the B4D_EXP_WMR builds: one pair per workgroup = 16-byte pieces costs the pass
//                   +25 us per 4k frame, 32-byte pieces +15 us, 64-byte pieces nothing; the L2 takes a partial-sector
//                   write as slowly as a whole one.)  max|rows of the pair| (np.nanmax(np.abs(padded)), filters.py:255)
//                   rides along, one value per pair -- 8 k same-address atomics per frame cost 50 us.
//   k_wmr_cols      one spectrum column per workgroup, resident in LDS: forward transform, times the transposed Wiener
//                   filter W[k][ky], inverse transform, back in place (the column never leaves the CU in between).
//                   Column 0 of every frame also reduces the pair maxima to max|frame|.
//   k_wmr_rows_inv  four pairs per workgroup again: 64-byte pieces gathered by four neighbouring lanes, Hermitian-extended
//                   to Ga + i Gb in LDS, ONE inverse transform per pair, real part -> row 2 s, imaginary part -> row
//                   2 s + 1; 1/(H W), the reference's normalise / clip / rescale (filters.py:259-266, 287-289) and the
//                   crop folded into the store.
// The row kernels keep 140 KB of LDS (4 x 34 KB row buffers), i.e. one 1024-lane workgroup per CU: they are persistent
// (grid = CUs, a loop over quads) and issue the next quad's frame loads before the current quad's store loop.
// The reference divides the padded frame by max|.| BEFORE the (linear) filter and multiplies back after the clip; here
// the division is applied to the filtered value just before the clip -- the same real-number result, different by one
// float32 rounding of the scale (the maximum is only known after the first pass over the frame).
// HBM/MALL traffic per frame at 4096^2, sigma 1.5: frame in 67 MB + spectrum out 67 + column in/out 135 + filter 67 +
// spectrum in 67 + frame out 67 = 470 MB (the route it replaces moved 938 MB in 9 kernels).
#include "b4d_fft2d.hpp"   // cross_power, argmax_merge
#include "b4d_mixed.hpp"
#include "b4d_wiener_mr.hpp"

namespace b4d {

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
// v mod n for 0 <= v < 2 n: the fftshift / mirror columns of the store loops (a `%` by a run-time length is an integer-division
// sequence of ~20 vector instructions PER ELEMENT)
__device__ __forceinline__ int wrap_idx(int v, int n) { return v >= n ? v - n : v; }

constexpr int WMR_Q = 4;   // row pairs per workgroup of the row kernels: 4 x 16-byte pieces = one 64-byte sector

template <class MX>
__host__ __device__ constexpr size_t wmr_rows_lds() {
    return sizeof(float2) * ((size_t)WMR_Q * MX::BUF + MX::M1) + sizeof(float) * (4 * (WMR_Q * MX::LANES / 64) + 8);
}

// Work index j -> quad.  Quads 2 m and 2 m + 1 of a frame share every 128-byte line of the transposed spectrum (their
// 64-byte pieces are neighbours), and workgroup b of a launch runs on XCD b % 8: items j and j + 8 -- the same XCD, adjacent
// dispatch slots -- are therefore mapped to such a pair, so that the two halves of a line meet in ONE L2 (measured before
// this map: FETCH_SIZE of k_wmr_rows_inv 2.05 x its algorithmic bytes, WRITE_SIZE of k_wmr_rows_fwd 1.23 x).
// Quads are numbered in a per-frame space padded to an even count (qpf2), items in groups of 16; invalid ones are skipped.
struct QuadRef {
    int f, qi;
    bool valid;
};
__device__ __forceinline__ QuadRef quad_of(int j, int qpf, int qpf2, int nframes) {
    const int xcd = j & 7, r = j >> 3;
    const int Q = ((r >> 1) << 4) + (xcd << 1) + (r & 1);
    QuadRef q;
    q.f = Q / qpf2;
    q.qi = Q - q.f * qpf2;
    q.valid = q.f < nframes && q.qi < qpf;
    return q;
}

template <class MX>
__global__ void __launch_bounds__(WMR_Q* MX::LANES) k_wmr_rows_fwd(const float* __restrict__ frames, float2* __restrict__ T,
                                                                    const float2* __restrict__ twN, float* __restrict__ pmax, WmrGeom g,
                                                                    int nframes, int qpf) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int R1 = MX::R1, M1 = MX::M1, L = MX::LANES, RD = MX::ROUNDS1, N = MX::N, WG = WMR_Q * L;
    const int tid = threadIdx.x, lt = tid % L;
    float2* tw2 = sm + (size_t)WMR_Q * MX::BUF;
    float* wmax = reinterpret_cast<float*>(tw2 + M1);   // [sub][wave of the group]
    for (int t = tid; t < M1; t += WG) tw2[t] = twN[R1 * t];

    float2 v[RD][R1];
    float mx = 0.f;
    // frame loads of quad q for this lane's pair (compute mapping: lane lt of group sub owns items lt, lt + L, ...)
    const int qpf2 = (qpf + 1) & ~1, nitems = (nframes * qpf2 + 15) & ~15;
    auto next_item = [&](int j) {   // next work item of this workgroup that maps to a real quad (nitems if none)
        for (j += gridDim.x; j < nitems && !quad_of(j, qpf, qpf2, nframes).valid; j += gridDim.x) {}
        return j;
    };
    auto load = [&](int j, int lt, int sub) {
        const QuadRef qr = quad_of(j, qpf, qpf2, nframes);
        const int f = qr.f, pr = WMR_Q * qr.qi + sub;
        const bool act = pr < g.hp;
        const int r0 = 2 * (act ? pr : 0);
        const bool has_b = act && r0 + 1 < g.H;
        const int y0 = reflect_idx(r0 - g.py, g.h), y1 = reflect_idx((has_b ? r0 + 1 : r0) - g.py, g.h);
        const float* fa = frames + ((size_t)f * g.h + y0) * g.w;
        const float* fb = frames + ((size_t)f * g.h + y1) * g.w;
        mx = 0.f;
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int mc = min(lt + r * L, M1 - 1);   // clamped: the loads are unconditional, the item is masked later
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                const int x = reflect_idx(M1 * n1 + mc - g.px, g.w);
                const float a = fa[x], b = fb[x];   // unconditional (fb = fa's row when the pair has no second row)
                v[r][n1] = make_float2(a, b);
                mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));   // fmaxf drops NaN operands: np.nanmax
            }
        }
        if (!has_b) {   // wave-uniform: a select per load would put every load under its own branch
#pragma unroll
            for (int r = 0; r < RD; ++r)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) v[r][n1].y = 0.f;
        }
    };
#ifndef B4D_WMR_PREFETCH
#define B4D_WMR_PREFETCH 1   // issue the next quad's global loads before the current quad's store loop (A/B: tools/dev_cfg5.py)
#endif
    int q = next_item((int)blockIdx.x - (int)gridDim.x);
    if (B4D_WMR_PREFETCH && q < nitems) load(q, (lt + 64 * (tid / L)) % L, tid / L);
    for (; q < nitems; q = next_item(q)) {
        // Everything derived from the lane index or the tables is invariant in q: left alone, the compiler precomputes all of
        // it (twiddle loads included) ahead of the quad loop and spills > 100 dwords around the radix stages.  Opaque
        // per-iteration copies keep those values short-lived.
        int ltq = lt, tidq = tid;
        const float2* twq = twN;
        asm volatile("" : "+v"(ltq), "+v"(tidq), "+s"(twq));
        const int subq = tidq / L;
        float2* bufq = sm + (size_t)subq * MX::BUF;
        const float2* tw2q = sm + (size_t)WMR_Q * MX::BUF;
        const QuadRef qr = quad_of(q, qpf, qpf2, nframes);
        const int f = qr.f, qi = qr.qi, pr = WMR_Q * qi + subq;
        const bool act = pr < g.hp;
        // The stages have M1 / M2 / M3 items for L lanes, so each leaves the group's last waves idle (radix 27: 152 items,
        // waves 2.4 .. 3).  Wave w of every group sits on SIMD w % 4: rotating the item <-> lane map by one wave per
        // group spreads the idle waves over the four SIMDs instead of parking all of them on SIMD 3.
        const int ltr = (ltq + 64 * subq) % L;
        if (!B4D_WMR_PREFETCH) load(q, ltr, subq);
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int m = ltr + r * L;
            if (m < M1) MX::stage1_item(v[r], m, bufq, twq);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
        if ((tidq & 63) == 0) wmax[tidq >> 6] = mx;
        __syncthreads();
        if (ltq == 0 && act) {
            float m2 = wmax[subq * (L / 64)];
#pragma unroll
            for (int i = 1; i < L / 64; ++i) m2 = fmaxf(m2, wmax[subq * (L / 64) + i]);
            pmax[(size_t)f * g.hp + pr] = m2;
        }
        MX::stage2(bufq, tw2q, ltr);
        __syncthreads();
        MX::stage3(bufq, ltr);
        __syncthreads();
        {
            const int qn = next_item(q);
            if (B4D_WMR_PREFETCH && qn < nitems) load(qn, ltr, subq);   // in flight under the store loop
        }
        // piece mapping: lanes 4 i .. 4 i + 3 hold the four pairs' pieces of ONE k: 64 contiguous bytes of T[k][.]
        const int j = tidq & (WMR_Q - 1), kk = tidq / WMR_Q;
        const float2* bj = sm + (size_t)j * MX::BUF;
        const bool wr = WMR_Q * qi + j < g.hp;
        float2* dst = T + (size_t)f * g.Wh * g.Hp + 2 * (WMR_Q * qi + j);
        typename MX::template PosIter<L> pk(kk), pn(kk == 0 ? 0 : N - kk);   // k upwards, N - k downwards
        for (int k = kk; k < g.Wh; k += L) {
            const float2 z = bj[pk.pos()], w = bj[pn.pos()];
            if (wr)
                *reinterpret_cast<float4*>(dst + (size_t)k * g.Hp) =
                    make_float4(0.5f * (z.x + w.x), 0.5f * (z.y - w.y), 0.5f * (z.y + w.y), 0.5f * (w.x - z.x));
            pk.up();
            if (k == 0) pn = typename MX::template PosIter<L>(N - L); else pn.down();
        }
        __syncthreads();   // the row buffers are free for the next quad
    }
}

// MODE 0: times the Wiener filter `filt` (deconvolve_psf).  MODE 1 (fft -> psd -> autocorr at general sizes,
// signal/fft.py:261-309 + signal/corr.py:256-320): P = |F|^2 (DC zeroed when flags & B4D_REMOVE_MEAN) replaces the product,
// is stored as a REAL column to Pt (same [k][ky] layout, pitch g.Hp floats) when Pt != null, and is what the inverse
// transform runs on (skipped when `inverse` == 0: PSD only).  MODE 2: forward transform only, F written back in natural
// order (the 2-D half spectra of images / templates for tracking and xcorr2d).
// MODE 0 with sep.y != null: the filter of a SEPARABLE, point-symmetric PSF (the Gaussian of deconvolve_psf) is not streamed from a
// (Wh, Hp) table but rebuilt per element from two 1-D tables -- H(k, ky) = hx[k] hy[ky] (real), Laplacian L = lx[k] + ly[ky],
// W = H / (H^2 + balance L^2) -- 67 MB per 4k frame that never cross HBM (the tables are 49 KB and stay in L2).
struct WmrSep {
    const float2* x;   // (Wh): {hx[k], lx[k]}
    const float2* y;   // (H):  {hy[ky], ly[ky]}
    float balance;
};
template <class MY, int MODE>
__global__ void __launch_bounds__(MY::LANES, 4) k_wmr_cols(float2* __restrict__ T, const float2* __restrict__ filt,
                                                            const float2* __restrict__ twN, const float* __restrict__ pmax,
                                                            float* __restrict__ amax, WmrGeom g, float* __restrict__ Pt, unsigned flags,
                                                            int inverse, WmrSep sep) {
    __shared__ __attribute__((aligned(16))) float2 buf[MY::BUF];
    __shared__ float2 tw2[MY::M1];
    constexpr int R1 = MY::R1, M1 = MY::M1, LANES = MY::LANES, RD = MY::ROUNDS1, N = MY::N;
    const int tid = threadIdx.x;
    const int col = blockIdx.x;
    const int f = col / g.Wh, k = col - f * g.Wh;
    float2* x = T + (size_t)col * g.Hp;
    const float2* fl = MODE == 0 ? (sep.y ? sep.y : filt + (size_t)k * g.Hp) : nullptr;
    float2 sx = make_float2(0.f, 0.f);
    if (MODE == 0 && sep.y) sx = sep.x[k];
    MY::build_tw2(tw2, twN, tid);
    if (MODE == 0 && k == 0 && tid < 64) {   // max|frame| from the pair maxima of k_wmr_rows_fwd (fmaxf drops NaN: np.nanmax)
        float mx = 0.f;
        for (int i = tid; i < g.hp; i += 64) mx = fmaxf(mx, pmax[(size_t)f * g.hp + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
        if (tid == 0) amax[f] = mx;
    }
    float2 v[RD][R1];
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int mc = min(tid + r * LANES, M1 - 1);
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) v[r][n1] = x[M1 * n1 + mc];
    }
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int m = tid + r * LANES;
        if (m < M1) MY::stage1_item(v[r], m, buf, twN);
    }
    // the filter values of the second transform's inputs: in flight under the first transform
    float2 fv[RD][R1];
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int mc = min(tid + r * LANES, M1 - 1);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) fv[r][n1] = fl[M1 * n1 + mc];
        }
    }
    __syncthreads();
    MY::stage2(buf, tw2, tid);
    __syncthreads();
    MY::stage3(buf, tid);
    __syncthreads();
    if (MODE == 2) {
        typename MY::template PosIter<LANES> pf(tid);
        for (int n = tid; n < N; n += LANES) {
            x[n] = buf[pf.pos()];
            pf.up();
        }
        return;
    }
    // inverse = conj(forward(conj(.))): inputs conj(X[n] W[n]) gathered from where the forward transform left X[n]
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int mc = min(tid + r * LANES, M1 - 1);
        typename MY::template PosIter<M1> pi(mc);   // n = mc, mc + M1, ...
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            if (MODE == 0) {
                if (sep.y) {   // fv = {hy, ly} of this row: W = t / (t^2 + balance s^2), t = hx hy, s = lx + ly (real)
                    const float t = sx.x * fv[r][n1].x, sl = sx.y + fv[r][n1].y;
                    const float wv = t * __builtin_amdgcn_rcpf(fmaf(t, t, sep.balance * sl * sl));   // hardware reciprocal (1 ulp): the pass is issue-bound
                    const float2 b = buf[pi.pos()];
                    v[r][n1] = make_float2(b.x * wv, -(b.y * wv));
                } else {
                    const float2 p = cmul(buf[pi.pos()], fv[r][n1]);
                    v[r][n1] = make_float2(p.x, -p.y);
                }
            } else {
                const float2 f = buf[pi.pos()];
                const int n = M1 * n1 + mc;
                float p = f.x * f.x + f.y * f.y;
                if (Pt && tid + r * LANES < M1) Pt[(size_t)col * g.Hp + n] = p;   // raw power: the scale is applied by the row pass
                if (n == 0 && k == 0 && (flags & B4D_REMOVE_MEAN)) p = 0.f;       // mean removal = DC bin of the power spectrum
                v[r][n1] = make_float2(p, 0.f);
            }
            pi.up();
        }
    }
    if (MODE == 1 && !inverse) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int m = tid + r * LANES;
        if (m < M1) MY::stage1_item(v[r], m, buf, twN);
    }
    __syncthreads();
    MY::stage2(buf, tw2, tid);
    __syncthreads();
    MY::stage3(buf, tid);
    __syncthreads();
    typename MY::template PosIter<LANES> pk(tid);
    for (int n = tid; n < N; n += LANES) {
        const float2 z = buf[pk.pos()];
        x[n] = make_float2(z.x, -z.y);
        pk.up();
    }
}

template <class MX>
__global__ void __launch_bounds__(WMR_Q* MX::LANES) k_wmr_rows_inv(const float2* __restrict__ T, float* __restrict__ out,
                                                                    const float2* __restrict__ twN, const float* __restrict__ amax, WmrGeom g,
                                                                    int nframes, int qpf) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int R1 = MX::R1, M1 = MX::M1, L = MX::LANES, RD = MX::ROUNDS1, N = MX::N, WG = WMR_Q * L;
    const int tid = threadIdx.x, lt = tid % L;
    float2* tw2 = sm + (size_t)WMR_Q * MX::BUF;
    for (int t = tid; t < M1; t += WG) tw2[t] = twN[R1 * t];
    // the 16-byte pieces of quad q this lane gathers (piece mapping: k = tid / 4 + i L, pair tid % 4), clamped addresses
    constexpr int NP = (N / 2 + 1 + L - 1) / L;
    float4 pc[NP];
    const int qpf2 = (qpf + 1) & ~1, nitems = (nframes * qpf2 + 15) & ~15;
    auto next_item = [&](int j) {
        for (j += gridDim.x; j < nitems && !quad_of(j, qpf, qpf2, nframes).valid; j += gridDim.x) {}
        return j;
    };
    auto fetch = [&](int j, int tidq) {
        const QuadRef qr = quad_of(j, qpf, qpf2, nframes);
        const int f = qr.f, pj = WMR_Q * qr.qi + (tidq & (WMR_Q - 1)), kk = tidq / WMR_Q;
        const float2* src = T + (size_t)f * g.Wh * g.Hp + 2 * min(pj, g.hp - 1);
#pragma unroll
        for (int i = 0; i < NP; ++i) pc[i] = *reinterpret_cast<const float4*>(src + (size_t)min(kk + i * L, g.Wh - 1) * g.Hp);
    };
    int q = next_item((int)blockIdx.x - (int)gridDim.x);
    if (B4D_WMR_PREFETCH && q < nitems) fetch(q, tid);
    for (; q < nitems; q = next_item(q)) {
        int ltq = lt, tidq = tid;   // opaque per-iteration copies (see k_wmr_rows_fwd)
        const float2* twq = twN;
        asm volatile("" : "+v"(ltq), "+v"(tidq), "+s"(twq));
        const int subq = tidq / L;
        float2* bufq = sm + (size_t)subq * MX::BUF;
        const float2* tw2q = sm + (size_t)WMR_Q * MX::BUF;
        const QuadRef qr = quad_of(q, qpf, qpf2, nframes);
        const int f = qr.f, qi = qr.qi, pr = WMR_Q * qi + subq;
        {   // piece mapping: four neighbouring lanes gather the four pairs' 16-byte pieces of ONE k (a 64-byte sector);
            // Ga + i Gb, Hermitian-extended beyond N/2, conjugated for the inverse, lands in natural order in the pair's buffer
            const int j = tidq & (WMR_Q - 1), kk = tidq / WMR_Q, pj = WMR_Q * qi + j;
            float2* bj = sm + (size_t)j * MX::BUF;
            const bool hb = 2 * pj + 1 < g.H;
            if (!B4D_WMR_PREFETCH) fetch(q, tidq);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int k = kk + i * L;
                if (k < g.Wh) {
                    const float4 p = pc[i];
                    const float bx = hb ? p.z : 0.f, by = hb ? p.w : 0.f;
                    bj[k] = make_float2(p.x - by, -(p.y + bx));
                    if (k != 0 && 2 * k != N) bj[N - k] = make_float2(p.x + by, p.y - bx);
                }
            }
        }
        __syncthreads();
        const int ltr = (ltq + 64 * subq) % L;   // item <-> lane map rotated by one wave per group (see k_wmr_rows_fwd)
        float2 v[RD][R1];
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int mc = min(ltr + r * L, M1 - 1);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[r][n1] = bufq[M1 * n1 + mc];
        }
        __syncthreads();   // every input is in registers: stage 1 may overwrite the buffer
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int m = ltr + r * L;
            if (m < M1) MX::stage1_item(v[r], m, bufq, twq);
        }
        __syncthreads();
        MX::stage2(bufq, tw2q, ltr);
        __syncthreads();
        MX::stage3(bufq, ltr);
        __syncthreads();
        // np: work = padded / scale; restored = clip(wiener(work)) * scale (filters.py:259-266, 287): the 1/(H W) of the
        // inverse transform and the division by the scale are one factor here
        const float fsc = amax[f];
        const bool fok = isfinite(fsc) && fsc != 0.f;
        const float sc = g.inv / fsc;
        const int r0 = 2 * pr, ya = r0 - g.py, yb = ya + 1;
        const bool act = pr < g.hp;
        const bool wa = act && ya >= 0 && ya < g.h, wb = act && r0 + 1 < g.H && yb >= 0 && yb < g.h;
        float* orow = out + ((size_t)f * g.h + ya) * g.w;
        {
            const int qn = next_item(q);
            if (B4D_WMR_PREFETCH && qn < nitems) fetch(qn, tidq);   // in flight under the store loop
        }
        typename MX::template PosIter<L> pk(ltq + g.px);
        for (int x = ltq; x < g.w; x += L) {
            const float2 z = bufq[pk.pos()];
            pk.up();
            float va = z.x * sc, vb = -z.y * sc;   // conj(buf): real part row a, imaginary part row b
            if (g.clip) {                          // np.clip: NaN stays NaN
                va = (va > 1.f ? 1.f : (va < -1.f ? -1.f : va));
                vb = (vb > 1.f ? 1.f : (vb < -1.f ? -1.f : vb));
            }
            if (wa) orow[x] = fok ? va * fsc : 0.f;
            if (wb) orow[g.w + x] = fok ? vb * fsc : 0.f;
        }
        __syncthreads();   // the row buffers are free for the next quad
    }
}

// Cross spectrum + inverse column transform of one (image, template) pair and column k: c = A conj(B) [/ (|c| + eps),
// signal/tracking.py:280-281], DC bin zeroed with B4D_REMOVE_MEAN (xcorr2d: both means removed), inverse along ky, G written
// to the pair's transposed workspace.  A, B: 2-D half spectra from k_wmr_cols<., 2>, indexed through ia / ib (null: the pair).
// grid (npairs * Wh), block LANES
template <class MY, bool WHITEN>
__global__ void __launch_bounds__(MY::LANES, 4) k_wmr_cols_prod(const float2* __restrict__ A, const float2* __restrict__ B,
                                                                 const int* __restrict__ ia, const int* __restrict__ ib, float2* __restrict__ G,
                                                                 const float2* __restrict__ twN, WmrGeom g, float eps, unsigned flags) {
    __shared__ __attribute__((aligned(16))) float2 buf[MY::BUF];
    __shared__ float2 tw2[MY::M1];
    constexpr int R1 = MY::R1, M1 = MY::M1, LANES = MY::LANES, RD = MY::ROUNDS1, N = MY::N;
    const int tid = threadIdx.x;
    const int col = blockIdx.x, pair = col / g.Wh, k = col - pair * g.Wh;
    const size_t sa = ia ? ia[pair] : pair, sb = ib ? ib[pair] : pair;
    const float2* xa = A + (sa * g.Wh + k) * g.Hp;
    const float2* xb = B + (sb * g.Wh + k) * g.Hp;
    float2* xo = G + (size_t)col * g.Hp;
    MY::build_tw2(tw2, twN, tid);
    float2 v[RD][R1];
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int mc = min(tid + r * LANES, M1 - 1);
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int n = M1 * n1 + mc;
            float2 c = cross_power<WHITEN>(xa[n], xb[n], eps);
            if (n == 0 && k == 0 && (flags & B4D_REMOVE_MEAN)) c = make_float2(0.f, 0.f);
            v[r][n1] = make_float2(c.x, -c.y);   // inverse = conj(forward(conj(.)))
        }
    }
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int m = tid + r * LANES;
        if (m < M1) MY::stage1_item(v[r], m, buf, twN);
    }
    __syncthreads();
    MY::stage2(buf, tw2, tid);
    __syncthreads();
    MY::stage3(buf, tid);
    __syncthreads();
    typename MY::template PosIter<LANES> pk(tid);
    for (int n = tid; n < N; n += LANES) {
        const float2 z = buf[pk.pos()];
        xo[n] = make_float2(z.x, -z.y);
        pk.up();
    }
}

// Last pass of general-size tracking: inverse row-pair transforms of G as in k_wmr_rows_out, epilogue = |value| / (H W)
// written fftshift-ed (signal/tracking.py:283-285) plus one arg-max partial per quad (first occurrence in row-major order).
template <class MX>
__global__ void __launch_bounds__(WMR_Q* MX::LANES) k_wmr_rows_mag(const float2* __restrict__ T, float* __restrict__ mag,
                                                                    float* __restrict__ part_val, int* __restrict__ part_idx,
                                                                    const float2* __restrict__ twN, WmrGeom g, int nframes, int qpf,
                                                                    unsigned* __restrict__ selw, int sel_stride, unsigned pred_bin,
                                                                    float* __restrict__ compact) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int R1 = MX::R1, M1 = MX::M1, L = MX::LANES, RD = MX::ROUNDS1, N = MX::N, WG = WMR_Q * L;
    const int tid = threadIdx.x, lt = tid % L;
    float2* tw2 = sm + (size_t)WMR_Q * MX::BUF;
    float* sv = reinterpret_cast<float*>(tw2 + M1);   // WG / 64 wave partials: values, then indices
    int* si = reinterpret_cast<int*>(sv + WG / 64);
    unsigned* hl = reinterpret_cast<unsigned*>(si + WG / 64);   // 2 (WG / 64) + 1 words: the median's expected-bin bookkeeping
    for (int t = tid; t < M1; t += WG) tw2[t] = twN[R1 * t];
    const int qpf2 = (qpf + 1) & ~1, nitems = (nframes * qpf2 + 15) & ~15;
    auto next_item = [&](int j) {
        for (j += gridDim.x; j < nitems && !quad_of(j, qpf, qpf2, nframes).valid; j += gridDim.x) {}
        return j;
    };
    const size_t fpix = (size_t)g.H * g.W;
    for (int q = next_item((int)blockIdx.x - (int)gridDim.x); q < nitems; q = next_item(q)) {
        int ltq = lt, tidq = tid;   // opaque per-iteration copies (see k_wmr_rows_fwd)
        const float2* twq = twN;
        asm volatile("" : "+v"(ltq), "+v"(tidq), "+s"(twq));
        const int subq = tidq / L;
        float2* bufq = sm + (size_t)subq * MX::BUF;
        const float2* tw2q = sm + (size_t)WMR_Q * MX::BUF;
        const QuadRef qr = quad_of(q, qpf, qpf2, nframes);
        const int f = qr.f, qi = qr.qi, pr = WMR_Q * qi + subq;
        const bool act = pr < g.hp;
        const int r0 = 2 * pr;
        const bool has_b = act && r0 + 1 < g.H;
        {
            const int j = tidq & (WMR_Q - 1), kk = tidq / WMR_Q, pj = WMR_Q * qi + j;
            float2* bj = sm + (size_t)j * MX::BUF;
            const bool hb = 2 * pj + 1 < g.H;
            const float2* src = T + (size_t)f * g.Wh * g.Hp + 2 * min(pj, g.hp - 1);
            for (int k = kk; k < g.Wh; k += L) {
                const float4 p = *reinterpret_cast<const float4*>(src + (size_t)k * g.Hp);
                const float bx = hb ? p.z : 0.f, by = hb ? p.w : 0.f;
                bj[k] = make_float2(p.x - by, -(p.y + bx));
                if (k != 0 && 2 * k != N) bj[N - k] = make_float2(p.x + by, p.y - bx);
            }
        }
        __syncthreads();
        const int ltr = (ltq + 64 * subq) % L;
        float2 v[RD][R1];
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int mc = min(ltr + r * L, M1 - 1);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[r][n1] = bufq[M1 * n1 + mc];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int m = ltr + r * L;
            if (m < M1) MX::stage1_item(v[r], m, bufq, twq);
        }
        __syncthreads();
        MX::stage2(bufq, tw2q, ltr);
        __syncthreads();
        MX::stage3(bufq, ltr);
        __syncthreads();
        const int ra = (r0 + g.H / 2) % g.H, rb = (r0 + 1 + g.H / 2) % g.H;
        float* mf = mag + (size_t)f * fpix;
        float bv = -1.0f;
        int bi = 0x7fffffff;
        unsigned cnt = 0, low = 0;   // elements of this lane inside / below the bin the median is expected in (b4d_track.hip)
        auto kbin = [](float m) { return (__float_as_uint(m) | 0x80000000u) >> 21; };   // f2key(m >= 0) >> 21
        typename MX::template PosIter<L> pk_it(ltq);
        for (int x = ltq; x < g.W; x += L) {
            const float2 z = bufq[pk_it.pos()];
            pk_it.up();
            const float ma = fabsf(z.x * g.inv), mb = fabsf(z.y * g.inv);   // conj(buf): real part row a, -imaginary part row b
            const int c = wrap_idx(x + g.W / 2, g.W);
            if (act) {
                mf[(size_t)ra * g.W + c] = ma;
                argmax_merge(bv, bi, ma, ra * g.W + c);
                cnt += (ma == ma && kbin(ma) == pred_bin) ? 1u : 0u;
                low += (ma == ma && kbin(ma) < pred_bin) ? 1u : 0u;
            }
            if (has_b) {
                mf[(size_t)rb * g.W + c] = mb;
                argmax_merge(bv, bi, mb, rb * g.W + c);
                cnt += (mb == mb && kbin(mb) == pred_bin) ? 1u : 0u;
                low += (mb == mb && kbin(mb) < pred_bin) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(bv, o, 64);
            const int oi = __shfl_down(bi, o, 64);
            argmax_merge(bv, bi, ov, oi);
        }
        const int lane = tidq & 63, wv = tidq >> 6;
        unsigned incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) low += __shfl_down(low, o, 64);
        if (lane == 0) {
            sv[wv] = bv;
            si[wv] = bi;
            hl[WG / 64 + 1 + wv] = low;
        }
        if (lane == 63) hl[wv] = incl;
        __syncthreads();
        if (tidq == 0) {
            for (int i = 1; i < WG / 64; ++i) argmax_merge(bv, bi, sv[i], si[i]);
            part_val[(size_t)f * qpf + qi] = bv;
            part_idx[(size_t)f * qpf + qi] = bi;
            unsigned run = 0, lo = 0;
            for (int i = 0; i < WG / 64; ++i) {
                const unsigned t = hl[i];
                hl[i] = run;
                run += t;
                lo += hl[WG / 64 + 1 + i];
            }
            unsigned base = 0;
            if (selw && pred_bin) {
                unsigned* sw = selw + (size_t)f * sel_stride;
                if (lo) atomicAdd(&sw[1], lo);
                if (run) {
                    atomicAdd(&sw[2], run);
                    base = atomicAdd(&sw[3], run);
                }
            }
            hl[WG / 64] = base;
        }
        __syncthreads();
        if (selw && pred_bin && cnt) {   // second walk over this lane's values: the bin's elements -> compact
            float* dst = compact + (size_t)f * fpix + hl[WG / 64] + hl[wv] + (incl - cnt);
            typename MX::template PosIter<L> p2(ltq);
            for (int x = ltq; x < g.W; x += L) {
                const float2 z = bufq[p2.pos()];
                p2.up();
                const float ma = fabsf(z.x * g.inv), mb = fabsf(z.y * g.inv);
                if (act && ma == ma && kbin(ma) == pred_bin) *dst++ = ma;
                if (has_b && mb == mb && kbin(mb) == pred_bin) *dst++ = mb;
            }
        }
        __syncthreads();   // the row buffers and the bookkeeping words are free for the next quad
    }
}

// fft2d of real frames at general sizes (signal/fft.py:198-237): the 2-D half spectrum S[k][ky] (k_wmr_cols<., 2>) -> the full
// fftshift-ed complex spectrum (nframes, H, W): every quad gathers its 8 rows ky as 64-byte pieces, stages them through the row
// buffers and writes, per row, the direct half (ky, kx) and the conjugate mirror (-ky, -kx) along kx.
template <class MX>
__global__ void __launch_bounds__(WMR_Q* MX::LANES) k_wmr_rows_spec(const float2* __restrict__ S, float2* __restrict__ out, WmrGeom g,
                                                                     int nframes, int qpf) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int L = MX::LANES;
    static_assert(MX::BUF >= MX::N + 2, "row buffer too small to stage two half rows");
    const int tid = threadIdx.x, lt = tid % L, sub = tid / L;
    const int qpf2 = (qpf + 1) & ~1, nitems = (nframes * qpf2 + 15) & ~15;
    auto next_item = [&](int j) {
        for (j += gridDim.x; j < nitems && !quad_of(j, qpf, qpf2, nframes).valid; j += gridDim.x) {}
        return j;
    };
    const size_t fpix = (size_t)g.H * g.W;
    for (int q = next_item((int)blockIdx.x - (int)gridDim.x); q < nitems; q = next_item(q)) {
        const QuadRef qr = quad_of(q, qpf, qpf2, nframes);
        const int f = qr.f, qi = qr.qi;
        {
            const int j = tid & (WMR_Q - 1), kk = tid / WMR_Q, pj = WMR_Q * qi + j;
            float2* bj = sm + (size_t)j * MX::BUF;
            const float2* src = S + (size_t)f * g.Wh * g.Hp + 2 * min(pj, g.hp - 1);
            for (int k = kk; k < g.Wh; k += L) {
                const float4 p = *reinterpret_cast<const float4*>(src + (size_t)k * g.Hp);
                bj[k] = make_float2(p.x, p.y);
                bj[g.Wh + k] = make_float2(p.z, p.w);
            }
        }
        __syncthreads();
        const int pr = WMR_Q * qi + sub;
        const float2* mine = sm + (size_t)sub * MX::BUF;
        float2* of = out + (size_t)f * fpix;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ky = 2 * pr + e;
            if (pr >= g.hp || ky >= g.H) continue;
            float2* drow = of + (size_t)((ky + g.H / 2) % g.H) * g.W;
            float2* mrow = of + (size_t)(((g.H - ky) % g.H + g.H / 2) % g.H) * g.W;
            const float2* prow = mine + e * g.Wh;
            for (int kx = lt; kx < g.Wh; kx += L) {
                const float2 v = prow[kx];
                drow[wrap_idx(kx + g.W / 2, g.W)] = v;
                if (kx > 0 && 2 * kx != g.W) mrow[wrap_idx(g.W - kx + g.W / 2, g.W)] = make_float2(v.x, -v.y);
            }
        }
        __syncthreads();
    }
}

// Zero-lag value of the (unscaled) autocorrelation of every frame from the column-inverted half spectrum G = T[k][y]:
// R[0,0] = sum over the FULL kx range of G[kx][0] = G[0] + 2 sum Re G[k] (+ G[W/2] for even W).  grid (nframes), block 256
__global__ void __launch_bounds__(256) k_wmr_peak(const float2* __restrict__ T, WmrGeom g, float* __restrict__ peak) {
    __shared__ double sh[4];
    const float2* t = T + (size_t)blockIdx.x * g.Wh * g.Hp;
    double acc = 0.0;
    for (int k = threadIdx.x; k < g.Wh; k += 256) {
        const double v = (double)t[(size_t)k * g.Hp].x;
        acc += (k == 0 || 2 * k == g.W) ? v : 2.0 * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) peak[blockIdx.x] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// Last pass of fft -> psd -> autocorr at general sizes.  Same quads, piece gathers and transforms as k_wmr_rows_inv; the
// epilogue writes the two real rows of every pair fftshift-ed in both axes, scaled by 1/(H W) or, with B4D_NORM_PEAK and a
// positive zero-lag value peak[f], divided by it with the zero lag forced to exactly 1 (signal/corr.py:247-250).  When
// psd != null the quad first turns its 8 rows ky of the transposed power spectrum Pt into PSD rows: 8-byte pieces of four
// neighbouring lanes (32 bytes) staged through the row buffers, then every pair writes its two rows along kx -- the direct
// half (ky, kx) and the Hermitian mirror (-ky, -kx), both fftshift-ed (signal/fft.py:300-309).
template <class MX>
__global__ void __launch_bounds__(WMR_Q* MX::LANES) k_wmr_rows_out(const float2* __restrict__ T, const float* __restrict__ Pt,
                                                                    float* __restrict__ autocorr, float* __restrict__ psd,
                                                                    const float2* __restrict__ twN, const float* __restrict__ peak, WmrGeom g,
                                                                    int nframes, int qpf, float psd_scale, unsigned flags) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    constexpr int R1 = MX::R1, M1 = MX::M1, L = MX::LANES, RD = MX::ROUNDS1, N = MX::N, WG = WMR_Q * L;
    const int tid = threadIdx.x, lt = tid % L;
    float2* tw2 = sm + (size_t)WMR_Q * MX::BUF;
    for (int t = tid; t < M1; t += WG) tw2[t] = twN[R1 * t];
    const int qpf2 = (qpf + 1) & ~1, nitems = (nframes * qpf2 + 15) & ~15;
    auto next_item = [&](int j) {
        for (j += gridDim.x; j < nitems && !quad_of(j, qpf, qpf2, nframes).valid; j += gridDim.x) {}
        return j;
    };
    const size_t fpix = (size_t)g.H * g.W;
    for (int q = next_item((int)blockIdx.x - (int)gridDim.x); q < nitems; q = next_item(q)) {
        int ltq = lt, tidq = tid;   // opaque per-iteration copies (see k_wmr_rows_fwd)
        const float2* twq = twN;
        asm volatile("" : "+v"(ltq), "+v"(tidq), "+s"(twq));
        const int subq = tidq / L;
        float2* bufq = sm + (size_t)subq * MX::BUF;
        const float2* tw2q = sm + (size_t)WMR_Q * MX::BUF;
        const QuadRef qr = quad_of(q, qpf, qpf2, nframes);
        const int f = qr.f, qi = qr.qi, pr = WMR_Q * qi + subq;
        const bool act = pr < g.hp;
        const int r0 = 2 * pr;
        const bool has_b = act && r0 + 1 < g.H;
        const int j = tidq & (WMR_Q - 1), kk = tidq / WMR_Q, pj = WMR_Q * qi + j;
        if (psd) {
            // ---- PSD rows ky = r0, r0 + 1 of every pair: gather (piece mapping) -> stage -> two half rows along kx
            float* stg = reinterpret_cast<float*>(sm + (size_t)j * MX::BUF);   // pair j: [row a | row b], g.Wh floats each
            const float* src = Pt + (size_t)f * g.Wh * g.Hp + 2 * min(pj, g.hp - 1);
            for (int k = kk; k < g.Wh; k += L) {
                const float2 p = *reinterpret_cast<const float2*>(src + (size_t)k * g.Hp);
                stg[k] = p.x;
                stg[g.Wh + k] = p.y;
            }
            __syncthreads();
            const float* mine = reinterpret_cast<const float*>(bufq);
            float* pf = psd + (size_t)f * fpix;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ky = r0 + e;
                if (!act || ky >= g.H) continue;
                float* drow = pf + (size_t)((ky + g.H / 2) % g.H) * g.W;                      // direct half: (ky, kx)
                float* mrow = pf + (size_t)(((g.H - ky) % g.H + g.H / 2) % g.H) * g.W;        // mirror: (-ky, -kx)
                const float* prow = mine + e * g.Wh;
                for (int kx = ltq; kx < g.Wh; kx += L) {
                    const float v = prow[kx] * psd_scale;
                    __builtin_nontemporal_store(v, drow + wrap_idx(kx + g.W / 2, g.W));   // outputs are written once: streaming stores
                    if (kx > 0 && 2 * kx != g.W) __builtin_nontemporal_store(v, mrow + wrap_idx(g.W - kx + g.W / 2, g.W));
                }
            }
            __syncthreads();
        }
        if (!autocorr) continue;
        {   // ---- inverse row transforms: piece gathers of G, Hermitian extension, conjugated for the inverse
            float2* bj = sm + (size_t)j * MX::BUF;
            const bool hb = 2 * pj + 1 < g.H;
            const float2* src = T + (size_t)f * g.Wh * g.Hp + 2 * min(pj, g.hp - 1);
            for (int k = kk; k < g.Wh; k += L) {
                const float4 p = *reinterpret_cast<const float4*>(src + (size_t)k * g.Hp);
                const float bx = hb ? p.z : 0.f, by = hb ? p.w : 0.f;
                bj[k] = make_float2(p.x - by, -(p.y + bx));
                if (k != 0 && 2 * k != N) bj[N - k] = make_float2(p.x + by, p.y - bx);
            }
        }
        __syncthreads();
        const int ltr = (ltq + 64 * subq) % L;
        float2 v[RD][R1];
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int mc = min(ltr + r * L, M1 - 1);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[r][n1] = bufq[M1 * n1 + mc];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int m = ltr + r * L;
            if (m < M1) MX::stage1_item(v[r], m, bufq, twq);
        }
        __syncthreads();
        MX::stage2(bufq, tw2q, ltr);
        __syncthreads();
        MX::stage3(bufq, ltr);
        __syncthreads();
        const float pk = (flags & B4D_NORM_PEAK) ? peak[f] : 0.f;
        const bool unit = (flags & B4D_NORM_PEAK) && pk > 0.f;
        const float se = unit ? 1.0f / pk : g.inv;
        float* af = autocorr + (size_t)f * fpix;
        float* rowa = af + (size_t)((r0 + g.H / 2) % g.H) * g.W;
        float* rowb = af + (size_t)((r0 + 1 + g.H / 2) % g.H) * g.W;
        typename MX::template PosIter<L> pk_it(ltq);
        for (int x = ltq; x < g.W; x += L) {
            const float2 z = bufq[pk_it.pos()];
            pk_it.up();
            float va = z.x * se;
            const float vb = -z.y * se;   // conj(buf): real part row a, imaginary part row b
            if (unit && r0 == 0 && x == 0) va = 1.0f;
            const int c = wrap_idx(x + g.W / 2, g.W);
            if (act) __builtin_nontemporal_store(va, rowa + c);
            if (has_b) __builtin_nontemporal_store(vb, rowb + c);
        }
        __syncthreads();
    }
}

// ---- instantiated lengths -----------------------------------------------------------------------------------------
//   4104 = 4096 + 8 (sigma 1.5 on 4k frames), 520 = 512 + 8, 264 = 256 + 8: Wiener padded sizes;
//   2560 x 2160 (sCMOS), 1280 x 720, 600: general-size fft -> psd -> autocorr (b4d_general.hip)
//   228, 171, 170: the tiles / sub-tiles speckle_stats and sharpness_stats cut 2048- and 512-px frames into (metrics/common.py:
//   75-106: round(linspace) edges; 227, the other sub-tile width, is prime and stays on the DFT-matrix products)
//   the other entries: sides of common area detectors / cameras and the powers of two that occur beside them in non-square
//   formats (1024 x 768, 2448 x 2048, 4096 x 3000 ...); radices are picked for one round of every stage where the lanes allow
#define B4D_WMR_LENGTHS(X)       \
    X(4104, 8, 27, 19, 256)      \
    X(4096, 16, 16, 16, 256)     \
    X(3840, 16, 16, 15, 256)     \
    X(3648, 16, 12, 19, 256)     \
    X(3200, 16, 20, 10, 256)     \
    X(3072, 16, 16, 12, 256)     \
    X(3000, 15, 10, 20, 256)     \
    X(2592, 16, 18, 9, 256)      \
    X(2560, 16, 16, 10, 256)     \
    X(2448, 16, 9, 17, 256)      \
    X(2400, 16, 15, 10, 256)     \
    X(2304, 16, 16, 9, 256)      \
    X(2160, 16, 27, 5, 256)      \
    X(2048, 16, 16, 8, 256)      \
    X(1944, 8, 27, 9, 256)       \
    X(1936, 16, 11, 11, 256)     \
    X(1920, 16, 12, 10, 256)     \
    X(1600, 16, 10, 10, 128)     \
    X(1536, 16, 16, 6, 128)      \
    X(1440, 12, 12, 10, 128)     \
    X(1280, 16, 16, 5, 128)      \
    X(1216, 16, 4, 19, 128)      \
    X(1200, 10, 12, 10, 128)     \
    X(1080, 12, 10, 9, 128)      \
    X(1024, 16, 8, 8, 128)       \
    X(1000, 10, 10, 10, 128)     \
    X(960, 16, 12, 5, 128)       \
    X(800, 16, 10, 5, 128)       \
    X(768, 8, 8, 12, 64)         \
    X(720, 16, 9, 5, 64)         \
    X(640, 16, 8, 5, 64)         \
    X(600, 8, 15, 5, 64)         \
    X(576, 16, 6, 6, 64)         \
    X(540, 12, 9, 5, 64)         \
    X(520, 8, 5, 13, 128)        \
    X(512, 8, 8, 8, 64)          \
    X(480, 8, 6, 10, 64)         \
    X(264, 8, 3, 11, 64)         \
    X(228, 4, 3, 19, 64)         \
    X(171, 3, 3, 19, 64)         \
    X(170, 5, 2, 17, 64)

bool wmr_supported(int n) {
#define X(N_, A_, B_, C_, L_) \
    if (n == N_) return true;
    B4D_WMR_LENGTHS(X)
#undef X
    return false;
}

static int wmr_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}

// persistent row kernels: one workgroup per CU when a workgroup needs most of the LDS, as many as fit otherwise.  The dynamic-LDS
// attribute is set once per (kernel address, device): ensure_dynamic_lds (b4d_common.hpp).
template <class MX, class K>
static int wmr_rows_launch(K kernel, int nquads, hipStream_t st, size_t* lds_out, int* grid_out) {
    const size_t lds = wmr_rows_lds<MX>();
    int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (rc) return rc;
    const int per_cu = std::max(1, std::min((int)((size_t)160 * 1024 / lds), 2048 / (WMR_Q * MX::LANES)));
    *lds_out = lds;
    *grid_out = std::min(nquads, wmr_cus() * per_cu);
    return B4D_OK;
}

int wmr_rows_fwd(const float* frames, float2* T, const float2* twx, float* pmax, const WmrGeom& g, int nframes, hipStream_t st) {
    const int qpf = (g.hp + WMR_Q - 1) / WMR_Q, nquads = nframes * qpf;
    size_t lds = 0;
    int grid = 0, rc;
    switch (g.W) {
#define X(N_, A_, B_, C_, L_)                                                                                                              \
    case N_: {                                                                                                                             \
        using MX = Mix3<A_, B_, C_, L_>;                                                                                                   \
        if ((rc = wmr_rows_launch<MX>(&k_wmr_rows_fwd<MX>, nquads, st, &lds, &grid))) return rc;                                          \
        hipLaunchKernelGGL((k_wmr_rows_fwd<MX>), dim3(grid), dim3(WMR_Q* L_), lds, st, frames, T, twx, pmax, g, nframes, qpf);              \
    } break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix row kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

template <int MODE>
static int wmr_cols_launch(float2* T, const float2* filt, const float2* twy, const float* pmax, float* amax, const WmrGeom& g,
                           int nframes, float* Pt, unsigned flags, int inverse, hipStream_t st, WmrSep sep = WmrSep{nullptr, nullptr, 0.f}) {
    const unsigned grid = (unsigned)nframes * (unsigned)g.Wh;
    switch (g.H) {
#define X(N_, A_, B_, C_, L_)                                                                                                            \
    case N_:                                                                                                                             \
        hipLaunchKernelGGL((k_wmr_cols<Mix3<A_, B_, C_, L_>, MODE>), dim3(grid), dim3(L_), 0, st, T, filt, twy, pmax, amax, g, Pt, flags, \
                           inverse, sep);                                                                                                \
        break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix column kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int wmr_cols(float2* T, const float2* filt, const float2* twy, const float* pmax, float* amax, const WmrGeom& g, int nframes,
             hipStream_t st, const float2* sep_x, const float2* sep_y, float balance) {
    return wmr_cols_launch<0>(T, filt, twy, pmax, amax, g, nframes, nullptr, 0u, 1, st, WmrSep{sep_x, sep_y, balance});
}

int wmr_psd_autocorr(const float* frames, int nframes, int ny, int nx, const float2* twx, const float2* twy, float2* T, float* Pt,
                     float* scratch, float* psd, float psd_scale, float* autocorr, unsigned flags, hipStream_t st) {
    WmrGeom g{};
    g.h = g.H = ny;
    g.w = g.W = nx;
    g.py = g.px = 0;
    g.Wh = nx / 2 + 1;
    g.Hp = wmr_pitch(ny);
    g.hp = (ny + 1) / 2;
    g.inv = 1.0f / ((float)ny * (float)nx);
    float* pmax = scratch;                       // (nframes, hp): by-product of the shared forward row kernel, unused here
    float* peak = scratch + (size_t)nframes * g.hp;
    int rc;
    if ((rc = wmr_rows_fwd(frames, T, twx, pmax, g, nframes, st))) return rc;
    if ((rc = wmr_cols_launch<1>(T, nullptr, twy, nullptr, nullptr, g, nframes, psd ? Pt : nullptr, flags, autocorr ? 1 : 0, st))) return rc;
    if (autocorr && (flags & B4D_NORM_PEAK)) {
        hipLaunchKernelGGL(k_wmr_peak, dim3(nframes), dim3(256), 0, st, (const float2*)T, g, peak);
        B4D_HIP(hipGetLastError());
    }
    const int qpf = (g.hp + WMR_Q - 1) / WMR_Q, nquads = nframes * qpf;
    size_t lds = 0;
    int grid = 0;
    switch (g.W) {
#define X(N_, A_, B_, C_, L_)                                                                                                             \
    case N_: {                                                                                                                            \
        using MX = Mix3<A_, B_, C_, L_>;                                                                                                  \
        if ((rc = wmr_rows_launch<MX>(&k_wmr_rows_out<MX>, nquads, st, &lds, &grid))) return rc;                                         \
        hipLaunchKernelGGL((k_wmr_rows_out<MX>), dim3(grid), dim3(WMR_Q* L_), lds, st, (const float2*)T, (const float*)Pt, autocorr, psd, \
                           twx, (const float*)peak, g, nframes, qpf, psd_scale, flags);                                                   \
    } break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix row kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int wmr_rows_inv(const float2* T, float* out, const float2* twx, const float* amax, const WmrGeom& g, int nframes, hipStream_t st) {
    const int qpf = (g.hp + WMR_Q - 1) / WMR_Q, nquads = nframes * qpf;
    size_t lds = 0;
    int grid = 0, rc;
    switch (g.W) {
#define X(N_, A_, B_, C_, L_)                                                                                                              \
    case N_: {                                                                                                                             \
        using MX = Mix3<A_, B_, C_, L_>;                                                                                                   \
        if ((rc = wmr_rows_launch<MX>(&k_wmr_rows_inv<MX>, nquads, st, &lds, &grid))) return rc;                                          \
        hipLaunchKernelGGL((k_wmr_rows_inv<MX>), dim3(grid), dim3(WMR_Q* L_), lds, st, T, out, twx, amax, g, nframes, qpf);                 \
    } break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix row kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

// ---- general-size correlation on the same passes (b4d_track.hip, b4d_general.hip) -----------------------------------------
static WmrGeom wmr_plain_geom(int ny, int nx) {
    WmrGeom g{};
    g.h = g.H = ny;
    g.w = g.W = nx;
    g.Wh = nx / 2 + 1;
    g.Hp = wmr_pitch(ny);
    g.hp = (ny + 1) / 2;
    g.inv = 1.0f / ((float)ny * (float)nx);
    return g;
}

size_t wmr_spectrum_elems(int ny, int nx) { return (size_t)(nx / 2 + 1) * wmr_pitch(ny); }

int wmr_forward_spectra(const float* frames, int nframes, int ny, int nx, const float2* twx, const float2* twy, float2* S, float* scratch,
                        hipStream_t st) {
    const WmrGeom g = wmr_plain_geom(ny, nx);
    int rc;
    if ((rc = wmr_rows_fwd(frames, S, twx, scratch, g, nframes, st))) return rc;
    return wmr_cols_launch<2>(S, nullptr, twy, nullptr, nullptr, g, nframes, nullptr, 0u, 0, st);
}

int wmr_product_inverse(const float2* A, const float2* B, const int* ia, const int* ib, int npairs, int ny, int nx, const float2* twy,
                        float2* G, int whiten, float eps, unsigned flags, hipStream_t st) {
    const WmrGeom g = wmr_plain_geom(ny, nx);
    const unsigned grid = (unsigned)npairs * (unsigned)g.Wh;
    switch (g.H) {
#define X(N_, A_, B_, C_, L_)                                                                                                            \
    case N_:                                                                                                                             \
        if (whiten)                                                                                                                      \
            hipLaunchKernelGGL((k_wmr_cols_prod<Mix3<A_, B_, C_, L_>, true>), dim3(grid), dim3(L_), 0, st, A, B, ia, ib, G, twy, g, eps, flags);  \
        else                                                                                                                             \
            hipLaunchKernelGGL((k_wmr_cols_prod<Mix3<A_, B_, C_, L_>, false>), dim3(grid), dim3(L_), 0, st, A, B, ia, ib, G, twy, g, eps, flags); \
        break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix column kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int wmr_quads_per_frame(int ny) { return ((ny + 1) / 2 + WMR_Q - 1) / WMR_Q; }

int wmr_rows_magnitude(const float2* G, int npairs, int ny, int nx, const float2* twx, float* mag, float* part_val, int* part_idx,
                       unsigned* selw, int sel_stride, unsigned pred_bin, float* compact, hipStream_t st) {
    const WmrGeom g = wmr_plain_geom(ny, nx);
    const int qpf = wmr_quads_per_frame(ny), nquads = npairs * qpf;
    size_t lds = 0;
    int grid = 0, rc;
    switch (g.W) {
#define X(N_, A_, B_, C_, L_)                                                                                                       \
    case N_: {                                                                                                                      \
        using MX = Mix3<A_, B_, C_, L_>;                                                                                            \
        if ((rc = wmr_rows_launch<MX>(&k_wmr_rows_mag<MX>, nquads, st, &lds, &grid))) return rc;                                   \
        hipLaunchKernelGGL((k_wmr_rows_mag<MX>), dim3(grid), dim3(WMR_Q* L_), lds, st, G, mag, part_val, part_idx, twx, g, npairs, qpf, \
                           selw, sel_stride, pred_bin, compact);                                                                    \
    } break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix row kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int wmr_rows_real_out(const float2* G, int nframes, int ny, int nx, const float2* twx, float* out, hipStream_t st) {
    const WmrGeom g = wmr_plain_geom(ny, nx);
    const int qpf = wmr_quads_per_frame(ny), nquads = nframes * qpf;
    size_t lds = 0;
    int grid = 0, rc;
    switch (g.W) {
#define X(N_, A_, B_, C_, L_)                                                                                                             \
    case N_: {                                                                                                                            \
        using MX = Mix3<A_, B_, C_, L_>;                                                                                                  \
        if ((rc = wmr_rows_launch<MX>(&k_wmr_rows_out<MX>, nquads, st, &lds, &grid))) return rc;                                         \
        hipLaunchKernelGGL((k_wmr_rows_out<MX>), dim3(grid), dim3(WMR_Q* L_), lds, st, G, (const float*)nullptr, out, (float*)nullptr,    \
                           twx, (const float*)nullptr, g, nframes, qpf, 1.0f, 0u);                                                        \
    } break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix row kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int wmr_fft2d(const float* frames, int nframes, int ny, int nx, const float2* twx, const float2* twy, float2* S, float* scratch,
              float2* out, hipStream_t st) {
    int rc = wmr_forward_spectra(frames, nframes, ny, nx, twx, twy, S, scratch, st);
    if (rc) return rc;
    const WmrGeom g = wmr_plain_geom(ny, nx);
    const int qpf = wmr_quads_per_frame(ny), nquads = nframes * qpf;
    size_t lds = 0;
    int grid = 0;
    switch (g.W) {
#define X(N_, A_, B_, C_, L_)                                                                                                  \
    case N_: {                                                                                                                 \
        using MX = Mix3<A_, B_, C_, L_>;                                                                                       \
        if ((rc = wmr_rows_launch<MX>(&k_wmr_rows_spec<MX>, nquads, st, &lds, &grid))) return rc;                             \
        hipLaunchKernelGGL((k_wmr_rows_spec<MX>), dim3(grid), dim3(WMR_Q* L_), lds, st, (const float2*)S, out, g, nframes, qpf); \
    } break;
        B4D_WMR_LENGTHS(X)
#undef X
        default: return fail(B4D_ESIZE, "no mixed-radix row kernel for this length");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

}  // namespace b4d

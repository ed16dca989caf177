// b4d_fft2d.hpp -- batched 2-D real FFT pipeline kernels (row R2C, fused column, row C2R), the plan
// object and the launch dispatchers.  Included by the translation units that launch them
// (b4d_kernels.hip: fft2d / psd2d / autocorr2d; b4d_track.hip: xcorr2d / phase correlation).
//
// Data flow for a batch of real (ny, nx) frames ("rows first"):
//
//   K1 row_r2c   two real rows are packed as one complex row (z = a + i b), one FFT of length nx,
//                Hermitian split -> half spectra of both rows, kx = 0..nx/2-1, written in a
//                column-tile-major layout: tile ct holds CT adjacent kx for all ny rows contiguously
//                ([ct][y][c]), so that K2 streams whole tiles.  The (real) Nyquist bin kx = nx/2 of every
//                row goes to a small side array (batch, ny) handled by k_nyq.
//   K2 col       one workgroup owns a tile (CT = 16 columns x ny rows, 256 KiB at 2048^2) entirely in
//                registers: forward FFT along y, |F|^2 (PSD written shifted, with its Hermitian mirror),
//                inverse FFT along y of the power spectrum, written back in place.  Fusing forward and
//                inverse column passes removes one full read+write of the spectrum (SURVEY.md §8d counts
//                4 passes); two REAL power columns share one complex inverse transform.
//   k_nyq        the same for the single Nyquist column of each frame (1/1024 of the work).
//   K3 row_c2r   rebuilds the two-row packing from the half spectra, one inverse FFT of length nx,
//                shift + normalise -> two autocorrelation rows.
//
// All of them are HBM-bandwidth bound; see DESIGN.md for the byte accounting.
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

// Translation units that build a kernel with their own compiler flags (b4d_passes.hpp) define B4D_UNIT_TAG before including
// this header: the tag is a template argument of the kernels they launch, so that their instantiations are symbols of their
// own -- two units instantiating the SAME kernel name share one host-side stub, and which unit's device code a launch then
// runs is decided by registration order, not by the caller.
#ifndef B4D_UNIT_TAG
#define B4D_UNIT_TAG 0
#endif
// Which launch dispatchers a unit compiles (bit mask).  Every kernel a dispatcher names is instantiated in the including unit, with
// THAT unit's flags, under a symbol the other units share unless it carries the tag: a unit built with flags of its own therefore
// compiles only the pass it exists for (b4d_colpass.hip: the column pass, b4d_rowout.hip: the inverse row pass), so that no kernel
// symbol has two different device codes in the library (`make check-symbols`).
#define B4D_PASS_R2C 1
#define B4D_PASS_COL 2
#define B4D_PASS_NYQ 4
#define B4D_PASS_C2R 8
#ifndef B4D_UNIT_PASSES
#define B4D_UNIT_PASSES (B4D_PASS_R2C | B4D_PASS_COL | B4D_PASS_NYQ | B4D_PASS_C2R)
#endif

namespace b4d {

constexpr int E16 = 16;

// radix plans: N = 16 * R2 * R3 with 16 points per lane
constexpr int radix2(int n) { return n / 16 < 16 ? n / 16 : 16; }
constexpr int radix3(int n) { return n / (16 * radix2(n)); }
template <int N>
using RowGeom = FftGeom<N, E16, 16, radix2(N), radix3(N), 1>;
template <int N, int CI>
using ColGeom = FftGeom<N, E16, 16, radix2(N), radix3(N), CI>;
// single transforms (two image rows, or one Nyquist column) per workgroup: 256 lanes, fewer where the
// 64 KiB static LDS limit binds
constexpr int row_seq(int n) { return n == 4096 ? 1 : (n == 64 ? 32 : 256 / (n / 16)); }

// spectrum element index in the tile-major layout
__device__ __forceinline__ size_t spec_index(size_t frame, int nt, int ny, int ct_w, int y, int kx) {
    return ((frame * nt + (kx / ct_w)) * (size_t)ny + y) * ct_w + (kx % ct_w);
}

// the same index for power-of-two tile widths (16, or 8 at ny = 4096; always the case) with shifts and a 24-bit multiply: uniform
// part (frame) in 64 bits, lane part in 32 bits (< ny nx / 2 <= 2^23)
__device__ __forceinline__ size_t spec_index_pow2(size_t frame, int nt, int ny, int ct_w, int y, int kx) {
    const int cs = __builtin_ctz(ct_w);
    const unsigned lane = ((__umul24((unsigned)kx >> cs, (unsigned)ny) + (unsigned)y) << cs) + ((unsigned)kx & (unsigned)(ct_w - 1));
    return ((frame * nt * (size_t)ny) << cs) + lane;
}

// c = a * conj(b), optionally whitened: c / (|c| + eps)   (signal/tracking.py:280-281)
template <bool WHITEN>
__device__ __forceinline__ float2 cross_power(float2 a, float2 b, float eps) {
    float2 c = make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
    if (WHITEN) {
        // hardware square root and reciprocal (1 ulp each) and two multiplies: the IEEE sqrtf and the two divisions this replaces
        // were ~35 of the ~70 vector instructions the cross-power column pass spends per element (half of k_col_prod's stream)
        const float inv = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(c.x, c.x, c.y * c.y)) + eps);
        c.x *= inv;
        c.y *= inv;
    }
    return c;
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

// grid (ceil(ny/2/(SEQ*ITER)), batch); block T*SEQ.  ct_w = tile width (complex columns) of the spectrum layout.
// ITER > 1 (full frames only): a workgroup transforms ITER consecutive groups of SEQ row pairs and issues the loads of ALL of
// them before the first transform -- ITER x the bytes in flight per workgroup at the same LDS footprint (the exchange buffer,
// not the register file, caps this kernel at four workgroups per CU).
#ifndef B4D_K1_ITER
#define B4D_K1_ITER 1
#endif
template <int NX, int SEQ, bool SRC, int ITER = 1>
__global__ void __launch_bounds__((NX / E16) * SEQ)
k_row_r2c(const float* __restrict__ in, float2* __restrict__ spec, float* __restrict__ nyq_rows,
          const float2* __restrict__ tw, int ny, int ct_w, const RowSrc* __restrict__ srcs) {
    using G = RowGeom<NX>;
    constexpr int T = G::T, E = E16;
    constexpr bool WV = T <= 64;   // one transform = (part of) one wavefront: no workgroup barriers (fft_sync)
    static_assert(ITER == 1 || !SRC, "ROI sources run one group per workgroup");
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const size_t frame = blockIdx.y;
    if (SRC) {   // rows outside the ROI are zero: a workgroup without a ROI row writes NOTHING -- the column pass and k_nyq
                 // know the ROI rows (ColArgs::srcs / NyqArgs::srcs) and never read the others (a 121-px template in a
                 // 1024^2 frame: 88 % of the half spectrum is neither written here nor read there)
        const int wa = 2 * blockIdx.x * SEQ, wb = wa + 2 * SEQ;
        if (wb <= srcs[frame].y0 || wa >= srcs[frame].y1) return;
    }
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    float2 va[ITER][E];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        float2(&v)[E] = va[it];
        const int pair = (blockIdx.x * ITER + it) * SEQ + seq;
        const bool live = 2 * pair < ny;  // ragged last workgroup: idle transforms still take part in the barriers
        if (!live) {
#pragma unroll
            for (int j = 0; j < E; ++j) v[j] = make_float2(0.f, 0.f);
        } else if (SRC) {
            const RowSrc sd = srcs[frame];
            const int ya = 2 * pair, yb = ya + 1;
            const bool ina = ya >= sd.y0 && ya < sd.y1, inb = yb >= sd.y0 && yb < sd.y1;
            const float* r0 = in + ((size_t)sd.frame * ny + ya) * NX;
            const float* r1 = r0 + NX;
            const float inv = 1.0f / sd.denom;   // one division per item instead of 32 per lane (z-score: signal/tracking.py:308-311)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int x = u + T * j;
                const bool inx = x >= sd.x0 && x < sd.x1;
                v[j].x = (ina && inx) ? (r0[x] - sd.mean) * inv : 0.f;
                v[j].y = (inb && inx) ? (r1[x] - sd.mean) * inv : 0.f;
            }
        } else {
            const float* r0 = in + (frame * ny + 2 * (size_t)pair) * NX;
            const float* r1 = r0 + NX;
#pragma unroll
            for (int j = 0; j < E; ++j) v[j] = make_float2(r0[u + T * j], r1[u + T * j]);
        }
    }
    const int nt = (NX / 2) / ct_w;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        float2(&v)[E] = va[it];
        const int pair = (blockIdx.x * ITER + it) * SEQ + seq;
        const bool live = 2 * pair < ny;
        if (it > 0) fft_sync<WV>();   // the previous group's Hermitian split has read the buffer
        Fft3<G, 1, WV>::run(v, v, u, 0, lds, tw);
        fft_sync<WV>();
#pragma unroll
        for (int j = 0; j < E; ++j) lds[u + T * j] = v[j];
        fft_sync<WV>();
        if (live) {
#pragma unroll
            for (int j = 0; j < E / 2; ++j) {
                const int k = u + T * j;
                const float2 z = v[j], zr = lds[(NX - k) & (NX - 1)];
                float2 a = make_float2(0.5f * (z.x + zr.x), 0.5f * (z.y - zr.y));
                float2 b = make_float2(0.5f * (z.y + zr.y), 0.5f * (zr.x - z.x));
                if (k == 0) {  // DC bins are real; the (real) Nyquist bins of both rows go to the side array
                    const float2 zn = lds[NX / 2];
                    a = make_float2(z.x, 0.f);
                    b = make_float2(z.y, 0.f);
                    nyq_rows[frame * ny + 2 * pair] = zn.x;
                    nyq_rows[frame * ny + 2 * pair + 1] = zn.y;
                }
                const size_t o = spec_index(frame, nt, ny, ct_w, 2 * pair, k);
                spec[o] = a;
                spec[o + ct_w] = b;  // next row of the same tile
            }
        }
    }
}

// ------------------------------------------------------------------------------------ K2
enum ColMode { COL_PSD_AC = 0, COL_FORWARD = 2 };   // (1 was the round-2 full-spectrum store: b4d_spectrum.hip replaced it)

struct ColArgs {
    float2* spec;     // tile-major half spectra, in/out
    float* psd;       // (batch, ny, nx) or null
    const float2* tw;
    const float2* tw_inv;  // the SAME table through a second, formally unrelated pointer: keeps the compiler from
                           // holding the forward pass's 30 twiddles in registers (or scratch) for the inverse pass
    float psd_scale;
    int nx;
    unsigned flags;
    int half_rows;    // COL_PSD_AC: the consumer uses R[-y,-x] = R[y,x]: only rows 0..ny/2+1 of the tile are stored
    const RowSrc* srcs;  // COL_FORWARD, optional: item i holds data in rows [srcs[i].y0, srcs[i].y1) only (zero-embedded ROI);
                         // the other rows are taken as zero WITHOUT being read (k_row_r2c does not write them)
#ifdef B4D_DIAG
    unsigned long long* diag;  // diagnostic build only: 8 s_memtime stamps per workgroup
#endif
};

#ifdef B4D_DIAG
// Diagnostic build (never shipped, never timed as a whole): phase stamps of wave 0 of each workgroup.
#define B4D_STAMP(i)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t__;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (p.diag && threadIdx.x == 0) p.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = t__; \
    } while (0)
#ifdef B4D_DIAG_DRAIN   // additionally separate issue from completion (perturbs the kernel: every phase waits for its memory traffic)
#define B4D_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define B4D_DRAIN() do {} while (0)
#endif
#else
#define B4D_STAMP(i) do {} while (0)
#define B4D_DRAIN() do {} while (0)
#endif

// Column-tile geometry: a lane holds NC = 2 adjacent columns (one 16-byte load per row), CPT lanes span the
// tile (CT = NC*CPT columns = whole 128-B lines per row at CT = 16), T = NY/16 lanes run along y.
//   NY <= 2048: CPT = 8 -> CT = 16; at 2048 the 256-KiB tile lives in the registers of 1024 lanes (16 waves,
//               <= 128 VGPRs each: 64 of data + butterflies; 4 waves per SIMD hide the LDS / twiddle latencies --
//               measured 1.23x faster than 512 lanes x 4 columns at 2 waves per SIMD)
//   NY == 4096: CPT = 4 -> CT = 8 (a 16-column tile would need the whole register file)
// __launch_bounds__(THREADS, 4) caps every variant at 128 VGPRs so that smaller NY run several workgroups per CU.
// SPLIT workgroups share one memory tile (each takes CT / SPLIT adjacent columns of it): the layout keeps whole
// 128-B lines per row while two 512-lane workgroups fit one CU and overlap each other's memory and butterfly phases.
template <int NY, int SPLIT = 1>
struct ColCfg {
    static constexpr int NC = 2;
    static constexpr int CPT = (NY == 4096 ? 4 : 8) / SPLIT;
    static constexpr int CT = NC * CPT * SPLIT;   // columns per MEMORY tile
    static constexpr int THREADS = CPT * (NY / E16);
    static constexpr int WAVES_PER_EU = THREADS >= 256 ? 4 : 1;
    using G = ColGeom<NY, CPT>;
    static constexpr int SB = 1;  // sets per LDS round trip (one exchange region = 132 KiB at NY = 2048)
    static constexpr size_t LDS_BYTES = sizeof(float2) * (size_t)G::LDS_ELEMS * CPT * SB;
    static constexpr bool SERIAL = true;  // at the 128-VGPR cap: pin the per-set order
};

// 4-byte-aligned 16-byte store (gfx950 runs in unaligned access mode: one global_store_dwordx4)
struct __attribute__((packed, aligned(4))) float4_u {
    float x, y, z, w;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));   // plain vector values for accesses through address-space pointers
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define B4D_GLOBAL __attribute__((address_space(1)))
// Address = uniform base + 32-bit BYTE offset of the lane: the form global_load/store take with the base in SGPRs and ONE offset
// VGPR.  With element indices the compiler cannot fold the scaling into a 32-bit offset (it could wrap) and builds 64-bit
// per-lane addresses: two VGPRs each, kept live across the transforms at the 128-register cap of the 1024-lane kernels.
template <class V>
__device__ __forceinline__ V B4D_GLOBAL* at_bytes(void* base, unsigned boff) {
    return (V B4D_GLOBAL*)((char B4D_GLOBAL*)base + boff);
}
template <class V>
__device__ __forceinline__ const V B4D_GLOBAL* at_bytes(const void* base, unsigned boff) {
    return (const V B4D_GLOBAL*)((const char B4D_GLOBAL*)base + boff);
}
// a uniform pointer pinned in an SGPR pair and opaque to reassociation: `sgpr_base(p + const) + lane offset` stays
// "scalar base + 32-bit lane offset" instead of being folded into a 64-bit per-lane address plus constants
template <class V>
__device__ __forceinline__ V* sgpr_base(V* p) {
    asm volatile("" : "+s"(p));
    return p;
}

template <int NC>
__device__ __forceinline__ void store_cols(f32x2 B4D_GLOBAL* rowp, const float2 (&c)[NC]) {
#pragma unroll
    for (int h = 0; h < NC / 2; ++h)
        *(f32x4 B4D_GLOBAL*)(rowp + 2 * h) = f32x4{c[2 * h].x, c[2 * h].y, c[2 * h + 1].x, c[2 * h + 1].y};
}
template <int NC>
__device__ __forceinline__ void store_cols(float2* __restrict__ rowp, const float2 (&c)[NC]) {
#pragma unroll
    for (int h = 0; h < NC / 2; ++h)
        *reinterpret_cast<float4*>(rowp + 2 * h) = make_float4(c[2 * h].x, c[2 * h].y, c[2 * h + 1].x, c[2 * h + 1].y);
}

// grid (nt, batch), block ColCfg<NY>::THREADS.
template <int NY, int MODE, int SPLIT = 1, int UNIT = 0>
__global__ void __launch_bounds__((ColCfg<NY, SPLIT>::THREADS), (ColCfg<NY, SPLIT>::WAVES_PER_EU)) k_col(ColArgs p) {
    using Cfg = ColCfg<NY, SPLIT>;
    using G = typename Cfg::G;
    constexpr int T = G::T, E = E16, NC = Cfg::NC, CPT = Cfg::CPT, CT = Cfg::CT;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int cp = threadIdx.x % CPT, u = threadIdx.x / CPT;
    const int nt = gridDim.x / SPLIT;
    // Workgroups are dealt round-robin over the 8 XCDs (linear workgroup id % 8).  Neighbouring column tiles write
    // the two 64-B halves of every 128-B PSD line and share the 64-B sectors of the Hermitian-mirror stores (the
    // mirror of 16 aligned columns is misaligned by one float): those merge only in a common L2.  With batch % 8 == 0
    // every tile of frame f runs on XCD f % 8 (measured -6% on the 2048^2 kernel against per-XCD tile ranges, which
    // leave one tile boundary in eight split across two L2s); otherwise each XCD takes a contiguous tile range.
    // Speed only: any placement is correct.  The SPLIT parts of a tile are consecutive workgroups of one XCD.
    int slot, fr;
    if (gridDim.y % 8 == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, k = lin / 8;
        fr = 8 * (k / gridDim.x) + lin % 8;
        slot = k % gridDim.x;
    } else {
        slot = (nt % 8 == 0) ? (blockIdx.x % 8) * (nt * SPLIT / 8) + blockIdx.x / 8 : blockIdx.x;
        fr = blockIdx.y;
    }
    const int ct = slot / SPLIT;
    const int cpm = cp + CPT * (slot % SPLIT);   // lane position across the memory tile
    const size_t frame = fr;
    const int nx = p.nx;
    float2* tile = p.spec + ((frame * nt + ct) * (size_t)NY) * CT;
    const unsigned toff = (unsigned)u * CT + NC * cpm;  // element offset of (row u, first column of this lane)
    float2 v[NC][E];
    B4D_STAMP(0);
    // column pairs in issue order: the first pair's loads complete first, so its butterflies start while the
    // second pair is still streaming in (loads return in order; the compiler's counted vmcnt does the rest)
    int ry0 = 0, ry1 = NY;
    if (MODE == COL_FORWARD && p.srcs) {
        ry0 = p.srcs[fr].y0;
        ry1 = p.srcs[fr].y1;
    }
#pragma unroll
    for (int h = 0; h < NC / 2; ++h) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (MODE == COL_FORWARD) {   // clamped row + select: the load stays unconditional (no branch per load), rows outside the
                                         // ROI re-read a ROI row that other lanes fetch anyway
                const int ky = u + T * j, kc = min(max(ky, ry0), ry1 - 1);
                float4 q = *reinterpret_cast<const float4*>(tile + (size_t)kc * CT + NC * cpm + 2 * h);
                if (ky < ry0 || ky >= ry1) q = make_float4(0.f, 0.f, 0.f, 0.f);
                v[2 * h][j] = make_float2(q.x, q.y);
                v[2 * h + 1][j] = make_float2(q.z, q.w);
            } else {
                const f32x4 q = *at_bytes<f32x4>(sgpr_base(tile + (size_t)(T * j * CT) + 2 * h), toff * 8u);
                v[2 * h][j] = make_float2(q.x, q.y);
                v[2 * h + 1][j] = make_float2(q.z, q.w);
            }
        }
    }
    B4D_DRAIN();
    B4D_STAMP(1);
    Fft3<G, 1>::template run_sets<NC, Cfg::SERIAL, MODE != COL_PSD_AC, Cfg::SB>(v, u, cp, lds, p.tw);
    B4D_STAMP(2);
    // v[c][j] = F[ky = u + T j][kx0 + c]   (COL_PSD_AC: after the stage-3 butterflies done below)
    const int kx0 = ct * CT + NC * cpm;

    if (MODE == COL_FORWARD) {  // keep the 2-D half spectrum in the tile
        // laundered offset: otherwise the 16 store addresses (= the load addresses) stay live in 32 registers across the
        // whole transform and the radix stages spill (8 dwords at 512 rows ... 28 at 4096)
        unsigned toff3 = toff;
        asm volatile("" : "+v"(toff3));
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float2 c[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) c[k] = v[k][j];
            store_cols<NC>(tile + (size_t)(T * j * CT) + toff3, c);
        }
        return;
    }

    // ---- COL_PSD_AC: power spectrum, optional PSD store (direct + Hermitian mirror), inverse transform.
    // Two REAL power columns ride one complex transform (Pa + i Pb), separated afterwards with the
    // Hermitian symmetry in y: half the butterflies and LDS traffic of a column-by-column inverse.
    // The last forward butterflies are interleaved with the |F|^2 epilogue (register pressure), and the
    // row index is laundered so that the 32 store offsets are not precomputed at kernel entry.
    const float s = p.psd_scale;
    float* psd = p.psd ? p.psd + frame * (size_t)NY * nx : nullptr;
    int uu = u;
    asm volatile("" : "+v"(uu));
    float2 w[NC / 2][E];
    constexpr int B3 = E / G::R3;
#pragma unroll
    for (int a = 0; a < B3; ++a) {
#pragma unroll
        for (int k = 0; k < NC; ++k) Fft3<G, 1>::stage3_one(v[k], a);
#pragma unroll
        for (int q = 0; q < G::R3; ++q) {
            const int j = a + B3 * q;
            const int ky = uu + T * j;
            float pw[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) pw[k] = v[k][j].x * v[k][j].x + v[k][j].y * v[k][j].y;
#ifdef B4D_EXP_PSD_TILED
            if (psd) {   // timing-only: same bytes, tile-contiguous addresses (DRAM page locality experiment)
                float* pt = psd + ((size_t)ct * NY + ky) * 32 + NC * cpm;
                *reinterpret_cast<float2*>(pt) = make_float2(pw[0] * s, pw[1] * s);
                *reinterpret_cast<float2*>(pt + 16) = make_float2(pw[1] * s, pw[0] * s);
            }
            if (false) {
#else
            if (psd) {
#endif
                const unsigned rd = (unsigned)((ky + NY / 2) & (NY - 1)) * nx, rm = (unsigned)((NY / 2 - ky) & (NY - 1)) * nx;
                if (NC == 4)
                    *reinterpret_cast<float4*>(&psd[rd + nx / 2 + kx0]) =
                        make_float4(pw[0] * s, pw[1] * s, pw[2 % NC] * s, pw[3 % NC] * s);
                else
                    *at_bytes<f32x2>(psd, (rd + nx / 2 + kx0) * 4u) = f32x2{pw[0] * s, pw[1] * s};
                // Hermitian mirror: columns nx/2 - kx0 - k, descending -> one reversed (4-byte aligned) vector store
                if (NC == 4 && kx0 >= 1) {
                    float4_u m;
                    m.x = pw[3 % NC] * s;
                    m.y = pw[2 % NC] * s;
                    m.z = pw[1] * s;
                    m.w = pw[0] * s;
                    *reinterpret_cast<float4_u*>(&psd[rm + nx / 2 - kx0 - 3]) = m;
                } else {
#if defined(B4D_EXP_NOMIRROR)
#elif defined(B4D_EXP_ALIGNED_MIRROR)
                    *reinterpret_cast<float2*>(&psd[rm + nx / 2 - kx0 - 2]) = make_float2(pw[1] * s, pw[0] * s);
#elif defined(B4D_PAIR_MIRROR)
                    // columns nx/2 - kx descend as kx ascends: the 8-byte aligned pairs are (kx0 + 2, kx0 + 1), i.e. the
                    // next lane's first column and this lane's second
                    const float nb = __shfl_down(pw[0], 1, 64);
                    if (cpm < CT / NC - 1)
                        *reinterpret_cast<float2*>(&psd[rm + nx / 2 - kx0 - 2]) = make_float2(nb * s, pw[1] * s);
                    else
                        psd[rm + nx / 2 - kx0 - 1] = pw[1] * s;
                    if (cpm == 0 && kx0 >= 1) psd[rm + nx / 2 - kx0] = pw[0] * s;
#else
                    // columns nx/2 - kx0 - 1, nx/2 - kx0 (column 0 has no mirror).  Two dword stores: ONE 8-byte store on the 4-byte
                    // boundary (the mirror of an aligned pair starts one float off) measured +6 % on the kernel
                    const unsigned mo = (rm + nx / 2 - kx0 - 1) * 4u;
                    *at_bytes<float>(psd, mo) = pw[1] * s;
                    if (kx0 >= 1) *at_bytes<float>(psd, mo + 4u) = pw[0] * s;
#endif
                }
            }
            // inverse inputs, already (im, re)-swapped so that the forward code inverts: Pa + i Pb -> (Pb, Pa)
#pragma unroll
            for (int h = 0; h < NC / 2; ++h) w[h][j] = make_float2(pw[2 * h + 1], pw[2 * h]);
        }
    }
    B4D_STAMP(3);
    B4D_DRAIN();
    B4D_STAMP(4);
    if (kx0 == 0 && uu == 0 && (p.flags & B4D_REMOVE_MEAN)) w[0][0].y = 0.f;  // DC bin: ky = 0 <-> u = 0, j = 0
    // The lane position is derived AGAIN from the (laundered) thread index: otherwise the compiler keeps the load addresses and
    // the ~40 twiddle / LDS byte offsets of the forward pass alive, or u, cp and the tile offset in three registers where one
    // does -- at the 128-register cap one spilled dword reloaded here would be a scratch load, i.e. a vmcnt(0) that drains
    // every PSD store before the inverse transform may start
    const float2* tw2 = p.tw_inv;
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int cp2 = tid2 % CPT, u2 = tid2 / CPT;
    const unsigned toff2 = (unsigned)u2 * CT + NC * (cp2 + CPT * (slot % SPLIT));
    __syncthreads();
    Fft3<G, 1>::template run_sets<NC / 2, Cfg::SERIAL, true, Cfg::SB>(w, u2, cp2, lds, tw2);
    B4D_STAMP(5);
    // V[y] = Ga[y] + i Gb[y] with Ga, Gb Hermitian in y: split with V[-y], one set at a time
    float2 vr[NC / 2][E];
    constexpr int SETE = G::LDS_ELEMS * CPT;  // one exchange region (complex elements)
#pragma unroll
    for (int b = 0; b < NC / 2; b += Cfg::SB) {
        __syncthreads();
#pragma unroll
        for (int h = b; h < b + Cfg::SB; ++h)
#pragma unroll
            for (int j = 0; j < E; ++j) lds[(h - b) * SETE + (u2 + T * j) * CPT + cp2] = make_float2(w[h][j].y, w[h][j].x);
        __syncthreads();
#pragma unroll
        for (int h = b; h < b + Cfg::SB; ++h)
#pragma unroll
            for (int j = 0; j < E; ++j) vr[h][j] = lds[(h - b) * SETE + ((NY - (u2 + T * j)) & (NY - 1)) * CPT + cp2];
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        float2 c[NC];
#pragma unroll
        for (int h = 0; h < NC / 2; ++h) {
            const float2 z = make_float2(w[h][j].y, w[h][j].x), zr = vr[h][j];
            c[2 * h] = make_float2(0.5f * (z.x + zr.x), 0.5f * (z.y - zr.y));
            c[2 * h + 1] = make_float2(0.5f * (z.y + zr.y), 0.5f * (zr.x - z.x));
        }
        // streaming (non-temporal) stores: the tile is read once more, by the row pass, after a gigabyte of other traffic -- kept
        // out of L2 it leaves the cache to the PSD half lines that do meet there (-3 % on this kernel; the PSD stores themselves
        // must stay cached: non-temporal they doubled the kernel)
        if (!p.half_rows || u2 + T * j <= NY / 2 + 1) {
            static_assert(NC == 2, "one 16-byte store per row");
            __builtin_nontemporal_store(f32x4{c[0].x, c[0].y, c[1].x, c[1].y},
                                        (f32x4 B4D_GLOBAL*)at_bytes<f32x2>(sgpr_base(tile + (size_t)(T * j * CT)), toff2 * 8u));
        }
    }
    B4D_STAMP(6);
    B4D_DRAIN();
    B4D_STAMP(7);
}

// ------------------------------------------------------------------------------------ Nyquist column
enum NyqMode { NYQ_PSD_AC = 0, NYQ_FORWARD = 2, NYQ_PROD = 3, NYQ_PROD_WHITEN = 4 };

struct NyqArgs {
    const float* rows;   // (items, NY) real Nyquist bins from K1                      [PSD_AC, FORWARD]
    float2* f_out;       // (items, NY) complex column spectra                          [FORWARD]
    float* g_out;        // (items, NY) real inverse-column output, consumed by K3      [PSD_AC, PROD]
    float* psd;          // (items, NY, nx) or null: column 0 of the shifted PSD        [PSD_AC]
    const float2* fa;    // column spectra of the two operands + per-pair indices       [PROD]
    const float2* fb;
    const int* idx_a;
    const int* idx_b;
    const float2* tw;
    float psd_scale, eps;
    int nx, items;
    const RowSrc* srcs;  // NYQ_FORWARD, optional: rows outside [srcs[i].y0, srcs[i].y1) are zero and were not written by k_row_r2c
};

// grid (ceil(items/SEQ)), block T*SEQ: one length-NY transform per T lanes.
template <int NY, int SEQ, int MODE>
__global__ void __launch_bounds__((NY / E16) * SEQ) k_nyq(NyqArgs p) {
    using G = RowGeom<NY>;
    constexpr int T = G::T, E = E16;
    constexpr bool WV = T <= 64;
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const int item = blockIdx.x * SEQ + seq;
    const bool live = item < p.items;
    const size_t it = live ? item : 0;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    float2 v[E];
    if (MODE == NYQ_PROD || MODE == NYQ_PROD_WHITEN) {
        const size_t ia = p.idx_a ? p.idx_a[it] : it, ib = p.idx_b ? p.idx_b[it] : it;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int ky = u + T * j;
            const float2 c = cross_power<MODE == NYQ_PROD_WHITEN>(p.fa[ia * NY + ky], p.fb[ib * NY + ky], p.eps);
            v[j] = make_float2(c.y, c.x);  // swapped: inverse transform
        }
        Fft3<G, 1, WV>::run(v, v, u, 0, lds, p.tw);
        if (live) {
#pragma unroll
            for (int j = 0; j < E; ++j) p.g_out[it * NY + u + T * j] = v[j].y;
        }
        return;
    }
    int ry0 = 0, ry1 = NY;
    if (MODE == NYQ_FORWARD && p.srcs) {
        ry0 = p.srcs[it].y0;
        ry1 = p.srcs[it].y1;
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int ky = u + T * j;
        v[j] = make_float2((ky >= ry0 && ky < ry1) ? p.rows[it * NY + ky] : 0.f, 0.f);
    }
    Fft3<G, 1, WV>::run(v, v, u, 0, lds, p.tw);
    if (MODE == NYQ_FORWARD) {
        if (live) {
#pragma unroll
            for (int j = 0; j < E; ++j) p.f_out[it * NY + u + T * j] = v[j];
        }
        return;
    }
    // NYQ_PSD_AC
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const float pw = v[j].x * v[j].x + v[j].y * v[j].y;
        if (live && p.psd) p.psd[(it * NY + ((u + T * j + NY / 2) & (NY - 1))) * (size_t)p.nx] = pw * p.psd_scale;
        v[j] = make_float2(0.f, pw);
    }
    fft_sync<WV>();
    Fft3<G, 1, WV>::run(v, v, u, 0, lds, p.tw);
    if (live) {
#pragma unroll
        for (int j = 0; j < E; ++j) p.g_out[it * NY + u + T * j] = v[j].y;
    }
}

// ------------------------------------------------------------------------------------ K3
struct RowOutArgs {
    const float2* g;     // tile-major inverse-column output
    const float* gnyq;   // (batch, ny) inverse-column output of the Nyquist column (real)
    float* out;          // (batch, ny, nx) float32 shifted
    float* peak;         // (batch) zero-lag values
    const float2* tw;
    float scale;         // used when !NORM_PEAK
    int ny, ct_w;
    unsigned flags;
    float* part_val;     // C2R_MAG: per-workgroup arg-max partials (batch, gridDim.x)
    int* part_idx;
    // C2R_MAG, optional (selw != null): the exact median's first select step, fused into the pass that produces the map.
    // pred_bin = top-11-bit key bin the median is EXPECTED in; per frame, at selw + frame * sel_stride: word 1 += elements
    // below that bin, word 2 += elements in it, word 3 = append cursor of compact + frame * ny * nx, which receives the bin's
    // elements (all zeroed by the caller).  b4d_track.hip checks the expectation and finishes on the gathered ~10 %.
    unsigned* selw;
    int sel_stride;
    unsigned pred_bin;
    float* compact;
    int half;            // C2R_OUT: the map is even (R[-y,-x] = R[y,x], autocorrelation): transform rows 0..ny/2 only
                         // and write every row together with its point mirror
    // C2R_MAG without the map (tracking): partials, counts and the gathered bin are everything the common path reads of the
    // 4 ny nx bytes -- except the 3 x 3 neighbourhood of the peak, which C2R_ROWS writes afterwards (three row pairs).
    int nomap;           // C2R_MAG: do not store the map
    int nblk;            // C2R_ROWS: partials per frame
    const unsigned* gate;   // C2R_MAG, optional: skip frame f when gate[f * gate_stride] != 0 (pairs that need no full map)
    int gate_stride;
};

enum RowOutMode { C2R_OUT = 0, C2R_PEAK = 1, C2R_MAG = 2, C2R_ROWS = 3 };

// (value, flat index) arg-max with NumPy's first-occurrence rule: larger value wins, ties go to the lower index
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// grid (ceil(ny/2/SEQ), batch) -- or (1, batch) for C2R_PEAK; block T*SEQ.
//   C2R_OUT   shifted real output, scaled (flags & NORM_PEAK: by 1/peak[frame], zero lag forced to 1)
//   C2R_PEAK  only the zero-lag value of each frame -> peak[frame]
//   C2R_MAG   |value| * scale (signal/tracking.py:283-285) + per-workgroup arg-max partials
template <int NX, int SEQ, int MODE, int UNIT = 0>
__global__ void __launch_bounds__((NX / E16) * SEQ) k_row_c2r(RowOutArgs p) {
    using G = RowGeom<NX>;
    constexpr int T = G::T, E = E16;
    constexpr bool WV = T <= 64;   // the transform's own exchanges are wave-local; the reductions ACROSS transforms below keep s_barrier
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const size_t frame = blockIdx.y;
    const int ny = p.ny, ct_w = p.ct_w, nt = (NX / 2) / ct_w;
    if (MODE == C2R_MAG && p.gate && p.gate[frame * p.gate_stride]) return;
    int peak_pair = 0;
    if (MODE == C2R_ROWS) {   // first-occurrence arg-max over the frame's partials -> the row pair that holds the peak
        float bv = -1.0f;
        int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < p.nblk; i += T * SEQ) argmax_merge(bv, bi, p.part_val[frame * p.nblk + i], p.part_idx[frame * p.nblk + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(bv, o, 64);
            const int oi = __shfl_down(bi, o, 64);
            argmax_merge(bv, bi, ov, oi);
        }
        float* sv = reinterpret_cast<float*>(lds_all);
        int* si = reinterpret_cast<int*>(lds_all) + 32;
        if ((threadIdx.x & 63) == 0) {
            sv[threadIdx.x >> 6] = bv;
            si[threadIdx.x >> 6] = bi;
        }
        __syncthreads();
        bv = sv[0];
        bi = si[0];
        for (int i = 1; i < (T * SEQ + 63) / 64; ++i) argmax_merge(bv, bi, sv[i], si[i]);
        __syncthreads();
        const int yr = bi == 0x7fffffff ? 0 : bi / NX;          // row of the shifted map
        peak_pair = ((yr + ny / 2) & (ny - 1)) >> 1;
    }
    const int nblk = gridDim.x, bx = blockIdx.x;
    // C2R_ROWS: the pairs peak - 1, peak, peak + 1 (cyclic) hold rows y - 1 .. y + 1 of the peak row y, whatever its parity
    const int pair = MODE == C2R_PEAK   ? 0
                     : MODE == C2R_ROWS ? (peak_pair + ny / 2 - 1 + min(bx * SEQ + seq, 2)) % (ny / 2)
                                        : bx * SEQ + seq;
    const bool live = (MODE == C2R_OUT && p.half) ? 2 * pair <= ny / 2 : 2 * pair < ny;
    const int yl = live ? 2 * pair : 0;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    float2 v[E];
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        const int k = u + T * j;
        // C2R_MAG (issue-bound, profiles/r03_pmc_cfg3_sq.txt): the tile index without the integer division by the run-time tile width
        // (a power of two) and with a 24-bit multiply -- the generic form costs ~40 quarter-rate multiplies per lane here.  The cfg2
        // instantiation keeps spec_index: there the same change cost K3 1 % (instruction order).
        const size_t o = MODE == C2R_MAG ? spec_index_pow2(frame, nt, ny, ct_w, yl, k) : spec_index(frame, nt, ny, ct_w, yl, k);
        // read once, written once: streaming hints on both sides of the row pass (-4 % on the cfg2 kernel)
        const f32x2 a_ = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p.g + o)),
                    b_ = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p.g + o + ct_w));
        const float2 a = make_float2(a_.x, a_.y), b = make_float2(b_.x, b_.y);
        if (k == 0) {  // DC and Nyquist bins of both rows are real
            v[j] = make_float2(b.x, a.x);                                                       // swap(A_dc + i B_dc)
            lds[NX / 2] = make_float2(p.gnyq[frame * ny + yl + 1], p.gnyq[frame * ny + yl]);  // swap(A_nyq + i B_nyq)
        } else {
            v[j] = make_float2(a.y + b.x, a.x - b.y);        // swap(A + iB)
            lds[NX - k] = make_float2(b.x - a.y, a.x + b.y);  // swap(conj A + i conj B)
        }
    }
    fft_sync<WV>();
#pragma unroll
    for (int j = E / 2; j < E; ++j) v[j] = lds[u + T * j];
    fft_sync<WV>();
    Fft3<G, 1, WV>::run(v, v, u, 0, lds, p.tw);
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
        bool unit_peak = false;   // corr / max|corr| only if the peak is > 0 (signal/corr.py:247-250): a constant frame stays 0
        if (norm) {
            const float pk = p.peak[frame];
            unit_peak = pk > 0.f;
            s = unit_peak ? 1.0f / pk : p.scale;
        }
        // half mode: rows 0..ny/2 are written directly, rows 1..ny/2-1 also as their point mirrors
        const bool wr1 = !p.half || y0 + 1 <= ny / 2;
        const bool mir0 = p.half && y0 >= 1 && y0 < ny / 2, mir1 = p.half && y0 + 1 < ny / 2;
        float* q0 = p.out + (frame * ny + ((ny / 2 - y0) & (ny - 1))) * (size_t)NX;
        float* q1 = p.out + (frame * ny + ((ny / 2 - y0 - 1) & (ny - 1))) * (size_t)NX;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int x = u + T * j, c = (x + NX / 2) & (NX - 1), cm = (NX / 2 - x) & (NX - 1);
            float r0 = v[j].y * s;
            const float r1 = v[j].x * s;
            if (unit_peak && pair == 0 && x == 0) r0 = 1.0f;  // peak normalisation: zero lag is 1 by definition
            __builtin_nontemporal_store(r0, o0 + c);
            if (wr1) __builtin_nontemporal_store(r1, o1 + c);
            if (mir0) __builtin_nontemporal_store(r0, q0 + cm);
            if (mir1) __builtin_nontemporal_store(r1, q1 + cm);
        }
        return;
    }
    if (MODE == C2R_ROWS) {   // the same values the map would hold, for the rows around the peak only
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int x = u + T * j, c = (x + NX / 2) & (NX - 1);
            o0[c] = fabsf(v[j].y * p.scale);
            o1[c] = fabsf(v[j].x * p.scale);
        }
        return;
    }
    // ---- C2R_MAG
    float bv = -1.0f;
    int bi = 0x7fffffff;
    // expected-median-bin bookkeeping (selw != null): per value one shift and two compares next to the store; bit j / 16 + j of
    // `hits` = value j of row a / b is in the bin (magnitudes are >= 0: the key's top 11 bits are 1024 + (bits >> 21))
    const unsigned pbx = (p.selw && p.pred_bin >= 1024u) ? p.pred_bin - 1024u : 0xffffffffu;
    unsigned cnt = 0, low = 0, hits = 0;
    if (live) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int x = u + T * j, c = (x + NX / 2) & (NX - 1);
            const float m0 = fabsf(v[j].y * p.scale), m1 = fabsf(v[j].x * p.scale);
            if (!p.nomap) {
                o0[c] = m0;
                o1[c] = m1;
            }
            if (pbx != 0xffffffffu) {
                // a NaN magnitude has the key bits 0x3fe: above every bin a finite median can be expected in -- neither counted
                // in the bin nor below it, without a select
                const unsigned b0 = __float_as_uint(m0) >> 21, b1 = __float_as_uint(m1) >> 21;
                hits |= (b0 == pbx ? 1u : 0u) << j | (b1 == pbx ? 1u : 0u) << (16 + j);
                low += (b0 < pbx ? 1u : 0u) + (b1 < pbx ? 1u : 0u);
            }
        }
        cnt = __popc(hits);
        // first-occurrence arg-max of the lane's 32 values: walked in ASCENDING flat index (row a before row b = row a + 1; inside a
        // row the shifted column of register j is u + T ((j + E/2) % E)), so a strict `>` keeps the first of equal maxima and the
        // index is rebuilt once from the winning slot -- three instructions per value instead of a (value, index) merge per value
        float lv = -1.0f;
        int ls = 0;
#pragma unroll
        for (int t = 0; t < 2 * E; ++t) {
            const int j = ((t % E) + E / 2) % E;
            const float m = fabsf((t < E ? v[j].y : v[j].x) * p.scale);
            if (m > lv) {   // NaN never wins (as argmax_merge)
                lv = m;
                ls = t;
            }
        }
        if (lv > -1.0f) {
            bv = lv;
            bi = (ls < E ? ra : rb) * NX + u + T * (ls % E);
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
        p.part_val[frame * nblk + bx] = bv;
        p.part_idx[frame * nblk + bx] = bi;
    }
    if (p.selw && p.pred_bin) {
        // counts of the magnitudes below / inside the expected median bin and the bin's elements themselves: lane counts,
        // a workgroup scan, THREE global atomics per workgroup (no per-element atomics, no second read of the map)
        constexpr int NW = (T * SEQ + 63) / 64;
        unsigned incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) low += __shfl_down(low, o, 64);
        unsigned* hl = reinterpret_cast<unsigned*>(lds_all) + 64;   // words 0..63: the arg-max partials above
        __syncthreads();
        if (lane == 63) hl[w] = incl;          // wave totals of the bin
        if (lane == 0) hl[NW + 1 + w] = low;   // wave totals below the bin
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned run = 0, lo = 0;
            for (int i = 0; i < NW; ++i) {
                const unsigned t = hl[i];
                hl[i] = run;
                run += t;
                lo += hl[NW + 1 + i];
            }
            unsigned* sw = p.selw + frame * p.sel_stride;
            if (lo) atomicAdd(&sw[1], lo);
            if (run) atomicAdd(&sw[2], run);
            hl[NW] = run ? atomicAdd(&sw[3], run) : 0u;
            hl[2 * NW + 1] = run;
        }
        __syncthreads();
        // The bin's elements go through the (free) exchange buffer: masked 4-byte LDS writes at the lane's offset, then one
        // coalesced copy to `compact` -- appended straight to global memory they cost as many (mostly empty) store
        // instructions as the map itself (+34 % on this kernel).
        constexpr int STAGE_WORDS = 2 * E * T * SEQ;   // every value of the workgroup may be in the bin
        static_assert(sizeof(lds_all) >= sizeof(unsigned) * (64 + 2 * NW + 2 + STAGE_WORDS), "exchange buffer too small for the staging");
        float* stg = reinterpret_cast<float*>(hl + 2 * NW + 2);
        if (cnt) {
            float* q = stg + hl[w] + (incl - cnt);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if (hits & (1u << j)) *q++ = fabsf(v[j].y * p.scale);
                if (hits & (1u << (16 + j))) *q++ = fabsf(v[j].x * p.scale);
            }
        }
        __syncthreads();
        const unsigned total = hl[2 * NW + 1];
        float* dst = p.compact + frame * (size_t)ny * NX + hl[NW];
        for (unsigned i = threadIdx.x; i < total; i += T * SEQ) dst[i] = stg[i];
    }
}

}  // namespace b4d

// ===================================================================================== host
using namespace b4d;

struct b4d_plan {
    std::recursive_mutex mu;   // host-side re-entrancy: one plan may be called from several host threads (one stream)
    int ny, nx, chunk, ct_w;
    float2* tw_x = nullptr;    // nx-point twiddles
    float2* tw_y = nullptr;    // ny-point twiddles
    float2* spec = nullptr;    // chunk * ny * nx/2
    float* nyq_rows = nullptr; // chunk * ny: Nyquist bins after the row pass
    float* gnyq = nullptr;     // chunk * ny: Nyquist column after the inverse column pass
    float* peak = nullptr;     // chunk
    size_t ws_bytes = 0;
    void* track_ws = nullptr;  // lazily grown arena of the xcorr / tracking entry points
    size_t track_bytes = 0;
    // general-length plans (b4d_general.hip): dense DFT matrices + three chunk-sized complex buffers
    bool general = false;
    float2* wx = nullptr;
    float2* wy = nullptr;
    float2* gbuf1 = nullptr;
    float2* gbuf2 = nullptr;
    float2* gbuf3 = nullptr;
    // general lengths beyond the DFT-matrix range: both axes through the fused P * A * B row transform (b4d_wiener.hip)
    bool large = false;
    // both sides have in-register three-radix kernels (b4d_wiener_mr.hip): the transform-based entry points take those passes,
    // whatever the size class (228-px aggregator tiles as well as 2560 x 2160 frames); tw_x / tw_y are built for it
    bool wmr = false;
    // second stream of the two-lane drivers (Lanes below; the library's shared lane_stream(0); fork / join by this plan's events)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

// Two-lane launch groups (b4d_kernels.hip).  A multi-pass transform whose intermediate is written and read once runs faster in
// groups whose workspace is about 64 MiB -- it stays in the 256-MB memory-side cache -- and with the groups dealt alternately to
// the caller's stream and a second one, so that the passes of one group run under those of the other (no drained chip between
// two launches, transform phases of one kernel under the memory phases of another).  Measured, frames/s (tools/dev_fft2d_chunk.py,
// tools/dev_pipe_chunk.py): fft2d 2048^2 61.7 k in 64-frame groups, 61.8 k in 4-frame groups, 66.5-67.7 k in 4-frame groups on two
// lanes; 1024^2 268 -> 277 k; 2160 x 2560 28.0 -> 31.4 k; psd + autocorr 2160 x 2560 20.0 -> 22.4 k, 1080 x 1920 55 -> 62.8 k,
// 228^2 1.72 -> 1.91 M.  (The power-of-two psd + autocorr pipeline gains 1 % at 2048^2 and keeps its single large groups.)
struct Lanes {
    b4d_plan* pl = nullptr;
    hipStream_t st = nullptr;
    bool two = false;
    int sub = 1;   // frames per group; lane l works in slot l of the workspaces (slot stride: `sub` frames)
    // inter_bytes: intermediate bytes per frame; two lanes need 2 * sub <= chunk and groups of at least min_group_bytes of input
    // (below that the launches themselves are what a group costs: 256^2 frames in two lanes of 64, 2.3 against 2.9 M frames/s)
    int open(b4d_plan* p, hipStream_t s, int batch, size_t inter_bytes, size_t frame_bytes, bool allow_two, size_t min_group_bytes);
    int fork(b4d_plan* p, hipStream_t s, bool want_two);   // the stream part alone (group sizes are the caller's)
    hipStream_t stream(int g) const { return two && (g & 1) ? pl->aux : st; }
    int slot(int g) const { return two ? (g & 1) : 0; }
    int close();   // joins the second lane into the caller's stream
};

// b4d_general.hip
int make_dft_matrix(int n, float2** out);
int general_psd_autocorr(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale, float* autocorr,
                         unsigned flags, hipStream_t st);
int general_fft2d(b4d_plan* pl, const float* frames, int batch, float2* out, hipStream_t st);
int general_xcorr(b4d_plan* pl, const float* a, const float* b, int batch, float* corr, unsigned flags, hipStream_t st);
int general_fft2d_c2c(b4d_plan* pl, const float2* in, int batch, int inverse, float2* out, hipStream_t st);
int general_dft2(const b4d_plan* pl, const void* X, bool x_real, int batch, int conj, float2* tmp, float2* F, hipStream_t st);
// b4d_wiener.hip: general-length 1-D engine (n = P * A * B, in-LDS mixed radix) and batched complex transpose
namespace b4d {
bool pm_fusable(int n);
bool pm_supported(int n);   // fused split, or Bluestein over a power-of-two fused transform (n <= 4096)
int pm_rows(const void* in, bool real_in, float2* out, int S, int n, const float2* tw, bool inverse, float scale, hipStream_t st);
int transpose_batch(const float2* in, float2* out, int rows, int cols, int batch, hipStream_t st);
int pm_rows_pair_fwd(const float* in, float2* half_out, int frames, int rows, int n, const float2* tw, hipStream_t st);
int pm_rows_pair_inv(const float2* half_in, float* real_out, int frames, int rows, int n, const float2* tw, float scale, const float* peak,
                     hipStream_t st);
}  // namespace b4d
// b4d_track.hip: x[b] /= max|x[b]| for `batch` maps of n floats; scratch holds >= 256 * batch floats
int normalise_by_absmax(float* x, size_t n, int batch, float* scratch, hipStream_t st);

static inline bool pow2_ok(int n) { return n >= 64 && n <= 4096 && (n & (n - 1)) == 0; }
template <int NY>
static inline int col_ct_of() { return ColCfg<NY>::CT; }
static inline int col_ct(int ny) {
    switch (ny) {
        case 64: return col_ct_of<64>();
        case 128: return col_ct_of<128>();
        case 256: return col_ct_of<256>();
        case 512: return col_ct_of<512>();
        case 1024: return col_ct_of<1024>();
        case 2048: return col_ct_of<2048>();
        case 4096: return col_ct_of<4096>();
    }
    return 16;
}
static inline bool general_ok(int n) { return n >= 2 && n <= 512; }

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

#define B4D_PLAN_LOCK(pl) std::lock_guard<std::recursive_mutex> b4d_plan_lk__((pl)->mu)
#define B4D_SIZE_SWITCH(n, CALL)        \
    switch (n) {                        \
        case 64: return CALL(64);       \
        case 128: return CALL(128);     \
        case 256: return CALL(256);     \
        case 512: return CALL(512);     \
        case 1024: return CALL(1024);   \
        case 2048: return CALL(2048);   \
        case 4096: return CALL(4096);   \
    }

#if B4D_UNIT_PASSES & B4D_PASS_COL
#ifndef B4D_COL_SPLIT
#define B4D_COL_SPLIT 1
#endif
template <int NY, int MODE>
static int launch_col(const ColArgs& a, int ntiles, int batch, hipStream_t st) {
    constexpr int SPLIT = (NY == 2048 && MODE == COL_PSD_AC) ? B4D_COL_SPLIT : 1;
    using Cfg = ColCfg<NY, SPLIT>;
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_col<NY, MODE, SPLIT, B4D_UNIT_TAG>), Cfg::LDS_BYTES);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL((k_col<NY, MODE, SPLIT, B4D_UNIT_TAG>), dim3(ntiles * SPLIT, batch), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
template <int MODE>
static int dispatch_col(const b4d_plan* pl, const ColArgs& a, int batch, hipStream_t st) {
    const int ntiles = (pl->nx / 2) / pl->ct_w;
#define B4D_CALL(N) launch_col<N, MODE>(a, ntiles, batch, st)
    B4D_SIZE_SWITCH(pl->ny, B4D_CALL)
#undef B4D_CALL
    return fail(B4D_ESIZE, "unsupported ny");
}
#endif  // B4D_PASS_COL

#if B4D_UNIT_PASSES & B4D_PASS_NYQ
template <int NY, int MODE>
static int launch_nyq(const NyqArgs& a, hipStream_t st) {
    constexpr int SEQ = row_seq(NY);
    hipLaunchKernelGGL((k_nyq<NY, SEQ, MODE>), dim3((a.items + SEQ - 1) / SEQ), dim3((NY / E16) * SEQ), 0, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
template <int MODE>
static int dispatch_nyq(const b4d_plan* pl, NyqArgs a, int items, hipStream_t st) {
    a.items = items;
    a.tw = pl->tw_y;
    a.nx = pl->nx;
#define B4D_CALL(N) launch_nyq<N, MODE>(a, st)
    B4D_SIZE_SWITCH(pl->ny, B4D_CALL)
#undef B4D_CALL
    return fail(B4D_ESIZE, "unsupported ny");
}
#endif  // B4D_PASS_NYQ

#if B4D_UNIT_PASSES & B4D_PASS_R2C
template <int NX>
static int launch_r2c(const b4d_plan* pl, const float* in, float2* spec, float* nyq_rows, const RowSrc* srcs, int batch,
                      hipStream_t st) {
    constexpr int SEQ = row_seq(NX);
    constexpr int ITER = NX >= 1024 ? B4D_K1_ITER : 1;   // groups of row pairs per workgroup, full frames only
    const dim3 block((NX / E16) * SEQ);
    if (srcs)
        hipLaunchKernelGGL((k_row_r2c<NX, SEQ, true>), dim3((pl->ny / 2 + SEQ - 1) / SEQ, batch), block, 0, st, in, spec, nyq_rows,
                           pl->tw_x, pl->ny, pl->ct_w, srcs);
    else
        hipLaunchKernelGGL((k_row_r2c<NX, SEQ, false, ITER>), dim3((pl->ny / 2 + SEQ * ITER - 1) / (SEQ * ITER), batch), block, 0, st,
                           in, spec, nyq_rows, pl->tw_x, pl->ny, pl->ct_w, srcs);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
// rows -> half spectra into `spec` / `nyq_rows` (default: the plan's chunk workspace); srcs != null selects
// ROI / z-score sources
static int dispatch_r2c(const b4d_plan* pl, const float* in, int batch, hipStream_t st, float2* spec = nullptr,
                        float* nyq_rows = nullptr, const RowSrc* srcs = nullptr) {
    if (!spec) spec = pl->spec;
    if (!nyq_rows) nyq_rows = pl->nyq_rows;
#define B4D_CALL(N) launch_r2c<N>(pl, in, spec, nyq_rows, srcs, batch, st)
    B4D_SIZE_SWITCH(pl->nx, B4D_CALL)
#undef B4D_CALL
    return fail(B4D_ESIZE, "unsupported nx");
}
#endif  // B4D_PASS_R2C

#if B4D_UNIT_PASSES & B4D_PASS_C2R
// mode: C2R_OUT (with a C2R_PEAK pre-pass when NORM_PEAK) or C2R_MAG.  ev: optional event sink (timed runs).
template <int NX>
static int launch_c2r(const b4d_plan* pl, const RowOutArgs& a, int batch, int mode, hipStream_t st,
                      std::vector<hipEvent_t>* ev, int* nblk) {
    constexpr int SEQ = row_seq(NX);
    const int npairs = (a.half && mode == C2R_OUT) ? pl->ny / 4 + 1 : pl->ny / 2;
    const dim3 grid((npairs + SEQ - 1) / SEQ, batch), block((NX / E16) * SEQ);
    if (nblk) *nblk = grid.x;
    if (batch < 1) return B4D_OK;
    if (mode == C2R_MAG) {
        // with a.gate (full maps for the few frames that need them) the workgroups of every other frame leave at once; a loop
        // over row blocks inside the kernel instead of the full grid was tried: the compiler hoists the transform's twiddle
        // and address set out of it (92 -> 236 VGPRs, half the occupancy of the main pass that shares the instantiation)
        hipLaunchKernelGGL((k_row_c2r<NX, SEQ, C2R_MAG, B4D_UNIT_TAG>), grid, block, 0, st, a);
        B4D_HIP(hipGetLastError());
        return B4D_OK;
    }
    if (mode == C2R_ROWS) {   // three row pairs around each frame's peak (a.nblk partials per frame)
        hipLaunchKernelGGL((k_row_c2r<NX, SEQ, C2R_ROWS, B4D_UNIT_TAG>), dim3((3 + SEQ - 1) / SEQ, batch), block, 0, st, a);
        B4D_HIP(hipGetLastError());
        return B4D_OK;
    }
    if (a.flags & B4D_NORM_PEAK) {
        hipLaunchKernelGGL((k_row_c2r<NX, SEQ, C2R_PEAK, B4D_UNIT_TAG>), dim3(1, batch), block, 0, st, a);
        B4D_HIP(hipGetLastError());
        if (ev) {
            hipEvent_t e;
            B4D_HIP(hipEventCreate(&e));
            ev->push_back(e);
            B4D_HIP(hipEventRecord(e, st));
        }
    }
    hipLaunchKernelGGL((k_row_c2r<NX, SEQ, C2R_OUT, B4D_UNIT_TAG>), grid, block, 0, st, a);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}
static int dispatch_c2r(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st, int mode = C2R_OUT,
                        std::vector<hipEvent_t>* ev = nullptr, int* nblk = nullptr) {
#define B4D_CALL(N) launch_c2r<N>(pl, a, batch, mode, st, ev, nblk)
    B4D_SIZE_SWITCH(pl->nx, B4D_CALL)
#undef B4D_CALL
    return fail(B4D_ESIZE, "unsupported nx");
}
#endif  // B4D_PASS_C2R

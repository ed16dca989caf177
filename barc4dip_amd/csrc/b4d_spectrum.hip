// b4d_spectrum.hip -- fft2d of real frames at power-of-two sizes (reference signal/fft.py:198-237:
// fftshift(fft2(image))), "columns first, rows last".
//
// The full shifted spectrum is twice the half spectrum: whichever pass runs LAST writes 8 of the 20 bytes per pixel this
// function moves.  A column pass cannot write them well: a 16-column tile owns a 128-byte piece of the direct half and a piece of
// the conjugate mirror that is shifted by one element against the lines (the mirror of an aligned run [a, a + 15] is
// [-a - 15, -a]) -- 8-byte mirror stores, three sectors per row (the round-2 route, k_col<COL_SPECTRUM>: 3.6 TB/s).  A row pass
// owns whole output rows: the direct row of ky and the conjugate, reversed row of -ky are both 16-KB runs.  So:
//
//   A  k_col_r2c   a tile of 32 REAL columns x ny rows lives in the registers of one workgroup (two complex columns per lane:
//                  z = x[:, c] + i x[:, c + 1]); forward FFT along y; Hermitian split -> F_y[ky][x] for ky = 0 .. ny/2 - 1, written
//                  row-major as `half` (ny/2, nx) complex -- whole 128-byte lines per row and instruction on both sides (the
//                  columns of a tile are stored in a fixed permuted order for that: half_pos).  Row 0 carries the two real rows
//                  ky = 0 and ky = ny/2 packed as re + i im.
//   B  k_row_full  one complex row of `half` per nx/16 lanes: FFT along x, then the row is written twice -- out[ky + ny/2] direct,
//                  out[ny/2 - ky] conjugated and reversed through the (free) exchange buffer.  Row 0 is un-packed by the Hermitian
//                  symmetry along kx into the two self-mirrored rows ky = 0 and ky = ny/2.
//
// Bytes per frame: A 4 ny nx in + 4 ny nx out, B 4 ny nx in + 8 ny nx out = 20 B per pixel (model of SURVEY.md §8(d): 12).
#define B4D_UNIT_PASSES 0   // none of the dispatchers of b4d_fft2d.hpp: this unit has its own kernels
#include "b4d_fft2d.hpp"

namespace b4d {

// ------------------------------------------------------------------------------------ A
// Geometry of the column pass: CPT lanes across a tile of RC = 4 CPT real columns (one 16-byte load per lane and row), T = NY / 16
// lanes along y.  A CU streams at its share of the HBM rate only while it has requests in flight, and a workgroup has none
// while it transforms: the tile is sized so that TWO workgroups fit a CU (LDS: NY x CPT x 8 B each) and one streams while the
// other computes -- CPT = 4 (64-byte row pieces) from 2048 rows on, CPT = 8 (128-byte pieces) below.
template <int NY>
struct ColR2cCfg {
    static constexpr int CPT = NY == 4096 ? 4 : 8;
    static constexpr int THREADS = CPT * (NY / E16);
    static constexpr int WAVES_PER_EU = THREADS >= 256 ? 4 : 1;
    using G = ColGeom<NY, CPT>;
    static constexpr size_t LDS_BYTES = sizeof(float2) * (size_t)G::LDS_ELEMS * CPT;
    static constexpr int BLK = 4 * CPT;   // columns per tile = period of the column order inside a row of `half`
};
constexpr int col_r2c_blk(int ny) { return ny == 4096 ? 16 : 32; }

// Column order of `half`: lane cp of a tile loads the real columns 4 cp .. 4 cp + 3 and transforms them as two complex columns
// (set s = columns 4 cp + 2 s + {0, 1}); after the split it owns the 32 contiguous output bytes of those four columns.  A 16-byte
// store instruction would write every other 16 bytes of them: instead set s of all lanes goes to the s-th half of the tile's
// run -- element 2 CPT s + 2 cp + e of a block holds column 4 cp + 2 s + e -- so that every store instruction writes 16 CPT
// contiguous bytes per row.  The row pass reads the columns back in natural order through the inverse map (half_pos).
template <int BLK>
__device__ __forceinline__ int half_pos(int x) {
    const int xl = x % BLK;
    return x - xl + (BLK / 2) * ((xl >> 1) & 1) + 2 * (xl >> 2) + (xl & 1);
}

// grid (nx / BLK, batch); block ColR2cCfg<NY>::THREADS.
template <int NY>
__global__ void __launch_bounds__((ColR2cCfg<NY>::THREADS), (ColR2cCfg<NY>::WAVES_PER_EU))
k_col_r2c(const float* __restrict__ frames, float2* __restrict__ half, const float2* __restrict__ tw, int nx) {
    using Cfg = ColR2cCfg<NY>;
    using G = typename Cfg::G;
    constexpr int T = G::T, E = E16, CPT = Cfg::CPT, RC = 4 * CPT, HALF = NY / 2;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int cp = threadIdx.x % CPT, u = threadIdx.x / CPT;
    const int ct = blockIdx.x;
    const size_t frame = blockIdx.y;
    const float* src = frames + (frame * NY) * (size_t)nx + RC * ct;
    const unsigned loff = ((unsigned)u * (unsigned)nx + 4u * cp) * 4u;
    float2 v[2][E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const f32x4 q = *at_bytes<f32x4>(sgpr_base(src + (size_t)(T * j) * nx), loff);
        v[0][j] = make_float2(q.x, q.y);
        v[1][j] = make_float2(q.z, q.w);
    }
    Fft3<G, 1>::template run_sets<2, true, true, 1>(v, u, cp, lds, tw);
    // v[s][j] = Z_s[ky = u + T j].  Columns a, b of z = a + i b:  A[ky] = (Z[ky] + conj Z[-ky]) / 2,  B[ky] = (Z[ky] - conj Z[-ky]) / 2i.
    // Only ky < ny/2 is kept (j < 8): the partners -ky live in the upper half (j >= 8), which goes through the exchange buffer
    // -- both sets at once (2 x ny/2 x CPT values fit the region of one transform set).
    int tid2 = threadIdx.x;   // lane position derived again: the forward pass's address set need not stay live (cf. k_col)
    asm volatile("" : "+v"(tid2));
    const int cp2 = tid2 % CPT, u2 = tid2 / CPT;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = E / 2; j < E; ++j) lds[(s * HALF + u2 + T * (j - E / 2)) * CPT + cp2] = v[s][j];
    __syncthreads();
    float2* dst = half + (frame * HALF) * (size_t)nx + RC * ct;
    const unsigned soff = ((unsigned)u2 * (unsigned)nx + 2u * cp2) * 8u;
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
        const int ky = u2 + T * j;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float2 z = v[s][j], zr = lds[(s * HALF + ((HALF - ky) & (HALF - 1))) * CPT + cp2];
            float2 a = make_float2(0.5f * (z.x + zr.x), 0.5f * (z.y - zr.y));
            float2 b = make_float2(0.5f * (z.y + zr.y), 0.5f * (zr.x - z.x));
            if (j == 0 && ky == 0) {   // rows ky = 0 and ky = ny/2 are real: packed as one complex row (u = 0 holds both: j = 0, 8)
                a = make_float2(z.x, v[s][E / 2].x);
                b = make_float2(z.y, v[s][E / 2].y);
            }
            *(f32x4 B4D_GLOBAL*)at_bytes<f32x2>(sgpr_base(dst + (size_t)(T * j) * nx + 2 * CPT * s), soff) = f32x4{a.x, a.y, b.x, b.y};
        }
    }
}

// ------------------------------------------------------------------------------------ B
// grid (ny / 2 / SEQ, batch); block (NX / 16) * SEQ.  BLK = column-order period of `half` (see half_pos).
template <int NX, int SEQ, int BLK, bool NTL = true>
__global__ void __launch_bounds__((NX / E16) * SEQ)
k_row_full(const float2* __restrict__ half, float2* __restrict__ out, const float2* __restrict__ tw, int ny) {
    using G = RowGeom<NX>;
    constexpr int T = G::T, E = E16;
    constexpr bool WV = T <= 64;   // one transform per wavefront (or two): no workgroup barriers
    __shared__ float2 lds_all[SEQ * G::LDS_ELEMS];
    const int seq = threadIdx.x / T, u = threadIdx.x % T;
    const size_t frame = blockIdx.y;
    const int r = blockIdx.x * SEQ + seq, hy = ny / 2;
    float2* lds = lds_all + seq * G::LDS_ELEMS;
    const float2* src = half + (frame * hy + r) * (size_t)NX;
    float2 v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const f32x2* qp = reinterpret_cast<const f32x2*>(src + half_pos<BLK>(u + T * j));
        const f32x2 q = NTL ? __builtin_nontemporal_load(qp) : *qp;
        v[j] = make_float2(q.x, q.y);
    }
    Fft3<G, 1, WV>::run(v, v, u, 0, lds, tw);
    // v[j] = Z[kx = u + T j].  The second output row needs Z[-kx]: reversed through the exchange buffer.
    fft_sync<WV>();
#pragma unroll
    for (int j = 0; j < E; ++j) lds[u + T * j] = v[j];
    fft_sync<WV>();
    // general row r = ky: out[ky + ny/2][c] = Z[kx], out[ny/2 - ky][c] = conj Z[-kx] at c = (kx + nx/2) % nx
    // row 0 (packed real rows ky = 0 and ny/2): out[ny/2][c] = (Z[kx] + conj Z[-kx]) / 2, out[0][c] = (Z[kx] - conj Z[-kx]) / 2i
    float2* oa = out + (frame * ny + (r == 0 ? hy : r + hy)) * (size_t)NX;
    float2* ob = out + (frame * ny + (r == 0 ? 0 : hy - r)) * (size_t)NX;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int kx = u + T * j, c = (kx + NX / 2) & (NX - 1);
        const float2 z = v[j], zm = lds[(NX - kx) & (NX - 1)];
        float2 a = z, b = make_float2(zm.x, -zm.y);
        if (r == 0) {
            a = make_float2(0.5f * (z.x + zm.x), 0.5f * (z.y - zm.y));
            b = make_float2(0.5f * (z.y + zm.y), 0.5f * (zm.x - z.x));
        }
        __builtin_nontemporal_store(f32x2{a.x, a.y}, reinterpret_cast<f32x2*>(oa + c));
        __builtin_nontemporal_store(f32x2{b.x, b.y}, reinterpret_cast<f32x2*>(ob + c));
    }
}

template <int NY>
static int launch_col_r2c(const float* frames, float2* half, const float2* tw, int nx, int batch, hipStream_t st) {
    using Cfg = ColR2cCfg<NY>;
    static_assert(Cfg::BLK == col_r2c_blk(NY), "the row pass reads the column order this pass writes");
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_col_r2c<NY>), Cfg::LDS_BYTES);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL((k_col_r2c<NY>), dim3(nx / Cfg::BLK, batch), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, frames, half, tw, nx);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

template <int NX>
static int launch_row_full(const float2* half, float2* out, const float2* tw, int ny, int batch, hipStream_t st) {
    constexpr int SEQ = row_seq(NX);
    const dim3 grid(ny / 2 / SEQ, batch), block((NX / E16) * SEQ);
    // measured and not kept (tools/dev_fft2d_var.py, interleaved): cached instead of streaming loads -2 %; 16-byte stores with
    // both rows re-read from the exchange buffer +-0 (the pass runs at 5.5 TB/s either way)
    // the half spectrum is read from the memory-side cache (b4d_fft2d's launch groups): cached loads there, +1.5 % at 1024^2 and
    // 2048^2; streaming loads stay for the 4096-row layout, whose single-frame groups leave the cache between the passes
    if (col_r2c_blk(ny) == 16)
        hipLaunchKernelGGL((k_row_full<NX, SEQ, 16, true>), grid, block, 0, st, half, out, tw, ny);
    else
        hipLaunchKernelGGL((k_row_full<NX, SEQ, 32, false>), grid, block, 0, st, half, out, tw, ny);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

int spectrum_rows_last(const b4d_plan* pl, float2* spec, const float* frames, int batch, float2* out, hipStream_t st) {
    int rc = B4D_ESIZE;
#define B4D_CALL(N) launch_col_r2c<N>(frames, spec, pl->tw_y, pl->nx, batch, st)
    switch (pl->ny) {
        case 64: rc = B4D_CALL(64); break;
        case 128: rc = B4D_CALL(128); break;
        case 256: rc = B4D_CALL(256); break;
        case 512: rc = B4D_CALL(512); break;
        case 1024: rc = B4D_CALL(1024); break;
        case 2048: rc = B4D_CALL(2048); break;
        case 4096: rc = B4D_CALL(4096); break;
    }
#undef B4D_CALL
    if (rc != B4D_OK) return rc == B4D_ESIZE ? fail(B4D_ESIZE, "unsupported ny") : rc;
#define B4D_CALL(N) launch_row_full<N>(spec, out, pl->tw_x, pl->ny, batch, st)
    B4D_SIZE_SWITCH(pl->nx, B4D_CALL)
#undef B4D_CALL
    return fail(B4D_ESIZE, "unsupported nx");
}

}  // namespace b4d

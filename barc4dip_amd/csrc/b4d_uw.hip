// b4d_uw.hip -- one Gibbs sweep of the unsupervised Wiener-Hunt sampler behind deconvolve_psf(method="uw")
// (reference: preprocessing/filters.py:278-286 -> skimage.restoration.unsupervised_wiener; Orieux, Giovannelli, Rodet,
// JOSA A 27(7), 2010, Eqs. 27-31).  Everything of a sweep that touches the image-sized arrays is ONE element-wise pass over the
// unitary half-plane spectrum (ny x nxh complex, nxh = nx/2 + 1, rfft2 layout):
//     precision = gn |H|^2 + gx |L|^2
//     x         = gn conj(H) / precision * Y + sqrt(0.5 / precision) (r1 + i r2)          (the sample, Eq. 27 / 30)
//     q1       += w |Y - x H|^2,  q2 += w |x|^2 |L|^2                                      (the two quadratic norms of Eq. 31)
//     post     += x  (after the burn-in);  d1 += |post/(n) - post_old/(n-1)|,  d2 += |post|  (the stopping rule)
// with the library's half-plane weights w (every column twice except column 0; only for a non-square array -- its own
// convention, kept).  The two Gamma draws per sweep and the loop control stay on the host (barc4dip_amd/preprocessing/filters.py).
// r1, r2: standard normals, either supplied (device pointers: a host stream uploaded by the caller, as the parity tests do) or
// generated here (Philox-4x32-10 keyed by `seed`, counter = element, sweep; Box-Muller).
#include "b4d_common.hpp"

namespace b4d {

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}
// two standard normals for (element, sweep)
__device__ inline float2 philox_normal2(unsigned long long seed, unsigned long long elem, unsigned sweep) {
    unsigned c[4] = {(unsigned)elem, (unsigned)(elem >> 32), sweep, 0x75770001u};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)c[0] + 0.5f) * 2.3283064365386963e-10f, u2 = ((float)c[1] + 0.5f) * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.0f * logf(fmaxf(u1, 1e-37f)));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    return make_float2(r * cs, r * sn);
}

struct UwArgs {
    const float2* y;      // unitary half-plane spectrum of the (normalised, padded) frame
    const float2* h;      // transfer function of the PSF (ir2tf, NOT unitary)
    const float* areg2;   // |L|^2 of the regulariser
    float2* x_out;        // optional: the sample of this sweep
    float2* post;         // running sum of the samples after the burn-in (in / out)
    const float* r1;      // optional supplied normals
    const float* r2;
    unsigned long long seed;
    int sweep, burnin;
    float gn, gx;
    int ny, nxh, herm;
    double* part;         // (gridDim.x, 4) partial sums
};

__global__ void __launch_bounds__(256) k_uw_step(UwArgs p) {
    const size_t n = (size_t)p.ny * p.nxh;
    double q1 = 0.0, q2 = 0.0, d1 = 0.0, d2 = 0.0;
    const bool acc = p.sweep > p.burnin, dl = p.sweep > p.burnin + 1;
    const float cn = acc ? 1.0f / (float)(p.sweep - p.burnin) : 0.f, cp = dl ? 1.0f / (float)(p.sweep - p.burnin - 1) : 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float2 y = p.y[e], h = p.h[e];
        const float a2 = p.areg2[e], t2 = h.x * h.x + h.y * h.y;
        const float prec = p.gn * t2 + p.gx * a2;
        float2 r;
        if (p.r1)
            r = make_float2(p.r1[e], p.r2[e]);
        else
            r = philox_normal2(p.seed, e, (unsigned)p.sweep);
        const float sd = sqrtf(0.5f / prec), g = p.gn / prec;
        // x = g conj(h) y + sd r
        const float2 x = make_float2(g * (h.x * y.x + h.y * y.y) + sd * r.x, g * (h.x * y.y - h.y * y.x) + sd * r.y);
        const float2 res = make_float2(y.x - (x.x * h.x - x.y * h.y), y.y - (x.x * h.y + x.y * h.x));
        const float w = (p.herm && (e % p.nxh) != 0) ? 2.0f : 1.0f;
        q1 += (double)(w * (res.x * res.x + res.y * res.y));
        q2 += (double)(w * (x.x * x.x + x.y * x.y) * a2);
        if (p.x_out) p.x_out[e] = x;
        if (acc) {
            const float2 po = p.post[e], pn = make_float2(po.x + x.x, po.y + x.y);
            p.post[e] = pn;
            if (dl) {
                const float dx = pn.x * cn - po.x * cp, dy = pn.y * cn - po.y * cp;
                d1 += (double)sqrtf(dx * dx + dy * dy);
                d2 += (double)sqrtf(pn.x * pn.x + pn.y * pn.y);
            }
        }
    }
    __shared__ double sh[4][4];
    double v[4] = {q1, q2, d1, d2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) p.part[(size_t)blockIdx.x * 4 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// fixed-order sum of the block partials: the same call gives the same sums bit for bit
__global__ void __launch_bounds__(64) k_uw_sum(const double* __restrict__ part, int nblk, double* __restrict__ out4) {
    const int k = threadIdx.x & 3, lane = threadIdx.x >> 2;   // 16 lanes per quantity
    double s = 0.0;
    for (int b = lane; b < nblk; b += 16) s += part[(size_t)b * 4 + k];
    __shared__ double sh[64];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int l = 0; l < 16; ++l) t += sh[l * 4 + threadIdx.x];
        out4[threadIdx.x] = t;
    }
}

}  // namespace b4d

using namespace b4d;

extern "C" int b4d_uw_step(const void* y, const void* tf, const float* areg2, void* x_sample, void* postmean, const float* r1,
                           const float* r2, unsigned long long seed, int sweep, int burnin, float gn, float gx, int ny, int nxh,
                           double* sums4, void* stream) {
    if (!y || !tf || !areg2 || !postmean || !sums4) return fail(B4D_EINVAL, "null argument");
    if (ny < 1 || nxh < 1 || sweep < 0 || burnin < 0) return fail(B4D_EINVAL, "bad sizes / sweep / burn-in");
    if ((r1 == nullptr) != (r2 == nullptr)) return fail(B4D_EINVAL, "supply both normal arrays or neither");
    if (!(gn > 0.f) || !(gx > 0.f)) return fail(B4D_EINVAL, "the precisions gn, gx must be > 0");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)ny * nxh;
    const int nblk = (int)std::min<size_t>((n + 255) / 256, 2048);
    B4D_SCRATCH_LOCK();
    void* scratch = nullptr;
    int rc = get_scratch(sizeof(double) * 4 * (size_t)nblk, &scratch, st);
    if (rc) return rc;
    UwArgs a{};
    a.y = static_cast<const float2*>(y);
    a.h = static_cast<const float2*>(tf);
    a.areg2 = areg2;
    a.x_out = static_cast<float2*>(x_sample);
    a.post = static_cast<float2*>(postmean);
    a.r1 = r1;
    a.r2 = r2;
    a.seed = seed;
    a.sweep = sweep;
    a.burnin = burnin;
    a.gn = gn;
    a.gx = gx;
    a.ny = ny;
    a.nxh = nxh;
    a.herm = nxh != ny ? 1 : 0;
    a.part = static_cast<double*>(scratch);
    hipLaunchKernelGGL(k_uw_step, dim3(nblk), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_uw_sum, dim3(1), dim3(64), 0, st, (const double*)a.part, nblk, sums4);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

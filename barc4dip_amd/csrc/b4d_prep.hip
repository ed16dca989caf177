// b4d_prep.hip -- flat-field (gain) correction, preprocessing/normalize.py:12-145 of the reference.
// Elementwise float32 arithmetic in the reference's order (one rounding per operation, no contraction), so
// the default path is bit-exact against NumPy: out = ((I - D) / (F - D)) * s, zero where F - D <= eps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/b4d.h"
#include "b4d_common.hpp"

namespace b4d {

// mean over axis 0 of a (T, npix) float32 stack the way NumPy reduces it: float32 adds in frame order, one division
__global__ void __launch_bounds__(256) k_stack_mean_f32(const float* __restrict__ stack, int T, size_t npix, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float acc = stack[i];
    for (int t = 1; t < T; ++t) acc = __fadd_rn(acc, stack[(size_t)t * npix + i]);
    out[i] = __fdiv_rn(acc, (float)T);
}

// den = F - D (D = 0 when null); mask_bad: pixels with den <= eps become NaN (the selection kernels skip NaN)
__global__ void __launch_bounds__(256) k_flat_den(const float* __restrict__ flat, const float* __restrict__ dark, size_t npix, float eps,
                                                  int mask_bad, float* __restrict__ den) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float d = __fsub_rn(flat[i], dark ? dark[i] : 0.f);
    den[i] = (mask_bad && d <= eps) ? __builtin_nanf("") : d;
}

// grid (ceil(npix/256), batch)
__global__ void __launch_bounds__(256) k_flat_field(const float* __restrict__ img, const float* __restrict__ flat,
                                                    const float* __restrict__ dark, size_t npix, float eps, float scale, int apply_scale,
                                                    float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t o = (size_t)blockIdx.y * npix + i;
    const float d = dark ? dark[i] : 0.f;
    const float num = dark ? __fsub_rn(img[o], d) : img[o];
    if (!flat) {  // darks only: I - D
        out[o] = num;
        return;
    }
    const float den = __fsub_rn(flat[i], d);
    if (den <= eps) {
        out[o] = 0.f;
        return;
    }
    float v = __fdiv_rn(num, den);
    if (apply_scale) v = __fmul_rn(v, scale);
    out[o] = v;
}

// 16-byte variant: a lane keeps the flat / dark values of its four pixels in registers and walks the frames of a slice of the
// batch (frame words are read and written once: streaming loads / stores).  Same per-element arithmetic as k_flat_field.
// grid (ceil(npix / 4 / 256), slices)
__global__ void __launch_bounds__(256) k_flat_field4(const float* __restrict__ img, const float* __restrict__ flat,
                                                     const float* __restrict__ dark, size_t npix, int batch, float eps, float scale,
                                                     int apply_scale, float* __restrict__ out) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= npix) return;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const v4f d = dark ? *reinterpret_cast<const v4f*>(dark + i) : zero;
    v4f den = zero;
    if (flat) {
        const v4f fl = *reinterpret_cast<const v4f*>(flat + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) den[k] = __fsub_rn(fl[k], d[k]);
    }
    const int per = (batch + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(batch, b0 + per);
    for (int b = b0; b < b1; ++b) {
        const size_t o = (size_t)b * npix + i;
        const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(img + o));
        v4f r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float num = dark ? __fsub_rn(x[k], d[k]) : x[k];
            float v = num;
            if (flat) {
                v = 0.f;
                if (!(den[k] <= eps)) {
                    v = __fdiv_rn(num, den[k]);
                    if (apply_scale) v = __fmul_rn(v, scale);
                }
            }
            r[k] = v;
        }
        __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(out + o));
    }
}

__device__ __forceinline__ void cswap_minmax(float& a, float& b) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}

// 3x3 median (scipy.ndimage.median_filter, mode="reflect": d c b a | a b c d | d c b a) of `frames` at the listed
// pixels, gathered into rep (batch, nbad).  grid (ceil(nbad/256), batch)
__global__ void __launch_bounds__(256) k_bad_median(const float* __restrict__ frames, int ny, int nx, const long long* __restrict__ idx,
                                                    int nbad, float* __restrict__ rep) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbad) return;
    const float* f = frames + (size_t)blockIdx.y * ny * nx;
    const int y = (int)(idx[k] / nx), x = (int)(idx[k] % nx);
    float v[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int yy = y + dy, xx = x + dx;
            yy = yy < 0 ? 0 : (yy >= ny ? ny - 1 : yy);   // half-sample symmetric reflection of a 1-pixel overhang
            xx = xx < 0 ? 0 : (xx >= nx ? nx - 1 : xx);
            v[(dy + 1) * 3 + dx + 1] = f[(size_t)yy * nx + xx];
        }
    // median-of-9 exchange network (19 compare-exchanges)
    cswap_minmax(v[1], v[2]); cswap_minmax(v[4], v[5]); cswap_minmax(v[7], v[8]);
    cswap_minmax(v[0], v[1]); cswap_minmax(v[3], v[4]); cswap_minmax(v[6], v[7]);
    cswap_minmax(v[1], v[2]); cswap_minmax(v[4], v[5]); cswap_minmax(v[7], v[8]);
    cswap_minmax(v[0], v[3]); cswap_minmax(v[5], v[8]); cswap_minmax(v[4], v[7]);
    cswap_minmax(v[3], v[6]); cswap_minmax(v[1], v[4]); cswap_minmax(v[2], v[5]);
    cswap_minmax(v[4], v[7]); cswap_minmax(v[4], v[2]); cswap_minmax(v[6], v[4]);
    cswap_minmax(v[4], v[2]);
    rep[(size_t)blockIdx.y * nbad + k] = v[4];
}

__global__ void __launch_bounds__(256) k_bad_scatter(float* __restrict__ frames, size_t npix, const long long* __restrict__ idx, int nbad,
                                                     const float* __restrict__ rep) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbad) return;
    frames[(size_t)blockIdx.y * npix + idx[k]] = rep[(size_t)blockIdx.y * nbad + k];
}

// raw detector words -> float32 (images.astype(np.float32), normalize.py:79 / io): one element per lane, 4 per iteration
template <typename T>
__global__ void __launch_bounds__(256) k_to_f32(const T* __restrict__ src, size_t n, float* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
// 16 source bytes per lane and iteration (16 / 8 / 4 elements for 1- / 2- / 4-byte words), streaming loads and stores: detector
// words are converted once.  src and dst 16-byte aligned; the tail (n % V) goes through k_to_f32.
template <typename T>
__global__ void __launch_bounds__(256) k_to_f32_vec(const T* __restrict__ src, size_t nvec, float* __restrict__ dst) {
    constexpr int V = 16 / sizeof(T);
    typedef T srcv __attribute__((ext_vector_type(V)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const srcv q = __builtin_nontemporal_load(reinterpret_cast<const srcv*>(src) + i);
        float* o = dst + i * V;
#pragma unroll
        for (int k = 0; k < V; k += 4)
            __builtin_nontemporal_store(v4f{(float)q[k], (float)q[k + 1], (float)q[k + 2], (float)q[k + 3]}, reinterpret_cast<v4f*>(o + k));
    }
}
template <typename T>
static void launch_to_f32(const T* src, size_t n, float* dst, hipStream_t st) {
    constexpr size_t V = 16 / sizeof(T);
    size_t done = 0;
    if constexpr (sizeof(T) <= 4) {
        if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && n >= V) {
            const size_t nvec = n / V;
            hipLaunchKernelGGL(k_to_f32_vec<T>, dim3((unsigned)std::min<size_t>((nvec + 255) / 256, 65535)), dim3(256), 0, st, src, nvec, dst);
            done = nvec * V;
        }
    }
    if (done < n)
        hipLaunchKernelGGL(k_to_f32<T>, dim3((unsigned)std::min<size_t>((n - done + 255) / 256, 65535)), dim3(256), 0, st, src + done, n - done,
                           dst + done);
}

}  // namespace b4d

using namespace b4d;

extern "C" int b4d_stack_mean_f32(const float* stack, int frames, size_t npix, float* out, void* stream) {
    if (!stack || !out || frames < 1 || npix < 1) return fail(B4D_EINVAL, "b4d_stack_mean_f32: bad argument");
    hipLaunchKernelGGL(k_stack_mean_f32, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stack, frames, npix, out);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

extern "C" int b4d_flat_den(const float* flat, const float* dark, size_t npix, float eps, int mask_bad, float* den, void* stream) {
    if (!flat || !den || npix < 1) return fail(B4D_EINVAL, "b4d_flat_den: bad argument");
    hipLaunchKernelGGL(k_flat_den, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, dark, npix, eps, mask_bad, den);
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

extern "C" int b4d_flat_field(const float* frames, int batch, size_t npix, const float* flat, const float* dark, float eps, float scale,
                              int apply_scale, float* out, void* stream) {
    if (!frames || !out || batch < 1 || npix < 1 || (!flat && !dark)) return fail(B4D_EINVAL, "b4d_flat_field: bad argument");
    const bool vec = (npix & 3) == 0 && ((reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(flat) |
                                          reinterpret_cast<uintptr_t>(dark)) & 15) == 0;
    if (vec) {
        const unsigned gx = (unsigned)((npix / 4 + 255) / 256);
        // enough workgroups to fill the chip several times over, as few batch slices as that takes (flat / dark are re-read per slice)
        const int slices = (int)std::min<size_t>((size_t)batch, std::max<size_t>(1, (size_t)8192 / std::max(1u, gx)));
        hipLaunchKernelGGL(k_flat_field4, dim3(gx, slices), dim3(256), 0, (hipStream_t)stream, frames, flat, dark, npix, batch, eps, scale,
                           apply_scale, out);
    } else {
        hipLaunchKernelGGL(k_flat_field, dim3((unsigned)((npix + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream, frames, flat, dark, npix,
                           eps, scale, apply_scale, out);
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

extern "C" int b4d_repair_pixels(float* frames, int batch, int ny, int nx, const long long* idx, int nbad, void* stream) {
    B4D_SCRATCH_LOCK();
    if (!frames || batch < 1 || ny < 1 || nx < 1 || nbad < 0 || (nbad > 0 && !idx)) return fail(B4D_EINVAL, "b4d_repair_pixels: bad argument");
    if (nbad == 0) return B4D_OK;
    hipStream_t st = (hipStream_t)stream;
    void* ws = nullptr;
    int rc = get_scratch(sizeof(float) * (size_t)batch * nbad, &ws, (hipStream_t)stream);
    if (rc) return rc;
    float* rep = static_cast<float*>(ws);
    hipLaunchKernelGGL(k_bad_median, dim3((nbad + 255) / 256, batch), dim3(256), 0, st, frames, ny, nx, idx, nbad, rep);
    hipLaunchKernelGGL(k_bad_scatter, dim3((nbad + 255) / 256, batch), dim3(256), 0, st, frames, (size_t)ny * nx, idx, nbad, rep);
    B4D_HIP(hipGetLastError());
    B4D_HIP(hipStreamSynchronize(st));  // the shared scratch holds the repaired values until the scatter has run
    return B4D_OK;
}

extern "C" int b4d_to_f32(const void* src, int dtype, size_t n, float* dst, void* stream) {
    if (!src || !dst || n < 1) return fail(B4D_EINVAL, "b4d_to_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case 0: launch_to_f32(static_cast<const unsigned char*>(src), n, dst, st); break;
        case 1: launch_to_f32(static_cast<const unsigned short*>(src), n, dst, st); break;
        case 2: launch_to_f32(static_cast<const short*>(src), n, dst, st); break;
        case 3: launch_to_f32(static_cast<const int*>(src), n, dst, st); break;
        case 4: launch_to_f32(static_cast<const unsigned int*>(src), n, dst, st); break;
        case 5: launch_to_f32(static_cast<const float*>(src), n, dst, st); break;
        case 6: launch_to_f32(static_cast<const double*>(src), n, dst, st); break;
        default: return fail(B4D_EINVAL, "b4d_to_f32: dtype code must be 0..6 (u8, u16, i16, i32, u32, f32, f64)");
    }
    B4D_HIP(hipGetLastError());
    return B4D_OK;
}

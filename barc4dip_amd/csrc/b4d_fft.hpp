// b4d_fft.hpp -- register/LDS FFT building blocks for gfx950 (wave64, 160 KiB LDS per CU).
//
// One transform of length N is held by T = N/E lanes, E complex points per lane, in the
// "strided" layout  lane u, register j  <->  element u + T*j   (input AND output).
// Three register stages (radices R1*R2*R3 = N) are separated by two LDS exchanges:
//
//   stage 1  n = n1 + M*n2 (M = N/R1):  DFT_R1 over n2, twiddle w_N^(n1*k2)
//   stage 2  n1 = m1 + R3*m2:           DFT_R2 over m2, twiddle w_M^(m1*q2)
//   stage 3                             DFT_R3 over m1        k = k2 + R1*(q2 + R2*q1)
//
// The exchange buffers are padded (S1, S2) so that every ds_read_b64 / ds_write_b64 of a
// wave is bank-conflict free; CI "interleaved" transforms (the column kernels keep CI
// column pairs in adjacent lanes so that global accesses are whole 128-B lines) share one
// buffer with the interleave index as the fastest LDS dimension.
//
// Inverse transforms use the same code on (im, re)-swapped data: ifft(x) = swap(fft(swap(x))).
#pragma once
#include <hip/hip_runtime.h>

#include "b4d_timing_only.hpp"

namespace b4d {

// Complex arithmetic on the packed-FP32 pipe (v_pk_add/mul/fma_f32: both halves of a complex value per instruction,
// the quarter-turn and the cross terms of a product ride the op_sel / neg modifiers).  The butterflies are VALU-issue
// bound at 4 waves per SIMD (SQ_ACTIVE_INST_VALU, profiles/): this halves their instruction count.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f to_v(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 to_f(v2f a) { return make_float2(a.x, a.y); }

#ifndef B4D_NO_PK
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {  // (a.x b.x - a.y b.y, a.y b.x + a.x b.y)
    v2f t, r, x = to_v(a), y = to_v(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(y));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(t));
    return to_f(r);
}
// product with a compile-time constant (cr, ci): the compiler folds the constants into scalar register pairs
__device__ __forceinline__ float2 cmulc(float2 a, float cr, float ci) {
    v2f x = to_v(a);
    v2f t = x * v2f{cr, cr};
    return to_f(__builtin_elementwise_fma(x.yx, v2f{-ci, ci}, t));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return to_f(to_v(a) + to_v(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return to_f(to_v(a) - to_v(b)); }
// a + (-i) b  and  a - (-i) b
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) {
    v2f r, x = to_v(a), y = to_v(b);
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return to_f(r);
}
__device__ __forceinline__ float2 csub_mi(float2 a, float2 b) {
    v2f r, x = to_v(a), y = to_v(b);
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return to_f(r);
}
#else
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float cr, float ci) { return cmul(a, make_float2(cr, ci)); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ float2 csub_mi(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }
#endif
__device__ __forceinline__ float2 cswap(float2 a) { return make_float2(a.y, a.x); }
// multiply by -i (forward-direction quarter turn)
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

#define B4D_SQRT1_2 0.70710678118654752440f
#define B4D_C1_16 0.92387953251128675613f  // cos(pi/8)
#define B4D_S1_16 0.38268343236508977173f  // sin(pi/8)
#define B4D_C1_32 0.98078528040323044913f  // cos(pi/16)
#define B4D_S1_32 0.19509032201612826785f  // sin(pi/16)
#define B4D_C3_32 0.83146961230254523708f  // cos(3pi/16)
#define B4D_S3_32 0.55557023301960222474f  // sin(3pi/16)

// ---- small DFTs, natural order in and out, forward sign exp(-2 pi i nk/R) -------------
template <int R>
struct Dft;

template <>
struct Dft<1> {
    static __device__ __forceinline__ void run(float2 (&x)[1]) {}
};

template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(float2 (&x)[2]) {
        float2 a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
    }
};

template <>
struct Dft<4> {
    static __device__ __forceinline__ void run(float2 (&x)[4]) {
        float2 t0 = cadd(x[0], x[2]), t1 = csub(x[0], x[2]);
        float2 t2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
        x[0] = cadd(t0, t2);
        x[2] = csub(t0, t2);
        x[1] = cadd_mi(t1, d);
        x[3] = csub_mi(t1, d);
    }
    // same with a pending factor -i on x[2]
    static __device__ __forceinline__ void run_mi2(float2 (&x)[4]) {
        float2 t0 = cadd_mi(x[0], x[2]), t1 = csub_mi(x[0], x[2]);
        float2 t2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
        x[0] = cadd(t0, t2);
        x[2] = csub(t0, t2);
        x[1] = cadd_mi(t1, d);
        x[3] = csub_mi(t1, d);
    }
};

template <>
struct Dft<8> {
    static __device__ __forceinline__ void run(float2 (&x)[8]) {
        float2 e[4] = {x[0], x[2], x[4], x[6]};
        float2 o[4] = {x[1], x[3], x[5], x[7]};
        Dft<4>::run(e);
        Dft<4>::run(o);
        // w8^1 = (1-i)/sqrt2, w8^2 = -i, w8^3 = (-1-i)/sqrt2
        float2 o1 = cmulc(o[1], B4D_SQRT1_2, -B4D_SQRT1_2);
        float2 o3 = cmulc(o[3], -B4D_SQRT1_2, -B4D_SQRT1_2);
        x[0] = cadd(e[0], o[0]);
        x[4] = csub(e[0], o[0]);
        x[1] = cadd(e[1], o1);
        x[5] = csub(e[1], o1);
        x[2] = cadd_mi(e[2], o[2]);
        x[6] = csub_mi(e[2], o[2]);
        x[3] = cadd(e[3], o3);
        x[7] = csub(e[3], o3);
    }
};

template <>
struct Dft<16> {
    static __device__ __forceinline__ void run(float2 (&x)[16]) {
        // decimation in time, 4 x 4: s_r[m] = x[4m + r]
        float2 s[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float2 t[4] = {x[r], x[r + 4], x[r + 8], x[r + 12]};
            Dft<4>::run(t);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[r][k] = t[k];
        }
        // twiddles w16^(r*k); s[2][2] keeps its factor -i for the final butterfly
        s[1][1] = cmulc(s[1][1], B4D_C1_16, -B4D_S1_16);
        s[1][2] = cmulc(s[1][2], B4D_SQRT1_2, -B4D_SQRT1_2);
        s[1][3] = cmulc(s[1][3], B4D_S1_16, -B4D_C1_16);
        s[2][1] = cmulc(s[2][1], B4D_SQRT1_2, -B4D_SQRT1_2);
        s[2][3] = cmulc(s[2][3], -B4D_SQRT1_2, -B4D_SQRT1_2);
        s[3][1] = cmulc(s[3][1], B4D_S1_16, -B4D_C1_16);
        s[3][2] = cmulc(s[3][2], -B4D_SQRT1_2, -B4D_SQRT1_2);
        s[3][3] = cmulc(s[3][3], -B4D_C1_16, B4D_S1_16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float2 t[4] = {s[0][k], s[1][k], s[2][k], s[3][k]};
            if (k == 2)
                Dft<4>::run_mi2(t);
            else
                Dft<4>::run(t);
#pragma unroll
            for (int q = 0; q < 4; ++q) x[k + 4 * q] = t[q];
        }
    }
};

// ---- plan geometry ---------------------------------------------------------------------
template <int N_, int E_, int R1_, int R2_, int R3_, int CI_>
struct FftGeom {
    static constexpr int N = N_, E = E_, R1 = R1_, R2 = R2_, R3 = R3_, CI = CI_;
    static constexpr int T = N / E;   // lanes per transform
    static constexpr int M = N / R1;  // length of the stage-2/3 sub-transform
    static_assert(R1 * R2 * R3 == N, "radices must multiply to N");
    static_assert(E % R1 == 0 && E % R2 == 0 && E % R3 == 0, "each radix must divide E");
    // exchange strides (complex elements), chosen for conflict-free b64 access:
    //  CI == 1 (lanes run along the transform):  S1 == 2 (mod 32), S2 == 16 (mod 32)
    //  CI >= 4 (interleave index fastest):        S1 odd,          S2 unpadded
    static constexpr int S1 = (CI == 1) ? (M + 2) : (M + 1);
    static constexpr int S2 = (CI == 1) ? (R1 * R3 + 16) : (R1 * R3);
    static constexpr int X1 = R1 * S1, X2 = R2 * S2;
    static constexpr int LDS_ELEMS = (X1 > X2 ? X1 : X2) > N ? (X1 > X2 ? X1 : X2) : N;  // per transform
};

// Synchronisation of the lanes that share one exchange buffer.  WAVE: every lane of the transform sits in ONE wavefront (row kernels
// with T <= 64 lanes per transform, laid out transform-major): the LDS executes a wave's instructions in order, so a write is
// visible to the reads the same wave issues after it -- no s_barrier, the waves of a workgroup (one or two transforms each) run
// independently of each other; the fences only pin the compiler's order.
template <bool WAVE>
__device__ __forceinline__ void fft_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Three-stage forward FFT on one or two register sets that share lane geometry.
// `u` = lane position along the transform (0..T-1); `lds` = this transform group's buffer;
// LDS element address = logical_address * CI + ci.   NV = number of register sets (1 or 2):
// with NV == 2 the two sets go through the buffer one after the other.
template <class G, int NV, bool WAVE = false>
struct Fft3 {
    static constexpr int E = G::E, T = G::T, R1 = G::R1, R2 = G::R2, R3 = G::R3, CI = G::CI;
    static_assert(!WAVE || (T <= 64 && CI == 1), "wave-local synchronisation needs the whole transform in one wavefront");
    static __device__ __forceinline__ void sync() { fft_sync<WAVE>(); }

    template <int R, int CNT, int STRIDE>
    static __device__ __forceinline__ void butterflies(float2 (&v)[E]) {
        // CNT butterflies of radix R on registers i + STRIDE*n  (i = 0..CNT-1, n = 0..R-1)
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            float2 t[R];
#pragma unroll
            for (int n = 0; n < R; ++n) t[n] = v[i + STRIDE * n];
            Dft<R>::run(t);
#pragma unroll
            for (int n = 0; n < R; ++n) v[i + STRIDE * n] = t[n];
        }
    }

    // stage 1 on one register set: register j = i + B1*n2, n1 = u + T*i; then twiddle w_N^(n1*k2)
    static __device__ __forceinline__ void stage1(float2 (&v)[E], int u, const float2* __restrict__ tw) {
        constexpr int B1 = E / R1;
        butterflies<R1, B1, B1>(v);
#pragma unroll
        for (int i = 0; i < B1; ++i) {
            const int n1 = u + T * i;
#pragma unroll
            for (int k2 = 1; k2 < R1; ++k2)
#ifdef B4D_EXP_NOTW
                v[i + B1 * k2] = cmulc(v[i + B1 * k2], 0.5f, 0.25f);
#else
                v[i + B1 * k2] = cmul(v[i + B1 * k2], tw[n1 * k2]);
#endif
        }
    }
    // exchange 1: addr1(k2, n1) = k2*S1 + n1 ; stage-2 combos c = k2 + R1*m1, lane u handles c = u + T*i2
    static __device__ __forceinline__ void xchg1_write(const float2 (&v)[E], int u, int ci, float2* __restrict__ lds) {
        constexpr int B1 = E / R1;
#pragma unroll
        for (int i = 0; i < B1; ++i)
#pragma unroll
            for (int k2 = 0; k2 < R1; ++k2) lds[(k2 * G::S1 + u + T * i) * CI + ci] = v[i + B1 * k2];
    }
    static __device__ __forceinline__ void xchg1_read(float2 (&v)[E], int u, int ci, const float2* __restrict__ lds) {
        constexpr int B2 = E / R2;
#pragma unroll
        for (int i2 = 0; i2 < B2; ++i2) {
            const int c = u + T * i2, k2 = c % R1, m1 = c / R1;
#pragma unroll
            for (int m2 = 0; m2 < R2; ++m2) v[i2 + B2 * m2] = lds[(k2 * G::S1 + m1 + R3 * m2) * CI + ci];
        }
    }
    static __device__ __forceinline__ void xchg1(float2 (&v)[E], int u, int ci, float2* __restrict__ lds) {
#ifndef B4D_EXP_NOXCHG
        xchg1_write(v, u, ci, lds);
#endif
#ifndef B4D_EXP_NOBAR
        sync();
#endif
#ifndef B4D_EXP_NOXCHG
        xchg1_read(v, u, ci, lds);
#endif
    }
    // stage 2: DFT over m2, twiddle w_M^(m1*q2) = w_N^(R1*m1*q2)
    static __device__ __forceinline__ void stage2(float2 (&v)[E], int u, const float2* __restrict__ tw) {
        constexpr int B2 = E / R2;
        butterflies<R2, B2, B2>(v);
#pragma unroll
        for (int i2 = 0; i2 < B2; ++i2) {
            const int m1 = (u + T * i2) / R1;
#pragma unroll
            for (int q2 = 1; q2 < R2; ++q2)
#ifdef B4D_EXP_NOTW
                v[i2 + B2 * q2] = cmulc(v[i2 + B2 * q2], 0.5f, 0.25f);
#else
                v[i2 + B2 * q2] = cmul(v[i2 + B2 * q2], tw[R1 * m1 * q2]);
#endif
        }
    }
    // exchange 2: addr2(q2, m1, k2) = q2*S2 + c ; stage-3 butterflies A = k2 + R1*q2, lane u handles A = u + T*a
    static __device__ __forceinline__ void xchg2_write(const float2 (&v)[E], int u, int ci, float2* __restrict__ lds) {
        constexpr int B2 = E / R2;
#pragma unroll
        for (int i2 = 0; i2 < B2; ++i2)
#pragma unroll
            for (int q2 = 0; q2 < R2; ++q2) lds[(q2 * G::S2 + u + T * i2) * CI + ci] = v[i2 + B2 * q2];
    }
    static __device__ __forceinline__ void xchg2_read(float2 (&v)[E], int u, int ci, const float2* __restrict__ lds) {
        constexpr int B3 = E / R3;
#pragma unroll
        for (int a = 0; a < B3; ++a) {
            const int A = u + T * a, k2 = A % R1, q2 = A / R1;
#pragma unroll
            for (int m1 = 0; m1 < R3; ++m1) v[a + B3 * m1] = lds[(q2 * G::S2 + m1 * R1 + k2) * CI + ci];
        }
    }
    static __device__ __forceinline__ void xchg2(float2 (&v)[E], int u, int ci, float2* __restrict__ lds) {
#ifndef B4D_EXP_NOXCHG
        xchg2_write(v, u, ci, lds);
#endif
#ifndef B4D_EXP_NOBAR
        sync();
#endif
#ifndef B4D_EXP_NOXCHG
        xchg2_read(v, u, ci, lds);
#endif
    }
    // stage 3: outputs q1 -> register a + B3*q1  <->  k = u + T*(a + B3*q1)
    static __device__ __forceinline__ void stage3(float2 (&v)[E]) { butterflies<R3, E / R3, E / R3>(v); }
    // one stage-3 butterfly: registers a + (E/R3)*q1, q1 = 0..R3-1 (lets a caller interleave its epilogue)
    static __device__ __forceinline__ void stage3_one(float2 (&v)[E], int a) {
        constexpr int B3 = E / R3;
        float2 t[R3];
#pragma unroll
        for (int n = 0; n < R3; ++n) t[n] = v[a + B3 * n];
        Dft<R3>::run(t);
#pragma unroll
        for (int n = 0; n < R3; ++n) v[a + B3 * n] = t[n];
    }

    // The caller guarantees nobody still reads `lds` on entry; on exit other lanes may still be
    // reading it (sync before reuse).  With NV == 2 the sets are processed strictly one after
    // the other (sched_barrier keeps the compiler from interleaving them: the column kernels run
    // at the 128-VGPR cap of a 1024-lane workgroup).
    static __device__ __forceinline__ void run(float2 (&va)[E], float2 (&vb)[E], int u, int ci,
                                               float2* __restrict__ lds, const float2* __restrict__ tw) {
        stage1(va, u, tw);
        xchg1(va, u, ci, lds);
        if (NV == 2) {
            __builtin_amdgcn_sched_barrier(0);
            stage1(vb, u, tw);
            sync();
            xchg1(vb, u, ci, lds);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage2(va, u, tw);
        sync();
        xchg2(va, u, ci, lds);
        if (NV == 2) {
            __builtin_amdgcn_sched_barrier(0);
            stage2(vb, u, tw);
            sync();
            xchg2(vb, u, ci, lds);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage3(va);
        if (NV == 2) stage3(vb);
    }

    // NS register sets that share lane geometry go through the exchange buffer one after the other.
    // Same entry/exit contract as run().  SERIAL pins the per-set order with sched_barriers (only
    // useful when the kernel sits at its VGPR cap); otherwise the compiler may overlap set s+1's
    // butterflies with set s's LDS round trip.
    // SB > 1: SB sets go through the buffer TOGETHER (the buffer holds SB regions of SET_ELEMS complex
    // values): half the barriers and twice the independent work between them.
    template <int NS, bool SERIAL, bool WITH_STAGE3 = true, int SB = 1>
    static __device__ __forceinline__ void run_sets(float2 (&v)[NS][E], int u, int ci, float2* __restrict__ lds,
                                                    const float2* __restrict__ tw) {
        if (SB > 1) {
            static_assert(NS % SB == 0, "batch must divide the set count");
            constexpr int SET_ELEMS = G::LDS_ELEMS * CI;
#pragma unroll
            for (int b = 0; b < NS; b += SB) {
#pragma unroll
                for (int s = 0; s < SB; ++s) stage1(v[b + s], u, tw);
                if (b > 0) sync();
#pragma unroll
                for (int s = 0; s < SB; ++s) xchg1_write(v[b + s], u, ci, lds + s * SET_ELEMS);
                sync();
#pragma unroll
                for (int s = 0; s < SB; ++s) xchg1_read(v[b + s], u, ci, lds + s * SET_ELEMS);
            }
#pragma unroll
            for (int b = 0; b < NS; b += SB) {
#pragma unroll
                for (int s = 0; s < SB; ++s) stage2(v[b + s], u, tw);
                sync();
#pragma unroll
                for (int s = 0; s < SB; ++s) xchg2_write(v[b + s], u, ci, lds + s * SET_ELEMS);
                sync();
#pragma unroll
                for (int s = 0; s < SB; ++s) xchg2_read(v[b + s], u, ci, lds + s * SET_ELEMS);
            }
            if (WITH_STAGE3) {
#pragma unroll
                for (int s = 0; s < NS; ++s) stage3(v[s]);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            stage1(v[s], u, tw);
            if (s > 0) sync();
            xchg1(v[s], u, ci, lds);
            if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            stage2(v[s], u, tw);
            sync();
            xchg2(v[s], u, ci, lds);
            if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        }
        if (WITH_STAGE3) {
#pragma unroll
            for (int s = 0; s < NS; ++s) stage3(v[s]);
        }
    }
};

}  // namespace b4d

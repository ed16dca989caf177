// b4d_radix.hpp -- in-register DFTs of arbitrary small lengths for the mixed-radix engine (b4d_mixed.hpp).
//
// The power-of-two butterflies live in b4d_fft.hpp (Dft<2/4/8/16>).  This header adds
//   * DftOdd<P>      any odd length P from the conjugate-pair form
//                        a_j = x_j + x_{P-j},  b_j = x_j - x_{P-j}            (j = 1 .. (P-1)/2)
//                        X_k, X_{P-k} = (x_0 + sum_j a_j cos(2 pi jk/P))  -/+  i (sum_j b_j sin(2 pi jk/P))
//                    every product is (complex) x (real constant): ONE packed FMA, (P-1)^2 / 2 of them instead of the
//                    4 P^2 real FMAs of a dense complex DFT row sum -- 8 x fewer for P = 19;
//   * DftComp<A, B>  Cooley-Tukey A x B on registers with compile-time twiddles (27 = 3 x 9, 9 = 3 x 3, 25 = 5 x 5 ...);
//   * Radix<R>       the dispatcher used by the engine.
// All loops are unrolled over integral constants, so every trigonometric factor is a literal evaluated by the compiler
// in double precision (constexpr Taylor series after octant reduction: no libm, no tables in memory).
// Forward sign exp(-2 pi i nk / R), natural order in and out.
#pragma once
#include <utility>

#include "b4d_fft.hpp"

namespace b4d {

// ---- compile-time trigonometry of 2 pi a / b ------------------------------------------------------------------
namespace ctrig {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double tcos(double x) {   // |x| <= pi/4
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int n = 1; n <= 12; ++n) {
        term *= -x2 / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
constexpr double tsin(double x) {   // |x| <= pi/4
    const double x2 = x * x;
    double term = x, sum = x;
    for (int n = 1; n <= 12; ++n) {
        term *= -x2 / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double cos2pi(long long a, long long b);
constexpr double sin2pi(long long a, long long b) {   // sin(2 pi a / b) = cos(2 pi (a/b - 1/4)) = cos(2 pi (4a - b) / (4b))
    return cos2pi(4 * a - b, 4 * b);
}
constexpr double cos2pi(long long a, long long b) {
    a %= b;
    if (a < 0) a += b;
    if (2 * a > b) a = b - a;                                  // cos(2 pi - x) = cos x          -> [0, 1/2]
    if (4 * a > b) return -cos2pi(b - 2 * a, 2 * b);           // cos(pi - x) = -cos x           -> [0, 1/4]
    if (8 * a > b) return tsin(2.0 * kPi * (double)(b - 4 * a) / (double)(4 * b));   // cos x = sin(pi/2 - x)
    return tcos(2.0 * kPi * (double)a / (double)b);
}
}  // namespace ctrig

// real and imaginary part of exp(-2 pi i T / P) as float literals
template <int P, int T>
struct Wc {
    static constexpr float re = (float)ctrig::cos2pi(T, P);
    static constexpr float im = (float)(-ctrig::sin2pi(T, P));
};

// ---- static_for: f(std::integral_constant<int, I>) for I = B .. E-1 --------------------------------------------
template <int B, int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (E > B) static_for_impl<B>(std::make_integer_sequence<int, E - B>{}, static_cast<F&&>(f));
}

// acc + x * c  (complex x, real constant c): one packed FMA
__device__ __forceinline__ float2 cfma_r(float2 acc, float2 x, float c) {
    return to_f(__builtin_elementwise_fma(to_v(x), v2f{c, c}, to_v(acc)));
}
__device__ __forceinline__ float2 cmul_r(float2 x, float c) { return to_f(to_v(x) * v2f{c, c}); }

template <int R>
struct Radix;

// ---- odd lengths ------------------------------------------------------------------------------------------------
template <int P>
struct DftOdd {
    static_assert(P % 2 == 1 && P >= 3, "odd length");
    static constexpr int H = (P - 1) / 2;
    static __device__ __forceinline__ void run(float2 (&x)[P]) {
        float2 a[H + 1], b[H + 1];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            a[j] = cadd(x[j], x[P - j]);
            b[j] = csub(x[j], x[P - j]);
        }
        const float2 x0 = x[0];
        float2 s0 = x0;
#pragma unroll
        for (int j = 1; j <= H; ++j) s0 = cadd(s0, a[j]);
        x[0] = s0;
        static_for<1, H + 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            float2 c = x0, s = make_float2(0.f, 0.f);
            static_for<1, H + 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr float cs = (float)ctrig::cos2pi((long long)j * k, P);
                constexpr float sn = (float)ctrig::sin2pi((long long)j * k, P);
                c = cfma_r(c, a[j], cs);
                s = (j == 1) ? cmul_r(b[j], sn) : cfma_r(s, b[j], sn);
            });
            x[k] = cadd_mi(c, s);        // c + (-i) s
            x[P - k] = csub_mi(c, s);    // c - (-i) s
        });
    }
};

// ---- A x B on registers: n = B n1 + n2, k = k1 + A k2 -------------------------------------------------------------
template <int A, int B>
struct DftComp {
    static constexpr int R = A * B;
    static __device__ __forceinline__ void run(float2 (&x)[R]) {
        float2 y[B][A];
        static_for<0, B>([&](auto n2c) {
            constexpr int n2 = decltype(n2c)::value;
            float2 t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) t[n1] = x[B * n1 + n2];
            Radix<A>::run(t);
            static_for<0, A>([&](auto k1c) {
                constexpr int k1 = decltype(k1c)::value;
                constexpr int e = (n2 * k1) % R;
                if constexpr (e == 0)
                    y[n2][k1] = t[k1];
                else if constexpr (4 * e == R)
                    y[n2][k1] = cmul_mi(t[k1]);
                else if constexpr (2 * e == R)
                    y[n2][k1] = make_float2(-t[k1].x, -t[k1].y);
                else if constexpr (4 * e == 3 * R)
                    y[n2][k1] = make_float2(-t[k1].y, t[k1].x);
                else
                    y[n2][k1] = cmulc(t[k1], Wc<R, e>::re, Wc<R, e>::im);
            });
        });
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            float2 t[B];
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) t[n2] = y[n2][k1];
            Radix<B>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) x[k1 + A * k2] = t[k2];
        }
    }
};

// ---- dispatcher ---------------------------------------------------------------------------------------------------
template <int R>
struct Radix {   // default: odd lengths without a dedicated factorisation
    static __device__ __forceinline__ void run(float2 (&x)[R]) { DftOdd<R>::run(x); }
};
#define B4D_RADIX_POW2(R_)                                                                  \
    template <>                                                                             \
    struct Radix<R_> {                                                                      \
        static __device__ __forceinline__ void run(float2 (&x)[R_]) { Dft<R_>::run(x); }   \
    };
B4D_RADIX_POW2(1)
B4D_RADIX_POW2(2)
B4D_RADIX_POW2(4)
B4D_RADIX_POW2(8)
B4D_RADIX_POW2(16)
#undef B4D_RADIX_POW2
#define B4D_RADIX_COMP(R_, A_, B_)                                                                   \
    template <>                                                                                      \
    struct Radix<R_> {                                                                               \
        static __device__ __forceinline__ void run(float2 (&x)[R_]) { DftComp<A_, B_>::run(x); }    \
    };
B4D_RADIX_COMP(6, 2, 3)
B4D_RADIX_COMP(9, 3, 3)
B4D_RADIX_COMP(10, 2, 5)
B4D_RADIX_COMP(12, 4, 3)
B4D_RADIX_COMP(14, 2, 7)
B4D_RADIX_COMP(15, 3, 5)
B4D_RADIX_COMP(18, 2, 9)
B4D_RADIX_COMP(20, 4, 5)
B4D_RADIX_COMP(21, 3, 7)
B4D_RADIX_COMP(22, 2, 11)
B4D_RADIX_COMP(24, 8, 3)
B4D_RADIX_COMP(25, 5, 5)
B4D_RADIX_COMP(26, 2, 13)
B4D_RADIX_COMP(27, 3, 9)
B4D_RADIX_COMP(32, 4, 8)
#undef B4D_RADIX_COMP

}  // namespace b4d

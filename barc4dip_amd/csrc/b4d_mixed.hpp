// b4d_mixed.hpp -- one length-N = R1 * R2 * R3 transform per workgroup, three in-register radix stages through ONE
// in-place LDS row buffer (gfx950: 64 banks x 4 B, ds_read/write_b64).
//
//   n = M1 n1 + m,  m = R3 n2 + n3          (M1 = R2 R3)          k = k1 + R1 (k2 + R2 k3)
//   stage 1  item m      : DFT_R1 over n1 (inputs x[M1 n1 + m], i.e. coalesced in m), twiddle W_N^{m k1}
//                          -> buf[k1 S1 + m]
//   stage 2  item (k1,n3): DFT_R2 over n2 on buf[k1 S1 + R3 n2 + n3], twiddle W_{M1}^{n3 k2}, written back IN PLACE
//   stage 3  item (k1,k2): DFT_R3 over n3 on buf[k1 S1 + R3 k2 + n3], written back IN PLACE
//   X[k] then sits at pos(k) = k1 S1 + R3 k2 + k3.
// In-place stages need one barrier each (3 per transform, none between a stage's reads and writes).  S1 >= M1 is padded
// to S1 == R3 (mod 32) so that the (k1, n3) lanes of stage 2 hit consecutive 8-byte bank pairs; stage 3 walks the
// buffer with the odd stride R3.  Every radix runs on registers (b4d_radix.hpp): a 4104-point row costs ~0.2 M packed
// FMAs where the dense 27- and 19-point DFT-matrix stages it replaces cost 0.75 M real FMAs on the matrix cores.
// The stage-2 twiddles come from a small LDS table (M1 entries) built once per workgroup; stage-1 twiddles are read
// from the N-entry global table (L2-resident, loads issued next to the data loads).
#pragma once
#include "b4d_radix.hpp"

namespace b4d {

template <int R1_, int R2_, int R3_, int LANES_>
struct Mix3 {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_, LANES = LANES_;
    static constexpr int N = R1 * R2 * R3, M1 = R2 * R3, M2 = R1 * R3, M3 = R1 * R2;
    static constexpr int S1 = M1 + (((R3 - M1) % 32) + 32) % 32;   // smallest S1 >= M1 with S1 == R3 (mod 32)
    static constexpr int BUF = R1 * S1;                             // complex words of the row buffer
    static constexpr int ROUNDS1 = (M1 + LANES - 1) / LANES;
    static_assert(R1 % 2 == 0 || R2 % 2 == 0 || R3 % 2 == 0 || true, "");

    // where X[k] sits after the three in-place stages
    static __device__ __forceinline__ int pos(int k) {
        const int k1 = k % R1, r = k / R1, k2 = r % R2, k3 = r / R2;
        return k1 * S1 + k2 * R3 + k3;
    }
    // pos(k) for k = k0, k0 + STEP, k0 + 2 STEP ... (or downwards) without a division per element: the mixed-radix digits
    // (k1, k2, k3) advance by the digits of STEP with carries (integer multiplies / divisions are quarter rate: a pos()
    // per element cost the output loops 4 x the butterflies' time)
    template <int STEP>
    struct PosIter {
        static constexpr int D1 = STEP % R1, D2 = (STEP / R1) % R2, D3 = STEP / (R1 * R2);
        int k1, k2, k3;
        __device__ __forceinline__ explicit PosIter(int k) : k1(k % R1), k2((k / R1) % R2), k3(k / (R1 * R2)) {}
        __device__ __forceinline__ int pos() const { return k1 * S1 + k2 * R3 + k3; }
        __device__ __forceinline__ void up() {
            k1 += D1;
            int c = k1 >= R1 ? 1 : 0;
            k1 -= c * R1;
            k2 += D2 + c;
            c = k2 >= R2 ? 1 : 0;
            k2 -= c * R2;
            k3 += D3 + c;
        }
        __device__ __forceinline__ void down() {
            k1 -= D1;
            int b = k1 < 0 ? 1 : 0;
            k1 += b * R1;
            k2 -= D2 + b;
            b = k2 < 0 ? 1 : 0;
            k2 += b * R2;
            k3 -= D3 + b;
        }
    };
    // tw2[t] = W_{M1}^t = W_N^{R1 t}
    static __device__ __forceinline__ void build_tw2(float2* __restrict__ tw2, const float2* __restrict__ twN, int tid) {
        for (int t = tid; t < M1; t += LANES) tw2[t] = twN[R1 * t];
    }
    static __device__ __forceinline__ void stage1_item(float2 (&v)[R1], int m, float2* __restrict__ buf, const float2* __restrict__ twN) {
        float2 w[R1];
#pragma unroll
        for (int k1 = 1; k1 < R1; ++k1) w[k1] = twN[__umul24(m, k1)];   // v_mul_u32_u24 is full rate, v_mul_lo_u32 a quarter
        Radix<R1>::run(v);
        buf[m] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < R1; ++k1) buf[k1 * S1 + m] = cmul(v[k1], w[k1]);
    }
    static __device__ __forceinline__ void stage2(float2* __restrict__ buf, const float2* __restrict__ tw2, int tid) {
        for (int it = tid; it < M2; it += LANES) {
            const int n3 = it % R3, k1 = it / R3;
            float2* p = buf + k1 * S1 + n3;
            float2 v[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) v[n2] = p[R3 * n2];
            Radix<R2>::run(v);
            p[0] = v[0];
            // tw2[n3 k2] through a running index: left to itself the compiler turns the additions back into one quarter-rate
            // 32-bit multiply per twiddle (and recomputes it rather than keep it in a register)
            int ti = 0;
#pragma unroll
            for (int k2 = 1; k2 < R2; ++k2) {
                ti += n3;
                asm volatile("" : "+v"(ti));
                p[R3 * k2] = cmul(v[k2], tw2[ti]);
            }
        }
    }
    static __device__ __forceinline__ void stage3(float2* __restrict__ buf, int tid) {
        for (int it = tid; it < M3; it += LANES) {
            const int k2 = it % R2, k1 = it / R2;
            float2* p = buf + k1 * S1 + R3 * k2;
            float2 v[R3];
#pragma unroll
            for (int n3 = 0; n3 < R3; ++n3) v[n3] = p[n3];
            Radix<R3>::run(v);
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) p[k3] = v[k3];
        }
    }
};

}  // namespace b4d

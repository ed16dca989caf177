// CPU check of b4d_radix.hpp / b4d_mixed.hpp against a float64 DFT (dev tool):
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -DB4D_NO_PK -Itools/host_check -Ibarc4dip_amd/csrc tools/host_check/check_mixed.cpp -o /tmp/check_mixed && /tmp/check_mixed
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "b4d_mixed.hpp"
using namespace b4d;
typedef std::complex<double> cd;

template <int R>
static double check_radix() {
    float2 x[R];
    std::vector<cd> in(R);
    for (int i = 0; i < R; ++i) {
        x[i] = make_float2((float)drand48() - 0.5f, (float)drand48() - 0.5f);
        in[i] = cd(x[i].x, x[i].y);
    }
    Radix<R>::run(x);
    double err = 0, nrm = 0;
    for (int k = 0; k < R; ++k) {
        cd s = 0;
        for (int n = 0; n < R; ++n) s += in[n] * std::polar(1.0, -2.0 * M_PI * (double)((long long)n * k % R) / R);
        err = std::max(err, std::abs(s - cd(x[k].x, x[k].y)));
        nrm = std::max(nrm, std::abs(s));
    }
    printf("radix %2d  err/max %.2e\n", R, err / nrm);
    return err / nrm;
}

template <class MX>
static double check_mix() {
    constexpr int N = MX::N;
    std::vector<float2> tw(N), buf(MX::BUF), tw2(MX::M1), x(N);
    for (int t = 0; t < N; ++t) tw[t] = make_float2((float)std::cos(-2.0 * M_PI * t / N), (float)std::sin(-2.0 * M_PI * t / N));
    std::vector<cd> in(N);
    for (int i = 0; i < N; ++i) {
        x[i] = make_float2((float)drand48() - 0.5f, (float)drand48() - 0.5f);
        in[i] = cd(x[i].x, x[i].y);
    }
    for (int tid = 0; tid < MX::LANES; ++tid) MX::build_tw2(tw2.data(), tw.data(), tid);
    for (int tid = 0; tid < MX::LANES; ++tid)
        for (int r = 0; r < MX::ROUNDS1; ++r) {
            const int m = tid + r * MX::LANES;
            if (m >= MX::M1) continue;
            float2 v[MX::R1];
            for (int n1 = 0; n1 < MX::R1; ++n1) v[n1] = x[MX::M1 * n1 + m];
            MX::stage1_item(v, m, buf.data(), tw.data());
        }
    for (int tid = 0; tid < MX::LANES; ++tid) MX::stage2(buf.data(), tw2.data(), tid);
    for (int tid = 0; tid < MX::LANES; ++tid) MX::stage3(buf.data(), tid);
    // reference: float64 DFT by the same factorisation-free definition on a subset of bins (N^2 is fine up to ~5k)
    double err = 0, nrm = 0;
    for (int k = 0; k < N; ++k) {
        cd s = 0;
        for (int n = 0; n < N; ++n) s += in[n] * std::polar(1.0, -2.0 * M_PI * (double)((long long)n * k % N) / N);
        const float2 g = buf[MX::pos(k)];
        err = std::max(err, std::abs(s - cd(g.x, g.y)));
        nrm = std::max(nrm, std::abs(s));
    }
    {   // digit iterators against pos()
        typename MX::template PosIter<MX::LANES> up(3), dn(N - 5);
        for (int k = 3, j = N - 5; k < N && j >= 0; k += MX::LANES, j -= MX::LANES) {
            if (up.pos() != MX::pos(k) || dn.pos() != MX::pos(j)) {
                printf("PosIter mismatch at %d / %d\n", k, j);
                err = 1;
            }
            up.up();
            dn.down();
        }
        typename MX::template PosIter<MX::M1> st(7);
        for (int n1 = 0; n1 < MX::R1; ++n1) {
            if (st.pos() != MX::pos(7 + MX::M1 * n1)) printf("PosIter<M1> mismatch\n"), err = 1;
            st.up();
        }
    }
    printf("mix %d = %d x %d x %d (lanes %d, S1 %d)  err/max %.2e\n", N, MX::R1, MX::R2, MX::R3, MX::LANES, MX::S1, err / nrm);
    return err / nrm;
}

int main() {
    double worst = 0;
    worst = std::max(worst, check_radix<3>());
    worst = std::max(worst, check_radix<5>());
    worst = std::max(worst, check_radix<6>());
    worst = std::max(worst, check_radix<7>());
    worst = std::max(worst, check_radix<9>());
    worst = std::max(worst, check_radix<10>());
    worst = std::max(worst, check_radix<11>());
    worst = std::max(worst, check_radix<12>());
    worst = std::max(worst, check_radix<13>());
    worst = std::max(worst, check_radix<15>());
    worst = std::max(worst, check_radix<17>());
    worst = std::max(worst, check_radix<19>());
    worst = std::max(worst, check_radix<18>());
    worst = std::max(worst, check_radix<20>());
    worst = std::max(worst, check_radix<22>());
    worst = std::max(worst, check_radix<24>());
    worst = std::max(worst, check_radix<25>());
    worst = std::max(worst, check_radix<27>());
    worst = std::max(worst, check_radix<32>());
    worst = std::max(worst, check_mix<Mix3<8, 27, 19, 256>>());
    worst = std::max(worst, check_mix<Mix3<19, 8, 27, 256>>());
    worst = std::max(worst, check_mix<Mix3<8, 5, 13, 256>>());
    worst = std::max(worst, check_mix<Mix3<8, 3, 11, 128>>());
    worst = std::max(worst, check_mix<Mix3<16, 27, 5, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 10, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 16, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 15, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 12, 19, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 20, 10, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 12, 256>>());
    worst = std::max(worst, check_mix<Mix3<15, 10, 20, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 18, 9, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 9, 17, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 15, 10, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 9, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 8, 256>>());
    worst = std::max(worst, check_mix<Mix3<8, 27, 9, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 11, 11, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 12, 10, 256>>());
    worst = std::max(worst, check_mix<Mix3<16, 10, 10, 128>>());
    worst = std::max(worst, check_mix<Mix3<16, 16, 6, 128>>());
    worst = std::max(worst, check_mix<Mix3<12, 12, 10, 128>>());
    worst = std::max(worst, check_mix<Mix3<16, 4, 19, 128>>());
    worst = std::max(worst, check_mix<Mix3<10, 12, 10, 128>>());
    worst = std::max(worst, check_mix<Mix3<12, 10, 9, 128>>());
    worst = std::max(worst, check_mix<Mix3<16, 8, 8, 128>>());
    worst = std::max(worst, check_mix<Mix3<10, 10, 10, 128>>());
    worst = std::max(worst, check_mix<Mix3<16, 12, 5, 128>>());
    worst = std::max(worst, check_mix<Mix3<16, 10, 5, 128>>());
    worst = std::max(worst, check_mix<Mix3<8, 8, 12, 64>>());
    worst = std::max(worst, check_mix<Mix3<16, 8, 5, 64>>());
    worst = std::max(worst, check_mix<Mix3<16, 6, 6, 64>>());
    worst = std::max(worst, check_mix<Mix3<12, 9, 5, 64>>());
    worst = std::max(worst, check_mix<Mix3<8, 8, 8, 64>>());
    worst = std::max(worst, check_mix<Mix3<8, 6, 10, 64>>());
    worst = std::max(worst, check_mix<Mix3<4, 3, 19, 64>>());    // 228: aggregator sub-tiles of 2048-px frames
    worst = std::max(worst, check_mix<Mix3<3, 3, 19, 64>>());    // 171 (odd): aggregator tiles of 512-px frames
    worst = std::max(worst, check_mix<Mix3<5, 2, 17, 64>>());    // 170
    printf(worst < 2e-6 ? "OK\n" : "FAIL\n");
    return worst < 2e-6 ? 0 : 1;
}

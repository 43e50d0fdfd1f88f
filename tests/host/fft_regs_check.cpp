// Host check of os2d_amd/csrc/fft_regs.h: every register DFT size and every two-stage factorisation used by fft.hip against a
// direct double-precision DFT (forward and inverse).  Built and run by tests/test_fft_regs_host.py (g++ / clang++, no GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_regs.h"

using namespace os2d_fft;

static double worst = 0.0;

template <int R, bool INV>
static void check_dft() {
  cf32 v[R];
  std::vector<double> re(R), im(R);
  for (int i = 0; i < R; ++i) {
    re[i] = std::sin(1.0 + 0.7 * i * i) + 0.1 * i;
    im[i] = std::cos(2.0 + 1.3 * i);
    v[i] = cf32{(float)re[i], (float)im[i]};
    re[i] = v[i][0];
    im[i] = v[i][1];
  }
  Dft<R, INV>::run(v);
  for (int k = 0; k < R; ++k) {
    double sr = 0, si = 0;
    for (int n = 0; n < R; ++n) {
      const double a = (INV ? 2.0 : -2.0) * M_PI * ((long long)k * n % R) / R;
      sr += re[n] * std::cos(a) - im[n] * std::sin(a);
      si += re[n] * std::sin(a) + im[n] * std::cos(a);
    }
    const double e = std::fmax(std::fabs(sr - v[k][0]), std::fabs(si - v[k][1]));
    if (e > worst) worst = e;
    if (e > 2e-5) {
      std::printf("Dft<%d,%d> bin %d: got (%g, %g), want (%g, %g)\n", R, (int)INV, k, v[k][0], v[k][1], sr, si);
      std::exit(1);
    }
  }
}

template <int R1, int R2, bool INV>
static void check_two_stage(int nfft) {
  constexpr int N = R1 * R2, THREADS = 64;
  const int sstride = N + 1, zstride = (R1 * ZRow<R2>::value) | 1;
  std::vector<cf32> src(nfft * sstride), Z(nfft * zstride), tw(N);
  for (int j = 0; j < N; ++j) tw[j] = cf32{(float)std::cos(-2.0 * M_PI * j / N), (float)std::sin(-2.0 * M_PI * j / N)};
  for (int f = 0; f < nfft; ++f)
    for (int n = 0; n < N; ++n) src[f * sstride + n] = cf32{(float)std::sin(0.3 * n * (f + 1) + f), (float)std::cos(0.11 * n * n - f)};
  std::vector<cf32> in = src;
  const unsigned inv = nfft > 1 ? (unsigned)(((1ull << 32) + nfft - 1) / nfft) : 0u;
  for (int tid = 0; tid < THREADS; ++tid) two_stage_first<R1, R2, INV, THREADS>(src.data(), sstride, Z.data(), zstride, nfft, inv, tw.data(), tid);
  for (int tid = 0; tid < THREADS; ++tid) two_stage_second<R1, R2, INV, THREADS>(Z.data(), zstride, src.data(), sstride, nfft, inv, tid);
  for (int f = 0; f < nfft; ++f)
    for (int k = 0; k < N; ++k) {
      double sr = 0, si = 0;
      for (int n = 0; n < N; ++n) {
        const double a = (INV ? 2.0 : -2.0) * M_PI * ((long long)k * n % N) / N;
        const double xr = in[f * sstride + n][0], xi = in[f * sstride + n][1];
        sr += xr * std::cos(a) - xi * std::sin(a);
        si += xr * std::sin(a) + xi * std::cos(a);
      }
      const cf32 got = src[f * sstride + k];
      const double e = std::fmax(std::fabs(sr - got[0]), std::fabs(si - got[1]));
      if (e > worst) worst = e;
      if (e > 2e-4) {
        std::printf("two_stage<%d,%d,%d> seq %d bin %d: got (%g, %g), want (%g, %g)\n", R1, R2, (int)INV, f, k, got[0], got[1], sr, si);
        std::exit(1);
      }
    }
}

template <int R1, int R2>
static void both(int nfft) {
  check_two_stage<R1, R2, false>(nfft);
  check_two_stage<R1, R2, true>(nfft);
}

int main() {
  check_dft<2, false>(); check_dft<3, false>(); check_dft<4, false>(); check_dft<6, false>(); check_dft<8, false>();
  check_dft<9, false>(); check_dft<12, false>(); check_dft<16, false>(); check_dft<7, false>(); check_dft<7, true>();
  check_dft<2, true>(); check_dft<3, true>(); check_dft<4, true>(); check_dft<6, true>(); check_dft<8, true>();
  check_dft<9, true>(); check_dft<12, true>(); check_dft<16, true>();
  for (int nfft : {1, 3, 30, 49}) {
    both<4, 4>(nfft); both<6, 4>(nfft); both<8, 4>(nfft); both<6, 6>(nfft); both<8, 6>(nfft); both<9, 6>(nfft); both<8, 8>(nfft);
    both<9, 8>(nfft); both<12, 8>(nfft); both<12, 7>(nfft); both<6, 7>(nfft); both<12, 9>(nfft); both<16, 8>(nfft); both<12, 12>(nfft);
  }
  std::printf("ok worst abs error %.3g\n", worst);
  return 0;
}

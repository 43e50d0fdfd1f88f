// Register-resident building blocks of the in-LDS FFTs (fft.hip): small DFTs of compile-time size evaluated entirely in
// registers and the two-stage decomposition N = R1 * R2 that needs ONE exchange through LDS per transform instead of one
// per radix-2/3/4 pass:
//
//     X[k1 + R1 k2] = sum_t  w_R2^(t k2)  [ w_N^(t k1)  sum_m w_R1^(m k1) x[t + R2 m] ]          w_n = exp(-+ 2 pi i / n)
//
//   stage 1   item (t, f): loads x_f[t + R2 m], m < R1, DFT of size R1 in registers, twiddles w_N^(t k1), stores Z_f[k1][t]
//   stage 2   item (k1, f): loads Z_f[k1][t], t < R2, DFT of size R2 in registers, stores X_f[k1 + R1 k2]
//
// The transform index f runs ACROSS the lanes (consecutive lanes = consecutive transforms at the same t / k1), so every
// LDS access of a wave has the transform stride between neighbouring lanes: with odd strides (in complex numbers) all
// 64-bit accesses are bank-conflict free - the radix passes of the Stockham formulation write with stride R and measured
// 40 % of their LDS cycles as conflict cycles (profiles/r02_fft_rocprof_summary.txt).
//
// The file compiles for the host as well (tests/host/fft_regs_check.cpp checks every size against a direct DFT).
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define OS2D_FFT_HD __host__ __device__ __forceinline__
#else
#define OS2D_FFT_HD inline
#endif

namespace os2d_fft {

typedef float cf32 __attribute__((ext_vector_type(2)));

template <int I>
struct IC {
  static constexpr int value = I;
};

// compile-time loop: f(IC<I>{}) for I = 0 .. N-1 (indices usable as template arguments / constant expressions)
template <int I, int N, class F>
OS2D_FFT_HD void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- roots of unity at compile time: double Taylor series on [-pi, pi], rounded once to float; the axes are exact
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double taylor_sin(double x) {
  double term = x, sum = x;
  for (int n = 1; n < 24; ++n) {
    term *= -x * x / ((2 * n) * (2 * n + 1));
    sum += term;
  }
  return sum;
}
constexpr double taylor_cos(double x) {
  double term = 1.0, sum = 1.0;
  for (int n = 1; n < 24; ++n) {
    term *= -x * x / ((2 * n - 1) * (2 * n));
    sum += term;
  }
  return sum;
}
constexpr double unit_angle(int j, int n) {   // 2 pi (j mod n) / n folded to (-pi, pi]
  j = ((j % n) + n) % n;
  return 2 * j <= n ? 2 * kPi * j / n : 2 * kPi * j / n - 2 * kPi;
}
constexpr double unit_cos(int j, int n) {
  j = ((j % n) + n) % n;
  if ((4 * j) % n == 0) return (4 * j / n) == 0 ? 1.0 : (4 * j / n) == 2 ? -1.0 : 0.0;
  return taylor_cos(unit_angle(j, n));
}
constexpr double unit_sin(int j, int n) {
  j = ((j % n) + n) % n;
  if ((4 * j) % n == 0) return (4 * j / n) == 1 ? 1.0 : (4 * j / n) == 3 ? -1.0 : 0.0;
  return taylor_sin(unit_angle(j, n));
}

OS2D_FFT_HD cf32 cmul(cf32 a, cf32 b) { return cf32{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }
OS2D_FFT_HD cf32 cconj(cf32 a) { return cf32{a[0], -a[1]}; }

// v * w_N^J (forward: exp(-2 pi i J / N); INV: the conjugate), J and N compile-time: multiplications by 1, -1, +-i are free
template <int J, int N, bool INV>
OS2D_FFT_HD cf32 mul_root(cf32 v) {
  constexpr int j = ((J % N) + N) % N;
  if constexpr (j == 0) return v;
  else if constexpr (2 * j == N) return cf32{-v[0], -v[1]};
  else if constexpr (4 * j == N) return INV ? cf32{-v[1], v[0]} : cf32{v[1], -v[0]};          // -i (forward), +i (inverse)
  else if constexpr (4 * j == 3 * N) return INV ? cf32{v[1], -v[0]} : cf32{-v[1], v[0]};      // +i (forward), -i (inverse)
  else {
    constexpr float c = (float)unit_cos(j, N), s = (float)(INV ? unit_sin(j, N) : -unit_sin(j, N));
    return cf32{v[0] * c - v[1] * s, v[0] * s + v[1] * c};
  }
}

// ---- DFT of compile-time size R in registers, in place, natural order in and out.  Base cases 2, 3, 4; composite sizes
// by decimation in time with the largest radix (4, 3, 2) as the combining butterfly: R = RA * RB,
//   X[k + RB q] = sum_s w_RA^(s q) [ w_R^(s k) S_s[k] ],   S_s = DFT_RB of x[s + RA m]
template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<1, INV> {
  static OS2D_FFT_HD void run(cf32 (&)[1]) {}
};
template <bool INV>
struct Dft<2, INV> {
  static OS2D_FFT_HD void run(cf32 (&v)[2]) {
    const cf32 a = v[0] + v[1], b = v[0] - v[1];
    v[0] = a;
    v[1] = b;
  }
};
template <bool INV>
struct Dft<3, INV> {
  static OS2D_FFT_HD void run(cf32 (&v)[3]) {
    constexpr float s = 0.86602540378443864676f;   // sin(pi / 3)
    const cf32 t1 = v[1] + v[2];
    const cf32 t2 = v[0] - 0.5f * t1;
    const cf32 d = v[1] - v[2];
    const cf32 t3 = INV ? cf32{-s * d[1], s * d[0]} : cf32{s * d[1], -s * d[0]};   // -+ i s d
    v[0] = v[0] + t1;
    v[1] = t2 + t3;
    v[2] = t2 - t3;
  }
};
template <bool INV>
struct Dft<4, INV> {
  static OS2D_FFT_HD void run(cf32 (&v)[4]) {
    const cf32 a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], e = v[1] - v[3];
    const cf32 d = INV ? cf32{-e[1], e[0]} : cf32{e[1], -e[0]};   // -+ i e
    v[0] = a + c;
    v[1] = b + d;
    v[2] = a - c;
    v[3] = b - d;
  }
};
// 7 points, directly: with a_n = x_n + x_(7-n), b_n = x_n - x_(7-n) (n = 1..3)
//   X_k, X_(7-k) = x_0 + sum_n a_n cos(2 pi n k / 7)  -+ i sum_n b_n sin(2 pi n k / 7)        (signs swapped for INV)
template <bool INV>
struct Dft<7, INV> {
  static OS2D_FFT_HD void run(cf32 (&v)[7]) {
    cf32 a[3], b[3];
    static_for<0, 3>([&](auto n) {
      a[n.value] = v[n.value + 1] + v[6 - n.value];
      b[n.value] = v[n.value + 1] - v[6 - n.value];
    });
    const cf32 x0 = v[0];
    v[0] = x0 + a[0] + a[1] + a[2];
    static_for<1, 4>([&](auto k) {
      cf32 re = x0, im = cf32{0.f, 0.f};
      static_for<0, 3>([&](auto n) {
        constexpr float c = (float)unit_cos((n.value + 1) * k.value, 7), s = (float)unit_sin((n.value + 1) * k.value, 7);
        re += c * a[n.value];
        im += s * b[n.value];
      });
      const cf32 rot = INV ? cf32{-im[1], im[0]} : cf32{im[1], -im[0]};     // -+ i im
      v[k.value] = re + rot;
      v[7 - k.value] = re - rot;
    });
  }
};
template <int R, bool INV>
struct Dft {
  static constexpr int RA = R % 4 == 0 ? 4 : R % 3 == 0 ? 3 : 2;
  static constexpr int RB = R / RA;
  static_assert(R > 4 && RA * RB == R, "composite sizes are products of 2s and 3s");
  static OS2D_FFT_HD void run(cf32 (&v)[R]) {
    cf32 sub[RA][RB];
    static_for<0, RA>([&](auto s) {
      static_for<0, RB>([&](auto m) { sub[s.value][m.value] = v[s.value + RA * m.value]; });
      Dft<RB, INV>::run(sub[s.value]);
    });
    static_for<0, RB>([&](auto k) {
      cf32 t[RA];
      static_for<0, RA>([&](auto s) { t[s.value] = mul_root<s.value * k.value, R, INV>(sub[s.value][k.value]); });
      Dft<RA, INV>::run(t);
      static_for<0, RA>([&](auto q) { v[k.value + RB * q.value] = t[q.value]; });
    });
  }
};

// ---- the two stages of an N = R1 * R2 point transform over `nfft` sequences.  item -> (t or k1, f) with f fastest:
// `inv_nfft` = ceil(2^32 / nfft) (0 for nfft == 1) turns the division into a multiply-high.  Z holds R1 rows of R2P = R2 | 1
// numbers per sequence (odd row length: the stage-2 loads of one item walk along a row, those of neighbouring lanes sit
// a whole sequence stride apart); tw = table of w_N^j, j < N (forward values; conjugated for INV).
template <int R2>
struct ZRow {
  static constexpr int value = R2 | 1;
};

template <int R1, int R2, bool INV, int THREADS, class TW>
OS2D_FFT_HD void two_stage_first(const cf32* src, int sstride, cf32* Z, int zstride, int nfft, unsigned inv_nfft, TW tw,
                                 int tid) {
  constexpr int R2P = ZRow<R2>::value;
  for (int item = tid; item < nfft * R2; item += THREADS) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int t = inv_nfft ? (int)__umulhi((unsigned)item, inv_nfft) : item;
#else
    const int t = inv_nfft ? (int)(((unsigned long long)(unsigned)item * inv_nfft) >> 32) : item;
#endif
    const int f = item - t * nfft;
    cf32 v[R1];
    const cf32* s = src + f * sstride + t;
    static_for<0, R1>([&](auto m) { v[m.value] = s[R2 * m.value]; });
    Dft<R1, INV>::run(v);
    cf32* z = Z + f * zstride + t;
    static_for<0, R1>([&](auto k1) {
      cf32 x = v[k1.value];
      if (k1.value > 0 && t > 0) {
        cf32 w = tw[t * k1.value];
        if (INV) w = cconj(w);
        x = cmul(x, w);
      }
      z[k1.value * R2P] = x;
    });
  }
}

template <int R1, int R2, bool INV, int THREADS>
OS2D_FFT_HD void two_stage_second(const cf32* Z, int zstride, cf32* dst, int dstride, int nfft, unsigned inv_nfft, int tid) {
  constexpr int R2P = ZRow<R2>::value;
  for (int item = tid; item < nfft * R1; item += THREADS) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int k1 = inv_nfft ? (int)__umulhi((unsigned)item, inv_nfft) : item;
#else
    const int k1 = inv_nfft ? (int)(((unsigned long long)(unsigned)item * inv_nfft) >> 32) : item;
#endif
    const int f = item - k1 * nfft;
    cf32 v[R2];
    const cf32* z = Z + f * zstride + k1 * R2P;
    static_for<0, R2>([&](auto t) { v[t.value] = z[t.value]; });
    Dft<R2, INV>::run(v);
    cf32* d = dst + f * dstride + k1;
    static_for<0, R2>([&](auto k2) { d[R1 * k2.value] = v[k2.value]; });
  }
}

}  // namespace os2d_fft

// Host-side SPMD emulator for the work-group kernels of os2d_amd/csrc/dft_mfma.h: every work item of a work-group is an OS
// thread, LDS is a per-group buffer, the work-group barrier is a pthread barrier, and the wave-level instructions the kernels
// use (v_mfma_f32_32x32x16_f16, cross-lane shuffles) are emulated by exchanging operands through a per-wave scratch area
// between two wave barriers.  The kernel source is compiled UNCHANGED (clang++ as a host compiler: ext_vector_type and
// _Float16 are available); only the handful of hardware hooks below differ from the device build.  Test infrastructure: the
// index arithmetic of the kernels (LDS layouts, fragment addressing, tile ownership, tilings) is checked on the CPU against a
// float64 DFT before a GPU minute is spent (tests/test_dft_mfma_host.py).
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <functional>
#include <thread>
#include <vector>

namespace emu {

constexpr int WAVE = 64;

struct WaveScratch {
  pthread_barrier_t bar;
  _Float16 a[WAVE][8], b[WAVE][8];
  float f[WAVE];
  int i[WAVE];
};

struct Group {
  int nthreads = 0;
  pthread_barrier_t bar;
  std::vector<unsigned char> lds;
  std::vector<WaveScratch> waves;
};

struct ThreadCtx {
  int tid = 0, bid = 0, grid = 0, nthreads = 0;
  Group* g = nullptr;
};
inline thread_local ThreadCtx ctx;

inline int tid() { return ctx.tid; }
inline int bid() { return ctx.bid; }
inline int grid() { return ctx.grid; }
inline unsigned char* lds() { return ctx.g->lds.data(); }
inline void group_barrier() { pthread_barrier_wait(&ctx.g->bar); }
inline WaveScratch& wave() { return ctx.g->waves[ctx.tid / WAVE]; }
inline void wave_barrier() { pthread_barrier_wait(&wave().bar); }

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x16_f16: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], i, j < 32, k < 16.  Lane l supplies A[i = l % 32][k = 8 (l / 32)
// + t] and B[k = 8 (l / 32) + t][j = l % 32], t < 8, and receives column j = l % 32, rows i = (r & 3) + 8 (r >> 2) + 4 (l / 32) in
// accumulator register r < 16 (the layout the product kernels rely on: spectral_f16.hip, conv_f16x3.hip).
inline f32x16 mfma_32x32x16_f16(half8 a, half8 b, f32x16 c) {
  WaveScratch& w = wave();
  const int l = ctx.tid % WAVE;
  for (int t = 0; t < 8; ++t) {
    w.a[l][t] = a[t];
    w.b[l][t] = b[t];
  }
  wave_barrier();
  const int j = l % 32, hw = l / 32;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hw;
    // the 16 products of a row are exact in fp32 (fp16 x fp16); they are summed here in double and rounded ONCE per
    // instruction - the hardware's internal order is not documented, this is the least-rounding model of it
    double s = c[r];
    for (int kh = 0; kh < 2; ++kh)
      for (int t = 0; t < 8; ++t) s += (double)((float)w.a[kh * 32 + i][t] * (float)w.b[kh * 32 + j][t]);
    c[r] = (float)s;
  }
  wave_barrier();
  return c;
}

inline float shfl_xor(float v, int mask) {
  WaveScratch& w = wave();
  const int l = ctx.tid % WAVE;
  w.f[l] = v;
  wave_barrier();
  const float r = w.f[l ^ mask];
  wave_barrier();
  return r;
}

inline unsigned shfl_xor_u32(unsigned v, int mask) {
  WaveScratch& w = wave();
  const int l = ctx.tid % WAVE;
  w.i[l] = (int)v;
  wave_barrier();
  const unsigned r = (unsigned)w.i[l ^ mask];
  wave_barrier();
  return r;
}

inline unsigned long long ballot(bool p) {
  WaveScratch& w = wave();
  const int l = ctx.tid % WAVE;
  w.i[l] = p ? 1 : 0;
  wave_barrier();
  unsigned long long m = 0;
  for (int k = 0; k < WAVE; ++k) m |= (unsigned long long)(w.i[k] != 0) << k;
  wave_barrier();
  return m;
}

// run `body` as a grid of `grid` work-groups of `nthreads` work items with `lds_bytes` of LDS (groups one after the other)
inline void launch(int grid, int nthreads, size_t lds_bytes, const std::function<void()>& body) {
  for (int b = 0; b < grid; ++b) {
    Group g;
    g.nthreads = nthreads;
    g.lds.assign(lds_bytes + 64, 0xA5);      // garbage, not zeros: a kernel must not rely on a cleared LDS
    pthread_barrier_init(&g.bar, nullptr, nthreads);
    g.waves = std::vector<WaveScratch>((nthreads + WAVE - 1) / WAVE);
    for (size_t w = 0; w < g.waves.size(); ++w) {
      const int n = std::min(WAVE, nthreads - (int)w * WAVE);
      pthread_barrier_init(&g.waves[w].bar, nullptr, n);
    }
    std::vector<std::thread> threads;
    for (int t = 0; t < nthreads; ++t)
      threads.emplace_back([&, t] {
        ctx = ThreadCtx{t, b, grid, nthreads, &g};
        body();
      });
    for (auto& th : threads) th.join();
    pthread_barrier_destroy(&g.bar);
    for (auto& w : g.waves) pthread_barrier_destroy(&w.bar);
  }
}

}  // namespace emu

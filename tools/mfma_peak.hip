// What does v_mfma_f32_32x32x16_f16 SUSTAIN on this chip?  Register-only loop: every wave keeps its A / B fragments in registers
// and issues NACC independent accumulators round-robin - no LDS, no memory, nothing to wait for but the matrix pipe.  Run with
// all-zero operands and with random normal fp16 data (the switching activity of real operands is what the power management reacts
// to: DESIGN 4.2).  Output: one line per (operands, waves per SIMD): TFLOP/s and the fraction of the 2.5 PFLOP/s nominal peak
// (256 CUs x 4 SIMDs x 1024 FLOP/cycle x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const half8* __restrict__ ab, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  half8 a[2], b[2];
  a[0] = ab[(t * 4 + 0) & 4095];
  a[1] = ab[(t * 4 + 1) & 4095];
  b[0] = ab[(t * 4 + 2) & 4095];
  b[1] = ab[(t * 4 + 3) & 4095];
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 1], b[(k >> 1) & 1], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) out[t] = s;      // never true: keeps the accumulators alive
}

template <int NACC>
double run(const half8* dab, float* dout, int waves_per_simd, int iters, int reps) {
  // 256 threads = 4 waves = one per SIMD; waves_per_simd work-groups per CU
  const int grid = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, dab, dout, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, dab, dout, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * grid * 4 /*waves*/ * iters * 4.0 * NACC * 32 * 32 * 16 * 2;
  return flops / (ms * 1e-3);
}

int main(int argc, char** argv) {
  // mfma_peak loop <zero|random> <seconds>: keep the loop running (for tools/power_probe.sh: package power while it does)
  const bool forever = argc >= 4 && std::string(argv[1]) == "loop";
  const int only_mode = forever ? (std::string(argv[2]) == "random" ? 1 : 0) : -1;
  const double seconds = forever ? atof(argv[3]) : 0.0;
  std::vector<_Float16> h(4096 * 8);
  half8* dab;
  float* dout;
  hipMalloc(&dab, h.size() * 2);
  hipMalloc(&dout, 1 << 22);
  for (int mode = 0; mode < 2; ++mode) {
    if (only_mode >= 0 && mode != only_mode) continue;
    srand(1);
    for (auto& x : h) {
      float u = 0.f;
      for (int k = 0; k < 12; ++k) u += rand() / (float)RAND_MAX;
      x = mode ? (_Float16)((u - 6.f) * 64.f) : (_Float16)0.f;        // ~N(0, 64^2): all mantissa / exponent bits busy
    }
    hipMemcpy(dab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    if (forever) {
      double done = 0.0, rate = 0.0;
      int n = 0;
      while (done < seconds) {
        rate = run<4>(dab, dout, 2, 20000, 20);
        done += 20.0 * 256 * 2 * 4 * 20000 * 4.0 * 4 * 32 * 32 * 16 * 2 / rate;
        ++n;
      }
      printf("LOOP mfma_%s iterations %d last rate %.1f TFLOP/s\n", mode ? "random" : "zero", n, rate / 1e12);
      continue;
    }
    for (int wps : {1, 2, 4}) {
      const double f4 = run<4>(dab, dout, wps, 20000, 5);
      const double f2 = run<2>(dab, dout, wps, 20000, 5);
      printf("operands %-6s waves/SIMD %d : 4 accumulators %7.1f TFLOP/s (%.3f of 2500)   2 accumulators %7.1f TFLOP/s (%.3f)\n",
             mode ? "random" : "zero", wps, f4 / 1e12, f4 / 2.5e15, f2 / 1e12, f2 / 2.5e15);
    }
  }
  return 0;
}

// Does the 3-instruction fp16 hi + lo split (v_cvt_pk_f16_f32 + v_fma_mixlo_f16 + v_fma_mixhi_f16) give the same bits as the 6-instruction
// one (v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, 2 x v_sub_f32, v_cvt_pk_f16_f32)?  Random floats over the whole exponent range the kernels
// meet, plus zeros, tiny values whose lo part is an fp16 subnormal, values beyond the fp16 range and NaN / Inf.
//   hipcc --offload-arch=gfx950 -O3 tools/split_mix_check.hip -o tools/bin/split_mix_check && tools/bin/split_mix_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned split_lo_mix(float x0, float x1, unsigned hi) {
  unsigned lo;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo)
      : "v"(x0), "v"(x1), "v"(hi));
  return lo;
}

__global__ void check(const float* x, unsigned* ref_hi, unsigned* ref_lo, unsigned* mix_lo, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const f32x2v v = {x[2 * i], x[2 * i + 1]};
  const half2v h = __builtin_convertvector(v, half2v);
  const half2v l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2v), half2v);
  const unsigned hp = __builtin_bit_cast(unsigned, h);
  ref_hi[i] = hp;
  ref_lo[i] = __builtin_bit_cast(unsigned, l);
  mix_lo[i] = split_lo_mix(v[0], v[1], hp);
}

int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    const int e = rand() % 60 - 40;                       // 2^-40 .. 2^19
    float m = 1.0f + (rand() & 0x7fffff) / 8388608.0f;
    h[i] = ldexpf(m, e) * ((rand() & 1) ? -1.f : 1.f);
  }
  const float special[] = {0.f, -0.f, 65504.f, 65520.f, 1e30f, -1e30f, 6.1e-5f, 5.96e-8f, 1e-10f, INFINITY, -INFINITY, NAN, 2049.f, 2051.f, 4097.5f, 0.33333334f};
  for (size_t k = 0; k < sizeof(special) / 4; ++k) h[k] = special[k];
  float* dx;
  unsigned *dh, *dl, *dm;
  hipMalloc(&dx, n * 4);
  hipMalloc(&dh, n * 2);
  hipMalloc(&dl, n * 2);
  hipMalloc(&dm, n * 2);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 512), dim3(256), 0, 0, dx, dh, dl, dm, n);
  std::vector<unsigned> rl(n / 2), ml(n / 2);
  hipMemcpy(rl.data(), dl, n * 2, hipMemcpyDeviceToHost);
  hipMemcpy(ml.data(), dm, n * 2, hipMemcpyDeviceToHost);
  long diff = 0, nan_only = 0;
  for (int i = 0; i < n / 2; ++i)
    if (rl[i] != ml[i]) {
      // NaN payloads may differ: compare "both NaN" per half
      bool real = false;
      for (int s = 0; s < 2; ++s) {
        const unsigned a = (rl[i] >> (16 * s)) & 0xffff, b = (ml[i] >> (16 * s)) & 0xffff;
        const bool an = (a & 0x7c00) == 0x7c00 && (a & 0x3ff), bn = (b & 0x7c00) == 0x7c00 && (b & 0x3ff);
        if (a != b && !(an && bn)) real = true;
      }
      if (real) {
        if (diff < 8) printf("DIFF pair %d: x = %g %g  ref lo %08x  mix lo %08x\n", i, h[2 * i], h[2 * i + 1], rl[i], ml[i]);
        ++diff;
      } else ++nan_only;
    }
  printf("split_mix_check: %d pairs, %ld differ (%ld differ in NaN payload only)\n", n / 2, diff, nan_only);
  return diff ? 1 : 0;
}

// Minimal victim / aggressor pair for the packed-FP32 finding of docs/DESIGN_HISTORY_r1-r3.md section 8 (MI355X, gfx950).
//
//   victim     a chain of complex multiply-adds held entirely in registers: no LDS, no barrier, no memory access inside the
//              loop.  Built twice from this file: with v_pk_{mul,add,fma}_f32 (default codegen) and with scalar v_fma_f32
//              (-Xclang -target-feature -Xclang -packed-fp32-ops).  Its result is a pure function of the thread index.
//   aggressor  half-precision MFMA at full rate (v_mfma_f32_32x32x16_f16 back to back) on OTHER streams.
//
// The victim runs alone first (reference), then R times next to the aggressors; every output word is compared bit for bit.
// Build (tools/diag_packed_fp32.sh does this):
//   hipcc --offload-arch=gfx950 -O3 tools/repro_packed_fp32.hip -o tools/bin/repro_pk_on
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops tools/repro_packed_fp32.hip -o tools/bin/repro_pk_off
// Run: repro_pk_on [rounds=200] [victim_iters=2000] [exclusive=0|1] [aggressor=0|1|2]
//   exclusive: the victim asks for 160 KB of LDS per group and the aggressor for 1 KB, so that no aggressor group can share
//              a CU with a victim group;  aggressor: 0 = fp16 MFMA (default), 1 = fp32 MFMA, 2 = scalar-FMA VALU loop (no MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NV = 8;                                   // complex registers per work item

__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) { return f32x2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }

__global__ __launch_bounds__(512) void victim(f32x2* __restrict__ out, int iters) {
  extern __shared__ char unused_lds[];                  // only its SIZE matters (CU exclusivity), never touched
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f32x2 a[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) a[k] = f32x2{1.0f + 1e-3f * (float)((t * NV + k) % 977), 0.5f - 1e-3f * (float)((t + 3 * k) % 811)};
  const f32x2 w = f32x2{0.99950656f, 0.03141076f};      // exp(i pi / 100): |w| = 1 up to rounding
  const f32x2 h = f32x2{0.4995f, 0.4995f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {                      // rotate, then mix with the neighbour register: values stay O(1)
      const f32x2 r = cmul(a[k], w);
      a[k] = r * h + a[(k + 1) % NV] * h;
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) out[(size_t)t * NV + k] = a[k];
}

typedef float f32x16b __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void aggressor_f32(float* __restrict__ out, int iters) {     // v_mfma_f32_32x32x2_f32
  __shared__ float pad[256];
  const int t = threadIdx.x;
  pad[t] = 0.f;
  const float a = 0.001f * (float)(t % 13), b = 0.002f * (float)(t % 7);
  f32x16b acc[4] = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = pad[t];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[j][k];
  out[blockIdx.x * 256 + t] = s;
}

__global__ __launch_bounds__(256) void aggressor_valu(float* __restrict__ out, int iters) {    // no matrix instructions
  __shared__ float pad[256];
  const int t = threadIdx.x;
  pad[t] = 0.f;
  float x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = 0.001f * (float)((t + k) % 13);
  const float m = 0.99999f, c = 1e-6f;
  for (int i = 0; i < iters * 8; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = __builtin_fmaf(x[k], m, c);
  }
  float s = pad[t];
#pragma unroll
  for (int k = 0; k < 16; ++k) s += x[k];
  out[blockIdx.x * 256 + t] = s;
}

__global__ __launch_bounds__(256) void aggressor(float* __restrict__ out, int iters) {
  __shared__ float pad[256];                            // 1 KB of LDS: cannot share a CU with a 160 KB victim group
  const int t = threadIdx.x;
  pad[t] = 0.f;
  half8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = (_Float16)(0.001f * (float)((t + k) % 13));
    b[k] = (_Float16)(0.002f * (float)((t * 3 + k) % 7));
  }
  f32x16 acc[4] = {};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
  }
  float s = pad[t];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[j][k];
  out[blockIdx.x * 256 + t] = s;
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200, viters = argc > 2 ? atoi(argv[2]) : 2000, excl = argc > 3 ? atoi(argv[3]) : 0, akind = argc > 4 ? atoi(argv[4]) : 0;
  const int vgrid = 1024, agrid = 512, NA = 3, NF = 4;
  const size_t nout = (size_t)vgrid * 512 * NV, lds = excl ? 160 * 1024 : 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  if (excl) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(victim), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t fs[NF], as[NA];
  f32x2 *ref, *got[NF];
  float* aout[NA];
  for (auto& s : fs) CK(hipStreamCreate(&s));
  for (auto& s : as) CK(hipStreamCreate(&s));
  CK(hipMalloc(&ref, nout * 8));
  for (auto& g : got) CK(hipMalloc(&g, nout * 8));
  for (auto& a : aout) CK(hipMalloc(&a, (size_t)agrid * 256 * 4));
  hipLaunchKernelGGL(victim, dim3(vgrid), dim3(512), lds, fs[0], ref, viters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> hr(nout * 2), hg(nout * 2);
  CK(hipMemcpy(hr.data(), ref, nout * 8, hipMemcpyDeviceToHost));
  int bad_runs = 0, runs = 0;
  for (int mode = 0; mode < 2; ++mode) {                // 0: the victim on four streams alone; 1: next to the aggressors
    int bad_mode = 0;
    for (int r = 0; r < rounds; ++r) {
      for (auto& g : got) CK(hipMemsetAsync(g, 0, nout * 8, 0));
      CK(hipDeviceSynchronize());
      for (int rep = 0; rep < 3; ++rep) {
        if (mode)
          for (int j = 0; j < NA; ++j) {
            if (akind == 0) hipLaunchKernelGGL(aggressor, dim3(agrid), dim3(256), 0, as[j], aout[j], 20000);
            else if (akind == 1) hipLaunchKernelGGL(aggressor_f32, dim3(agrid), dim3(256), 0, as[j], aout[j], 10000);
            else hipLaunchKernelGGL(aggressor_valu, dim3(agrid), dim3(256), 0, as[j], aout[j], 20000);
          }
        for (int i = 0; i < NF; ++i) hipLaunchKernelGGL(victim, dim3(vgrid), dim3(512), lds, fs[i], got[i], viters);
      }
      CK(hipDeviceSynchronize());
      for (int i = 0; i < NF; ++i) {
        CK(hipMemcpy(hg.data(), got[i], nout * 8, hipMemcpyDeviceToHost));
        ++runs;
        if (memcmp(hg.data(), hr.data(), nout * 8) == 0) continue;
        ++bad_mode;
        if (bad_mode <= 6) {                            // where: (work-group, wave, lane range, register word)
          size_t n = 0, first = 0, last = 0;
          for (size_t w = 0; w < nout * 2; ++w)
            if (hg[w] != hr[w]) {
              if (!n) first = w;
              last = w;
              ++n;
            }
          const size_t t0 = first / (2 * NV), t1 = last / (2 * NV);
          printf("  mode %d round %d stream %d: %zu words differ; first: thread %zu (group %zu wave %zu lane %zu) reg word %zu got %08x want %08x; "
                 "last: thread %zu (lane %zu) reg word %zu\n", mode, r, i, n, t0, t0 / 512, (t0 % 512) / 64, t0 % 64, first % (2 * NV),
                 hg[first], hr[first], t1, t1 % 64, last % (2 * NV));
        }
      }
    }
    static const char* anames[] = {"fp16-MFMA", "fp32-MFMA", "scalar-FMA VALU (no MFMA)"};
    printf("%s: victim %s%s: %d of %d runs differ (device %s, %d CUs, victim iters %d, %s)\n", argv[0],
           mode ? "next to aggressors: " : "alone on four streams", mode ? anames[akind] : "", bad_mode, rounds * NF, prop.gcnArchName,
           prop.multiProcessorCount, viters, excl ? "CU-exclusive victim" : "shared CUs");
    bad_runs += bad_mode;
  }
  return bad_runs ? 1 : 0;
}

// Dependent-launch boundary cost on this box (round 6): N back-to-back launches on one stream, wall clock / N, for
//   trivial kernels (256 work-groups, no memory), kernels that leave D MB dirty in the L2s (plain 16-byte stores), the same with
//   LDS-heavy work-groups (128 KB dynamic LDS, 512 threads: what the head's large kernels look like to the dispatcher).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/launch_gap tools/launch_gap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_trivial(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_dirty(u32x4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = u32x4{1u, 2u, 3u, 4u};
}
__global__ void k_lds(u32x4* p, size_t n) {
  extern __shared__ u32x4 sm[];
  sm[threadIdx.x] = u32x4{threadIdx.x, 0u, 0u, 0u};
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = sm[(threadIdx.x + 1) & 511];
}
template <class F> double run(F launch, int N, hipStream_t s) {
  for (int i = 0; i < 50; ++i) launch();
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) launch();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  u32x4* buf; hipMalloc(&buf, 512u << 20);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const int N = 2000;
  printf("trivial 256 x 256 threads        : %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, s, (int*)nullptr); }, N, s));
  printf("trivial 1024 x 512 threads       : %.2f us per launch\n", run([&] { hipLaunchKernelGGL(k_trivial, dim3(1024), dim3(512), 0, s, (int*)nullptr); }, N, s));
  for (int mb : {1, 8, 32, 128}) {
    const size_t n = (size_t)mb * (1u << 20) / 16;
    const double t = run([&] { hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, s, buf, n); }, N, s);
    printf("plain stores, %3d MB per launch   : %.2f us per launch (%.2f TB/s)\n", mb, t, mb * 1.048576e6 / t * 1e-6);
  }
  for (int mb : {8, 32}) {
    const size_t n = (size_t)mb * (1u << 20) / 16;
    const double t = run([&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 128 * 1024, s, buf, n); }, N, s);
    printf("128 KB LDS groups, %3d MB         : %.2f us per launch\n", mb, t);
  }
  // host launch cost alone: how fast can the host enqueue (no dependency wait visible: measure enqueue time of N launches)
  {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, s, buf, (size_t)(32u << 20) / 16);
    const double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    hipStreamSynchronize(s);
    printf("host enqueue time                 : %.2f us per launch\n", enq);
  }
  return 0;
}

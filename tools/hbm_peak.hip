// What does a plain stream sustain on this chip?  Calibration for the HBM fractions of DESIGN.md (peak 8 TB/s): a grid-stride kernel
// with 16-byte accesses over 4 GiB buffers - read only (sum), write only (fill), copy (read + write) - and the same copy in
// 128-byte runs scattered at a 7,424-byte stride (the run length and pitch of the spectra the transform kernels and the per-bin GEMM
// exchange).  Output: achieved TB/s per pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_peak.hip -o tools/bin/hbm_peak && tools/bin/hbm_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ a, float* __restrict__ out, size_t n) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ b, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// runs of 8 units (128 bytes) at a pitch of 464 units (7,424 bytes): unit i of the dense order -> run i / 8 of `lines` interleaved rows
__global__ __launch_bounds__(256) void k_copy_runs(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n, size_t pitch) {
  const size_t rows = n / pitch;                       // rows of `pitch` units; a pass takes run r of every row before run r + 1
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rows * (pitch / 8) * 8; i += (size_t)gridDim.x * 256) {
    const size_t run = i / 8, u = i - run * 8;
    const size_t row = run % rows, r = run / rows;
    const size_t idx = row * pitch + r * 8 + u;
    b[idx] = a[idx];
  }
}

template <class F>
double timeit(F launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 / reps;
}

int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  f32x4 *a, *b;
  float* out;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(a, 0, bytes);
  (void)hipMemset(b, 0, bytes);
  for (int grid : {256 * 8, 256 * 32}) {
    const double tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n); }, 10);
    const double tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); }, 10);
    const double tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10);
    const double ts = timeit([&] { hipLaunchKernelGGL(k_copy_runs, dim3(grid), dim3(256), 0, 0, a, b, n, (size_t)464); }, 10);
    printf("grid %5d x 256: read %.2f TB/s   write %.2f TB/s   copy %.2f TB/s (read + write)   copy in 128-byte runs at a 7424-byte pitch %.2f TB/s\n",
           grid, bytes / tr / 1e12, bytes / tw / 1e12, 2.0 * bytes / tc / 1e12, 2.0 * (n / 464) * 464 * 16 / ts / 1e12);
  }
  return 0;
}

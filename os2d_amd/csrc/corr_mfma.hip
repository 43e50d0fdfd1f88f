// All-pairs feature correlation (reference os2d/modeling/head.py:339-350) fused with the TransformNet input
// normalisation relu -> L2 over the 225 correlation channels (head.py:650, eps 1e-6), fp32 MFMA, gfx950.
//
//   corr[a,b,k=j*15+i,h,w] = sum_c q_hat[b,c,i,j] * f[a,c,h,w] / (||f[a,:,h,w]|| + 1e-5)
//
// GEMM view per (image a, class b):  C[m, n] = sum_c Qp[b][c][m] * F[a][c][n],  m = 225 rows padded to 256 with
// zero rows (prep.hip packs Qp = q_hat permuted to the x-major channel order), n = h*W+w, K = C.  Both operands
// are "k-outer": for a fixed c the 256 m (resp. the n) are contiguous, which is exactly the lane order of the
// v_mfma_f32_32x32x2_f32 A/B fragments, so LDS tiles are plain copies and fragment reads are conflict-free.
//
// Work-group = 256 threads (4 waves as 2x2), tile = 256 (all rows of one class) x NT=128 positions, so the
// whole 225-channel column of every position lives in one group and the channel L2-norm is a cross-wave LDS
// reduction in the epilogue.  Outputs:
//   corr  [A*B][225][H*W]      raw correlation (input of the resampling kernel)
//   rpad  [A*B][226][PLANE]    relu+L2-normalised, in the zero-bordered plane layout the conv kernels read
#include "os2d_common.h"

namespace {

constexpr int NT = 128;
constexpr int KC = 16;  // k-chunk staged per barrier

__global__ __launch_bounds__(256, 2) void corr_mfma_kernel(const float* __restrict__ fm,     // [A][C][HW]
                                                           const float* __restrict__ qp,     // [B][C][256]
                                                           const float* __restrict__ sumsq,  // [A][HW]
                                                           float* __restrict__ corr, float* __restrict__ rpad,
                                                           int B, int C, int H, int W, int PLANE) {
  __shared__ __attribute__((aligned(16))) float ldsA[KC * OS2D_QROWS];
  __shared__ __attribute__((aligned(16))) float ldsB[KC * NT];
  __shared__ float red[2][NT];

  const int HW = H * W;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;  // wave tile: rows [wm*128, +128), cols [wn*64, +64)
  const int n0 = blockIdx.x * NT;
  const int b = blockIdx.y, a = blockIdx.z;
  const int nb = a * B + b;

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const float* aBase = ldsA + hi * OS2D_QROWS + wm * 128 + l31;
  const float* bBase = ldsB + hi * NT + wn * 64 + l31;

  for (int k0 = 0; k0 < C; k0 += KC) {
    {  // A: KC consecutive channels of the packed class map = KC*256 contiguous floats
      const float4* src = reinterpret_cast<const float4*>(qp + ((size_t)b * C + k0) * OS2D_QROWS);
      float4* dst = reinterpret_cast<float4*>(ldsA);
      const int nvalid = (min(KC, C - k0) * OS2D_QROWS) >> 2;
      for (int i = tid; i < (KC * OS2D_QROWS) / 4; i += 256)
        dst[i] = i < nvalid ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {  // B: KC rows of NT positions (dword loads: H*W need not be a multiple of 4)
      for (int i = tid; i < KC * NT; i += 256) {
        const int kk = i / NT, j = i - kk * NT;
        const int n = n0 + j;
        float v = 0.f;
        if (k0 + kk < C && n < HW) v = fm[((size_t)a * C + k0 + kk) * HW + n];
        ldsB[i] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < KC / 2; ++p) {
      float av[4], bv[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) av[mi] = aBase[2 * p * OS2D_QROWS + mi * 32];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bv[ni] = bBase[2 * p * NT + ni * 32];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue
  const int Ws = os2d_ws(W), BASE = os2d_base(W);
  float part[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    const bool nin = n < HW;
    // image-feature normalisation folded in as a per-column scale (head.py:339, eps 1e-5)
    const float inv_f = nin ? 1.0f / (sqrtf(sumsq[(size_t)a * HW + n]) + 1e-5f) : 0.f;
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[mi][ni][r] * inv_f;
        acc[mi][ni][r] = v;
        const int m = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (nin && m < OS2D_K) corr[((size_t)nb * OS2D_K + m) * HW + n] = v;
        const float rl = fmaxf(v, 0.f);
        s += rl * rl;
      }
    s += __shfl_xor(s, 32);  // the other 16 rows of each 32x32 block live in lane^32
    part[ni] = s;
  }
  if (hi == 0) {
    red[wm][wn * 64 + l31] = part[0];
    red[wm][wn * 64 + 32 + l31] = part[1];
  }
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = wn * 64 + ni * 32 + l31;
    const int n = n0 + col;
    if (n >= HW) continue;
    const float inv_r = 1.0f / (sqrtf(red[0][col] + red[1][col]) + 1e-6f);  // head.py:650,597 (eps 1e-6)
    const int h = n / W, w = n - h * W;
    const size_t cell = (size_t)BASE + (size_t)h * Ws + w;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m < OS2D_K) rpad[((size_t)nb * OS2D_KP + m) * PLANE + cell] = fmaxf(acc[mi][ni][r], 0.f) * inv_r;
      }
  }
}

}  // namespace

int os2d_launch_corr(const float* fm, const float* qp, const float* sumsq, float* corr, float* rpad, int A, int B,
                     int C, int H, int W, hipStream_t stream) {
  const int HW = H * W;
  dim3 grid((HW + NT - 1) / NT, B, A);
  hipLaunchKernelGGL(corr_mfma_kernel, grid, dim3(256), 0, stream, fm, qp, sumsq, corr, rpad, B, C, H, W,
                     os2d_plane(H, W));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("corr launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

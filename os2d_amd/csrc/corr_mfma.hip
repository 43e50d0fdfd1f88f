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
// reduction in the epilogue.  K is consumed in chunks of KC=16 channels through double-buffered LDS with a
// register prefetch (same pipeline as conv_mfma.hip: loads of chunk t+1 are issued before the MFMAs of chunk t,
// written to the other buffer after them, one barrier per chunk).  Outputs:
//   corr  [A*B][225][H*W]      raw correlation (input of the resampling kernel)
//   rpad  [A*B][226][PLANE]    relu+L2-normalised, in the zero-bordered plane layout the conv kernels read
#include "os2d_common.h"

namespace {

constexpr int NT = 128;
constexpr int KC = 16;                        // k-chunk per pipeline stage
constexpr int A4 = KC * OS2D_QROWS / 4 / 256;  // float4 per thread of an A chunk (4)
constexpr int B4 = KC * NT / 4 / 256;          // float4 per thread of a B chunk (2)
constexpr int B1 = KC * NT / 256;              // dwords per thread of a B chunk (8)

template <bool VEC4, bool SHB>
__global__ __launch_bounds__(256, 2) void corr_mfma_kernel(const float* __restrict__ fm,     // [A][C][HW]
                                                           const float* __restrict__ qp,     // [B][C][256]
                                                           const float* __restrict__ sumsq,  // [A][HW]
                                                           float* __restrict__ corr, float* __restrict__ rpad /*or NULL*/,
                                                           float* __restrict__ invn /*[A*B][HW] 1/(norm+eps), or NULL*/,
                                                           int B, int C, int H, int W, int PLANE) {
  __shared__ __attribute__((aligned(16))) float ldsA[2][KC * OS2D_QROWS];
  __shared__ __attribute__((aligned(16))) float ldsB[2][KC * NT];
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

  const int aOff = hi * OS2D_QROWS + wm * 128 + l31;
  const int bOff = hi * NT + wn * 64 + l31;
  const float* qb = qp + (size_t)b * C * OS2D_QROWS;
  const float* fa = fm + (size_t)a * C * HW;
  const int nchunks = (C + KC - 1) / KC;

  f32x4 pfA[A4];
  f32x4 pfB4[B4];
  float pfB1[B1];

  // Unconditional, clamped prefetch loads (a load under a divergent branch would be waited for at the branch end);
  // rows past C and columns past H*W are zeroed when the registers are written to LDS.
#define CORR_LOAD(T)                                                                                              \
  {                                                                                                               \
    const int k0_ = (T)*KC;                                                                                       \
    _Pragma("unroll") for (int k = 0; k < A4; ++k) {                                                              \
      const int i_ = tid + k * 256; /* float4 index: row = i_/64 */                                               \
      const int row_ = min(k0_ + (i_ >> 6), C - 1);                                                               \
      pfA[k] = *reinterpret_cast<const f32x4*>(qb + (size_t)row_ * OS2D_QROWS + ((i_ & 63) << 2));                \
    }                                                                                                             \
    if (VEC4) {                                                                                                   \
      _Pragma("unroll") for (int k = 0; k < B4; ++k) {                                                            \
        const int i_ = tid + k * 256; /* float4 index: row = i_/32 */                                             \
        const int row_ = min(k0_ + (i_ >> 5), C - 1);                                                             \
        const int n_ = min(n0 + ((i_ & 31) << 2), HW - 4);                                                        \
        pfB4[k] = *reinterpret_cast<const f32x4*>(fa + (size_t)row_ * HW + n_);                                   \
      }                                                                                                           \
    } else {                                                                                                      \
      _Pragma("unroll") for (int k = 0; k < B1; ++k) {                                                            \
        const int i_ = tid + k * 256; /* dword index: row = i_/128 */                                             \
        const int row_ = min(k0_ + (i_ >> 7), C - 1);                                                             \
        const int n_ = min(n0 + (i_ & 127), HW - 1);                                                              \
        pfB1[k] = fa[(size_t)row_ * HW + n_];                                                                     \
      }                                                                                                           \
    }                                                                                                             \
  }
#define CORR_STORE(T)                                                                                             \
  {                                                                                                               \
    const int k0_ = (T)*KC;                                                                                       \
    const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                                        \
    _Pragma("unroll") for (int k = 0; k < A4; ++k) {                                                              \
      const int i_ = tid + k * 256;                                                                               \
      reinterpret_cast<f32x4*>(ldsA[(T)&1])[i_] = (k0_ + (i_ >> 6) < C) ? pfA[k] : z_;                            \
    }                                                                                                             \
    if (VEC4) {                                                                                                   \
      _Pragma("unroll") for (int k = 0; k < B4; ++k) {                                                            \
        const int i_ = tid + k * 256;                                                                             \
        const bool ok_ = (k0_ + (i_ >> 5) < C) && (n0 + ((i_ & 31) << 2) < HW); /* HW % 4 == 0 here */            \
        reinterpret_cast<f32x4*>(ldsB[(T)&1])[i_] = ok_ ? pfB4[k] : z_;                                           \
      }                                                                                                           \
    } else {                                                                                                      \
      _Pragma("unroll") for (int k = 0; k < B1; ++k) {                                                            \
        const int i_ = tid + k * 256;                                                                             \
        const bool ok_ = (k0_ + (i_ >> 7) < C) && (n0 + (i_ & 127) < HW);                                         \
        ldsB[(T)&1][i_] = ok_ ? pfB1[k] : 0.f;                                                                    \
      }                                                                                                           \
    }                                                                                                             \
  }

  CORR_LOAD(0)
  CORR_STORE(0)
  __syncthreads();
#define CORR_COMPUTE(T)                                                                                           \
  {                                                                                                               \
    const float* aBase_ = ldsA[(T)&1] + aOff;                                                                     \
    const float* bBase_ = ldsB[(T)&1] + bOff;                                                                     \
    _Pragma("unroll") for (int p = 0; p < KC / 2; ++p) {                                                          \
      float av_[4], bv_[2];                                                                                       \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) av_[mi] = aBase_[2 * p * OS2D_QROWS + mi * 32];            \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) bv_[ni] = bBase_[2 * p * NT + ni * 32];                    \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                            \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                          \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[mi], bv_[ni], acc[mi][ni], 0, 0, 0);             \
    }                                                                                                             \
  }
  // The last chunk is peeled so that load -> MFMA -> store is branch-free inside the loop: with the store under
  // "if (t+1 < nchunks)" hipcc sinks the (speculatable) prefetch loads into that branch, i.e. behind the MFMAs.
  for (int t = 0; t + 1 < nchunks; ++t) {
    CORR_LOAD(t + 1)
    __builtin_amdgcn_sched_barrier(0);  // and keep the scheduler from moving them below the MFMAs
    CORR_COMPUTE(t)
    __builtin_amdgcn_sched_barrier(0);
    CORR_STORE(t + 1)
    __syncthreads();
  }
  CORR_COMPUTE(nchunks - 1)
  __syncthreads();  // red[] / LDS reuse below
#undef CORR_COMPUTE
#undef CORR_LOAD
#undef CORR_STORE

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
        const float rl = os2d_relu(v);
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
    if (invn != nullptr && wm == 0 && hi == 0) invn[(size_t)nb * HW + n] = inv_r;   // for the frequency-domain 7x7 layer
    if (rpad == nullptr) continue;                                                   // ... which reads corr + invn only
    const int h = n / W, w = n - h * W;
    const size_t cell = (size_t)BASE + (size_t)h * Ws + w;
    if (!SHB) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (m < OS2D_K) rpad[((size_t)nb * OS2D_KP + m) * PLANE + cell] = os2d_relu(acc[mi][ni][r]) * inv_r;
        }
    } else {
      // split-half blocked output for conv_f16x3.hip: [nb][29 groups][hi|lo][PLANE] units of 8 halves.  Registers
      // 4q..4q+3 of a lane are rows m0..m0+3 = channels 4*hi..4*hi+3 of group m0/8: one 8-byte store each for hi, lo.
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      char* base = reinterpret_cast<char*>(rpad);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m0 = wm * 128 + mi * 32 + 8 * q + 4 * hi;
          const int grp = m0 >> 3;
          if (grp >= OS2D_G) continue;
          half4 h4, l4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = (m0 + k < OS2D_K) ? os2d_relu(acc[mi][ni][4 * q + k]) * inv_r : 0.f;
            const _Float16 hv = (_Float16)v;
            h4[k] = hv;
            l4[k] = (_Float16)(v - (float)hv);
          }
          char* o = base + (((size_t)nb * OS2D_G + grp) * 2 * PLANE + cell) * 16 + hi * 8;
          *reinterpret_cast<half4*>(o) = h4;
          *reinterpret_cast<half4*>(o + (size_t)PLANE * 16) = l4;
        }
    }
  }
}

}  // namespace

int os2d_launch_corr(const float* fm, const float* qp, const float* sumsq, float* corr, void* rnorm, float* invn, int A, int B,
                     int C, int H, int W, int shb, hipStream_t stream) {
  const int HW = H * W;
  dim3 grid((HW + NT - 1) / NT, B, A);
  // float4 loads of the image map need 16-byte aligned rows: H*W % 4 == 0 (and a 16-byte aligned base)
  const bool vec4 = (HW % 4 == 0) && HW >= 4 && ((reinterpret_cast<uintptr_t>(fm) & 15) == 0);
  float* rp = reinterpret_cast<float*>(rnorm);
  const int PL = os2d_plane(H, W);
#define CORR_GO(V, S) \
  hipLaunchKernelGGL((corr_mfma_kernel<V, S>), grid, dim3(256), 0, stream, fm, qp, sumsq, corr, rp, invn, B, C, H, W, PL)
  if (vec4 && shb) CORR_GO(true, true);
  else if (vec4) CORR_GO(true, false);
  else if (shb) CORR_GO(false, true);
  else CORR_GO(false, false);
#undef CORR_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("corr launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

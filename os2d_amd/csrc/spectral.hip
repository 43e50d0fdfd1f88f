// The 7x7 TransformNet layer in the frequency domain (gfx950) - building blocks, round 2 (docs/DESIGN_HISTORY_r1-r3.md section 8, "next").
//
// A zero-padded 7x7 convolution of a 60x80 map is a pointwise product of 72x96 spectra: for every frequency bin the
// layer (reference os2d/modeling/head.py:619-623: Conv 225 -> 128 k7 + BatchNorm) is one small COMPLEX matrix product
//       Y[o][b] = sum_c K[o][c] * X[c][b]          o = 128 output channels, c = 225 input channels, b = classes
// with the weight spectra K shared by all classes - 16.7x fewer multiply-adds than the direct form, and the fp32 result is
// closer to an fp64 convolution than an fp32 direct convolution is (tools/proto_fft_conv.py).
//
// spectral_gemm_kernel: all bins as one launch on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products,
// fp32 accumulation; one instruction = the two real products of one complex multiply-add: k = {re, im}):
//       Yr += [Kr | -Ki] . [Xr ; Xi]        Yi += [Ki | Kr] . [Xr ; Xi]
// Work-group = 512 threads = 8 waves = 8 consecutive bins x 64 output channels x 64 classes; wave w owns bin w
// (accumulators 2x2 blocks for Yr and Yi = 128 registers).  K runs over the 225 channels in chunks of 9 through
// double-buffered LDS: the weights (packed [bin group][o half][c][bin][o], 36 KB per chunk, contiguous) arrive by LDS-DMA,
// the spectra (natural layout [class][c][bin]: 64-byte pieces) through registers with the transposition to [c][bin][class]
// on the way, so every fragment read is a conflict-free ds_read_b64 of 64 consecutive complex numbers.
#include "os2d_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_BINS = 8;     // bins per group of the packed weight layout
constexpr int SG_CC = 9;       // channels per K chunk (225 = 25 * 9)
constexpr int SG_OH = 64;      // output channels per work-group (half of 128)
constexpr int SG_NB = 64;      // classes per work-group
#ifndef OS2D_SG_WB
#define OS2D_SG_WB 4
#endif
// WB = bins (= waves) per work-group.  8: one work-group of 512 threads per CU (144 KB of LDS).  4: two independent
// work-groups of 256 threads per CU (72 KB each, the two halves of a bin group): while one waits at its chunk barrier for
// the next operands, the other keeps the matrix pipes busy.
constexpr int SG_WB = OS2D_SG_WB;
constexpr int SG_THR = SG_WB * 64;
constexpr int SG_NBH = SG_BINS / SG_WB;                  // work-groups per bin group
constexpr int SG_WUNITS = SG_CC * SG_WB * SG_OH / 2;     // 16-byte units of a weight chunk (2 complex each)
constexpr int SG_XUNITS = SG_CC * SG_NB * SG_WB / 2;     // ... of a spectra chunk
constexpr int SG_XPF = (SG_XUNITS + SG_THR - 1) / SG_THR;  // 5 units per thread
constexpr int SG_SLOTS = 256 * SG_NBH;                   // work-groups resident on the chip at a time

// NBLK = 2: a work-group owns a whole unit (8 bins x 64 output channels x 64 classes).  NBLK = 1: it owns one 32 x 32
// QUARTER of a unit (same staging, a quarter of the matrix instructions) - used for the units left over after the last
// full round of 256 work-groups, so that the tail of the launch costs a quarter of a round instead of a whole one.
template <int NBLK>
__global__ __launch_bounds__(SG_THR, 2) void spectral_gemm_kernel(const f32x2* wspec,  // [G][2][C][8][64] complex
                                                                  const f32x2* __restrict__ X,  // [C][NB][NBINS]
                                                                  f32x2* __restrict__ Y,        // [NB][Cout][NBINS]
                                                                  int NB, int C, int Cout, int NBINS, int G, int unit0,
                                                                  int nunits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x2* ldsW = reinterpret_cast<f32x2*>(smem);                                      // [2][CC][8][64]
  f32x2* ldsX = reinterpret_cast<f32x2*>(smem + 2 * SG_WUNITS * 16);                 // [2][CC][8][64]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hw = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order (work-group L runs on XCD L % 8; each XCD has its own L2): every XCD takes a contiguous range of
  // logical indices = (bin group, class tile, o half, part of the bin group), last fastest.  The work-groups that share
  // cache lines run on one XCD at about the same time: the parts / halves of a bin group share the input spectra, the
  // class tiles of a bin group share the weight spectra (which therefore leave HBM once, not once per class tile)
  const int per = gridDim.x >> 3;
  const int lidx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  constexpr int SUB = NBLK == 2 ? 1 : 4;                 // work-groups per unit
  if (lidx >= nunits * SUB) return;
  const int logical = unit0 + lidx / SUB, sub = lidx % SUB;
  const int oq = NBLK == 2 ? 0 : (sub & 1), bq = NBLK == 2 ? 0 : (sub >> 1);   // quarter: o block / class block
  const int bh = logical % SG_NBH, lg = logical / SG_NBH;                      // part of the bin group
  const int nbt = (NB + SG_NB - 1) / SG_NB;
  const int half = lg & 1, bt = (lg >> 1) % nbt, g = (lg >> 1) / nbt;
  const int nb0 = bt * SG_NB, bin0 = g * SG_BINS + bh * SG_WB;
  const int nchunks = (C + SG_CC - 1) / SG_CC;

  f32x16 yr[NBLK][NBLK], yi[NBLK][NBLK];
#pragma unroll
  for (int a = 0; a < NBLK; ++a)
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        yr[a][b][r] = 0.f;
        yi[a][b][r] = 0.f;
      }

  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  const char* wbase = reinterpret_cast<const char*>(wspec) + (((size_t)(g * 2 + half) * C) * SG_BINS + bh * SG_WB) * SG_OH * 8;
  u32x4 pfx[SG_XPF];

  // weights of chunk t: WB * 512 contiguous bytes per channel, 1 KB per wave instruction straight into LDS
#define SG_DMA_W(T)                                                                                               \
  {                                                                                                               \
    const int c0_ = (T)*SG_CC;                                                                                    \
    const int units_ = min(SG_CC, C - c0_) * SG_WB * SG_OH / 2;                                                   \
    const char* src_ = wbase + (size_t)c0_ * SG_BINS * SG_OH * 8;                                                 \
    for (int u_ = wv * 64; u_ < units_; u_ += SG_THR) {                                                           \
      const int cl_ = u_ / (SG_WB * 32), r_ = u_ % (SG_WB * 32);                                                  \
      __builtin_amdgcn_global_load_lds((gptr_t)(src_ + ((size_t)cl_ * SG_BINS * 32 + r_ + lane) * 16),            \
                                       (lptr_t)(reinterpret_cast<char*>(ldsW) + (((T)&1) * SG_WUNITS + u_) * 16), \
                                       16, 0, 0);                                                                 \
    }                                                                                                             \
  }
  // spectra of chunk t: unit u = (c, class, bin pair); out-of-range classes / channels read a valid address and are
  // zeroed when stored to LDS
#define SG_LOAD_X(T)                                                                                              \
  {                                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < SG_XPF; ++k_) {                                                       \
      const int u_ = min(tid + k_ * SG_THR, SG_XUNITS - 1);                                                       \
      const int c_ = min((T)*SG_CC + u_ / (SG_NB * (SG_WB / 2)), C - 1);                                          \
      const int b_ = min(nb0 + (u_ / (SG_WB / 2)) % SG_NB, NB - 1);                                               \
      pfx[k_] = *reinterpret_cast<const u32x4*>(X + ((size_t)c_ * NB + b_) * NBINS + bin0 + 2 * (u_ % (SG_WB / 2))); \
    }                                                                                                             \
  }
#define SG_STORE_X(T)                                                                                             \
  {                                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < SG_XPF; ++k_) {                                                       \
      const int u_ = tid + k_ * SG_THR;                                                                           \
      if (u_ < SG_XUNITS) {                                                                                       \
        const int cl_ = u_ / (SG_NB * (SG_WB / 2)), bl_ = (u_ / (SG_WB / 2)) % SG_NB, q_ = u_ % (SG_WB / 2);      \
        const bool ok_ = (T)*SG_CC + cl_ < C && nb0 + bl_ < NB;                                                   \
        f32x2 e0_ = {__uint_as_float(pfx[k_][0]), __uint_as_float(pfx[k_][1])};                                   \
        f32x2 e1_ = {__uint_as_float(pfx[k_][2]), __uint_as_float(pfx[k_][3])};                                   \
        if (!ok_) e0_ = e1_ = f32x2{0.f, 0.f};                                                                    \
        f32x2* dst_ = ldsX + ((T)&1) * (SG_XUNITS * 2) + (cl_ * SG_WB + 2 * q_) * SG_NB + bl_;                    \
        dst_[0] = e0_;                                                                                            \
        dst_[SG_NB] = e1_;                                                                                        \
      }                                                                                                           \
    }                                                                                                             \
  }

  SG_DMA_W(0)
  SG_LOAD_X(0)
  SG_STORE_X(0)
  __syncthreads();
  for (int t = 0; t < nchunks; ++t) {
    const bool more = t + 1 < nchunks;
    if (more) {
      SG_DMA_W(t + 1)
      SG_LOAD_X(t + 1)
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x2* wB = ldsW + (t & 1) * (SG_WUNITS * 2) + wv * SG_OH + oq * 32 + l31;      // [c][bin = wv][o]
    const f32x2* xB = ldsX + (t & 1) * (SG_XUNITS * 2) + wv * SG_NB + bq * 32 + l31;      // [c][bin = wv][class]
    const int cc = min(SG_CC, C - t * SG_CC);
    // fragments of channel c+1 are read while the 8 matrix instructions of channel c issue (two register sets)
    f32x2 kf[2][NBLK], xf[2][NBLK];
#pragma unroll
    for (int a = 0; a < NBLK; ++a) kf[0][a] = wB[a * 32];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) xf[0][b] = xB[b * 32];
#pragma unroll
    for (int c = 0; c < SG_CC; ++c) {
      if (c < cc) {
        const int cur = c & 1, nxt = cur ^ 1;
        if (c + 1 < cc) {
#pragma unroll
          for (int a = 0; a < NBLK; ++a) kf[nxt][a] = wB[(c + 1) * SG_WB * SG_OH + a * 32];
#pragma unroll
          for (int b = 0; b < NBLK; ++b) xf[nxt][b] = xB[(c + 1) * SG_WB * SG_NB + b * 32];
        }
#pragma unroll
        for (int a = 0; a < NBLK; ++a) {
          const float ar = hw ? -kf[cur][a][1] : kf[cur][a][0];   // row of [Kr | -Ki]
          const float ai = hw ? kf[cur][a][0] : kf[cur][a][1];    // row of [Ki |  Kr]
#pragma unroll
          for (int b = 0; b < NBLK; ++b) {
            const float xb = hw ? xf[cur][b][1] : xf[cur][b][0];  // column of [Xr ; Xi]
            // the spectra are the ROW operand: accumulator registers run over the pairs, lanes over the output channels, so a
            // store instruction stays inside one pair's rows (22 KB apart) instead of touching 32 pairs (3 MB apart)
            yr[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb, ar, yr[a][b], 0, 0, 0);
            yi[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb, ai, yi[a][b], 0, 0, 0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) SG_STORE_X(t + 1)
    __syncthreads();
  }
#undef SG_DMA_W
#undef SG_LOAD_X
#undef SG_STORE_X

  // ---- epilogue: Y[pair][o][bin]
  const int bin = bin0 + wv;
#pragma unroll
  for (int a = 0; a < NBLK; ++a) {
    const int o = half * SG_OH + (oq + a) * 32 + l31;
    if (o >= Cout) continue;
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nb = nb0 + (bq + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hw;
        if (nb < NB) {
          const float vr = yr[a][b][r], vi = yi[a][b][r];   // (copies: see the ext-vector element note in corr_f16x3.hip)
          Y[((size_t)nb * Cout + o) * NBINS + bin] = f32x2{vr, vi};
        }
      }
  }
}

}  // namespace

size_t os2d_spectral_weight_floats(int C, int Cout, int NBINS) {
  return (size_t)(NBINS / SG_BINS) * 2 * C * SG_BINS * SG_OH * 2;
}

int os2d_launch_spectral_gemm(const float* wspec, const float* X, float* Y, int NB, int C, int Cout, int NBINS,
                              hipStream_t stream) {
  if (NBINS % SG_BINS || Cout > 2 * SG_OH) {
    os2d_set_error("spectral_gemm: NBINS %d must be a multiple of %d and Cout %d <= %d", NBINS, SG_BINS, Cout, 2 * SG_OH);
    return -3;
  }
  const int G = NBINS / SG_BINS, nbt = (NB + SG_NB - 1) / SG_NB;
  const size_t lds = (size_t)(2 * SG_WUNITS + 2 * SG_XUNITS) * 16;
  for (const void* k : {reinterpret_cast<const void*>(spectral_gemm_kernel<2>), reinterpret_cast<const void*>(spectral_gemm_kernel<1>)}) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      os2d_set_error("hipFuncSetAttribute(spectral_gemm): %s", hipGetErrorString(e));
      return -4;
    }
  }
  // units = (class tile, bin group, o half, part of the bin group); 8 / WB units fill a CU.  Whole rounds of the resident
  // work-groups go to the full-tile kernel; what is left (2 % of the work at 64 classes, which would cost a whole extra
  // round) is cut into 32 x 32 quarters so that the tail takes a quarter of a round.
  const long long units = 2LL * G * nbt * SG_NBH;
  // a small remainder (up to an eighth of a round) goes to the quarter-tile kernel; a larger one is cheaper as one more
  // partly filled round of full tiles (quarter tiles load the operands of a whole unit for a quarter of its arithmetic)
  long long main_units = units >= SG_SLOTS ? units / SG_SLOTS * SG_SLOTS : 0;
  if (units - main_units > SG_SLOTS / 8) main_units = units;
  const f32x2* w = reinterpret_cast<const f32x2*>(wspec);
  const f32x2* x = reinterpret_cast<const f32x2*>(X);
  f32x2* y = reinterpret_cast<f32x2*>(Y);
  if (main_units > 0) {
    dim3 grid((unsigned)((main_units + 7) / 8 * 8));
    hipLaunchKernelGGL(spectral_gemm_kernel<2>, grid, dim3(SG_THR), lds, stream, w, x, y, NB, C, Cout, NBINS, G, 0, (int)main_units);
  }
  if (units > main_units) {
    const long long tail = units - main_units;
    dim3 grid((unsigned)((tail * 4 + 7) / 8 * 8));
    hipLaunchKernelGGL(spectral_gemm_kernel<1>, grid, dim3(SG_THR), lds, stream, w, x, y, NB, C, Cout, NBINS, G, (int)main_units,
                       (int)tail);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("spectral_gemm launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

// TransformNet convolutions (reference os2d/modeling/head.py:612-629,654) as fp32 MFMA implicit GEMM
// for gfx950.  One kernel template serves conv 7x7 225->128 (+BN+ReLU), conv 5x5 128->64 (+BN+ReLU) and
// conv 5x5 64->P; BatchNorm is folded into the packed weights (prep.hip).
//
// GEMM view per class plane:  OUT[o, n] = sum_{kk} Wp[kk, o] * IN[kk, n],  kk = (channel, dy, dx),
// n = flat index in the zero-bordered plane (os2d_common.h).  v_mfma_f32_32x32x2_f32 consumes two k per
// instruction: lanes 0-31 hold k, lanes 32-63 hold k+1.  We pair CHANNELS (2cp, 2cp+1) at the same tap so
// that both half-waves use the same tap shift: the per-lane LDS addresses are fixed for a whole channel
// pair and every tap / 32-column block is a compile-time immediate (plus one add per kernel row dy).
//
// Work-group = 256 threads (4 waves), tile = MT output channels x NT=256 plane positions starting at the first
// data cell (tiles cover exactly the H*WS data rows; the first / last tile also zero the pad rows above / below).
//
// Software pipeline (one barrier per stage, global latency hidden behind the MFMAs of the previous stage):
//   stage        = one channel pair x RS kernel rows  (conv1: RS=1 -> 7 taps x 8 MFMA = 3.6k matrix cycles per wave)
//   LDS (x2 buf) = A slab [RS*KS][2][MT] packed weights (contiguous in HBM/L2)  +  B slab [2][NT+2*HALO] input rows
//   iteration s  : issue the global loads of stage s+1 into registers -> MFMAs of stage s from buffer s&1 ->
//                  write the registers into buffer (s+1)&1 -> barrier.
// The B slab only changes with the channel pair, so it is prefetched with the last row-stage of the previous pair.
// conv1 needs 2*(7 KB + 6 KB) = 27 KB of LDS per group, so occupancy is set by registers (2 waves / SIMD).
// Arithmetic is exact fp32 (the MFMA is a k-ordered fmaf chain).
#include "os2d_common.h"

namespace {

constexpr int NBPF = 3;  // max float4 per thread for the B-slab prefetch (SLAB <= 1536 floats)

// STRIP: maps wider than the linear slab takes (SLAB <= 1536 floats) are cut into column strips - the strip-plane geometry of
// conv_f16x3.hip (os2d_strip_cell): scalar slab loads through the index translation, the matrix loop unchanged with the strip
// pitch SP for the row pitch.
__device__ __forceinline__ int conv_strip_cell(int np, int SP, int c0mR, int H, int W, int Ws, int BASE) {
  const int h = np / SP, c = c0mR + (np - h * SP);
  return (np >= 0 && h < H && c >= 0 && c < W) ? BASE + h * Ws + c : 0;
}

template <int KS, int RS, int MT, int WM, int WN, int NT, bool RELU, bool COMPACT, bool STRIP = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const float* __restrict__ in,   // [NB][CinP][PLANE]
                                                           const float* __restrict__ wp,   // [CinP/2][KS*KS][2][MT]
                                                           const float* __restrict__ bp,   // [MT]
                                                           float* __restrict__ out, int CinP, int CoutStore,
                                                           int H, int W, int PLANE, int HALO, int TILES, int NB,
                                                           int SPITCH /*STRIP: row pitch of a strip-plane*/, int TPS /*STRIP: tiles per strip*/) {
  constexpr int R = KS / 2;
  constexpr int SP = KS / RS;  // stages per channel pair
  static_assert(SP * RS == KS, "RS must divide KS");
  constexpr int MW = MT / WM, NW = NT / WN;
  constexpr int MI = MW / 32, NI = NW / 32;
  static_assert(WM * WN == 4, "4 waves per work-group");
  static_assert(MW % 32 == 0 && NW % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA block");
  constexpr int ASTAGE = RS * KS * 2 * MT;          // floats of one A stage slab
  constexpr int NAPF = (ASTAGE / 4 + 255) / 256;    // float4 per thread for the A prefetch

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Ws = W + OS2D_PAD;
  const int BASE = os2d_base(W);
  const int DATA = H * Ws;  // flat extent of the data rows
  const int SLAB = NT + 2 * HALO;
  float* ldsA = smem;                 // [2][ASTAGE]
  float* ldsB = smem + 2 * ASTAGE;    // [2][2*SLAB]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  // XCD-aware work mapping (see conv_f16x3.hip): XCD x = work-groups L with L % 8 == x gets a contiguous range of the
  // logical order (plane, tile), so the tiles of one plane - whose input slabs overlap by 2/3 - share one L2
  const int per = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (logical >= TILES * NB) return;
  const int tile = logical % TILES;
  const int nb = logical / TILES;
  const int PW = STRIP ? SPITCH : Ws;                            // row pitch of the cells in the LDS slab
  const int strip = STRIP ? tile / TPS : 0;
  const int c0mR = STRIP ? strip * (SPITCH - 2 * R) - R : 0;     // map column of strip-plane column 0
  const int n0 = STRIP ? (tile - strip * TPS) * NT : BASE + tile * NT;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int aOff = hi * MT + wm * MW + l31;
  const int bOff = hi * SLAB + wn * NW + l31 + HALO - R * PW - R;
  const int npairs = CinP >> 1;
  const int nstages = npairs * SP;
  const int q4 = SLAB >> 2;  // float4 per channel of the B slab

  f32x4 pfA[NAPF], pfB[NBPF];
  const float* inb = in + (size_t)nb * CinP * PLANE;

  // NOTE: the prefetch loads are UNCONDITIONAL (indices clamped, out-of-plane reads redirected to offset 0 and
  // zeroed at store time).  A load under a divergent branch makes hipcc wait for it at the end of the branch
  // (s_waitcnt vmcnt(0) before the MFMAs), which would serialise the very latency this pipeline hides.
  // (Written as macros, not lambdas: by-reference capture of the register arrays sent them to scratch memory.)
#define OS2D_LOAD_STAGE(S)                                                                                         \
  {                                                                                                                \
    const f32x4* src_ = reinterpret_cast<const f32x4*>(wp + (size_t)(S)*ASTAGE);                                  \
    _Pragma("unroll") for (int k = 0; k < NAPF; ++k) pfA[k] = src_[min(tid + k * 256, ASTAGE / 4 - 1)];           \
    if ((S) % SP == 0) {                                                                                           \
      const int cp_ = (S) / SP;                                                                                    \
      _Pragma("unroll") for (int k = 0; k < NBPF; ++k) {                                                          \
        const int i_ = min(tid + k * 256, 2 * q4 - 1);                                                             \
        const int h2_ = i_ >= q4 ? 1 : 0;                                                                          \
        int g_ = n0 - HALO + ((i_ - h2_ * q4) << 2);                                                               \
        if (STRIP) {                                                                                               \
          const float* src_b_ = inb + (size_t)(2 * cp_ + h2_) * PLANE;                                             \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) pfB[k][e] = src_b_[conv_strip_cell(g_ + e, SPITCH, c0mR, H, W, Ws, BASE)]; \
        } else {                                                                                                   \
          g_ = (g_ >= 0 && g_ < PLANE) ? g_ : 0;                                                                   \
          pfB[k] = *reinterpret_cast<const f32x4*>(inb + (size_t)(2 * cp_ + h2_) * PLANE + g_);                    \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }
#define OS2D_STORE_STAGE(S)                                                                                        \
  {                                                                                                                \
    f32x4* dstA_ = reinterpret_cast<f32x4*>(ldsA + ((S)&1) * ASTAGE);                                              \
    _Pragma("unroll") for (int k = 0; k < NAPF; ++k) {                                                            \
      const int i_ = tid + k * 256;                                                                                \
      if (i_ < ASTAGE / 4) dstA_[i_] = pfA[k];                                                                     \
    }                                                                                                              \
    if ((S) % SP == 0) {                                                                                           \
      f32x4* dstB_ = reinterpret_cast<f32x4*>(ldsB + (((S) / SP) & 1) * 2 * SLAB);                                 \
      _Pragma("unroll") for (int k = 0; k < NBPF; ++k) {                                                          \
        const int i_ = tid + k * 256;                                                                              \
        if (i_ < 2 * q4) {                                                                                         \
          const int h2_ = i_ >= q4 ? 1 : 0;                                                                        \
          const int g_ = n0 - HALO + ((i_ - h2_ * q4) << 2);                                                       \
          const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                                   \
          if (STRIP) {                                                                                             \
            f32x4 v_ = pfB[k];                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                          \
              if (conv_strip_cell(g_ + e, SPITCH, c0mR, H, W, Ws, BASE) == 0) v_[e] = 0.f;                         \
            dstB_[i_] = v_;                                                                                        \
          } else {                                                                                                 \
            dstB_[i_] = (g_ >= 0 && g_ < PLANE) ? pfB[k] : z_;                                                     \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }

  OS2D_LOAD_STAGE(0)
  OS2D_STORE_STAGE(0)
  __syncthreads();
#define OS2D_COMPUTE_STAGE(S)                                                                                      \
  {                                                                                                                \
    const int cp_ = (S) / SP, rg_ = (S)-cp_ * SP; /* row group of this stage */                                    \
    const float* aBase_ = ldsA + ((S)&1) * ASTAGE + aOff;                                                          \
    const float* bBase_ = ldsB + (cp_ & 1) * 2 * SLAB + bOff + rg_ * RS * PW;                                      \
    _Pragma("unroll") for (int ry = 0; ry < RS; ++ry) {                                                           \
      const float* bRow_ = bBase_ + ry * PW;                                                                       \
      _Pragma("unroll") for (int dx = 0; dx < KS; ++dx) {                                                         \
        float a_[MI], b_[NI];                                                                                      \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) a_[mi] = aBase_[(ry * KS + dx) * 2 * MT + mi * 32];     \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) b_[ni] = bRow_[dx + ni * 32];                           \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                         \
          _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                       \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[mi], b_[ni], acc[mi][ni], 0, 0, 0);              \
      }                                                                                                            \
    }                                                                                                              \
  }
  // The last stage is peeled so that load -> MFMA -> store is branch-free inside the loop: with the store under
  // "if (s+1 < nstages)" hipcc may sink the (speculatable) prefetch loads into that branch, behind the MFMAs.
  for (int s = 0; s + 1 < nstages; ++s) {
    OS2D_LOAD_STAGE(s + 1)
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads ahead of the MFMAs
    OS2D_COMPUTE_STAGE(s)
    __builtin_amdgcn_sched_barrier(0);
    OS2D_STORE_STAGE(s + 1)
    __syncthreads();
  }
  OS2D_COMPUTE_STAGE(nstages - 1)
#undef OS2D_LOAD_STAGE
#undef OS2D_STORE_STAGE
#undef OS2D_COMPUTE_STAGE

  // ---- epilogue: bias (+ReLU); pad cells of a plane-layout output are written as exact zeros
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int n = n0 + wn * NW + ni * 32 + l31;
    int hr, wc;
    if (STRIP) {       // strip-plane index -> map cell; only the strip's own output columns (incl. the map's pad columns) are stored
      hr = n / SPITCH;
      const int j = n - hr * SPITCH;
      wc = c0mR + j;
      if (j < R || j >= SPITCH - R || hr >= H || wc >= Ws) continue;
      n = BASE + hr * Ws + wc;
    } else {
      const int r = n - BASE;
      hr = r / Ws;
      wc = r - hr * Ws;
    }
    const bool valid = hr < H && wc < W;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int m = wm * MW + mi * 32 + (k & 3) + 8 * (k >> 2) + 4 * hi;
        if (m >= CoutStore) continue;
        float v = acc[mi][ni][k] + bp[m];
        if (RELU) v = os2d_relu(v);
        if (COMPACT) {
          if (valid) out[((size_t)nb * CoutStore + m) * (H * W) + hr * W + wc] = v;
        } else if (n < PLANE) {
          out[((size_t)nb * CoutStore + m) * PLANE + n] = valid ? v : 0.f;
        }
      }
    }
  }
  if (!COMPACT) {
    // pad rows above the data (first tile) and whatever lies beyond the last tile
    if (tile == 0)
      for (int i = tid; i < CoutStore * BASE; i += 256) out[((size_t)nb * CoutStore + i / BASE) * PLANE + i % BASE] = 0.f;
    if (tile == TILES - 1) {
      const int tail0 = STRIP ? BASE + DATA : BASE + TILES * NT, tail = PLANE - tail0;
      if (tail > 0)
        for (int i = tid; i < CoutStore * tail; i += 256)
          out[((size_t)nb * CoutStore + i / tail) * PLANE + tail0 + i % tail] = 0.f;
    }
  }
}

template <int KS, int RS, int MT, int WM, int WN, bool RELU, bool COMPACT, bool STRIP = false>
int launch(const float* in, const float* wp, const float* bp, float* out, int NB, int CinP, int CoutStore, int H,
           int W, hipStream_t stream) {
  constexpr int NT = 256;
  constexpr int R = KS / 2;
  const int Ws = os2d_ws(W), PLANE = os2d_plane(H, W);
  int NS = 1, SPITCH = 0;
  os2d_conv_strips(W, R, &NS, &SPITCH);
  const int HALO = os2d_round_up(R * (STRIP ? SPITCH : Ws) + R, 4);
  const int SLAB = NT + 2 * HALO;
  if (2 * (SLAB / 4) > NBPF * 256) {
    os2d_set_error("conv%dx%d: feature map too wide for the B-slab prefetch (W=%d)", KS, KS, W);
    return -3;
  }
  const size_t lds = (size_t)(2 * RS * KS * 2 * MT + 4 * SLAB) * sizeof(float);
  auto kern = conv_mfma_kernel<KS, RS, MT, WM, WN, NT, RELU, COMPACT, STRIP>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(conv): %s", hipGetErrorString(e));
    return -4;
  }
  const int TPS = STRIP ? (H * SPITCH + NT - 1) / NT : 0;
  const int tiles = STRIP ? NS * TPS : (H * Ws + NT - 1) / NT;
  const long long groups = (long long)tiles * NB;
  dim3 grid((unsigned)((groups + 7) / 8 * 8));  // multiple of 8: every XCD gets the same number of logical slots
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, in, wp, bp, out, CinP, CoutStore, H, W, PLANE, HALO, tiles, NB, SPITCH, TPS);
  e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("conv launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

// layer 1: 7x7 225(226)->128 +ReLU, plane out; layer 2: 5x5 128->64 +ReLU, plane out;
// layer 3: 5x5 64->P (rows padded to 32 in the packed weights), compact [NB][P][H*W] out.
int os2d_launch_conv(int layer, const float* in, const float* wp, const float* bp, float* out, int NB, int P, int H,
                     int W, hipStream_t stream) {
  if (layer != 1 && W > OS2D_MAX_W_LINEAR5) {     // wider than the linear slab takes: column strips
    if (layer == 2) return launch<5, 5, 64, 1, 4, true, false, true>(in, wp, bp, out, NB, 128, 64, H, W, stream);
    if (layer == 3) return launch<5, 5, 32, 1, 4, false, true, true>(in, wp, bp, out, NB, 64, P, H, W, stream);
  }
  switch (layer) {
    case 1: return launch<7, 1, 128, 2, 2, true, false>(in, wp, bp, out, NB, OS2D_KP, 128, H, W, stream);
    case 2: return launch<5, 5, 64, 1, 4, true, false>(in, wp, bp, out, NB, 128, 64, H, W, stream);
    case 3: return launch<5, 5, 32, 1, 4, false, true>(in, wp, bp, out, NB, 64, P, H, W, stream);
    default: os2d_set_error("os2d_launch_conv: bad layer %d", layer); return -1;
  }
}

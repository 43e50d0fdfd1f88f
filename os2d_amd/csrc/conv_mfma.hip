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
// Work-group = 256 threads (4 waves), tile = MT output channels x NT=256 plane positions.
// Per channel pair the group stages in LDS:  A slab  [KS*KS][2][MT]  (packed weights, contiguous in HBM/L2)
//                                            B slab  [2][NT + 2*HALO] (input rows with halo)
// then issues KS*KS taps x (MI x NI) MFMAs per wave.  conv1: 49 taps x 8 MFMA x 64 cycles = 25k cycles of
// matrix work per 56 KB staged, i.e. the kernel is MFMA-bound; two groups per CU overlap each other's
// staging.  Arithmetic is exact fp32 (the MFMA is a k-ordered fmaf chain).
#include "os2d_common.h"

namespace {

template <int KS, int MT, int WM, int WN, int NT, bool RELU, bool COMPACT>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const float* __restrict__ in,   // [NB][CinP][PLANE]
                                                           const float* __restrict__ wp,   // [CinP/2][KS*KS][2][MT]
                                                           const float* __restrict__ bp,   // [MT]
                                                           float* __restrict__ out, int CinP, int CoutStore,
                                                           int H, int W, int PLANE, int HALO) {
  constexpr int R = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int MW = MT / WM, NW = NT / WN;
  constexpr int MI = MW / 32, NI = NW / 32;
  static_assert(WM * WN == 4, "4 waves per work-group");
  static_assert(MW % 32 == 0 && NW % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA block");
  constexpr int ASLAB = TAPS * 2 * MT;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ldsA = smem;
  float* ldsB = smem + ASLAB;

  const int Ws = W + 2 * OS2D_PAD, Hp = H + 2 * OS2D_PAD;
  const int SLAB = NT + 2 * HALO;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int nb = blockIdx.y;
  const int n0 = blockIdx.x * NT;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // tiles that only cover border rows have nothing to accumulate (block-uniform)
  const bool has_work = (n0 < (H + OS2D_PAD) * Ws) && (n0 + NT > OS2D_PAD * Ws);
  if (has_work) {
    const float* aBase = ldsA + hi * MT + wm * MW + l31;
    const float* bBase = ldsB + hi * SLAB + wn * NW + l31 + HALO - R * Ws - R;
    const int npairs = CinP >> 1;
    for (int cp = 0; cp < npairs; ++cp) {
      // ---- stage A: contiguous packed weights of this channel pair
      {
        const float4* src = reinterpret_cast<const float4*>(wp + (size_t)cp * ASLAB);
        float4* dst = reinterpret_cast<float4*>(ldsA);
        for (int i = tid; i < ASLAB / 4; i += 256) dst[i] = src[i];
      }
      // ---- stage B: two input channels, plane range [n0-HALO, n0+NT+HALO)
      {
        const int q4 = SLAB >> 2;
        for (int i = tid; i < 2 * q4; i += 256) {
          const int h2 = i >= q4 ? 1 : 0;
          const int j = (i - h2 * q4) << 2;
          const int g = n0 - HALO + j;  // plane-relative, multiple of 4
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g >= 0 && g + 3 < PLANE)
            v = *reinterpret_cast<const float4*>(in + ((size_t)nb * CinP + 2 * cp + h2) * PLANE + g);
          *reinterpret_cast<float4*>(ldsB + h2 * SLAB + j) = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        const float* bRow = bBase + dy * Ws;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
          const int tap = dy * KS + dx;
          float a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = aBase[tap * 2 * MT + mi * 32];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b[ni] = bRow[dx + ni * 32];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: bias (+ReLU); border cells of a padded output are written as exact zeros
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * NW + ni * 32 + l31;
    if (n >= Hp * Ws) continue;
    const int hr = n / Ws, wc = n - hr * Ws;
    const bool valid = hr >= OS2D_PAD && hr < H + OS2D_PAD && wc >= OS2D_PAD && wc < W + OS2D_PAD;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * MW + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= CoutStore) continue;
        float v = acc[mi][ni][r] + bp[m];
        if (RELU) v = fmaxf(v, 0.f);
        if (COMPACT) {
          if (valid) out[((size_t)nb * CoutStore + m) * (H * W) + (hr - OS2D_PAD) * W + (wc - OS2D_PAD)] = v;
        } else {
          out[((size_t)nb * CoutStore + m) * PLANE + n] = valid ? v : 0.f;
        }
      }
    }
  }
}

template <int KS, int MT, int WM, int WN, bool RELU, bool COMPACT>
int launch(const float* in, const float* wp, const float* bp, float* out, int NB, int CinP, int CoutStore, int H,
           int W, hipStream_t stream) {
  constexpr int NT = 256;
  constexpr int R = KS / 2;
  const int Ws = os2d_ws(W), Hp = os2d_hp(H), PLANE = os2d_plane(H, W);
  const int HALO = os2d_round_up(R * Ws + R, 4);
  const size_t lds = (size_t)(KS * KS * 2 * MT + 2 * (NT + 2 * HALO)) * sizeof(float);
  if (lds > 160 * 1024) {
    os2d_set_error("conv%dx%d: feature map too wide for the LDS halo (W=%d needs %zu B)", KS, KS, W, lds);
    return -3;
  }
  auto kern = conv_mfma_kernel<KS, MT, WM, WN, NT, RELU, COMPACT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(conv): %s", hipGetErrorString(e));
    return -4;
  }
  dim3 grid((Hp * Ws + NT - 1) / NT, NB);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, in, wp, bp, out, CinP, CoutStore, H, W, PLANE, HALO);
  e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("conv launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

// layer 1: 7x7 225(226)->128 +ReLU, padded out; layer 2: 5x5 128->64 +ReLU, padded out;
// layer 3: 5x5 64->P (rows padded to 32 in the packed weights), compact [NB][P][H*W] out.
int os2d_launch_conv(int layer, const float* in, const float* wp, const float* bp, float* out, int NB, int P, int H,
                     int W, hipStream_t stream) {
  switch (layer) {
    case 1: return launch<7, 128, 2, 2, true, false>(in, wp, bp, out, NB, OS2D_KP, 128, H, W, stream);
    case 2: return launch<5, 64, 1, 4, true, false>(in, wp, bp, out, NB, 128, 64, H, W, stream);
    case 3: return launch<5, 32, 1, 4, false, true>(in, wp, bp, out, NB, 64, P, H, W, stream);
    default: os2d_set_error("os2d_launch_conv: bad layer %d", layer); return -1;
  }
}

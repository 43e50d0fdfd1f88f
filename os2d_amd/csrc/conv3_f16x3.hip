// The last TransformNet layer (reference os2d/modeling/head.py:629: Conv 64 -> P, 5x5, P = 6 or 4) on the half-precision
// matrix cores with split operands (the f16x3 arithmetic of conv_f16x3.hip), as its own kernel: the generic kernel pads the
// P output rows to the 32 rows of v_mfma_f32_32x32x16_f16 (5.3x more matrix work than needed: 0.08 ms at 64 classes).
// v_mfma_f32_16x16x32_f16 has 16 rows and K = 32: one k-step = 8 input channels x FOUR taps - lanes 16q .. 16q+15 take tap
// 4 p + q of the k-step p - so the 25 taps are 7 k-steps per 8-channel group (3 of 28 tap slots carry zero weights).
//
// Layouts are conv_f16x3.hip's: input = the split-half blocked activations [NB][8 groups][hi|lo][PLANE] units of 8 channels,
// weights = what os2d_pack_conv_f16x3 packs for layer 3, [8 groups][28 taps][hi|lo][32 rows] units (rows >= 16 unused here),
// packed_b = [3][32] (bias | 2^-weight_exp | -); output = compact fp32 [NB][P][H*W].  A work-group (4 waves) owns 256 plane
// cells of one map: per channel group the input slab (256 + 2 x (2 rows + 2) halo cells, hi and lo) and the group's weights
// (28 x 2 x 16 units) arrive by LDS-DMA into the other half of a double buffer while the current group is multiplied.
// Every output element accumulates its products in ONE order whatever the batch size (one kernel shape for all).
// Maps wider than OS2D_MAX_W_LINEAR5 run in column strips (SP > 0: the strip-plane geometry of conv_f16x3.hip).
//
// FUSE (round 6): the layer's P outputs of a location are all the alignment epilogue needs for that location
// (sample_decode.h: theta, resample + pool, box, corners, loc) - so the head's launch does that epilogue right here: the accumulators
// go through LDS to one lane per cell ([P][256] floats in the operand buffer, which is free by then) and every lane runs
// os2d_sample_decode_location for its cell.  The parameters never touch HBM, one launch and one dependent-launch gap (~11 us per
// step at 64 classes) less, and the gather-bound epilogue of one work-group overlaps the load-bound matrix phase of its neighbours
// on the CU.  The standalone kernels remain for the per-stage entry points (os2d_transform_conv_f16x3, os2d_sample_decode).
#include "os2d_common.h"
#include "sample_decode.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C3_THR = 256;
constexpr int C3_NT = 256;          // plane cells per work-group (64 per wave = 4 blocks of 16)
constexpr int C3_G = 8;             // 8-channel input groups (64 channels)
constexpr int C3_TAPS = 28;         // tap slots: 7 k-steps x 4
constexpr int C3_ROWS = 8;          // output rows kept in LDS (P <= 8; MFMA rows 8 .. 15 re-read rows 0 .. 7 and are dropped)
constexpr int C3_WUNITS = C3_TAPS * 2 * C3_ROWS;   // weight units of one group kept in LDS: 448
constexpr int C3_WG = C3_TAPS * 2 * 32;       // ... in the packed global layout (32 rows)

#ifndef OS2D_C3_BUFS
#define OS2D_C3_BUFS 1
#endif
#ifndef OS2D_C3_FUSE_GROUPS
#define OS2D_C3_FUSE_GROUPS 5      /* work-groups per CU the register budget of the fused kernel is set for */
#endif
// BUFS = 2: the next group's operands land in the other half of a double buffer while this one is multiplied (53 KB: three
// work-groups per CU).  BUFS = 1: one buffer (26.6 KB), load -> barrier -> multiply -> barrier, and FIVE work-groups per CU
// cover each other's load latency (a group is only 0.6 us of matrix work against ~2 us of load latency); 1280 slots are
// exactly the 1280 work-groups of 64 classes at 60 x 80.  Measured: 0.058 ms double-buffered, 0.049 ms with four single-
// buffered groups per CU (16 weight rows), 8 weight rows + five groups below.
constexpr int C3_BUFS = OS2D_C3_BUFS;
struct C3Fuse {            // what the fused epilogue needs on top of the layer's own arguments (os2d_launch_sample_decode's)
  const float* corr;       // [NB][225][HW]
  float *loc, *cls, *corners;
  const int* flags;        // [A + 1] range words of the call or NULL
  int* host_status;
  int inverse, Bc, Btot, b0, A, epoch;
  float stride, half_box;
};
template <bool FUSE>
__global__ __launch_bounds__(C3_THR, C3_BUFS == 1 ? (FUSE ? OS2D_C3_FUSE_GROUPS : 5) : 2) void conv3_f16x3_kernel(
    const u32x4* in, const u32x4* wp, const float* __restrict__ bp, float* __restrict__ out, int P, int H, int W, int PLANE, int TILES, int NB,
    int SP /*0: linear; else row pitch of a strip-plane*/, int TPS /*tiles per strip*/, C3Fuse fz) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  const int Ws = W + OS2D_PAD, BASE = os2d_base(W), DATA = H * Ws;
  const int PW = SP ? SP : Ws;                     // row pitch of the cells in the LDS slab
  const int HALO = 2 * PW + 2, SLAB = C3_NT + 2 * HALO;
  const int SLABP = (2 * SLAB + 63) & ~63;         // slab area rounded up to whole DMA instructions (64 units)
  const int STAGE = SLABP + C3_WUNITS;             // units of one double-buffer half: slab hi | slab lo | pad | weights
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);   // XCD-aware order, see conv_f16x3.hip
  if (logical >= TILES * NB) return;
  const int tile = logical % TILES, nb = logical / TILES;
  const int strip = SP ? tile / TPS : 0;
  const int c0mR = SP ? strip * (SP - 4) - 2 : 0;  // map column of strip-plane column 0
  const int n0 = SP ? (tile - strip * TPS) * C3_NT : BASE + tile * C3_NT;
  const u32x4* inb = in + (size_t)nb * C3_G * 2 * PLANE;

  typedef const void __attribute__((address_space(1))) * gptr_t;
  typedef void __attribute__((address_space(3))) * lptr_t;
  // one group's operands -> LDS half (G & 1): slab cells n0 - HALO .. n0 + NT + HALO of the hi and the lo plane (clamped to
  // the plane: what lies outside is border = zero, and the last cell of a plane is a zero cell), then the 896 weight units
  const int nslab = (2 * SLAB + 63) / 64;           // wave instructions for the slab (both parts back to back)
#define C3_DMA(GRP)                                                                                               \
  {                                                                                                               \
    u32x4* dst_ = smem16 + ((GRP) & (C3_BUFS - 1)) * STAGE;                                                       \
    for (int i_ = wv; i_ < nslab; i_ += C3_THR / 64) {                                                            \
      const int u_ = i_ * 64 + lane;                     /* unit in [0, 2 SLAB): part = u / SLAB */               \
      const int part_ = u_ >= SLAB ? 1 : 0;                                                                       \
      int c_ = n0 - HALO + (u_ - part_ * SLAB);                                                                   \
      if (SP) {                                          /* strip-plane index -> map cell, 0 (a zero cell) outside */ \
        const int h_ = c_ / SP, cc_ = c0mR + (c_ - h_ * SP);                                                      \
        c_ = (c_ >= 0 && u_ < 2 * SLAB && h_ < H && cc_ >= 0 && cc_ < W) ? BASE + h_ * Ws + cc_ : 0;              \
      } else {                                                                                                    \
        c_ = min(max(c_, 0), PLANE - 1);                                                                          \
      }                                                                                                           \
      const u32x4* src_ = inb + ((size_t)(GRP)*2 + part_) * PLANE + c_;                                           \
      __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(dst_ + i_ * 64), 16, 0, 0);                         \
    }                                                                                                             \
    for (int i_ = wv; i_ < C3_WUNITS / 64; i_ += C3_THR / 64) {                                                   \
      const int u_ = i_ * 64 + lane;                     /* (tap, part, row < 8) */                               \
      const u32x4* src_ = wp + (size_t)(GRP)*C3_WG + (u_ / C3_ROWS) * 32 + (u_ % C3_ROWS);                        \
      __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(dst_ + SLABP + i_ * 64), 16, 0, 0);                  \
    }                                                                                                             \
  }

  // per-lane cell offsets of the 7 k-steps: tap t = 4 p + kq -> (dy, dx) = (t / 5, t % 5); slots 25 .. 27 have zero
  // weights (any valid cell will do)
  int toff[7];
#pragma unroll
  for (int p = 0; p < 7; ++p) {
    const int t = 4 * p + kq;
    toff[p] = t < 25 ? (t / 5 - 2) * PW + (t % 5 - 2) : 0;
  }
  f32x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (C3_BUFS == 2) {
    C3_DMA(0)
    __syncthreads();
  }
  for (int g = 0; g < C3_G; ++g) {
    if (C3_BUFS == 2) {
      if (g + 1 < C3_G) C3_DMA(g + 1)
    } else {
      C3_DMA(g)
      __syncthreads();   // landed (vmcnt(0) is part of the barrier)
    }
    __builtin_amdgcn_sched_barrier(0);
    const u32x4* sl = smem16 + (g & (C3_BUFS - 1)) * STAGE;
    const u32x4* bB = sl + HALO + wv * 64 + l15;          // this lane's cell of column block 0, hi plane
    const u32x4* wB = sl + SLABP + (l15 & (C3_ROWS - 1)); // row l15 (rows >= 8 of the instruction are not used)
#pragma unroll
    for (int p = 0; p < 7; ++p) {
      const int t = 4 * p + kq;
      const half8 ah = *reinterpret_cast<const half8*>(wB + (t * 2 + 0) * C3_ROWS);
      const half8 al = *reinterpret_cast<const half8*>(wB + (t * 2 + 1) * C3_ROWS);
      half8 bh[4], bl[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        bh[cb] = *reinterpret_cast<const half8*>(bB + toff[p] + cb * 16);
        bl[cb] = *reinterpret_cast<const half8*>(bB + toff[p] + cb * 16 + SLAB);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[cb], acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[cb], acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[cb], acc[cb], 0, 0, 0);
    }
    __syncthreads();   // the other half has landed (vmcnt(0) is part of the barrier) and this one is free again
  }
#undef C3_DMA

  // ---- epilogue: a lane holds rows 4 kq .. 4 kq + 3 of column l15 of every block: undo the weight scale, add the bias
  if (FUSE) {
    // parameters -> LDS [8 rows][256 cells] (the operand buffer is free: the loop ended in a barrier), then one lane per cell
    float* pl = reinterpret_cast<float*>(smem16);
    if (kq < 2) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = 4 * kq + k;
          const float a = acc[cb][k];
          pl[m * C3_NT + wv * 64 + cb * 16 + l15] = a * bp[32 + (m & 7)] + bp[m & 7];
        }
    }
    __syncthreads();
    const int n = n0 + tid;
    int hr, wc;
    bool ok;
    if (SP) {
      hr = n / SP;
      const int j = n - hr * SP;
      wc = c0mR + j;
      ok = !(j < 2 || j >= SP - 2 || hr >= H || wc >= W);
    } else {
      const int r = n - BASE;
      hr = r / Ws;
      wc = r - hr * Ws;
      ok = !(n >= PLANE || r >= DATA || wc >= W);
    }
    if (!ok) return;
    const int HW = H * W, nl = hr * W + wc;
    const int img = nb / fz.Bc;
    const size_t ob = (size_t)img * fz.Btot + fz.b0 + (nb - img * fz.Bc);
    if (fz.flags != nullptr && (fz.flags[img] == fz.epoch || fz.flags[fz.A] == fz.epoch)) {      // non-finite input: see sample_decode.hip
      os2d_sample_decode_poison(HW, ob, nl, fz.loc, fz.cls, fz.corners);
      if (fz.host_status != nullptr && tile == 0 && tid == (SP ? 2 : 0))
        __hip_atomic_store(fz.host_status, OS2D_STATUS_F16_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    os2d_sample_decode_location(fz.corr + (size_t)nb * OS2D_K * HW, pl + tid, C3_NT, H, W, P, fz.inverse, fz.stride, fz.half_box, ob, nl, hr, wc,
                                fz.loc, fz.cls, fz.corners);
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int n = n0 + wv * 64 + cb * 16 + l15;
    int hr, wc;
    if (SP) {
      hr = n / SP;
      const int j = n - hr * SP;
      wc = c0mR + j;
      if (j < 2 || j >= SP - 2 || hr >= H || wc >= W) continue;
    } else {
      const int r = n - BASE;
      hr = r / Ws;
      wc = r - hr * Ws;
      if (n >= PLANE || r >= DATA || wc >= W) continue;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = 4 * kq + k;
      if (m < P) {
        const float a = acc[cb][k];   // (scalar copy: see the ext-vector note in corr_f16x3.hip)
        out[((size_t)nb * P + m) * (H * W) + hr * W + wc] = a * bp[32 + m] + bp[m];
      }
    }
  }
}

}  // namespace

namespace {
int launch_conv3(bool fuse, const C3Fuse& fz, const void* in, const void* wp, const float* bp, void* out, int NB, int P, int H, int W,
                 hipStream_t stream) {
  const int Ws = os2d_ws(W), PLANE = os2d_plane(H, W);
  int NS = 1, SP = 0;
  if (W > OS2D_MAX_W_LINEAR5) os2d_conv_strips(W, 2, &NS, &SP);
  const int SLAB = C3_NT + 2 * (2 * (SP ? SP : Ws) + 2);
  const size_t lds = (size_t)C3_BUFS * (((2 * SLAB + 63) & ~63) + C3_WUNITS) * 16;
  if (lds > 160 * 1024) {
    os2d_set_error("conv3 (f16x3): feature map too wide for the input slab (W=%d)", W);
    return -3;
  }
  auto kern = fuse ? conv3_f16x3_kernel<true> : conv3_f16x3_kernel<false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(conv3 f16x3): %s", hipGetErrorString(e));
    return -4;
  }
  const int TPS = SP ? (H * SP + C3_NT - 1) / C3_NT : 0;
  const int tiles = SP ? NS * TPS : (H * Ws + C3_NT - 1) / C3_NT;
  const long long groups = (long long)tiles * NB;
  if (groups + 7 > 0x7fffffffLL) {
    os2d_set_error("conv3 f16x3: too many work-groups (%lld)", groups);
    return -3;
  }
  dim3 grid((unsigned)((groups + 7) / 8 * 8));
  hipLaunchKernelGGL(kern, grid, dim3(C3_THR), lds, stream, static_cast<const u32x4*>(in), static_cast<const u32x4*>(wp), bp,
                     static_cast<float*>(out), P, H, W, PLANE, tiles, NB, SP, TPS, fz);
  e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("conv3 f16x3 launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}
}  // namespace

int os2d_launch_conv3_f16x3(const void* in, const void* wp, const float* bp, void* out, int NB, int P, int H, int W, hipStream_t stream) {
  return launch_conv3(false, C3Fuse{}, in, wp, bp, out, NB, P, H, W, stream);
}

// the last layer + the alignment epilogue of the head in one launch (arguments of os2d_launch_conv3_f16x3 and os2d_launch_sample_decode;
// the parameters are not written anywhere)
int os2d_launch_conv3_sample_decode(const void* in, const void* wp, const float* bp, const float* corr, int NB, int H, int W, int P,
                                    int inverse, int stride, int rec_field, int Bc, int Btot, int b0, float* loc, float* cls, float* corners,
                                    const int* flags, int epoch, int* host_status, hipStream_t stream) {
  C3Fuse fz;
  fz.corr = corr;
  fz.loc = loc;
  fz.cls = cls;
  fz.corners = corners;
  fz.flags = flags;
  fz.host_status = host_status;
  fz.inverse = inverse;
  fz.Bc = Bc;
  fz.Btot = Btot;
  fz.b0 = b0;
  fz.A = NB / Bc;
  fz.epoch = epoch;
  fz.stride = (float)stride;
  fz.half_box = 0.5f * (float)(stride * (OS2D_T - 1) + rec_field);      // head.py:236-237
  return launch_conv3(true, fz, in, wp, bp, nullptr, NB, P, H, W, stream);
}

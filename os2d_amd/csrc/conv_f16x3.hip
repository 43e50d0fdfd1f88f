// TransformNet convolutions on the half-precision matrix cores with fp32-equivalent accuracy ("f16x3"), gfx950.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (157 TFLOP/s); v_mfma_f32_32x32x16_f16 is 16x faster.
// Every fp32 operand x is stored as TWO halves, hi = rn16(x), lo = rn16(x - hi) (22 mantissa bits together), and a
// product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with exact fp16 products accumulated in fp32 by the
// MFMA - three instructions instead of sixteen-times-slower one.  Weights are pre-multiplied by a power of two so that
// their lo parts stay normal fp16 numbers (undone exactly in the epilogue).  On the reference's golden vectors the
// resulting TransformNet parameters differ from the fp32 path by <= 3e-7 (tests/test_head_gpu.py, precision=f16x3).
//
// Layouts ("split-half blocked", SHB): activations [NB][G = C/8][2 = hi|lo][PLANE] units of 16 B = 8 channels of one
// plane cell (zero-bordered plane geometry of os2d_common.h).  One MFMA k-step = 16 k = 8 channels x 2 TAPS:
// lanes 0-31 take tap 2p, lanes 32-63 tap 2p+1 of the same channel group, so a work-group only needs an 8-channel
// input slab in LDS (the halo makes that slab 3x the tile).  Packed weights [G][STEPS][2 half-wave][2 hi|lo][MT] units.
//
// Work-group = 512 threads (8 waves, 2 per SIMD) on MT x NT = (WM*MI*32) x (WN*NI*32) outputs.  Pipeline per channel
// group: the weight slab streams through double-buffered LDS stages of SS k-steps (global -> registers issued before the
// MFMAs of the previous stage, registers -> LDS after them, one barrier per stage); the input slab of the next group is
// fetched with the last stage of the current one.  All prefetch indices are clamped instead of predicated so the loop
// is branch-free (hipcc otherwise sinks the loads behind the MFMAs).
#include "os2d_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-byte unit (8 halves); ext vector: stays in VGPRs
#define U32X4_ZERO (u32x4{0u, 0u, 0u, 0u})
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_half(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// STRIP mode (maps wider than the linear slab allows: 256 + 2 * HALO <= 1536 units, W <= 316 for a 5x5 layer; the reference has
// no width limit, head.py:622-629).  The map is cut into NS column strips of SW output columns.  A strip is treated as a
// zero-bordered plane of its own with row pitch SP = SW + 2 R: strip-plane cell (h, j) is map cell (h, c0 - R + j), zero
// outside the map - so the R columns either side of the strip's outputs hold the NEIGHBOURING strips' data instead of
// zeros - and the convolution is again a shift-and-accumulate over the flat strip-plane index n' = h * SP + j, exact for
// the output columns R <= j < SP - R.  The matrix loop is the linear one with SP for the row pitch; only the slab loads and
// the epilogue translate n' into map cells.  Returns the map-plane cell of strip-plane index np, or 0 - a border cell of
// every plane, zero by contract - when np lies outside the map.
__device__ __forceinline__ int os2d_strip_cell(int np, int SP, int c0mR, int H, int W, int Ws, int BASE) {
  const int h = np / SP, c = c0mR + (np - h * SP);
  return (np >= 0 && h < H && c >= 0 && c < W) ? BASE + h * Ws + c : 0;
}

// MTP = output rows of the packed weights (all output channels, padded), MT = rows handled by ONE work-group
// (blockIdx.z selects the slice): splitting the output channels over two independent 4-wave groups per CU lets one
// group's barrier / staging bubble be filled by the other's MFMAs.
// TERMS = 3: fp32-equivalent product (above).  TERMS = 2 drops a_lo*b_hi, i.e. the WEIGHTS enter as their fp16 roundings
// only (activations keep both halves): two thirds of the matrix-core work.  Used for the 7x7 layer under precision
// "f16x2", where the averaging over K = 11025 keeps the box regression within 5e-5 of the fp32 result.
template <int KS, int MTP, int MT, int WM, int WN, int NI, int SS, bool RELU, int OUT_MODE /*0 SHB, 1 fp32 plane, 2 fp32 compact [NB][Cout][H*W]*/,
          int NBPF /*16-byte units per thread for the input-slab prefetch*/, int TERMS, int MINW /*waves per SIMD the register budget allows*/,
          bool STRIP = false>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_f16x3_kernel(const u32x4* in,  // SHB [NB][G][2][PLANE] (no __restrict__: invariant loads get
                                                             const u32x4* wp,  // rematerialised BEHIND the MFMAs by the register allocator)
                                                             const float* __restrict__ bp,  // [3][MTP] fp32 per output row: folded bias | 2^-weight_exp | 2^out_exp
                                                             Os2dRangeFlag status,
                                                             void* __restrict__ outv, int G,
                                                             int CoutStore, int H, int W, int PLANE, int HALO,
                                                             int TILES, int NB, int SP /*STRIP: row pitch of a strip-plane*/,
                                                             int TPS /*STRIP: tiles per strip*/) {
  constexpr int R = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int STEPS = (TAPS + 1) / 2;
  constexpr int NST = (STEPS + SS - 1) / SS;  // stages per channel group
  constexpr int MW = MT / WM, MI = MW / 32;
  constexpr int NW = NI * 32, NT = WN * NW;
  constexpr int NTHR = 64 * WM * WN;
  static_assert(MW % 32 == 0 && MTP % MT == 0, "wave tile rows must be a multiple of 32");
  constexpr int AP = TERMS == 3 ? 2 : 1;      // weight parts staged in LDS (hi|lo or hi only)
  constexpr int ASTAGE = SS * 2 * AP * MT;    // 16-byte units per weight stage in LDS (this group's rows only)
  constexpr int ASTAGE_G = SS * 4 * MTP;  // ... and in the packed global layout
  constexpr int NAPF = (ASTAGE + NTHR - 1) / NTHR;

  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  const int Ws = W + OS2D_PAD;
  const int BASE = os2d_base(W);
  const int DATA = H * Ws;
  const int SLAB = NT + 2 * HALO;
  u32x4* ldsA = smem16;               // [2][ASTAGE]
  u32x4* ldsB = smem16 + 2 * ASTAGE;  // [2 parts][SLAB]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hw = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  // XCD-aware work mapping: the dispatcher places work-group L on XCD L % 8, so XCD x is given the contiguous range
  // [x*per, (x+1)*per) of the logical order (plane, tile, channel half).  The ~40 groups of one plane - whose input slabs
  // overlap by 2/3 (halo) and are shared by both channel halves - then run on ONE XCD at about the same time and the
  // plane is fetched into that L2 once instead of ~6 times (FETCH_SIZE 2.0 -> 0.6 GB per launch at 64 classes).
  constexpr int ZG = MTP / MT;
  const int per = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (logical >= TILES * NB * ZG) return;
  const int zg = logical % ZG, tile = (logical / ZG) % TILES;
  const int nb = logical / (ZG * TILES);
  const int PW = STRIP ? SP : Ws;                        // row pitch of the cells in the LDS slab
  const int strip = STRIP ? tile / TPS : 0;
  const int c0mR = STRIP ? strip * (SP - 2 * R) - R : 0; // map column of strip-plane column 0
  const int n0 = STRIP ? (tile - strip * TPS) * NT : BASE + tile * NT;     // first output cell (strip-plane | plane index)
  const int mOff = zg * MT;  // first output row of this group

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const u32x4* inb = in + (size_t)nb * G * 2 * PLANE;
  const int aLane = wm * MW + l31;
  // B lane bases (units): same-row tap pair -> upper half-wave reads the next cell; row-crossing pair -> next row start
  const int bLane = wn * NW + l31 + HALO - R * PW - R;
  const int bSame = bLane + hw;
  const int bCross = bLane + hw * (PW - (KS - 1));

  u32x4 pfA[NAPF], pfB[NBPF];
  half8 fah[2][MI], fal[2][MI], fbh[2][NI], fbl[2][NI];  // double-buffered MFMA fragments

#define F16_LOAD_A(S)                                                                                             \
  {                                                                                                               \
    const u32x4* src_ = wp + (size_t)(S)*ASTAGE_G + mOff;                                                         \
    _Pragma("unroll") for (int k = 0; k < NAPF; ++k) {                                                            \
      const int i_ = min(tid + k * NTHR, ASTAGE - 1);                                                             \
      pfA[k] = src_[(i_ / MT) * (2 / AP) * MTP + (i_ % MT)];                                                      \
    }                                                                                                             \
  }
#define F16_LOAD_A1(S, K)                                                                                         \
  {                                                                                                               \
    const int i_ = min(tid + (K)*NTHR, ASTAGE - 1);                                                               \
    pfA[K] = (wp + (size_t)(S)*ASTAGE_G + mOff)[(i_ / MT) * (2 / AP) * MTP + (i_ % MT)];                          \
  }
#define F16_LOAD_B1(GRP, K)                                                                                       \
  {                                                                                                               \
    const int i_ = min(tid + (K)*NTHR, 2 * SLAB - 1);                                                             \
    const int part_ = i_ >= SLAB ? 1 : 0;                                                                         \
    int g_ = n0 - HALO + (i_ - part_ * SLAB);                                                                     \
    g_ = STRIP ? os2d_strip_cell(g_, SP, c0mR, H, W, Ws, BASE) : ((g_ >= 0 && g_ < PLANE) ? g_ : 0);              \
    pfB[K] = inb[((size_t)(GRP)*2 + part_) * PLANE + g_];                                                         \
  }
#define F16_STORE_A(S)                                                                                            \
  {                                                                                                               \
    u32x4* dst_ = ldsA + ((S)&1) * ASTAGE;                                                                        \
    _Pragma("unroll") for (int k = 0; k < NAPF; ++k) {                                                            \
      const int i_ = tid + k * NTHR;                                                                              \
      if (i_ < ASTAGE) dst_[i_] = pfA[k];                                                                         \
    }                                                                                                             \
  }
#define F16_LOAD_B(GRP)                                                                                           \
  {                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < NBPF; ++k) {                                                            \
      const int i_ = min(tid + k * NTHR, 2 * SLAB - 1);                                                           \
      const int part_ = i_ >= SLAB ? 1 : 0;                                                                       \
      int g_ = n0 - HALO + (i_ - part_ * SLAB);                                                                   \
      g_ = STRIP ? os2d_strip_cell(g_, SP, c0mR, H, W, Ws, BASE) : ((g_ >= 0 && g_ < PLANE) ? g_ : 0);            \
      pfB[k] = inb[((size_t)(GRP)*2 + part_) * PLANE + g_];                                                       \
    }                                                                                                             \
  }
#define F16_STORE_B()                                                                                             \
  {                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < NBPF; ++k) {                                                            \
      const int i_ = tid + k * NTHR;                                                                              \
      if (i_ < 2 * SLAB) {                                                                                        \
        const int part_ = i_ >= SLAB ? 1 : 0;                                                                     \
        const int g_ = n0 - HALO + (i_ - part_ * SLAB);                                                           \
        const bool in_ = STRIP ? os2d_strip_cell(g_, SP, c0mR, H, W, Ws, BASE) != 0 : (g_ >= 0 && g_ < PLANE);    \
        ldsB[i_] = in_ ? pfB[k] : U32X4_ZERO;                                                                     \
      }                                                                                                           \
    }                                                                                                             \
  }
  // MFMAs of stage ST (compile-time) of the current group from weight buffer BUF (runtime 0/1)
  // Fragment reads of k-step P (compile-time, within stage ST) into register set SET
#define F16_FRAGS(ST, BUF, P, SET)                                                                                \
  {                                                                                                               \
    const int ps_ = (ST)*SS + (P);                                                                                \
    const int t0_ = 2 * ps_, t1_ = (2 * ps_ + 1 < TAPS) ? 2 * ps_ + 1 : 2 * ps_;                                  \
    const int dy0_ = t0_ / KS, dx0_ = t0_ % KS, dy1_ = t1_ / KS;                                                  \
    const int bsel_ = (t1_ == t0_) ? bLane : (dy1_ == dy0_ ? bSame : bCross);                                     \
    const u32x4* bS_ = ldsB + bsel_ + dy0_ * PW + dx0_;                                                           \
    const u32x4* aS_ = ldsA + (BUF)*ASTAGE + aLane + (((P)*2 + hw) * AP) * MT;                                    \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                           \
      fah[SET][mi] = *reinterpret_cast<const half8*>(aS_ + mi * 32);                                              \
      if (TERMS == 3) fal[SET][mi] = *reinterpret_cast<const half8*>(aS_ + MT + mi * 32);                         \
    }                                                                                                             \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                           \
      fbh[SET][ni] = *reinterpret_cast<const half8*>(bS_ + ni * 32);                                              \
      fbl[SET][ni] = *reinterpret_cast<const half8*>(bS_ + SLAB + ni * 32);                                       \
    }                                                                                                             \
  }
  // three passes over the MI x NI blocks: consecutive MFMAs never touch the same accumulator
#define F16_MFMAS(SET)                                                                                            \
  {                                                                                                               \
    if (TERMS == 3) {                                                                                             \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                           \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                         \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[SET][mi], fbh[SET][ni], acc[mi][ni], 0, 0, 0); \
    }                                                                                                             \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                             \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                           \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[SET][mi], fbl[SET][ni], acc[mi][ni], 0, 0, 0);   \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                             \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                           \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[SET][mi], fbh[SET][ni], acc[mi][ni], 0, 0, 0);   \
  }
  // MFMAs of stage ST (compile-time) of the current group from weight buffer BUF (runtime 0/1); the LDS fragment
  // reads of k-step p+1 are issued BEFORE the 3*MI*NI MFMAs of k-step p (two register sets), and the scheduler is
  // pinned to that order: a wave then waits for LDS only once per stage instead of once per k-step.
#define F16_COMPUTE(ST, BUF)                                                                                      \
  {                                                                                                               \
    const int nsteps_ = (STEPS - (ST)*SS) < SS ? (STEPS - (ST)*SS) : SS; /* folds: ST is an unrolled index */     \
    F16_FRAGS(ST, BUF, 0, 0)                                                                                      \
    _Pragma("unroll") for (int p = 0; p < SS; ++p) {                                                              \
      if (p < nsteps_) {                                                                                          \
        if (p + 1 < nsteps_) {                                                                                    \
          F16_FRAGS(ST, BUF, p + 1, (p + 1) & 1)                                                                  \
          __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                         \
        F16_MFMAS(p & 1)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        F16_PF(ST, p)                                                                                             \
      }                                                                                                           \
    }                                                                                                             \
  }
  // Prefetch pieces issued right after the MFMAs of k-step P of stage ST: one weight unit of the next stage, and in the
  // last stage of a channel group a share of the next group's input slab.  Spreading the loads over the stage instead
  // of bursting them at its start keeps the CU's load path short for everybody (see corr_f16x3.hip).
  constexpr int LASTN = STEPS - (NST - 1) * SS;        // k-steps of the last stage of a group
  constexpr int BPS = (NBPF + LASTN - 1) / LASTN;      // slab units per k-step there
#define F16_PF(ST, P)                                                                                             \
  {                                                                                                               \
    if ((P) < NAPF) F16_LOAD_A1(s1, P)                                                                            \
    if ((ST) == NST - 1) {                                                                                        \
      _Pragma("unroll") for (int kk = 0; kk < BPS; ++kk)                                                          \
        if ((P)*BPS + kk < NBPF) F16_LOAD_B1(g1, (P)*BPS + kk)                                                    \
    }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  }

  // ---- prologue: group 0 slab + stage 0 weights
  F16_LOAD_B(0)
  F16_LOAD_A(0)
  F16_STORE_B()
  F16_STORE_A(0)
  __syncthreads();
  const int nstages = G * NST;
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int s = g * NST + st;
      const int s1 = min(s + 1, nstages - 1);  // clamped: the very last iteration re-loads its own stage (harmless)
      const int g1 = min(g + 1, G - 1);
      {  // weight units beyond this stage's k-steps (one unit rides behind each k-step) go first
        const int nst_ = (st == NST - 1) ? (STEPS - (NST - 1) * SS) : SS;
        _Pragma("unroll") for (int k = 0; k < NAPF; ++k)
          if (k >= nst_) F16_LOAD_A1(s1, k)
      }
      __builtin_amdgcn_sched_barrier(0);
      F16_COMPUTE(st, (s & 1))
      __builtin_amdgcn_sched_barrier(0);
      if (st == NST - 1) {
        __syncthreads();  // every wave is done with this group's input slab
        F16_STORE_B()
      }
      F16_STORE_A(s + 1)
      __syncthreads();
    }
  }
#undef F16_LOAD_A
#undef F16_LOAD_A1
#undef F16_LOAD_B1
#undef F16_PF
#undef F16_STORE_A
#undef F16_LOAD_B
#undef F16_STORE_B
#undef F16_COMPUTE
#undef F16_FRAGS
#undef F16_MFMAS

  // ---- epilogue: undo the weight scale of the output channel (the input channels' scales are folded into the packed
  // weights), bias (+ReLU), apply the output channel's scale of the split activation buffer; pad cells are written as
  // exact zeros.  The output scales come from a rigorous per-channel bound of the layer's outputs
  // (os2d_amd/modeling/head.py: TransformationNet.range_plan), so a FINITE input cannot leave the fp16 range; anything
  // that still does (Inf / NaN inputs) raises the sticky status flag instead of being clamped silently.
  bool out_of_range = false;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int n = n0 + wn * NW + ni * 32 + l31;
    int hr, wc;
    if (STRIP) {       // strip-plane index -> map cell; only the strip's own output columns (incl. the map's pad columns) are stored
      hr = n / SP;
      const int j = n - hr * SP;
      wc = c0mR + j;
      if (j < R || j >= SP - R || hr >= H || wc >= Ws) continue;
      n = BASE + hr * Ws + wc;
    } else {
      const int r = n - BASE;
      hr = r / Ws;
      wc = r - hr * Ws;
      if (n >= PLANE) continue;
    }
    const bool valid = hr < H && wc < W;      // (linear: r < DATA <=> hr < H)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m0 = mOff + wm * MW + mi * 32 + 8 * q + 4 * hw;  // this lane holds rows m0..m0+3 (regs 4q..4q+3)
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = acc[mi][ni][4 * q + k] * bp[MTP + m0 + k] + bp[m0 + k];
          const bool live = valid && m0 + k < CoutStore;
          if (OUT_MODE == 0 && live && t != t) out_of_range = true;   // before the ReLU: fmaxf(NaN, 0) = 0
          if (RELU) t = fmaxf(t, 0.f);
          if (OUT_MODE == 0) {
            t *= bp[2 * MTP + m0 + k];
            if (live && !(fabsf(t) <= 65504.f)) out_of_range = true;
          }
          v[k] = live ? t : 0.f;
        }
        if (OUT_MODE == 0) {
          // SHB: 8-channel group = rows 8*(m0/8) .. +7; this lane writes channels 4*hw .. 4*hw+3 (8 bytes) of hi and lo
          const int grp = m0 >> 3;
          if (grp * 8 >= CoutStore) continue;
          half4 hi4, lo4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            _Float16 h_, l_;
            split_half(v[k], h_, l_);
            hi4[k] = h_;
            lo4[k] = l_;
          }
          const int Gout = (CoutStore + 7) >> 3;
          // lanes l and l + 32 hold channels 0-3 / 4-7 of the same unit of the same cell (n depends on l31 only, grp on neither
          // half): v_permlane32_swap hands the upper lane's hi half to the lower lane and the lower lane's lo half to the upper
          // one - the lower lane stores the complete hi unit, the upper lane the complete lo unit: one 16-byte store per lane
          // instead of two 8-byte halves of two units (round 6: the same time, half the store instructions, no partially
          // written lines; profiles/r06/stages_store_policies_and_corr_order.txt).
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 hw_ = __builtin_bit_cast(u32x2, hi4), lw_ = __builtin_bit_cast(u32x2, lo4);
          const u32x2 s0 = __builtin_amdgcn_permlane32_swap(hw_[0], lw_[0], false, false);
          const u32x2 s1 = __builtin_amdgcn_permlane32_swap(hw_[1], lw_[1], false, false);
          const u32x4 unit = {s0[0], s1[0], s0[1], s1[1]};
          char* o = reinterpret_cast<char*>(outv) + ((((size_t)nb * Gout + grp) * 2 + hw) * PLANE + n) * 16;
          *reinterpret_cast<u32x4*>(o) = unit;
        } else if (OUT_MODE == 1) {
          float* o = reinterpret_cast<float*>(outv);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (m0 + k < CoutStore) o[((size_t)nb * CoutStore + m0 + k) * PLANE + n] = v[k];
        } else {
          float* o = reinterpret_cast<float*>(outv);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (valid && m0 + k < CoutStore) o[((size_t)nb * CoutStore + m0 + k) * (H * W) + hr * W + wc] = v[k];
        }
      }
    }
  }
  if (OUT_MODE == 0 && status.word != nullptr && __builtin_amdgcn_ballot_w64(out_of_range) != 0ull) {
    if (lane == 0) os2d_raise(status);
  }
  // pad rows above the data (first tile) and whatever lies beyond the last tile
  {
    const int tail0 = STRIP ? BASE + DATA : BASE + TILES * NT, tail = PLANE - tail0;
    const bool first = tile == 0 && zg == 0, last = tile == TILES - 1 && zg == 0;
    if (OUT_MODE == 0) {
      const int planes = ((CoutStore + 7) >> 3) * 2;
      u32x4* o = reinterpret_cast<u32x4*>(outv) + (size_t)nb * planes * PLANE;
      if (first)
        for (int i = tid; i < planes * BASE; i += NTHR) o[(size_t)(i / BASE) * PLANE + i % BASE] = U32X4_ZERO;
      if (last && tail > 0)
        for (int i = tid; i < planes * tail; i += NTHR)
          o[(size_t)(i / tail) * PLANE + tail0 + i % tail] = U32X4_ZERO;
    } else if (OUT_MODE == 1) {
      float* o = reinterpret_cast<float*>(outv) + (size_t)nb * CoutStore * PLANE;
      if (first)
        for (int i = tid; i < CoutStore * BASE; i += NTHR) o[(size_t)(i / BASE) * PLANE + i % BASE] = 0.f;
      if (last && tail > 0)
        for (int i = tid; i < CoutStore * tail; i += NTHR) o[(size_t)(i / tail) * PLANE + tail0 + i % tail] = 0.f;
    }
  }
}

template <int KS, int MTP, int MT, int WM, int WN, int NI, int SS, bool RELU, int OUT_MODE, int TERMS = 3, int NBPF = 0, int MINW = 2,
          bool STRIP = false>
int launch(const void* in, const void* wp, const float* bp, Os2dRangeFlag status, void* out, int NB, int G, int CoutStore, int H,
           int W, hipStream_t stream) {
  constexpr int R = KS / 2;
  constexpr int NT = WN * NI * 32;
  const int Ws = os2d_ws(W), PLANE = os2d_plane(H, W);
  int SP = 0, TPS = 0, NS = 1;
  os2d_conv_strips(W, R, &NS, &SP);     // strips of equal width, at most 256 output columns each
  const int HALO = STRIP ? R * SP + R : R * Ws + R;
  const int SLAB = NT + 2 * HALO;
  constexpr int NTHR = 64 * WM * WN;
  if (NBPF == 0) {  // pick the slab-prefetch depth: 8 units/thread up to W = 124 (fewer registers), 12 up to W = 209
    if (2 * SLAB <= 8 * NTHR) return launch<KS, MTP, MT, WM, WN, NI, SS, RELU, OUT_MODE, TERMS, 8, MINW, STRIP>(in, wp, bp, status, out, NB, G, CoutStore, H, W, stream);
    if (2 * SLAB <= 12 * NTHR) return launch<KS, MTP, MT, WM, WN, NI, SS, RELU, OUT_MODE, TERMS, 12, MINW, STRIP>(in, wp, bp, status, out, NB, G, CoutStore, H, W, stream);
    os2d_set_error("conv%dx%d (f16x3): feature map too wide for the input-slab prefetch (W=%d)", KS, KS, W);
    return -3;
  }
  const size_t lds = (size_t)(2 * SS * 2 * (TERMS == 3 ? 2 : 1) * MT + 2 * SLAB) * 16;
  if (lds > 160 * 1024) {
    os2d_set_error("conv%dx%d (f16x3): LDS budget exceeded (%zu B, W=%d)", KS, KS, lds, W);
    return -3;
  }
  auto kern = conv_f16x3_kernel<KS, MTP, MT, WM, WN, NI, SS, RELU, OUT_MODE, (NBPF ? NBPF : 8), TERMS, MINW, STRIP>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(conv f16x3): %s", hipGetErrorString(e));
    return -4;
  }
  TPS = STRIP ? (H * SP + NT - 1) / NT : 0;
  const int tiles = STRIP ? NS * TPS : (H * Ws + NT - 1) / NT;
  const long long groups = (long long)tiles * NB * (MTP / MT);
  if (groups + 7 > 0x7fffffffLL) {
    os2d_set_error("conv f16x3: too many work-groups (%lld)", groups);
    return -3;
  }
  dim3 grid((unsigned)((groups + 7) / 8 * 8));  // multiple of 8: every XCD gets the same number of logical slots
  hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, stream, reinterpret_cast<const u32x4*>(in),
                     reinterpret_cast<const u32x4*>(wp), bp, status, out, G, CoutStore, H, W, PLANE, HALO, tiles, NB, SP, TPS);
  e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("conv f16x3 launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

// layer 1: 7x7, 29 input groups (225 ch), 128 out, SHB out;  layer 2: 5x5, 16 groups, 64 out, SHB out;
// layer 3: 5x5, 8 groups, P out (rows padded to 32), compact fp32 [NB][P][H*W] out.
//
// Small grids (a handful of classes: fewer work-groups of the standard shape than 1.5 per CU) use finer work-groups -
// 128 positions instead of 256 and 32 output channels per group - so that a call with ONE class (the reference's own
// calling pattern, evaluate.py:323-331) still spreads over 160 / 80 / 40 groups instead of 40 / 20 / 20 and a group's
// serial K loop issues 3 instead of 12 MFMAs per k-step.  Every output element accumulates the same products in the
// same order in both shapes, so results do not depend on which one ran (tests: small batch == slice of a large batch).
int os2d_launch_conv_f16x3(int layer, const void* in, const void* wp, const float* bp, Os2dRangeFlag status, void* out, int NB,
                           int P, int H, int W, int terms, hipStream_t stream) {
  // the last layer has its own kernel (16-row MFMA, conv3_f16x3.hip), one shape for every batch size
  if (layer == 3) return os2d_launch_conv3_f16x3(in, wp, bp, out, NB, P, H, W, stream);
  if (layer != 1 && W > OS2D_MAX_W_LINEAR5) {    // wider than the linear slab takes: column strips, one shape for every batch size
    if (layer == 2) return launch<5, 64, 64, 1, 4, 2, 7, true, 0, 3, 0, 2, true>(in, wp, bp, status, out, NB, 16, 64, H, W, stream);
    return launch<5, 32, 32, 1, 4, 2, 7, false, 2, 3, 0, 2, true>(in, wp, bp, status, out, NB, 8, P, H, W, stream);
  }
  const long long std_groups = (long long)((H * os2d_ws(W) + 255) / 256) * NB;
  if (std_groups * (layer == 1 ? 2 : 1) < 384) {  // measured crossover at 60x80: finer shapes win up to 9 classes
    if (layer == 1 && terms == 2)
      return launch<7, 128, 32, 1, 4, 1, 5, true, 0, 2>(in, wp, bp, status, out, NB, 29, 128, H, W, stream);
    switch (layer) {
      case 1: return launch<7, 128, 32, 1, 4, 1, 5, true, 0>(in, wp, bp, status, out, NB, 29, 128, H, W, stream);
      case 2: return launch<5, 64, 32, 1, 4, 1, 7, true, 0>(in, wp, bp, status, out, NB, 16, 64, H, W, stream);
      case 3: return launch<5, 32, 32, 1, 4, 1, 7, false, 2>(in, wp, bp, status, out, NB, 8, P, H, W, stream);
      default: break;
    }
  }
  if (layer == 1 && terms == 2)
    return launch<7, 128, 64, 1, 4, 2, 5, true, 0, 2>(in, wp, bp, status, out, NB, 29, 128, H, W, stream);
  switch (layer) {
    case 1: return launch<7, 128, 64, 1, 4, 2, 5, true, 0>(in, wp, bp, status, out, NB, 29, 128, H, W, stream);
    case 2: return launch<5, 64, 64, 1, 4, 2, 7, true, 0>(in, wp, bp, status, out, NB, 16, 64, H, W, stream);
    case 3: return launch<5, 32, 32, 1, 4, 2, 7, false, 2>(in, wp, bp, status, out, NB, 8, P, H, W, stream);
    default: os2d_set_error("os2d_launch_conv_f16x3: bad layer %d", layer); return -1;
  }
}

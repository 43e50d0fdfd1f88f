// The per-bin complex GEMM of the frequency-domain 7x7 layer (see spectral.hip) on the HALF-PRECISION matrix cores with
// split operands - precision "fftx3".  The fp32-MFMA version sits at the ridge of its two roofs (0.27 ms of matrix time,
// ~0.28 ms of HBM time at 64 classes, measured 0.57 ms); v_mfma_f32_32x32x16_f16 is 16x faster, three of them per product
// (hi*hi + hi*lo + lo*hi, fp32 accumulation: the arithmetic of the f16x3 mode) leave ~0.06 ms of matrix time and the kernel
// becomes a stream of its 1.1 GB of operands.
//
// As a real GEMM per bin:  [Yr; Yi] = [[Kr, -Ki], [Ki, Kr]] . [Xr; Xi]  with k = (channel, re | im) - the interleaved storage
// of a complex number IS the k order.  One k-step of the instruction = 16 k = 8 channels; lanes 0-31 take channels 0-3,
// lanes 32-63 channels 4-7 (one 16-byte unit of 8 halves = 4 complex numbers each).
//   B operand (spectra of the correlation maps): staged global -> registers -> LDS; the fp32 complex values are scaled by
//     2^xexp (|X| <= H*W by construction - every sample is <= 1 after the per-location normalisation - so 2^xexp * H * W
//     <= 65504: no overflow), split into fp16 hi + lo on the way and laid out [bin][channel group][hi|lo][pair] units;
//   A operand (weight spectra, host data): stored ONCE as [Kr, Ki] units, pre-split, row o scaled by 2^wexp[o] (its largest
//     |Kr|, |Ki| -> <= 32768); the two row types are derived in registers: [Kr, -Ki] = sign flip of the odd halves,
//     [Ki, Kr] = half swap within each dword (8 VALU operations per fragment pair against 6 matrix instructions);
//   the accumulators are multiplied by 2^-(wexp[o] + xexp) before the store, so Y is what the fp32 kernel produces (up to
//     the 2^-22 relative error of a split product) and the inverse transform is unchanged.
//
// Work decomposition (round 3; the round-2 kernel and the intermediate steps are timed in profiles/r03_spectral_gemm_*.txt):
//   work-group = 8 waves = 4 consecutive bins x ALL 128 output channels x 64 pairs, one per CU (96 KB of LDS: weight stages
//     2 x 32 KB + spectra 2 x 16 KB; both operands go global -> registers -> LDS two k-steps ahead).
//     Round 2 ran two 4-wave groups of 64 output channels each, and each fetched, scaled and split the same spectra: component-removal builds (tools/diag_spectral.sh) put the spectra loads at 0.10 of the launch's
//     0.39 ms at 64 pairs and at 2.5 of 5.6 ms at 1024;
//   wave = a 32 x 32 (pairs x output channels) tile of all FOUR bins (ot = wv & 3, pt = wv >> 2): a lane owns the 4 bins of a
//     (pair, channel) row = 32 contiguous bytes (round 2: a wave = one bin of the whole tile, 8-byte stores; the stores were
//     0.12 of the 0.39 ms, 2.3 of the 5.6 ms);
//   output spectra in QUADS of bins (include/os2d_hip.h, OS2D_SPECTRA_QUADS): Y [bins/4][pair][o][4] - a store instruction
//     writes two 1 KB runs (32 output channels x 32 B for two pairs) instead of 64 pieces 22 KB apart; the inverse transform
//     gathers 32-byte pieces at no measurable cost (its loads run a whole image ahead).  The INPUT spectra stay in rows
//     X [c][pair][bin]: in quads this kernel's loads were 0.03 ms cheaper per 64 pairs, but the forward transform's stores
//     (688 pieces of 32 bytes per image, 460 KB apart) 0.04 ms dearer (profiles/r03_spectral_layouts.txt).
// XCD-aware order: the pair tiles of a bin group (which share the weight spectra) run on one XCD at about the same time.
#include "os2d_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int SH_THR = 512;
constexpr int SH_WB = 4;        // bins per work-group
constexpr int SH_BINS = 8;      // bins per group of the packed weight layout
constexpr int SH_OH = 64, SH_NB = 64;
constexpr int SH_KC = 8;        // channels per k-step
constexpr int SH_STAGE = SH_WB * 2 * 2 * 64;   // 16-byte units of one operand stage of 64 rows: [bin][group][hi|lo][row]: 16 KB
constexpr int SH_WSTAGE = 2 * SH_STAGE;        // weights of both output-channel halves: 32 KB
constexpr int SH_WRING = 2;           // weight stages in LDS: k-step s + 1 is stored while k-step s is multiplied (both operands
                                      // run two k-steps ahead in registers)

#define SH_MM(x_, y_, acc_) __builtin_amdgcn_mfma_f32_32x32x16_f16(x_, y_, acc_, 0, 0, 0)
// barrier for data exchanged through LDS that leaves the wave's global loads in flight (see fft.hip)
__device__ __forceinline__ void sh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XQ: the input spectra in quads of bins too - X [NBINS/4][NB][Cpad][4] (what the matrix-product forward transform of
// dft_mfma.hip writes, 4 channels x 4 bins = 128 contiguous bytes per work-group iteration): a k-step of 8 channels of one pair
// is ONE 256-byte run, 8 lanes of a wave walk it in two 16-byte loads each - instead of the 32-byte pieces 22 KB apart of the
// row layout X [C][NB][NBINS] (round 3: 2.5 of the kernel's 5.6 ms at 1024 pairs were spent on those).
template <bool XQ>
__global__ __launch_bounds__(SH_THR, 1) void spectral_gemm_f16_kernel(const u32x4* w16,            // [G][2][KS][8][2][2][64] units
                                                                      const float* __restrict__ wscale,  // [128] 2^-wexp[o]
                                                                      const f32x2* __restrict__ X,       // [C][NB][NBINS] | quads
                                                                      f32x2* __restrict__ Y,             // [NBINS/4][NB][Cout][4] (XQ: blocked)
                                                                      int NB, int C, int Cpad, int Cout, int NBINS, int G, float xscale,
                                                                      int nunits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* ldsW = reinterpret_cast<u32x4*>(smem);                 // [SH_WRING][SH_WSTAGE]
  u32x4* ldsX = ldsW + SH_WRING * SH_WSTAGE;                    // [2][SH_STAGE]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hw = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = gridDim.x >> 3;
  const int lidx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);   // XCD-aware order
  if (lidx >= nunits) return;
  constexpr int NBH = SH_BINS / SH_WB;
  const int bh = lidx % NBH, lg = lidx / NBH;
  const int nbt = (NB + SH_NB - 1) / SH_NB;
  const int bt = lg % nbt, g = lg / nbt;
  const int nb0 = bt * SH_NB, bin0 = g * SH_BINS + bh * SH_WB;
  const int KS = (C + SH_KC - 1) / SH_KC;
  const int ot = wv & 3, pt = wv >> 2;        // the wave's tile: output channels 32 ot .., pairs 32 pt ..
  const int wq = wv & 3, wh = wv >> 2;        // its share of the staging: quarter wq of output-channel half wh / channels 2 wh ..

  f32x16 yr[SH_WB], yi[SH_WB];
#pragma unroll
  for (int j = 0; j < SH_WB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      yr[j][r] = 0.f;
      yi[j][r] = 0.f;
    }

  // weights of k-step s for this work-group's 4 bins: 1024 contiguous units per output-channel half (waves 0-3 stage half 0,
  // waves 4-7 half 1)
  const u32x4* wbase = w16 + ((size_t)(g * 2 + wh) * KS) * (SH_BINS * 256) + bh * SH_STAGE;
  // REGISTER staging (round 4).  The LDS-DMA of round 3 (global_load_lds) looked like a 2-step-deep pipeline and was none: the
  // compiler treats a FLAT-encoded instruction that may touch LDS as returning out of order, and from the first DMA on every
  // vmcnt wait it inserts is vmcnt(0).  The one wait this loop needs - for the spectra registers before their conversion - thus
  // also waited for the weight DMA and the spectra loads issued at the top of the SAME k-step: each of the 29 k-steps paid the
  // full load latency minus its 0.75 us of matrix work (2.3 us per k-step at 1024 pairs; 1.07 with every load removed,
  // profiles/r04/spectral_gemm_components.txt).  With plain loads the counter is exact: a k-step requests the weights and the
  // spectra of step S + 2 into one of two register sets, multiplies step S, then moves the set requested a step ago (S + 1)
  // into the other LDS stage - vmcnt(6): this step's 4 + 2 requests stay in flight through the barrier and the next k-step's
  // matrix work.  Costs 32 registers and 4 LDS stores per thread and k-step; the LDS ring shrinks to 2 stages (96 KB in all).
  u32x4 wra[4], wrb[4];
#define SH_LOAD_W(S, wr)                                                                                            \
  {                                                                                                                 \
    const u32x4* src_ = wbase + (size_t)(S) * (SH_BINS * 256);                                                      \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) wr[k_] = src_[(wq * 4 + k_) * 64 + lane];                      \
  }
#define SH_LOAD_W1(S, wr, K) wr[K] = (wbase + (size_t)(S) * (SH_BINS * 256))[(wq * 4 + (K)) * 64 + lane];
#define SH_STORE_W(S, wr)                                                                                           \
  {                                                                                                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_)                                                                \
        ldsW[((S) % SH_WRING) * SH_WSTAGE + wh * SH_STAGE + (wq * 4 + k_) * 64 + lane] = wr[k_];                    \
  }
  // spectra of k-step s: this thread owns pair xn, bins 2 xj / 2 xj + 1 and TWO channels (xg * 4 + wh * 2 + {0, 1}).  A wave
  // covers 16 pairs x (2 bin pairs x 2 channel groups): with one pair per lane a load instruction would touch 64 rows 5 MB
  // apart and spend its time in address translation, not in the memory system
  // (XQ: a wave covers 8 pairs x the 8 sixteen-byte pieces (channel pair, bin pair) of a k-step's 256-byte run; the thread's
  // second load is the neighbouring channel, 32 bytes on: wh = which channel pair of the group comes from the LANE then)
  const int xn = XQ ? wv * 8 + (lane >> 3) : wq * 16 + (lane & 15);
  const int xj = XQ ? (lane & 1) : (lane >> 4) & 1, xg = XQ ? (lane >> 2) & 1 : lane >> 5;
  const int xwh = XQ ? (lane >> 1) & 1 : wh;
  const bool xn_ok = nb0 + xn < NB;
  // XQ: both spectra in blocks of 64 pairs = this work-group's pair tile: [pair tile][bins / 4][pair in tile][channel][4]
  // (dft_mfma.h, dft_spectra_pair0): the operands of a work-group are one contiguous slab whatever the batch size
  const int nbl = min(SH_NB, NB - nb0);                                        // pairs in this tile (the last one may be short)
  const size_t qblk = XQ ? (size_t)nb0 * (NBINS >> 2) + (size_t)(bin0 >> 2) * nbl : 0;   // in pairs: start of (tile, quad)
  const f32x2* xrow = XQ ? X + ((qblk + min(xn, nbl - 1)) * Cpad) * 4 + 2 * xj
                         : X + (size_t)min(nb0 + xn, NB - 1) * NBINS + bin0 + 2 * xj;
  const size_t xcs = XQ ? 4 : (size_t)NB * NBINS;      // channel stride: 4 bins | X [C][NB][NBINS]
  u32x4 pfa[2], pfb[2];     // two k-steps of spectra in flight (even / odd k-steps)
#define SH_LOAD_X(S, pfx)                                                                                           \
  {                                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                              \
      const int c_ = min((S)*SH_KC + xg * 4 + xwh * 2 + i_, C - 1);                                                 \
      pfx[i_] = *reinterpret_cast<const u32x4*>(xrow + (size_t)c_ * xcs);                                           \
    }                                                                                                               \
  }
#define SH_LOAD_X1(S, pfx, I)                                                                                       \
  {                                                                                                                 \
    const int c_ = min((S)*SH_KC + xg * 4 + xwh * 2 + (I), C - 1);                                                  \
    pfx[I] = *reinterpret_cast<const u32x4*>(xrow + (size_t)c_ * xcs);                                              \
  }
  // (the four values of a store as ONE vector conversion: v_cvt_pk_f16_f32 converts two values per instruction - a third fewer
  // conversion instructions than value by value; same roundings)
#define SH_STORE_X(S, pfx)                                                                                          \
  {                                                                                                                 \
    _Pragma("unroll") for (int b2_ = 0; b2_ < 2; ++b2_) {                                                           \
      f32x4 v4_;                                                                                                    \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                            \
        const bool ok_ = xn_ok && (S)*SH_KC + xg * 4 + xwh * 2 + i_ < C;                                            \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                          \
          const unsigned raw_ = pfx[i_][2 * b2_ + p_];   /* (scalar copy first: see the ext-vector note in corr_f16x3.hip) */ \
          v4_[2 * i_ + p_] = ok_ ? __uint_as_float(raw_) * xscale : 0.f;                                            \
        }                                                                                                           \
      }                                                                                                             \
      const half4 h_ = __builtin_convertvector(v4_, half4);                                                         \
      const u32x2 hp_ = __builtin_bit_cast(u32x2, h_);     /* lo halves: one mixed-precision instruction each (os2d_common.h) */ \
      const u32x2 lp_ = {os2d_split_lo_pair(v4_[0], v4_[1], hp_[0]), os2d_split_lo_pair(v4_[2], v4_[3], hp_[1])};    \
      const half4 l_ = __builtin_bit_cast(half4, lp_);                                                              \
      char* dst_ = reinterpret_cast<char*>(ldsX + ((S)&1) * SH_STAGE + (((2 * xj + b2_) * 2 + xg) * 2) * 64 + xn) + xwh * 8; \
      *reinterpret_cast<half4*>(dst_) = h_;              /* channels 2 wh, 2 wh + 1 of the unit (re, im each) */      \
      *reinterpret_cast<half4*>(dst_ + 64 * 16) = l_;                                                               \
    }                                                                                                               \
  }
  // The fragments of bin j + 1 are requested before the matrix instructions of bin j are issued, the scheduler fenced to that order
  // (round 6: left to the compiler every bin's three reads sat right in front of its first matrix instruction - four exposed LDS
  // latencies per k-step and wave; 2.84 against 2.91 ms at 1024 pairs, 0.275 against 0.282 at 64).  Behind the matrix instructions
  // of bins 0, 1, 2 the k-step's six global requests (step SN into the sets WN, PN) go out two at a time: a vector-memory
  // instruction occupies its wave's issue slot for 60 - 180 cycles (MI355X guide, LDS-DMA issue cost), and as one burst in front of
  // the k-step - rounds 4 - 5 - all eight waves paid that with the matrix pipes idle (0.270 against 0.279 ms at 64 pairs, 2.76
  // against 2.81 at 1024: profiles/r06/gemm_burst_vs_spread_requests.txt).  Same order as the burst: the wait in front of the
  // stores stays vmcnt(6).
#define SH_REQUESTS_HOOK(SN, WN, PN, J)                                                                             \
  if ((J) == 0) {                                                                                                   \
    SH_LOAD_W1(SN, WN, 0)                                                                                           \
    SH_LOAD_W1(SN, WN, 1)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  } else if ((J) == 1) {                                                                                            \
    SH_LOAD_W1(SN, WN, 2)                                                                                           \
    SH_LOAD_W1(SN, WN, 3)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  } else if ((J) == 2) {                                                                                            \
    SH_LOAD_X1(SN, PN, 0)                                                                                           \
    SH_LOAD_X1(SN, PN, 1)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
#define SH_COMPUTE(S, SN, WN, PN)                                                                                         \
  {                                                                                                                 \
    /* [half][bin][group = hw][hi|lo][o 64] and [bin][group][hi|lo][pair 64] */                                     \
    const u32x4* aB = ldsW + ((S) % SH_WRING) * SH_WSTAGE + (ot >> 1) * SH_STAGE + (hw * 2) * 64 + (ot & 1) * 32 + l31; \
    const u32x4* bB = ldsX + ((S)&1) * SH_STAGE + (hw * 2) * 64 + pt * 32 + l31;                                    \
    half8 bq[2][2];                                                                                                 \
    u32x4 kq[2][2];                                                                                                 \
    bq[0][0] = *reinterpret_cast<const half8*>(bB);                                                                 \
    bq[0][1] = *reinterpret_cast<const half8*>(bB + 64);                                                            \
    kq[0][0] = aB[0];                                                                                               \
    kq[0][1] = aB[64];                                                                                              \
    _Pragma("unroll") for (int j = 0; j < SH_WB; ++j) {                                                             \
      if (j + 1 < SH_WB) {                                                                                          \
        bq[(j + 1) & 1][0] = *reinterpret_cast<const half8*>(bB + (j + 1) * 256);                                   \
        bq[(j + 1) & 1][1] = *reinterpret_cast<const half8*>(bB + (j + 1) * 256 + 64);                              \
        kq[(j + 1) & 1][0] = aB[(j + 1) * 256];                                                                     \
        kq[(j + 1) & 1][1] = aB[(j + 1) * 256 + 64];                                                                \
      }                                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      const half8 bhf = bq[j & 1][0], blf = bq[j & 1][1];                                                           \
      const u32x4 kh = kq[j & 1][0], kl = kq[j & 1][1]; /* [Kr, Ki] x 4 channels, hi and lo */                      \
      u32x4 rh, rl, ih, il;                                                                                         \
      _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                                               \
        rh[d] = kh[d] ^ 0x80000000u; /* [Kr, -Ki] */                                                                \
        rl[d] = kl[d] ^ 0x80000000u;                                                                                \
        ih[d] = __builtin_amdgcn_alignbit(kh[d], kh[d], 16); /* [Ki, Kr] */                                         \
        il[d] = __builtin_amdgcn_alignbit(kl[d], kl[d], 16);                                                        \
      }                                                                                                             \
      const half8 arh = __builtin_bit_cast(half8, rh), arl = __builtin_bit_cast(half8, rl);                         \
      const half8 aih = __builtin_bit_cast(half8, ih), ail = __builtin_bit_cast(half8, il);                         \
      /* the spectra are the ROW operand: the accumulator registers run over the pairs, the lanes over the output channels */ \
      yr[j] = SH_MM(bhf, arl, yr[j]);                                                                               \
      yr[j] = SH_MM(blf, arh, yr[j]);                                                                               \
      yr[j] = SH_MM(bhf, arh, yr[j]);                                                                               \
      yi[j] = SH_MM(bhf, ail, yi[j]);                                                                               \
      yi[j] = SH_MM(blf, aih, yi[j]);                                                                               \
      yi[j] = SH_MM(bhf, aih, yi[j]);                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      SH_REQUESTS_HOOK(SN, WN, PN, j)                                                                               \
    }                                                                                                               \
  }
  // k-step S: requests of step S + 2 -> set N (weights WN, spectra PN); multiply step S; the set requested a step ago (WC, PC:
  // step S + 1) -> the other LDS stage; barrier.  The sets alternate, so the loop runs two k-steps per pass.
  // Every request is unconditional, its k-step index clamped to the last one: behind "if (S + 2 < KS)" the compiler cannot tell how
  // many loads are in flight when it needs the older set and waits for all of them (vmcnt(0) instead of vmcnt(6)); the store of a
  // step beyond the last goes to the stage nobody reads.
#define SH_STEP(S, WC, PC, WN, PN)                                                                                  \
  {                                                                                                                 \
    const int sn_ = min((S) + 2, KS - 1);                                                                           \
    SH_COMPUTE(S, sn_, WN, PN)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    SH_STORE_W((S) + 1, WC)                                                                                         \
    SH_STORE_X((S) + 1, PC)                                                                                         \
    sh_lds_barrier();                                                                                               \
  }
  SH_LOAD_W(0, wrb)
  SH_LOAD_X(0, pfb)
  SH_LOAD_W(min(1, KS - 1), wra)
  SH_LOAD_X(min(1, KS - 1), pfa)
  SH_STORE_W(0, wrb)
  SH_STORE_X(0, pfb)
  sh_lds_barrier();
  int s = 0;
  for (; s + 1 < KS; s += 2) {
    SH_STEP(s, wra, pfa, wrb, pfb)
    SH_STEP(s + 1, wrb, pfb, wra, pfa)
  }
  if (s < KS) SH_STEP(s, wra, pfa, wrb, pfb)
#undef SH_STEP
#undef SH_REQUESTS_HOOK
#undef SH_LOAD_W1
#undef SH_LOAD_X1
#undef SH_LOAD_W
#undef SH_STORE_W
#undef SH_COMPUTE
#undef SH_LOAD_X
#undef SH_STORE_X

  // ---- epilogue: undo the operand scales; Y[bin / 4][pair][o][bin % 4]: the lane's 4 bins are 32 contiguous bytes, the 32
  // lanes of a half-wave (32 consecutive output channels of one pair) 1 KB
  const float inv_x = 1.0f / xscale;
  const int o = ot * 32 + l31;
  if (o < Cout) {
    const float sc = wscale[o] * inv_x;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nb = nb0 + pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hw;
      if (nb < NB) {
        float4* dst = XQ ? reinterpret_cast<float4*>(Y + ((qblk + (nb - nb0)) * Cout + o) * 4)
                         : reinterpret_cast<float4*>(Y + (((size_t)(bin0 >> 2) * NB + nb) * Cout + o) * 4);
        dst[0] = make_float4(yr[0][r] * sc, yi[0][r] * sc, yr[1][r] * sc, yi[1][r] * sc);
        dst[1] = make_float4(yr[2][r] * sc, yi[2][r] * sc, yr[3][r] * sc, yi[3][r] * sc);
      }
    }
  }
}

}  // namespace

// bytes of the split weight spectra: [NBINS/8][2][KS][8][2][2][64] units of 16 B, followed by the 128 row scales (floats)
size_t os2d_spectral_weight16_size(int C, int NBINS) {
  const size_t KS = (size_t)(C + SH_KC - 1) / SH_KC;
  return (size_t)(NBINS / SH_BINS) * 2 * KS * SH_BINS * 256 * 16 + 128 * sizeof(float);
}

// largest power-of-two scale of the input spectra that cannot overflow fp16: |X| <= H * W (samples of one transform window)
float os2d_spectral_xscale_for(int H, int W) {
  float s = 1.0f;
  while (s * 2.0f * (float)H * (float)W <= 65504.0f) s *= 2.0f;
  return s;
}

int os2d_launch_spectral_gemm_f16(const void* w16, const float* X, float* Y, int NB, int C, int Cout, int NBINS, float xscale,
                                  int x_quads, int Cpad, hipStream_t stream) {
  if (NBINS % SH_BINS || Cout > 2 * SH_OH) {
    os2d_set_error("spectral_gemm_f16: NBINS %d must be a multiple of %d and Cout %d <= %d", NBINS, SH_BINS, Cout, 2 * SH_OH);
    return -3;
  }
  const int G = NBINS / SH_BINS, nbt = (NB + SH_NB - 1) / SH_NB;
  const size_t lds = (size_t)(SH_WRING * SH_WSTAGE + 2 * SH_STAGE) * 16;
  if (x_quads && Cpad < C) {
    os2d_set_error("spectral_gemm_f16: channel stride %d < C %d", Cpad, C);
    return -1;
  }
  auto kern = x_quads ? spectral_gemm_f16_kernel<true> : spectral_gemm_f16_kernel<false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(spectral_gemm_f16): %s", hipGetErrorString(e));
    return -4;
  }
  const long long units = (long long)G * nbt * (SH_BINS / SH_WB);
  const size_t KS = (size_t)(C + SH_KC - 1) / SH_KC;
  const float* wscale = reinterpret_cast<const float*>(static_cast<const char*>(w16) + (size_t)G * 2 * KS * SH_BINS * 256 * 16);
  dim3 grid((unsigned)((units + 7) / 8 * 8));
  hipLaunchKernelGGL(kern, grid, dim3(SH_THR), lds, stream, static_cast<const u32x4*>(w16), wscale,
                     reinterpret_cast<const f32x2*>(X), reinterpret_cast<f32x2*>(Y), NB, C, Cpad, Cout, NBINS, G, xscale, (int)units);
  e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("spectral_gemm_f16 launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

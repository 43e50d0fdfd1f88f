// All-pairs feature correlation (reference os2d/modeling/head.py:339-350) + TransformNet input normalisation
// (head.py:650) on the half-precision matrix cores with fp32-equivalent accuracy ("f16x3", see conv_f16x3.hip).
//
// Operands are pre-split once per image / per class into the split-half blocked layout, both scaled by 2^12
// (values are L2-normalised, |x| <= 1, so hi <= 4096 and lo stays a normal fp16 number):
//   fs [A][C/8][hi|lo][H*W]  units of 8 halves   image features, ALREADY L2-normalised over channels (split_fm_kernel)
//   qs [B][C/8][hi|lo][256]  units of 8 halves   class features in the x-major channel order, rows 225..255 zero
// Both are padded with ZERO channel groups to a multiple of 4 groups (one K chunk of 32 channels: os2d_corr_groups(C)).
// One v_mfma_f32_32x32x16_f16 k-step = 16 channels = two 8-channel groups (lanes 0-31 / 32-63).
//
// Work-group = 512 threads (8 waves as 2 x 4), tile = 256 rows (one class) x 256 positions, wave tile 128 x 64.
// K runs in chunks of 32 channels through double-buffered LDS with the register prefetch pipeline of the other
// kernels.  With 256-wide tiles a group moves 2 MB per 134 MFLOP (3 x that in executed half-precision FLOPs), i.e. the
// kernel needs ~5 TB/s of L2->LDS traffic at full matrix rate: it is the most bandwidth-hungry kernel of the head.
// Outputs: corr [A*B][225][H*W] fp32 (for the resampler) and the relu+L2-normalised tensor in SHB layout.
//
// STACK (round 4): with a class per 256-row tile, rows 225..255 - 12 % of the matrix instructions - are
// padding.  The frequency-domain route (no normalised SHB tensor to write) packs the classes along M instead: class b owns the
// stacked rows [228 b, 228 b + 225) (a stride of 228 = 4 x 57 keeps every class 4-row aligned: 1.3 % padding) and a work-group
// takes 256 CONSECUTIVE stacked rows, whatever classes they belong to - B x 228 / 256 row tiles instead of B.  The class operand
// keeps its per-class layout: the LDS-DMA already takes a per-lane global offset, so a lane simply fetches row p of class b.
// The per-position sum of relu^2 over a class's 225 rows now crosses wave and work-group boundaries, and the result must not
// depend on where in a batch a class sits (slices of a batch are bit-identical to the class on its own).  So the sum is made
// ORDER-INDEPENDENT: a lane adds the 4 rows of an accumulator run in fp32 (a run = rows 4 k .. 4 k + 3 of ONE class, the same
// four values in the same order for every placement), converts the group sum to 2^-44 fixed point (exact for sums >= 2^-20;
// below that the dropped bits are < 2^-44 absolute) and accumulates integers: lanes, waves and work-groups combine with 64-bit
// integer atomics, which commute and associate.  corr_norm_finalize_kernel turns the sums into 1 / (sqrt(s) + 1e-6) and
// clears them for the next call.
#include "os2d_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define U32X4_ZERO (u32x4{0u, 0u, 0u, 0u})
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int GC = 4;                    // 8-channel groups per K chunk (32 channels)
constexpr int STACK_STRIDE = 228;        // stacked rows per class (225 rounded up to a multiple of 4)
constexpr int TM = 256;                  // rows per work-group: two waves of 128
constexpr int WNW = 4;                   // waves along the positions: 8 waves of 128 x (32 NI)

// LDS-DMA as inline assembly (round 6): 64 lanes x 16 bytes from (scalar base + the lane's 32-bit byte offset) to the 1 KB of LDS at
// ``lds`` (M0).  The builtin (__builtin_amdgcn_global_load_lds) is a FLAT-encoded instruction that "may access LDS" to LLVM's
// wait-count pass, which then treats every later counter wait as out of order: with one such instruction in a loop, every wait it
// inserts is vmcnt(0) / lgkmcnt(0) - a fragment read requested a group ahead was waited for where it was issued, and the software
// pipeline of the matrix loop below was none (the ISA of the first build).  As assembly the pass does not see it; the counters of
// the DMAs are handled by hand (cf_barrier_drain), the LDS reads keep exact compiler-made counts.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void cf_dma16(const void* sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
// end of a K chunk: this wave's DMAs have landed and its fragment reads have returned (vmcnt(0), lgkmcnt(0)), then the barrier
// orders both for every wave of the group
__device__ __forceinline__ void cf_barrier_drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One tile: 256 rows x (128 NI) positions.  NI = 32-column tiles per wave: 2 -> 256 positions (the throughput shape: 128
// accumulators per lane, two waves per SIMD); 1 -> 128 positions (half the matrix instructions per K chunk for the same class
// operand: for a handful of classes, where a group's serial K loop is what a call waits for, and for the TAIL of a launch, see
// the kernel).  Every output accumulates the same products in the same order in both shapes.
// (Retired shapes, measured slower: 4 waves of 128 x 128 with the whole register file; 128-row half tiles, two groups per CU -
// tools/patches/corr_f16x3_variants.patch, profiles/r04/corr_fixed_cost.txt.)
template <int NI, bool STACK>
__device__ __forceinline__ void corr_tile(const u32x4* fs,  // [A][CGP][2][HW]  (no __restrict__, see conv_f16x3.hip)
                                          const u32x4* qs,  // [B][CGP][2][256]
                                          float* __restrict__ corr, char* __restrict__ rshb,
                                          float* __restrict__ invn /*[A*B][HW] 1/(norm+eps) or NULL*/,
                                          unsigned long long* __restrict__ sumfx /*STACK: [A*B][HW] fixed-point sums*/,
                                          int B, int CGP /*channel groups, padded to a multiple of GC*/, int H, int W, int PLANE,
                                          float unscale, int a, int b /*class | STACK: row tile of the stacked matrix*/, int n0,
                                          u32x4* smem16, unsigned long long (*red)[256]) {
  constexpr int KC = GC;
  constexpr int NTHR = 64 * 2 * WNW;
  constexpr int NT = WNW * 32 * NI;      // positions per work-group
  constexpr int AUNITS = KC * 2 * TM;    // 16-byte units of one class-operand chunk
  constexpr int BUNITS = KC * 2 * NT;    // 16-byte units of one image-operand chunk
  constexpr int NPF = AUNITS / NTHR;     // class units per thread
  constexpr int NPFB = BUNITS / NTHR;    // image units per thread
  static_assert((NPF > NPFB ? NPF : NPFB) <= (KC / 2) * 4, "one DMA piece per matrix-instruction group of a chunk");
  u32x4* ldsA = smem16;                 // [2][AUNITS]
  u32x4* ldsB = smem16 + 2 * AUNITS;    // [2][BUNITS]

  const int HW = H * W;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hw = lane >> 5;
  const int wm = wid / WNW, wn = wid % WNW;  // wave tile: rows [wm*128,+128), cols [wn*32*NI,+32*NI)
  const int nb = a * B + b;
  const int R0 = b * TM;                         // STACK: first stacked row of this work-group

  f32x16 acc[4][NI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int bfirst = STACK ? R0 / STACK_STRIDE : b;                 // first class this work-group touches
  const u32x4* qb = qs + (size_t)bfirst * CGP * 2 * 256;
  const u32x4* fa = fs + (size_t)a * CGP * 2 * HW;
  const int nchunks = CGP / KC;
  // ---- staging: global -> LDS directly (LDS-DMA, global_load_lds_dwordx4): no staging registers, no ds_write pass.
  // A wave instruction writes 64 consecutive 16-byte units starting at a wave-uniform LDS address, which is exactly how
  // a chunk is laid out (unit i = row i/256 = (group, part), column i%256; 64 consecutive threads = 64 consecutive
  // columns of one row).  Everything but the lane's column is WAVE-UNIFORM and kept in scalar registers: the global
  // address is (scalar base of the chunk / row) + (per-lane byte offset, loop invariant), the LDS address goes to M0
  // from a scalar - one DMA costs a few SALU instructions and no VALU (the per-lane 64-bit pointer + v_readfirstlane
  // form cost ~14 instructions and an M0 dependency stall each).  Both operands are padded with zero channel groups to a
  // whole number of chunks (split_fm / split_qp), columns past H*W read the last valid column instead: their products
  // land in accumulator columns that are never stored.
  typedef void __attribute__((address_space(3))) * lptr_t;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as a scalar
  // STACK: this thread always stages the same stacked row (its units are 256 apart): row p of class bfirst + d, clamped to the
  // last class (rows 225 .. 255 of every class are zero in the operand: the 3 padding rows of the stride need nothing else)
  // (unit u of a class chunk in LDS = [(group, part) = u / TM][row u % TM]; in memory a (group, part) row is 256 units long)
  const int rowA = (wv * 64 + lane) & (TM - 1), RA = min(R0 + rowA, B * STACK_STRIDE - 1);
  const int bA = RA / STACK_STRIDE, pA = RA - bA * STACK_STRIDE;
  const unsigned voffA = STACK ? (unsigned)(bA - bfirst) * (unsigned)(CGP * 2 * 256 * 16) + (unsigned)pA * 16u : (unsigned)rowA * 16u;
  const int colB = (wv * 64) % NT + lane;
  const unsigned voffB = (unsigned)min(colB, HW - 1 - n0) * 16u;
  const char* baseA0 = reinterpret_cast<const char*>(qb);
  const unsigned ldsA0 = (unsigned)(uintptr_t)(lptr_t)ldsA, ldsB0 = (unsigned)(uintptr_t)(lptr_t)ldsB;      // LDS byte addresses (M0)
  const char* baseB0 = reinterpret_cast<const char*>(fa) + ((size_t)((wv * 64) / NT) * HW + n0) * 16;
#define CF_DMA1(T, K)                                                                                             \
  {                                                                                                               \
    if ((K) < NPF) {                                                                                              \
      const char* ga_ = baseA0 + ((size_t)((T)*KC * 2 + ((K)*NTHR + wv * 64) / TM) * 256) * 16;                    \
      cf_dma16(ga_, voffA, ldsA0 + (unsigned)((((T)&1) * AUNITS + (K)*NTHR + wv * 64) * 16));                      \
    }                                                                                                             \
    if ((K) < NPFB) {                                                                                             \
      const char* gb_ = baseB0 + ((size_t)((T)*KC * 2 + ((K)*NTHR) / NT) * HW) * 16;                              \
      cf_dma16(gb_, voffB, ldsB0 + (unsigned)((((T)&1) * BUNITS + (K)*NTHR + wv * 64) * 16));                      \
    }                                                                                                             \
  }
#define CF_DMA(T)                                                                                                 \
  {                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < (NPF > NPFB ? NPF : NPFB); ++k) CF_DMA1(T, k)                            \
  }
  // The matrix loop, software-pipelined in the source (round 6).  A K chunk is 8 GROUPS of 3 NI matrix instructions - group g =
  // (k-step g >> 2, row tile g & 3): one class fragment pair (hi, lo) against the k-step's NI image fragment pairs.  Left to the
  // compiler, every group's class fragments were read right in front of its first matrix instruction (ds_read_b128, s_waitcnt
  // lgkmcnt(0), v_mfma ...: the ISA of rounds 2 - 5) - eight exposed LDS latencies per chunk and wave, covered only by the other
  // wave of the SIMD, and the matrix pipe 55 - 58 % busy.  Now the fragments of group g + 1 (and of the next k-step's image
  // columns) are requested into the OTHER of two register sets before the matrix instructions of group g are issued, and the
  // scheduler is fenced to that order (as in dft_mfma.h, round 5).  The barrier that ends a chunk sits in front of the LAST group's
  // matrix instructions: that group's fragments have landed (they were requested a group earlier; the barrier's lgkmcnt(0)), so
  // every wave is done reading the stage, the next chunk's first fragments are requested right behind the barrier, and the last
  // group's 3 NI instructions cover their latency.  Same products, same order per accumulator: same bits.
  half8 fa_[2][2], fb_[2][NI][2];      // [set][hi | lo], [set = k-step parity][ni][hi | lo]
#define CF_READ_A(T, G, SET)                                                                                      \
  {                                                                                                               \
    const u32x4* aB_ = ldsA + ((T)&1) * AUNITS + wm * 128 + l31 + ((G)&3) * 32;                                    \
    fa_[SET][0] = *reinterpret_cast<const half8*>(aB_ + ((2 * ((G) >> 2) + hw) * 2 + 0) * TM);                    \
    fa_[SET][1] = *reinterpret_cast<const half8*>(aB_ + ((2 * ((G) >> 2) + hw) * 2 + 1) * TM);                    \
  }
#define CF_READ_B(T, KS, SET)                                                                                     \
  {                                                                                                               \
    const u32x4* bB_ = ldsB + ((T)&1) * BUNITS + wn * (32 * NI) + l31;                                            \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                           \
      fb_[SET][ni][0] = *reinterpret_cast<const half8*>(bB_ + ((2 * (KS) + hw) * 2 + 0) * NT + ni * 32);          \
      fb_[SET][ni][1] = *reinterpret_cast<const half8*>(bB_ + ((2 * (KS) + hw) * 2 + 1) * NT + ni * 32);          \
    }                                                                                                             \
  }
#define CF_MMA(G)                                                                                                 \
  {                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                           \
      acc[(G)&3][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[(G)&1][1], fb_[(G) >> 2][ni][0], acc[(G)&3][ni], 0, 0, 0); \
      acc[(G)&3][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[(G)&1][0], fb_[(G) >> 2][ni][1], acc[(G)&3][ni], 0, 0, 0); \
      acc[(G)&3][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[(G)&1][0], fb_[(G) >> 2][ni][0], acc[(G)&3][ni], 0, 0, 0); \
    }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
  }
  static_assert(KC == 4, "a chunk is two k-steps x four row tiles: eight groups");
  // The DMA of chunk t+1 is issued IN PIECES between the MFMA groups of chunk t (one class unit + one image unit per
  // thread after each of the first four groups): eight waves bursting 72 KB of loads right after the barrier queue up
  // behind each other in the CU's load path; spread out, the issue slots hide behind the matrix pipe.  The barrier of a
  // chunk drains the wave's own DMAs (cf_barrier_drain) and then orders them for everybody's fragment reads of the next chunk.
  CF_DMA(0)
  cf_barrier_drain();
  CF_READ_B(0, 0, 0)
  CF_READ_A(0, 0, 0)
  // (the last chunk is peeled: behind "if (more)" the wait in front of the last group's matrix instructions became lgkmcnt(0) -
  // it waited for the next chunk's fragments it was meant to cover)
  for (int t = 0; t + 1 < nchunks; ++t) {
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      CF_READ_A(t, g + 1, (g + 1) & 1)
      if (g == 2) CF_READ_B(t, 1, 1)         // the second k-step's image columns, a group and a half ahead
      CF_MMA(g)
      if (g < (NPF > NPFB ? NPF : NPFB)) {
        CF_DMA1(t + 1, g)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cf_barrier_drain();
    CF_READ_B(t + 1, 0, 0)
    CF_READ_A(t + 1, 0, 0)
    CF_MMA(7)
  }
  {
    const int t = nchunks - 1;
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      CF_READ_A(t, g + 1, (g + 1) & 1)
      if (g == 2) CF_READ_B(t, 1, 1)
      CF_MMA(g)
    }
    CF_MMA(7)
  }
#undef CF_MMA
#undef CF_READ_B
#undef CF_READ_A
#undef CF_DMA
#undef CF_DMA1

  // ---- epilogue (features were normalised before the split, so only the 2^-24 operand scale is undone).
  // A lane holds a 32x32 block as 4 consecutive rows x ONE column per register quadruple: stored as is, that is 128 dword
  // stores for the correlation and 64 eight-byte stores for the normalised tensor per lane, and the epilogue (17 % of a
  // work-group's life) is bound by issuing them.  Both go out as 16-byte stores instead:
  //   * correlation: every 4x4 block (4 rows in 4 registers x the 4 lanes of a quad) is transposed inside the quad with
  //     two DPP exchange stages, after which a lane owns ONE row and 4 consecutive columns - one dwordx4 store
  //     (needs H*W % 4 == 0 for the alignment; otherwise the scalar stores);
  //   * normalised tensor: lanes l and l+32 hold channels 0-3 / 4-7 of the same 8-channel unit; they swap halves
  //     (ds_bpermute) so that the lower lane stores the complete hi unit and the upper lane the complete lo unit.
  const int Ws = os2d_ws(W), BASE = os2d_base(W);
  const bool vec4 = (HW & 3) == 0;
  const int qc = lane & 3;                    // column of this lane inside its quad = the row it owns after the transpose
  // Sums of relu^2 over a class's rows, per position, in 2^-44 FIXED POINT - in both forms of the kernel, so that they give the
  // same inverse norms bit for bit and the launcher may pick either: a lane adds the 4 rows of an accumulator run in fp32 (rows
  // 4 k .. 4 k + 3 of ONE class, the same values in the same order wherever the class sits), converts the run's sum and adds
  // integers from there on.  Slot 1 (STACK only): the second class of the wave's 128 rows (they touch at most two: class cw0 up
  // to row `bound`, cw0 + 1 after).
  const int Rw = R0 + wm * 128, cw0 = Rw / STACK_STRIDE, bound = (cw0 + 1) * STACK_STRIDE;
  constexpr int NSL = STACK ? 2 : 1;
  unsigned fxh[NSL][NI], fxl[NSL][NI];
  bool fxbad[NSL][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int ncol0 = n0 + wn * (32 * NI) + ni * 32;
    const int n = ncol0 + l31;
    const bool nin = n < HW;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      fxh[sl][ni] = fxl[sl][ni] = 0u;
      fxbad[sl][ni] = false;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = acc[mi][ni][4 * q + k] * unscale;
          acc[mi][ni][4 * q + k] = v;
          const float rl = fmaxf(v, 0.f);
          g = fmaf(rl, rl, g);
        }
        // g <= 4 (cosines); anything else is a non-finite input: flagged, the norm then becomes NaN like the reference's
        const bool gbad = !(g < 1024.f);
        const float ga = fminf(g, 1024.f) * 1048576.f;                   // 2^20
        const unsigned gh = (unsigned)ga;
        const unsigned gl = (unsigned)((ga - (float)gh) * 16777216.f);   // the next 24 bits
        if (STACK) {
          const bool second = Rw + mi * 32 + 8 * q + 4 * hw >= bound;    // rows .. + 3: one class (4-row alignment)
          fxh[0][ni] += second ? 0u : gh;
          fxl[0][ni] += second ? 0u : gl;
          fxh[NSL - 1][ni] += second ? gh : 0u;
          fxl[NSL - 1][ni] += second ? gl : 0u;
          fxbad[0][ni] |= gbad && !second;
          fxbad[NSL - 1][ni] |= gbad && second;
        } else {
          fxh[0][ni] += gh;              // rows 225 .. 255 of the tile are zero operand rows: they add nothing
          fxl[0][ni] += gl;
          fxbad[0][ni] |= gbad;
        }
      }
      if (vec4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (copies first: __builtin_bit_cast applied directly to an ext-vector ELEMENT reads the wrong element with hipcc 7.2)
          const float f0 = acc[mi][ni][4 * q + 0], f1 = acc[mi][ni][4 * q + 1];
          const float f2 = acc[mi][ni][4 * q + 2], f3 = acc[mi][ni][4 * q + 3];
          int t0 = __float_as_int(f0), t1 = __float_as_int(f1), t2 = __float_as_int(f2), t3 = __float_as_int(f3);
          {  // stage 1: 2x2 blocks, lanes c <-> c^1 (quad_perm [1,0,3,2])
            const bool odd = qc & 1;
            const int x01 = odd ? t0 : t1, x23 = odd ? t2 : t3;
            const int y01 = __builtin_amdgcn_update_dpp(0, x01, 0xB1, 0xF, 0xF, false);
            const int y23 = __builtin_amdgcn_update_dpp(0, x23, 0xB1, 0xF, 0xF, false);
            t0 = odd ? y01 : t0;
            t1 = odd ? t1 : y01;
            t2 = odd ? y23 : t2;
            t3 = odd ? t3 : y23;
          }
          {  // stage 2: lanes c <-> c^2 (quad_perm [2,3,0,1])
            const bool up = qc & 2;
            const int x02 = up ? t0 : t2, x13 = up ? t1 : t3;
            const int y02 = __builtin_amdgcn_update_dpp(0, x02, 0x4E, 0xF, 0xF, false);
            const int y13 = __builtin_amdgcn_update_dpp(0, x13, 0x4E, 0xF, 0xF, false);
            t0 = up ? y02 : t0;
            t2 = up ? t2 : y02;
            t1 = up ? y13 : t1;
            t3 = up ? t3 : y13;
          }
          // this lane now holds row (.. + qc) at the 4 columns of its quad
          const int m = wm * 128 + mi * 32 + 8 * q + 4 * hw + qc;
          const int nq = ncol0 + (l31 & ~3);
          bool mok = m < OS2D_K;
          size_t orow = (size_t)nb * OS2D_K + m;
          if (STACK) {     // stacked row -> (class, template cell): the run of 8 rows starts in class cg (uniform), a lane may be one on
            const int Rg = Rw + mi * 32 + 8 * q, cg = Rg / STACK_STRIDE;
            int pc = Rg - cg * STACK_STRIDE + 4 * hw + qc, cc = cg;
            if (pc >= STACK_STRIDE) {
              pc -= STACK_STRIDE;
              ++cc;
            }
            mok = pc < OS2D_K && cc < B;
            orow = ((size_t)a * B + cc) * OS2D_K + pc;
          }
          if (mok && nq < HW) {
            f32x4 o4 = {__int_as_float(t0), __int_as_float(t1), __int_as_float(t2), __int_as_float(t3)};
            os2d_stream_store(reinterpret_cast<f32x4*>(corr + orow * HW + nq), o4);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hw;
          bool mok = m < OS2D_K;
          size_t orow = (size_t)nb * OS2D_K + m;
          if (STACK) {
            const int Rr = R0 + m, cc = Rr / STACK_STRIDE, pc = Rr - cc * STACK_STRIDE;
            mok = pc < OS2D_K && cc < B;
            orow = ((size_t)a * B + cc) * OS2D_K + pc;
          }
          if (nin && mok) corr[orow * HW + n] = acc[mi][ni][r];
        }
      }
    }
  }
  if (STACK) {
    // both half-waves hold rows of the same columns: add them (integers), then one 64-bit atomic per (class, position) from the
    // lower half-wave.  A class spans at most two row tiles x two waves: <= 4 contributions per position, any order.
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (32 * NI) + ni * 32 + l31;
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        const unsigned h2 = fxh[sl][ni] + (unsigned)__shfl_xor((int)fxh[sl][ni], 32);
        const unsigned l2 = fxl[sl][ni] + (unsigned)__shfl_xor((int)fxl[sl][ni], 32);
        const bool bad2 = (((int)fxbad[sl][ni]) | __shfl_xor((int)fxbad[sl][ni], 32)) != 0;
        const int cc = cw0 + sl;
        const bool live = cc < B && (sl == 0 || bound < Rw + 128);          // wave-uniform
        if (live && hw == 0 && n < HW) {
          unsigned long long* dst = sumfx + ((size_t)a * B + cc) * HW + n;
          atomicAdd(dst, ((unsigned long long)h2 << 24) + (unsigned long long)l2);
          if (bad2) atomicOr(dst, 1ull << 62);
        }
      }
    }
    return;
  }
  // one class per tile: the two half-waves, then the two waves of a column (LDS), all integers
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const unsigned h2 = fxh[0][ni] + (unsigned)__shfl_xor((int)fxh[0][ni], 32);
    const unsigned l2 = fxl[0][ni] + (unsigned)__shfl_xor((int)fxl[0][ni], 32);
    const bool bad2 = (((int)fxbad[0][ni]) | __shfl_xor((int)fxbad[0][ni], 32)) != 0;
    if (hw == 0)
      red[wm][wn * (32 * NI) + ni * 32 + l31] = (((unsigned long long)h2 << 24) + (unsigned long long)l2) | (bad2 ? 1ull << 62 : 0ull);
  }
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = wn * (32 * NI) + ni * 32 + l31;
    const int n = n0 + col;
    const bool nin = n < HW;   // lanes l and l+32 share n: both take part in the exchange below or neither stores
    // head.py:650,597 (eps 1e-6); the normalised values (<= 1) are stored scaled by 2^OS2D_RNORM_EXP so that their lo halves
    // stay normal fp16 numbers (the conv 7x7 epilogue undoes the scale exactly)
    const unsigned long long vs = red[0][col] + red[1][col];      // < 2^53; bit 62 / 63: a non-finite term
    const float ssum = (vs >> 62) ? __builtin_nanf("") : (float)((double)vs * 5.6843418860808015e-14);    // 2^-44, as os2d_corr_norm_finalize_one
    const float inv_r = 1.0f / (sqrtf(ssum) + 1e-6f);
    if (invn != nullptr && wm == 0 && hw == 0 && nin) invn[(size_t)nb * HW + n] = inv_r;   // for the frequency-domain 7x7 layer
    const float rscale = ldexpf(1.0f, OS2D_RNORM_EXP);
    const int nc = nin ? n : 0;
    const int h = nc / W, w = nc - h * W;
    const size_t cell = (size_t)BASE + (size_t)h * Ws + w;
    if (rshb == nullptr) continue;   // frequency-domain 7x7 layer: it reads corr + invn, the split activations are not needed
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m0 = wm * 128 + mi * 32 + 8 * q + 4 * hw;
        const int grp = m0 >> 3;
        if (grp >= OS2D_G) continue;   // wave-uniform (depends on wm, mi, q only)
        half4 h4, l4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = (m0 + k < OS2D_K) ? fmaxf(acc[mi][ni][4 * q + k], 0.f) * inv_r * rscale : 0.f;
          const _Float16 hv = (_Float16)v;
          h4[k] = hv;
          l4[k] = (_Float16)(v - (float)hv);
        }
        // lower half-wave: channels 0-3, upper: channels 4-7 of the unit.  The lower lane sends its lo half and receives the
        // partner's hi half; the upper lane sends its hi half and receives the partner's lo half.
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 send = hw ? __builtin_bit_cast(u32x2, h4) : __builtin_bit_cast(u32x2, l4);
        u32x2 recv;
        recv[0] = (unsigned)__shfl_xor((int)send[0], 32);
        recv[1] = (unsigned)__shfl_xor((int)send[1], 32);
        const u32x2 keep = hw ? __builtin_bit_cast(u32x2, l4) : __builtin_bit_cast(u32x2, h4);
        const u32x4 unit = hw ? u32x4{recv[0], recv[1], keep[0], keep[1]} : u32x4{keep[0], keep[1], recv[0], recv[1]};
        if (nin) {
          char* o = rshb + ((((size_t)nb * OS2D_G + grp) * 2 + hw) * PLANE + cell) * 16;   // part hw: 0 = hi plane, 1 = lo
          *reinterpret_cast<u32x4*>(o) = unit;
        }
      }
  }
}

// The launch.  Work-group L runs on XCD L % 8; XCD x takes the contiguous range [x tpx, (x + 1) tpx) of the logical tile order
// (image, group of 4 row tiles, position tile, row tile in the group): the 32 groups resident on an XCD (one per CU) are then ~8
// position tiles x 4 classes marching through K together - 12 MB of distinct operand bytes per 32 groups in that XCD's L2 instead
// of 20+ MB with classes or tiles spread round-robin over the XCDs.  STACK: "b" is a ROW TILE of the stacked class matrix (RT of
// them), not a class.
// TAIL (round 6).  Every tile costs the same, so a launch takes ceil(tiles / 256) rounds of the chip however
// few tiles its last round holds (64 classes, padded rows: 1216 tiles = 4.75 rounds, paid as 5).  The last ``rsplit`` tiles of every
// XCD's range are therefore cut into two 128-position halves (the NI = 1 shape of the same code: same products, same order, same
// bits), dispatched last: the partial round then costs about half a round per 32 halves.  NIM = NI of the main shape (1: every
// tile is a 128-position tile and there is no tail).
template <bool STACK, int NIM>
__global__ __launch_bounds__(512, 2) void corr_f16x3_kernel(const u32x4* fs, const u32x4* qs, float* __restrict__ corr, char* __restrict__ rshb,
                                                            float* __restrict__ invn, unsigned long long* __restrict__ sumfx, int A, int B,
                                                            int CGP, int H, int W, int PLANE, float unscale, int tpx /*tiles per XCD*/,
                                                            int rsplit /*of them, at the end of the range: cut into halves*/) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  __shared__ unsigned long long red[2][256];
  const int HW = H * W;
  const int RT = STACK ? (B * STACK_STRIDE + TM - 1) / TM : B;
  const int tiles = (HW + 128 * NIM - 1) / (128 * NIM);
  const int x = blockIdx.x & 7, s = blockIdx.x >> 3, nfull = tpx - rsplit;
  int logical = x * tpx + s, half = -1;
  if (NIM == 2 && s >= nfull) {
    logical = x * tpx + nfull + ((s - nfull) >> 1);
    half = (s - nfull) & 1;
  }
  if (s >= nfull + 2 * rsplit || logical >= min((x + 1) * tpx, tiles * RT * A)) return;
  const int a = logical / (tiles * RT);
  const int r_ = logical - a * tiles * RT;
  const int gb0 = (r_ / (4 * tiles)) * 4, gsz = min(4, RT - gb0);
  const int r2_ = r_ - gb0 * tiles;
  const int b = gb0 + r2_ % gsz;
  const int n0 = (r2_ / gsz) * (128 * NIM);
  if (NIM == 2 && half >= 0) {
    if (n0 + half * 128 < HW) corr_tile<1, STACK>(fs, qs, corr, rshb, invn, sumfx, B, CGP, H, W, PLANE, unscale, a, b, n0 + half * 128, smem16, red);
  } else {
    corr_tile<NIM, STACK>(fs, qs, corr, rshb, invn, sumfx, B, CGP, H, W, PLANE, unscale, a, b, n0, smem16, red);
  }
}

// image features [A][C][HW] fp32 -> L2-normalised over channels (head.py:339, eps 1e-5), scaled, split, blocked
__global__ __launch_bounds__(256) void split_fm_kernel(const float* __restrict__ fm, const float* __restrict__ sumsq,
                                                       u32x4* __restrict__ fs, int C, int HW, float scale,
                                                       unsigned long long* __restrict__ clear, size_t clear_words,
                                                       Os2dRangeFlag status) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y, a = blockIdx.z;
  if (clear_words) {       // the packed correlation kernel's sums, zeroed on the way (grid-stride over all work items)
    const size_t nthr = (size_t)gridDim.x * gridDim.y * gridDim.z * 256;
    for (size_t i = ((size_t)(a * gridDim.y + g) * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < clear_words; i += nthr) clear[i] = 0ull;
  }
  if (n >= HW) return;
  const int CG = os2d_round_up((C + 7) / 8, GC);  // zero groups pad the channel dimension to whole K chunks
  const float ss = sumsq[(size_t)a * HW + n];
  // A non-finite feature (NaN / Inf anywhere in the 1024 channels of this location) makes the reference's outputs NaN through
  // relu(NaN) = NaN (torch).  The ReLUs of this path are fmaxf(x, 0), which DROPS a NaN (the forward transform would hand finite
  // zeros to the 7x7 layer - ADVICE r4), so the one place that sees every input value raises the range word of ITS IMAGE: the last
  // kernel of the call (sample_decode_kernel) then writes NaN into every output of that image (round 6).
  if (status.word != nullptr && g == 0 && __builtin_amdgcn_ballot_w64(!(ss <= 3.4028234e38f)) != 0ull && (threadIdx.x & 63) == 0)
    os2d_raise(Os2dRangeFlag{status.word + a, status.value});
  const float inv = scale / (sqrtf(ss) + 1e-5f);
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = g * 8 + j;
    const float v = c < C ? fm[((size_t)a * C + c) * HW + n] * inv : 0.f;
    const _Float16 hv = (_Float16)v;
    hi[j] = hv;
    lo[j] = (_Float16)(v - (float)hv);
  }
  u32x4* o = fs + ((size_t)a * CG + g) * 2 * HW + n;
  *reinterpret_cast<half8*>(o) = hi;
  *reinterpret_cast<half8*>(o + HW) = lo;
}

// (Round 6 measured fm_sumsq_kernel + this kernel as ONE launch - VERDICT r5 item 1a: 17.5 us against 8.8 + 10.5 us, but the launch
// gaps around it grew from 6 + 0 + 6 to 15 + 20 us - all of its 19.7 MB of stores are still dirty in the L2s when it ends - and the
// step did not move: 1.574 - 1.579 against 1.572 ms, profiles/r06/trace_64_fused_split_and_tail_vs_separate.txt.  Not kept.)

// class operand [C][256] fp32 (os2d_class_prepare) -> [C/8][hi|lo][256] units, scaled
__global__ __launch_bounds__(256) void split_qp_kernel(const float* __restrict__ qp, u32x4* __restrict__ qs, int C,
                                                       float scale) {
  const int m = threadIdx.x;  // 0..255
  const int g = blockIdx.x, b = blockIdx.y;
  const int CG = os2d_round_up((C + 7) / 8, GC);
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = g * 8 + j;
    const float v = c < C ? qp[((size_t)b * C + c) * OS2D_QROWS + m] * scale : 0.f;
    const _Float16 hv = (_Float16)v;
    hi[j] = hv;
    lo[j] = (_Float16)(v - (float)hv);
  }
  u32x4* o = qs + ((size_t)b * CG + g) * 2 * 256 + m;
  *reinterpret_cast<half8*>(o) = hi;
  *reinterpret_cast<half8*>(o + 256) = lo;
}

// STACK: fixed-point sums of relu(corr)^2 -> 1 / (sqrt(s) + 1e-6) (head.py:650, 597: what the per-class kernel writes), and the
// sums are cleared for the next launch.  acc = s * 2^44 < 2^53: the conversion to double is exact, the one to float rounds once.
__global__ __launch_bounds__(256) void corr_norm_finalize_kernel(unsigned long long* __restrict__ sumfx, float* __restrict__ invn,
                                                                 size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) os2d_corr_norm_finalize_one(sumfx, invn, i);
}

constexpr int SCALE_LOG2 = 12;  // operands are L2-normalised (|x| <= 1): hi <= 4096, lo >= 2^-11 * 2^12 * x stays normal

int check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("%s launch: %s", what, hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

int os2d_corr_groups(int C) { return os2d_round_up((C + 7) / 8, GC); }

int os2d_launch_split_fm(const float* fm, const float* sumsq, void* fs, int A, int C, int HW, void* clear, size_t clear_words,
                         Os2dRangeFlag status, hipStream_t stream) {
  hipLaunchKernelGGL(split_fm_kernel, dim3((HW + 255) / 256, os2d_round_up((C + 7) / 8, GC), A), dim3(256), 0, stream, fm, sumsq,
                     reinterpret_cast<u32x4*>(fs), C, HW, ldexpf(1.0f, SCALE_LOG2), static_cast<unsigned long long*>(clear),
                     clear ? clear_words : (size_t)0, status);
  return check("split_fm");
}

int os2d_launch_split_qp(const float* qp, void* qs, int B, int C, hipStream_t stream) {
  hipLaunchKernelGGL(split_qp_kernel, dim3(os2d_round_up((C + 7) / 8, GC), B), dim3(256), 0, stream, qp, reinterpret_cast<u32x4*>(qs), C,
                     ldexpf(1.0f, SCALE_LOG2));
  return check("split_qp");
}

namespace {

// tiles per XCD and how many of them (at the end of every XCD's range) are cut into halves: the partial round of an XCD's 32 CUs
// is split when its halves still fit the 32 CUs in ONE round (up to 16 tiles): a half takes 0.67 of a full tile's time (measured:
// 4 full rounds + 16 halves per XCD = 0.4055 ms against 0.434 ms for 5 rounds, profiles/r06/stages_corr_tail_*.txt), so a second
// round of halves would cost more than the round of full tiles it replaces.  Quarter tiles (64 positions, 4 x 2 waves of 64 x 32)
// for tails of up to 8 tiles were built and measured no faster than halves (1.532 against 1.523 ms per step: a quarter keeps the
// whole class operand and the barriers of a tile's K loop).  The same figures feed the cost model of os2d_corr_f16x3_use_packed.
// $OS2D_CORR_TAIL=0 disables the split (measurements).
struct CorrGrid {
  int tpx, rsplit;
};
CorrGrid corr_grid(long long tiles_total, bool allow_split) {
  static const bool tail_enabled = [] {
    const char* e = getenv("OS2D_CORR_TAIL");
    return !(e && e[0] == '0');
  }();
  CorrGrid g;
  g.tpx = (int)((tiles_total + 7) / 8);
  const int r = g.tpx % 32;
  g.rsplit = (allow_split && tail_enabled && g.tpx > 32 && r > 0 && r <= 16) ? r : 0;
  return g;
}
// rounds of the chip a launch takes, in units of one full-tile round
double corr_rounds(long long tiles_total, bool allow_split) {
  const CorrGrid g = corr_grid(tiles_total, allow_split);
  const int nfull = g.tpx - g.rsplit;
  return (double)((nfull + 31) / 32) + (g.rsplit ? 0.67 : 0.0);
}

template <bool STACK, int NIM>
int launch_corr(const void* fs, const void* qs, float* corr, void* rshb, float* invn, unsigned long long* sumfx, int flags, int A,
                int B, int C, int H, int W, hipStream_t stream) {
  const int HW = H * W;
  const size_t lds = (size_t)(2 * GC * 2 * TM + 2 * GC * 2 * (128 * NIM)) * 16;  // 128 KB (NIM = 2) / 96 KB dynamic (+ 4 KB static)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_f16x3_kernel<STACK, NIM>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(corr f16x3): %s", hipGetErrorString(e));
    return -4;
  }
  const int RT = STACK ? (B * STACK_STRIDE + TM - 1) / TM : B;       // row tiles: stacked classes | one per class
  const long long tiles_total = (long long)((HW + 128 * NIM - 1) / (128 * NIM)) * RT * A;
  const CorrGrid g = corr_grid(tiles_total, NIM == 2 && !(flags & 2));
  dim3 grid((unsigned)(8 * (g.tpx + g.rsplit)));      // per XCD: tpx - rsplit full tiles, then 2 rsplit halves
  hipLaunchKernelGGL((corr_f16x3_kernel<STACK, NIM>), grid, dim3(512), lds, stream, reinterpret_cast<const u32x4*>(fs),
                     reinterpret_cast<const u32x4*>(qs), corr, reinterpret_cast<char*>(rshb), invn, sumfx, A, B,
                     os2d_round_up((C + 7) / 8, GC), H, W, os2d_plane(H, W), ldexpf(1.0f, -2 * SCALE_LOG2), g.tpx, g.rsplit);
  int rc = check("corr_f16x3");
  if (rc || !STACK || (flags & 1)) return rc;
  const size_t n = (size_t)A * B * HW;
  hipLaunchKernelGGL(corr_norm_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sumfx, invn, n);
  return check("corr_norm_finalize");
}

}  // namespace

// Which form the head runs.  Both give the same bits (correlation AND inverse norms), so this is a pure scheduling decision: the
// packed form executes 228 / 256 of the matrix instructions, but its sums cross work-groups as atomics and need the norms pass
// (a small launch of its own in front of the forward transform: ~0.1 of a round at 64 classes).  Decided on the rounds of the chip
// each form takes WITH the tail of its launch cut into half tiles (round 6; round 5 compared whole rounds only, which kept the
// padded form at 64 classes: 1216 tiles = 5 rounds either way - now 5 against 4.67 + 0.1; measured 0.428 -> 0.403 ms).
// $OS2D_CORR_PACKED = 0 | 1 forces one form (measurements).
int os2d_corr_f16x3_use_packed(int A, int B, int H, int W) {
  static const int forced = [] {
    const char* e = getenv("OS2D_CORR_PACKED");
    return !e ? -1 : (e[0] == '0' ? 0 : 1);
  }();
  if (forced >= 0) return forced;
  const long long tiles = (H * W + 255) / 256;
  const double plain = corr_rounds(tiles * B * A, true), packed = corr_rounds(tiles * ((B * STACK_STRIDE + 255) / 256) * A, true) + 0.1;
  return packed < plain ? 1 : 0;
}

// the packed sums of the stacked kernel: cleared once per buffer (the finalize kernel leaves them cleared)
int os2d_launch_corr_sums_clear(void* sumfx, int A, int B, int H, int W, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(sumfx, 0, (size_t)A * B * H * W * sizeof(unsigned long long), stream);
  if (e != hipSuccess) {
    os2d_set_error("hipMemsetAsync(corr sums): %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

// sumfx != NULL: the classes packed along M (STACK; needs rshb == NULL and invn != NULL: the frequency-domain route), sumfx =
// A * B * H * W cleared 64-bit words (os2d_launch_corr_sums_clear once; every launch leaves them cleared again).
// flags & 1 (packed form only): the sums stay in sumfx, the caller's next launch turns them into invn
// (os2d_launch_border_zero_shb_planes_norms); flags & 2: no half tiles at the tail of the launch (measurements / tests).
int os2d_launch_corr_f16x3(const void* fs, const void* qs, float* corr, void* rshb, float* invn, void* sumfx, int flags, int A,
                           int B, int C, int H, int W, hipStream_t stream) {
  unsigned long long* sx = static_cast<unsigned long long*>(sumfx);
  if (sx && (rshb || !invn)) {
    os2d_set_error("corr_f16x3: the packed form writes inverse norms only (rshb must be NULL, invn not)");
    return -1;
  }
  // the 128-position shape as long as its work-groups still fit the chip in one round (see corr_tile)
  const int RT = sx ? (B * STACK_STRIDE + 255) / 256 : B;
  if ((long long)((H * W + 127) / 128) * RT * A <= 256)
    return sx ? launch_corr<true, 1>(fs, qs, corr, rshb, invn, sx, flags, A, B, C, H, W, stream)
              : launch_corr<false, 1>(fs, qs, corr, rshb, invn, sx, flags, A, B, C, H, W, stream);
  return sx ? launch_corr<true, 2>(fs, qs, corr, rshb, invn, sx, flags, A, B, C, H, W, stream)
            : launch_corr<false, 2>(fs, qs, corr, rshb, invn, sx, flags, A, B, C, H, W, stream);
}

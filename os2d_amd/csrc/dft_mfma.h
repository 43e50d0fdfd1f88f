// The two transforms of the frequency-domain 7x7 TransformNet layer (reference os2d/modeling/head.py:619-623, 650) as DENSE
// MATRIX PRODUCTS on the half-precision matrix cores of gfx950 - precision "fftx3", round 4.  Included by dft_mfma.hip (device
// build) and by tests/host/dft_mfma_check.cpp (host build on the SPMD emulator, tests/host/spmd_emu.h): the includer provides
// the hardware hooks DFT_* below, everything else is the same source.
//
// Why matrix products.  Rounds 2 - 3 transformed one image (pair, channel) at a time with register FFTs in LDS: ~10 work-group
// barriers per image and phases of a few hundred active lanes each - the kernels ran at 2.3 / 1.8 TB/s, bound by that serial
// chain, with the matrix cores idle.  A 64 x 84 transform is small enough that the O(N^2) form costs less
// TIME than the O(N log N) one once it runs on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s): per image 2 x 3.1 MFLOP, three
// half-precision products per fp32-equivalent product (operands split into fp16 hi + lo, fp32 accumulation: the arithmetic of
// f16x3 / of the split-half spectral GEMM), ~2,300 matrix cycles per image and CU against ~3,000 cycles of HBM time - and a
// work-group iteration is 4 images behind 6 barriers.
//
//   forward   x[h][w] = relu(corr[nb][c][h][w]) * inv_norm[nb][h][w]   (head.py:650 folded into the load), zero-padded
//     step 1  R^T[(img, h)][(v, re|im)] = x[(img, h)][w] . FqT[w][(v, re|im)]            rows of the map -> half spectrum in v
//     step 2  X[(u, re|im)][(img, v)]   = Fp2[(u, re|im)][(re|im, h)] . R[(re|im, h)][(img, v)]   columns: complex as real 2 x 2
//   inverse   Y[(re|im, u)][(img, v)]
//     step A  T[(h, re|im)][(img, v)]   = E2[(h, re|im)][(re|im, u)] . Y[(re|im, u)][(img, v)]
//     step B  y[(h, img)][w]            = T^T[(h, img)][(v, re|im)] . Gq[(v, re|im)][w]   Hermitian half spectrum -> real rows
//   + the layer's epilogue (1 / (P Q), bias, ReLU, the channel's power-of-two scale, fp16 hi | lo split into the activation
//   buffer of the 5x5 layer), as in fft.hip.
// G = 4 images per work-group iteration: 4 consecutive channels of one pair (forward) / 4 consecutive output channels (inverse).
// That makes the spectra layouts of BOTH sides of the per-bin GEMM "quads x channels": X [bins/4][pair'][Cpad][4] and
// Y [bins/4][pair'][Cout][4] complex64 (in blocks of 64 pairs: dft_spectra_pair0) with bin = v * P + u (u fastest, P % 4 == 0: a quad never straddles v) - the forward
// kernel writes 128-byte runs (4 channels x 4 bins), the GEMM reads 256-byte runs per (pair, k-step of 8 channels) instead of
// 32-byte pieces 22 KB apart, its Y stores stay 1 KB runs, and the inverse kernel reads 128-byte runs.
//
// Operand format.  Every matrix operand lives in "units" of 16 bytes = 8 halves = 8 consecutive k of one row / column, hi and
// lo parts in separate units: arrays [k / 8][hi | lo][row or column].  A fragment of the instruction is then ONE 16-byte read
// per lane, 32 consecutive units per half-wave (conflict-free), lanes 32 - 63 take the next k group.  The constant matrices
// (FqT, Fp2, E2, Gq: functions of (P, Q) only) are built once per transform size by dft_matrices_kernel from float64 tables.
//
// Scales (all powers of two; exact): x * 2^15 (x <= 1), DFT matrices * 2^14 (Gq: 2^13, its entries reach 2), the row
// spectrum R * 2^8 (|R| <= window width <= 96), forward output X = acc * 2^-22; inverse: every image's spectrum is scaled by
// its own power of two that puts its largest |component| into [2^13, 2^14) (the dynamic range of Y is far larger than that of
// the rigorous bound), T * 2^-(14 + ceil(log2 P) + 1), output = acc * 2^(ceil(log2 P) + 1 - 13) / (scale * P * Q).
#pragma once

#ifndef DFT_DEV
#error "the includer defines DFT_DEV, DFT_TID, DFT_BID, DFT_GRID, DFT_LDS, DFT_BARRIER, DFT_MFMA, DFT_SHFL_XOR, DFT_SHFL_XOR_U32, DFT_BALLOT, DFT_FLAG, DFT_FLAG_SET, DFT_RAISE, DFT_UNIFORM"
#endif

#ifndef DFT_HD
#define DFT_HD static inline      /* host + device helpers (the device build says __host__ __device__) */
#endif
#ifndef DFT_MEMBER
#define DFT_MEMBER inline         /* member functions (the device build says __device__ __forceinline__) */
#endif
#ifndef DFT_LANDED
#define DFT_LANDED(X)             /* device build: an empty asm that reads and writes X - see the Fp2 fragments in the forward kernel */
#endif
#ifndef DFT_FWD_PREFETCH_AFTER
#define DFT_FWD_PREFETCH_AFTER 2  /* the next window's loads right AFTER step 2 instead of in front of it: 0 never, 1 always, 2 for the
                                     transforms of up to 6 k-steps (measured: 4 - 5 % faster for (44,54), neutral for (56,70), slower for
                                     (64,84): profiles/r05/dft_phases_prefetch_after.txt) */
#endif
#ifndef DFT_INV_PREFETCH_AFTER
#define DFT_INV_PREFETCH_AFTER 2  /* the same for the next spectra and step A */
#endif
#ifndef DFT_PIPE_REGA
#define DFT_PIPE_REGA 1           /* 1: step 2 / step A read the fragments of k-step ks + 1 before the matrix instructions of ks */
#endif
#ifndef DFT_PIPE_LDS
#define DFT_PIPE_LDS 1            /* 1: the same in step 1 (every product) / step B (the row-sharing products: registers) */
#endif
#ifndef DFT_SCHED_FENCE
#define DFT_SCHED_FENCE()         /* device build: __builtin_amdgcn_sched_barrier(0) - the scheduler moves nothing across it */
#endif
#ifndef DFT_STREAM_STORE
#define DFT_STREAM_STORE(P, V) (*(P) = (V))      /* device build: a non-temporal store (os2d_stream_store, os2d_common.h) */
#endif
#ifndef DFT_STAMP
#define DFT_STAMP(K)              /* diagnostic builds (-DOS2D_DIAG_DFT_STAMPS): time since the previous stamp -> phase K */
#define DFT_STAMP_BEGIN()
#define DFT_STAMP_END(BASE)
#endif

namespace os2d_dft {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // 4 floats at any float address (one dwordx4 load)

constexpr int DFT_THR = 512, DFT_WAVES = 8;
constexpr int DFT_G = 4;            // images per work-group iteration
constexpr int DFT_MAXP = 64;        // transform rows (P <= 64: the 2P rows of step 2 / step A are at most four 32-row tiles)
constexpr int DFT_MAXV = 48;        // half-spectrum columns (2V <= 96: three 32-column tiles; Q <= 94)
constexpr int DFT_MAXWK = 96;       // window columns rounded up to 16
constexpr int DFT_KREG = 8;         // k-steps of the register-resident operand (2 * Pp / 16 <= 8)

struct DftPlan {
  int H, W;                        // map
  int P, Q, V, NBINS;              // transform size, V = Q/2 + 1; bins = v * P + u padded to a multiple of 8
  int T, TY, TX, TH, TW;           // overlap-save tiles (T = TY * TX = 1: the whole map in one transform)
  int oy, ox;                      // 3 along a tiled axis (window starts 3 cells before the tile; outputs sit at offset 3), else 0
  int LH, LW;                      // rows / columns of the window a tile loads (TH + 6 | H, TW + 6 | W)
  int RH;                          // rows of the inverse that are needed (oy + TH | H)
  int Pp;                          // P rounded up to 8: rows of a window in the operand arrays (k of step 2 / step A = 2 * Pp)
  int Wk;                          // LW rounded up to 16: k of step 1
  int N1;                          // 2 V rounded up to 32: columns of step 1
  int M2;                          // 2 P rounded up to 32: rows of step 2
  int Mx;                          // G * Pp rounded up to 32: rows of step 1 = (img, h)
  int N2;                          // G * V rounded up to 32: columns of step 2 / step A = (img, v)
  int MA;                          // 2 * RH rounded up to 32: rows of step A = (h, re | im)
  int MB;                          // G * (MA / 2) : rows of step B = (h, img)
  int KB;                          // 2 V rounded up to 16: k of step B
  int NBo;                         // columns of step B that are needed (ox + TW | W) rounded up to 32
  int G;                           // images per work-group iteration of both kernels: 4, or 8 for the small transforms (dft_plan_g8)
  int eT;                          // ceil(log2 P) + 1
  int fast;                        // 1: W % 4 == 0 and untiled (16-byte loads of the correlation rows)
  unsigned inv_cg, inv_t, inv_tx, inv_c4, inv_v, inv_pq, inv_og, inv_kg, inv_pp;   // ceil(2^32 / d) (0 where d == 1)
  // LDS (bytes)
  int lds_const, lds_union, lds_total;
};

DFT_HD int dft_round_up(int x, int m) { return (x + m - 1) / m * m; }
DFT_HD unsigned dft_magic(unsigned d) { return d > 1 ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u; }

// sizes of the constant operand arrays of a (P, Q) transform, in 16-byte units: they depend on P and Q only, so every map
// (and every tiling) that uses the transform shares them
DFT_HD int dft_units_fqt(int P, int Q) { return (dft_round_up(Q, 16) / 8) * 2 * dft_round_up(Q + 2, 32); }
DFT_HD int dft_units_fp2(int P, int Q) { return (2 * dft_round_up(P, 8) / 8) * 2 * dft_round_up(2 * P, 32); }
DFT_HD int dft_units_e2(int P, int Q) { return (2 * dft_round_up(P, 8) / 8) * 2 * dft_round_up(2 * P, 32); }
DFT_HD int dft_units_gq(int P, int Q) { return (dft_round_up(Q + 2, 16) / 8) * 2 * dft_round_up(Q, 32); }
DFT_HD size_t dft_matrices_units(int P, int Q) {
  return (size_t)dft_units_fqt(P, Q) + dft_units_fp2(P, Q) + dft_units_e2(P, Q) + dft_units_gq(P, Q);
}

// plan of ONE transform: window LH x LW -> P x Q, RH rows of the inverse needed, NBo output columns needed
static inline bool dft_plan_transform(int LH, int LW, int RH, int ncols, int minP, int minQ, int G, DftPlan* pl) {
  pl->G = G;
  pl->P = dft_round_up(minP, 4);
  pl->Q = dft_round_up(minQ, 2);
  pl->V = pl->Q / 2 + 1;
  pl->NBINS = dft_round_up(pl->P * pl->V, 8);
  pl->LH = LH;
  pl->LW = LW;
  pl->RH = RH;
  pl->Pp = dft_round_up(pl->P, 8);
  pl->Wk = dft_round_up(LW, 16);
  pl->N1 = dft_round_up(2 * pl->V, 32);
  pl->M2 = dft_round_up(2 * pl->P, 32);
  pl->Mx = dft_round_up(G * pl->Pp, 32);
  pl->N2 = dft_round_up(G * pl->V, 32);
  pl->MA = dft_round_up(2 * RH, 32);
  pl->MB = G * (pl->MA / 2);
  pl->KB = dft_round_up(2 * pl->V, 16);
  pl->NBo = dft_round_up(ncols, 32);
  int e = 0;
  while ((1 << e) < pl->P) ++e;
  pl->eT = e + 1;
  if (pl->P > DFT_MAXP || pl->V > DFT_MAXV || pl->Wk > DFT_MAXWK || LH > pl->P || LW > pl->Q || RH > pl->P) return false;
  // LDS: constants FqT | Gq are resident (whichever kernel runs takes its own: the larger decides the common figure);
  // the union region holds, one after the other, x | R2 | X staging (forward) and Y2 | Tt (inverse)
  const int fqt = (pl->Wk / 8) * 2 * pl->N1 * 16, gq = (pl->KB / 8) * 2 * pl->NBo * 16;
  const int x = (pl->Wk / 8) * 2 * (pl->Mx + 1) * 16, r2 = (2 * pl->Pp / 8) * 2 * (pl->N2 + 1) * 16;
  const int xs = pl->V * (pl->P * 8 * G + 16);
  const int y2 = r2, tt = (pl->KB / 8) * 2 * (pl->MB + 1) * 16;
  int u = x;
  if (r2 > u) u = r2;
  if (xs > u) u = xs;
  if (y2 > u) u = y2;
  if (tt > u) u = tt;
  if (G == 8) {
    // EIGHT images per iteration (round 6): the fixed phases of an iteration - six barriers, the fill and drain of
    // four products, the burst of the next window's requests - do not shrink with the transform, so the small transforms of the
    // pyramid paid 1.3 - 1.9x per location.  With 8 images the products have 20 - 24 tiles (three per wave: the tile grids of
    // step 2 / step A no longer match the fixed "row tile in registers" ownership, so Fp2 / E2 live in LDS next to FqT / Gq
    // and every product takes the LDS-LDS path), the inverse kernel owns all 8 channels of an activation unit and writes whole
    // 16-byte units.  Fits the transforms up to 48 x 62.
    const int fp2 = (2 * pl->Pp / 8) * 2 * pl->M2 * 16;      // = the size of E2 (MAfull = M2 rows)
    const int ca = fqt + fp2, cb = gq + fp2;
    pl->lds_const = ca > cb ? ca : cb;
    pl->lds_union = dft_round_up(u, 256);
    pl->lds_total = pl->lds_const + pl->lds_union + 1024;
    const int t1 = (pl->Mx / 32) * (pl->N1 / 32), t2 = (pl->M2 / 32) * (pl->N2 / 32), tA = (pl->MA / 32) * (pl->N2 / 32),
              tB = (pl->MB / 32) * (pl->NBo / 32);
    const int npos = pl->Pp * (pl->Wk / 4), nitem = G * pl->V * (pl->Pp / 8);
    return pl->lds_total <= 160 * 1024 && t1 <= 24 && t2 <= 24 && tA <= 24 && tB <= 24 && npos <= 2 * DFT_THR && nitem <= 3 * DFT_THR;
  }
  pl->lds_const = fqt > gq ? fqt : gq;
  pl->lds_union = dft_round_up(u, 256);
  pl->lds_total = pl->lds_const + pl->lds_union + 1024;      // + per-image maxima / scales
  return pl->lds_total <= 160 * 1024;
}

static inline void dft_set_tiles(DftPlan* pl, int H, int W, int TY, int TX, int TH, int TW) {
  pl->H = H;
  pl->W = W;
  pl->TY = TY;
  pl->TX = TX;
  pl->T = TY * TX;
  pl->TH = TH;
  pl->TW = TW;
  pl->oy = TY > 1 ? 3 : 0;
  pl->ox = TX > 1 ? 3 : 0;
  pl->fast = (pl->T == 1 && (W & 3) == 0) ? 1 : 0;
  pl->inv_t = dft_magic((unsigned)pl->T);
  pl->inv_tx = dft_magic((unsigned)TX);
  pl->inv_c4 = dft_magic((unsigned)(pl->Wk / 4));
  pl->inv_v = dft_magic((unsigned)pl->V);
  pl->inv_pq = dft_magic((unsigned)(pl->P / 4 * pl->G * 2));
  pl->inv_kg = dft_magic((unsigned)(pl->Pp / 8));
  pl->inv_pp = dft_magic((unsigned)pl->Pp);
}

// Transform sizes.  The weight spectra of the per-bin GEMM cost 0.24 MB per bin and transform size (654 MB for 64 x 84), and a
// dataset fed at its own aspect ratios and 7 pyramid scales meets hundreds of map sizes (reference os2d/data/dataloader.py:326,
// os2d/config.py:194).  Since every map can be cut into overlap-save tiles, the planner may restrict itself to a handful of
// CANONICAL sizes and pay in bins instead of in resident spectra (35.6 GB cached for 52 FFT-friendly sizes):
//   policy 1 (default)  the six sizes below - the exact transforms of the 7-scale pyramid of a 1280 x 960 image (levels 30x40 ..
//                       60x80 whole, 72x96 / 84x112 / 96x128 as 2 x 2 tiles): 10,220 bins = 2.4 GB of weight spectra in total
//   policy 0            any P % 4 == 0, even Q: the smallest transform per map (fewest bins; one set of spectra per size)
// $OS2D_DFT_SIZES = canonical | exact selects it for the process (read once).
struct DftSize {
  int P, Q;
};
static const DftSize DFT_CANONICAL[6] = {{36, 46}, {44, 54}, {48, 62}, {52, 68}, {56, 70}, {64, 84}};

// The whole map in one transform when it fits (P >= H + 3, Q >= W + 3: the zero padding is the halo), otherwise the tiling
// with the fewest bins in total; an axis is either untiled or cut into >= 2 tiles of ceil(n / k) outputs whose window is 6
// longer.
static inline bool dft_make_plan_policy(int H, int W, int canonical, DftPlan* out, int G = DFT_G) {
  bool found = false;
  long best = 0;
  for (int TY = 1; TY <= 48; ++TY)
    for (int TX = 1; TX <= 48; ++TX) {
      const int TH = (H + TY - 1) / TY, TW = (W + TX - 1) / TX;
      if ((TY > 1 && (TY - 1) * TH >= H) || (TX > 1 && (TX - 1) * TW >= W)) continue;     // an empty last tile
      const int LH = TY > 1 ? TH + 6 : H, LW = TX > 1 ? TW + 6 : W;
      const int minP = TY > 1 ? TH + 6 : H + 3, minQ = TX > 1 ? TW + 6 : W + 3;
      if (minP > DFT_MAXP || minQ > 2 * (DFT_MAXV - 1)) continue;
      for (int k = 0; k < (canonical ? 6 : 1); ++k) {
        if (canonical && (DFT_CANONICAL[k].P < minP || DFT_CANONICAL[k].Q < minQ)) continue;
        DftPlan c = {};
        if (!dft_plan_transform(LH, LW, TY > 1 ? TH + 3 : H, TX > 1 ? TW + 3 : W, canonical ? DFT_CANONICAL[k].P : minP,
                                canonical ? DFT_CANONICAL[k].Q : minQ, G, &c))
          continue;
        dft_set_tiles(&c, H, W, TY, TX, TH, TW);
        const long cost = (long)c.T * c.NBINS;
        if (!found || cost < best) {
          found = true;
          best = cost;
          *out = c;
        }
      }
    }
  return found;
}

#ifndef OS2D_HOST_EMU
static inline int dft_size_policy() {
  static const int policy = [] {
    const char* e = getenv("OS2D_DFT_SIZES");
    return (e && (e[0] == 'e' || e[0] == '0')) ? 0 : 1;
  }();
  return policy;
}
#else
static inline int dft_size_policy() { return emu_dft_policy; }
#endif
static inline bool dft_make_plan(int H, int W, DftPlan* out) { return dft_make_plan_policy(H, W, dft_size_policy(), out); }
// The plan of the same map with 8 images per iteration: the SAME tiling and transform size as ``base`` (= dft_make_plan: the spectra
// layouts do not depend on G), its own operand sizes; false when the transform is too large for it (then both kernels take 4).
static inline bool dft_plan_g8(const DftPlan& base, DftPlan* out) {
  DftPlan c = {};
  if (!dft_plan_transform(base.LH, base.LW, base.RH, base.TX > 1 ? base.TW + 3 : base.W, base.P, base.Q, 8, &c)) return false;
  dft_set_tiles(&c, base.H, base.W, base.TY, base.TX, base.TH, base.TW);
  *out = c;
  return true;
}

// Spectra in quads of bins, BLOCKED by 64 pairs (the pair tile of the per-bin GEMM):
//   [pair' / 64][bins / 4][pair' % 64][channel stride][4] complex64, the last block holding NBT % 64 pairs (no padding: the
//   buffers keep their size NBT x bins x channels).
// With one block - up to 64 pairs - this is [bins / 4][pair'][channel][4].  With 1024 pairs the quads of one pair were 7.6 MB
// (X) / 4 MB (Y) apart: every 128-byte run of a transform iteration in its own DRAM page and translation entry; blocked, the
// distance is that of the 64-pair case (475 / 262 KB) whatever the batch, and a GEMM work-group's operands are one contiguous
// slab.  Returns the float offset of (pair', quad 0, channel 0); *qstride = floats between consecutive quads of that pair.
constexpr int DFT_PBLK = 64;
DFT_HD size_t dft_spectra_pair0(int pr, int NBT, int NQ, int cstride, size_t* qstride) {
  const int pb = pr / DFT_PBLK, pl = pr - pb * DFT_PBLK;
  const int nbl = NBT - pb * DFT_PBLK < DFT_PBLK ? NBT - pb * DFT_PBLK : DFT_PBLK;
  *qstride = (size_t)nbl * cstride * 8;
  return ((size_t)pb * DFT_PBLK * NQ + pl) * cstride * 8;
}

// ---------------------------------------------------------------------------------------------------- device helpers
DFT_DEV int dft_div(int x, unsigned magic) { return magic ? (int)(((unsigned long long)(unsigned)x * magic) >> 32) : x; }

// fp16 hi + lo of four values: two 8-byte halves-of-a-unit.  Vector conversions: gfx950 has v_cvt_pk_f16_f32 (two values per
// instruction, round to nearest even like the scalar form) - 12 instructions per call instead of ~20 with scalar conversions
// and integer packing; the split conversions are a third of the VALU work of the W / R / WY / WT phases.
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
DFT_DEV void dft_split4(float a, float b, float c, float d, u32x2v* hi, u32x2v* lo) {
  const f32x4v x = {a, b, c, d};
  const half4v h = __builtin_convertvector(x, half4v);
  *hi = __builtin_bit_cast(u32x2v, h);
#ifdef DFT_SPLIT_LO_PAIR      /* device build: one mixed-precision instruction per lo half (os2d_split_lo_pair), same bits */
  u32x2v l;
  l[0] = DFT_SPLIT_LO_PAIR(a, b, (*hi)[0]);
  l[1] = DFT_SPLIT_LO_PAIR(c, d, (*hi)[1]);
  *lo = l;
#else
  const half4v l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4v), half4v);
  *lo = __builtin_bit_cast(u32x2v, l);
#endif
}

DFT_DEV half8 dft_frag(const u32x4v* p) { return __builtin_bit_cast(half8, *p); }

// one split product step: acc += A_hi B_lo + A_lo B_hi + A_hi B_hi (fp32 accumulation; the dropped lo x lo term is 2^-22 relative)
DFT_DEV f32x16v dft_mma3(half8 ah, half8 al, half8 bh, half8 bl, f32x16v acc) {
  acc = DFT_MFMA(ah, bl, acc);
  acc = DFT_MFMA(al, bh, acc);
  acc = DFT_MFMA(ah, bh, acc);
  return acc;
}

// acc[j] += A . B[tile j] for this wave's NT column tiles, the row operand A resident in REGISTERS (its k-steps as fragment
// arrays).  Software-pipelined in the source (round 5): the B fragments of k-step ks + 1 are read into the OTHER of two register
// sets before the matrix instructions of k-step ks are issued, and the scheduler is fenced to that order.  Left to itself the
// compiler placed every fragment read one or two instructions in front of its first use - each of the 3 NT matrix instructions of
// a k-step then waited out the LDS latency (~100+ cycles) with one other wave per SIMD to cover it: the matrix phases ran at
// ~45 % of the instruction rate (step 2: 5.3 us for 1.9 us of matrix time), spills or no spills.
// KS > 0: the number of k-steps is a template parameter (the canonical transform sizes: 2 Pp / 16 = 5 .. 8) - no branch at all
// between the first fragment read and the last matrix instruction; KS == 0: any count up to DFT_KREG behind uniform guards.
// The tiles of a k-step are INDEPENDENT accumulators interleaved in the matrix pipe: one tile after the other - a chain of 3 K
// dependent instructions - measured 5.7 against 5.1 us.
// (The next iteration's global loads spread over the k-steps of this product instead of one burst next to it: measured slower,
// tools/patches/dft_mfma_variants.patch.)
template <int NT, int KS>
DFT_DEV void dft_product_rega(f32x16v* acc, const half8* ah, const half8* al, int ksn, const u32x4v* B, int bstride, int nt0,
                              int ntstep, int l31, int hw) {
  constexpr int KR = KS ? KS : DFT_KREG;
  half8 bh[2][NT], bl[2][NT];
#define DFT_REGA_READ(KSTEP, SET)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                         \
    bh[SET][j] = dft_frag(B + (size_t)((2 * (KSTEP) + hw) * 2 + 0) * bstride + (nt0 + ntstep * j) * 32 + l31);             \
    bl[SET][j] = dft_frag(B + (size_t)((2 * (KSTEP) + hw) * 2 + 1) * bstride + (nt0 + ntstep * j) * 32 + l31);             \
  }
  if (DFT_PIPE_REGA) DFT_REGA_READ(0, 0)
#pragma unroll
  for (int ks = 0; ks < KR; ++ks) {
    if (KS || ks < ksn) {
      if (DFT_PIPE_REGA) {
        if (ks + 1 < KR && (KS || ks + 1 < ksn)) DFT_REGA_READ(ks + 1, (ks + 1) & 1)
        DFT_SCHED_FENCE();
      } else {
        DFT_REGA_READ(ks, ks & 1)
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = dft_mma3(ah[ks], al[ks], bh[ks & 1][j], bl[ks & 1][j], acc[j]);
      if (DFT_PIPE_REGA) DFT_SCHED_FENCE();
    }
  }
#undef DFT_REGA_READ
}

template <int KS>
DFT_DEV void dft_product_rega_any(f32x16v* acc, int ntiles, const half8* ah, const half8* al, int ksn, const u32x4v* B, int bstride,
                                  int nt0, int ntstep, int l31, int hw) {
  if (ntiles == 3) dft_product_rega<3, KS>(acc, ah, al, ksn, B, bstride, nt0, ntstep, l31, hw);
  else if (ntiles == 2) dft_product_rega<2, KS>(acc, ah, al, ksn, B, bstride, nt0, ntstep, l31, hw);
  else if (ntiles == 1) dft_product_rega<1, KS>(acc, ah, al, ksn, B, bstride, nt0, ntstep, l31, hw);
}

// acc[j] += A[row tile mt_j] . B[column tile nt_j], both operands in LDS, tiles t = t0 + NW j of a grid of mtn row tiles;
// SHARE: all NT tiles have the same row tile (mtn == NW): its A fragment is read once per k-step.  Pipelined like the product
// above: two fragment sets, the k loop unrolled by two so that the sets keep their registers, the reads of k-step ks + 1 in front
// of the matrix instructions of k-step ks (an index beyond the last k-step is clamped: one harmless re-read).
template <int NT, bool SHARE, int NW, int PIPE_MODE>
DFT_DEV void dft_product_lds(f32x16v* acc, int ks0, int ks1, const u32x4v* A, int astride, const u32x4v* B, int bstride, int t0, int mtn,
                             int l31, int hw) {
  constexpr int NA = SHARE ? 1 : NT;
  int tm[NT], tn[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + NW * j;
    tn[j] = t / mtn;
    tm[j] = t - tn[j] * mtn;
  }
  half8 ah[2][NA], al[2][NA], bh[2][NT], bl[2][NT];
#define DFT_LDS_READ(KSTEP, SET)                                                                                           \
  {                                                                                                                        \
    const int kc_ = (KSTEP) < ks1 ? (KSTEP) : ks1 - 1;                                                                     \
    _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                                       \
      ah[SET][j] = dft_frag(A + (size_t)((2 * kc_ + hw) * 2 + 0) * astride + tm[j] * 32 + l31);                            \
      al[SET][j] = dft_frag(A + (size_t)((2 * kc_ + hw) * 2 + 1) * astride + tm[j] * 32 + l31);                            \
    }                                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                       \
      bh[SET][j] = dft_frag(B + (size_t)((2 * kc_ + hw) * 2 + 0) * bstride + tn[j] * 32 + l31);                            \
      bl[SET][j] = dft_frag(B + (size_t)((2 * kc_ + hw) * 2 + 1) * bstride + tn[j] * 32 + l31);                            \
    }                                                                                                                      \
  }
#define DFT_LDS_MMA(SET)                                                                                                   \
  DFT_SCHED_FENCE();                                                                                                       \
  _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                           \
    acc[j] = dft_mma3(ah[SET][SHARE ? 0 : j], al[SET][SHARE ? 0 : j], bh[SET][j], bl[SET][j], acc[j]);                    \
  DFT_SCHED_FENCE();
  if (ks0 >= ks1) return;
  // (two sets of NA + NT fragment pairs: 64 registers when the tiles share their row tile, up to 96 when they do not - more than
  // the inverse kernel has left next to its prefetched spectra: PIPE_MODE 2 = every product, 1 = the row-sharing ones, 0 = none)
  constexpr bool PIPE = PIPE_MODE == 2 || (PIPE_MODE == 1 && SHARE);
  if (!PIPE) {
    for (int ks = ks0; ks < ks1; ++ks) {
      DFT_LDS_READ(ks, 0)
      _Pragma("unroll") for (int j = 0; j < NT; ++j)
        acc[j] = dft_mma3(ah[0][SHARE ? 0 : j], al[0][SHARE ? 0 : j], bh[0][j], bl[0][j], acc[j]);
    }
    return;
  }
  DFT_LDS_READ(ks0, 0)
  int ks = ks0;
  for (; ks + 1 < ks1; ks += 2) {
    DFT_LDS_READ(ks + 1, 1)
    DFT_LDS_MMA(0)
    DFT_LDS_READ(ks + 2, 0)
    DFT_LDS_MMA(1)
  }
  if (ks < ks1) {
    DFT_LDS_MMA(0)
  }
#undef DFT_LDS_READ
#undef DFT_LDS_MMA
}

template <int NW, int PIPE_MODE>
DFT_DEV void dft_product_lds_any(f32x16v* acc, int ntiles, int ks0, int ks1, const u32x4v* A, int astride, const u32x4v* B, int bstride,
                                 int t0, int mtn, int l31, int hw) {
  const bool share = mtn == NW;
  if (ntiles == 3) {
    if (share) dft_product_lds<3, true, NW, PIPE_MODE>(acc, ks0, ks1, A, astride, B, bstride, t0, mtn, l31, hw);
    else dft_product_lds<3, false, NW, PIPE_MODE>(acc, ks0, ks1, A, astride, B, bstride, t0, mtn, l31, hw);
  } else if (ntiles == 2) {
    if (share) dft_product_lds<2, true, NW, PIPE_MODE>(acc, ks0, ks1, A, astride, B, bstride, t0, mtn, l31, hw);
    else dft_product_lds<2, false, NW, PIPE_MODE>(acc, ks0, ks1, A, astride, B, bstride, t0, mtn, l31, hw);
  } else if (ntiles == 1) {
    dft_product_lds<1, true, NW, PIPE_MODE>(acc, ks0, ks1, A, astride, B, bstride, t0, mtn, l31, hw);
  }
}

// ---------------------------------------------------------------------------------------------------- constant matrices
// thread per unit; tw tables: double [n][2] = (cos, sin) of -2 pi m / n.  Layouts (units of 8 halves along k):
//   FqT [k = w / 8][hi|lo][n = 2 v + ri < N1]         value(w, n) = E_Q[v w].{re, im}                     * 2^14
//   Fp2 [k = (ri * Pp + h) / 8][hi|lo][m = 2 u + ro]  value = ro == 0 ? (ri == 0 ? cos : sin) : (ri == 0 ? -sin : cos) of 2 pi u h / P
//   E2  [k = (ri * Pp + u) / 8][hi|lo][m = 2 h + ro]  value = ro == 0 ? (ri == 0 ? cos : -sin) : (ri == 0 ? sin : cos)
//   Gq  [k = (2 v + ri) / 8][hi|lo][n = w]            value = a_v * E_Q[v w].{re, im}, a_0 = a_{Q/2} = 1, else 2   * 2^13
DFT_DEV void dft_matrix_unit(int which, int unit, int P, int Q, const double* twP, const double* twQ, u32x4v* out) {
  const int V = Q / 2 + 1, Pp = dft_round_up(P, 8);
  int cols, kidx, hl, col;
  double vals[8];
  if (which == 0) cols = dft_round_up(2 * V, 32);
  else if (which == 1 || which == 2) cols = dft_round_up(2 * P, 32);
  else cols = dft_round_up(Q, 32);
  col = unit % cols;
  hl = (unit / cols) & 1;
  kidx = unit / (2 * cols);
  for (int t = 0; t < 8; ++t) {
    const int k = kidx * 8 + t;
    double x = 0.0;
    if (which == 0) {                  // FqT: k = w, col = 2 v + ri
      const int v = col >> 1, ri = col & 1;
      if (k < Q && v < V) x = twQ[2 * (int)(((long long)v * k) % Q) + ri] * 16384.0;
    } else if (which == 1) {           // Fp2: k = ri * Pp + h, col (row of the product) = 2 u + ro
      const int ri = k / Pp, h = k - ri * Pp, u = col >> 1, ro = col & 1;
      if (h < P && u < P && ri < 2) {
        const int a = (int)(((long long)u * h) % P);
        const double c = twP[2 * a], s = -twP[2 * a + 1];          // cos, sin of +2 pi u h / P
        x = (ro == 0 ? (ri == 0 ? c : s) : (ri == 0 ? -s : c)) * 16384.0;
      }
    } else if (which == 2) {           // E2: k = ri * Pp + u, col = 2 h + ro
      const int ri = k / Pp, u = k - ri * Pp, h = col >> 1, ro = col & 1;
      if (u < P && h < P && ri < 2) {
        const int a = (int)(((long long)u * h) % P);
        const double c = twP[2 * a], s = -twP[2 * a + 1];
        x = (ro == 0 ? (ri == 0 ? c : -s) : (ri == 0 ? s : c)) * 16384.0;
      }
    } else {                           // Gq: k = 2 v + ri, col = w
      const int v = k >> 1, ri = k & 1;
      if (v < V && col < Q) {
        const double a = (v == 0 || 2 * v == Q) ? 1.0 : 2.0;
        x = a * twQ[2 * (int)(((long long)v * col) % Q) + ri] * 8192.0;
      }
    }
    vals[t] = x;
  }
  unsigned w[4];
  for (int t = 0; t < 4; ++t) {
    unsigned short b[2];
    for (int e = 0; e < 2; ++e) {
      const double x = vals[2 * t + e];
      const _Float16 h = (_Float16)x;
      const _Float16 r = hl == 0 ? h : (_Float16)(x - (double)h);
      b[e] = __builtin_bit_cast(unsigned short, r);
    }
    w[t] = (unsigned)b[0] | ((unsigned)b[1] << 16);
  }
  u32x4v o;
  o[0] = w[0];
  o[1] = w[1];
  o[2] = w[2];
  o[3] = w[3];
  *out = o;
}

// ---------------------------------------------------------------------------------------------------- forward transform
// iteration it -> (pair' = it / CG, channel group cg = it % CG), pair' = nb * T + tile; channels G cg .. G cg + G - 1
// G = 4: Fp2 in registers (the wave owns one row tile of step 2); G = 8 (dft_plan_transform): Fp2 in LDS, free tile ownership
template <bool TILED, bool FAST, int G, int NW, int KS2 = 0>
DFT_DEV void dft_forward_body(const float* corr,      // [NB][C][H * W]
                              const float* invn,      // [NB][H * W]
                              float* X,               // [NBT / 64][NBINS / 4][64][Cpad][4][2] (dft_spectra_pair0)
                              const u32x4v* FqT, const u32x4v* Fp2, const DftPlan& pl, int C, int Cpad, int NBT, int iters) {
  constexpr int THR = NW * 64;
  constexpr bool ALDS = G == 8;    // the row operand of step 2 lives in LDS
  unsigned char* smem = DFT_LDS;
  const int tid = DFT_TID, lane = tid & 63, l31 = lane & 31, hw = lane >> 5;
  const int wv = DFT_UNIFORM(tid >> 6);      // wave-uniform (a scalar register): the tile ownership below is real branching, not exec masks
  const int P = pl.P, V = pl.V, Pp = pl.Pp, Wk = pl.Wk, N1 = pl.N1, Mx = pl.Mx, N2 = pl.N2, LH = pl.LH, LW = pl.LW, W = pl.W, H = pl.H;
  const int MxS = Mx + 1, N2S = N2 + 1;
  const int CG = (C + G - 1) / G, HW = H * W;
  u32x4v* ldsF = reinterpret_cast<u32x4v*>(smem);                                   // FqT [Wk / 8][2][N1]
  u32x4v* ldsU = reinterpret_cast<u32x4v*>(smem + pl.lds_const);                    // x | R2 | X staging
  unsigned char* ldsUb = smem + pl.lds_const;
  const int nF = (Wk / 8) * 2 * N1;
  for (int i = tid; i < nF; i += THR) ldsF[i] = FqT[i];
  u32x4v* ldsP = ldsF + nF;                                                         // ALDS: Fp2 [2 Pp / 8][2][M2]
  if (ALDS)
    for (int i = tid; i < (2 * Pp / 8) * 2 * pl.M2; i += THR) ldsP[i] = Fp2[i];

  // the wave's tiles.  step 1: (row tile of x, column tile of FqT), round robin - with 8 row tiles a wave keeps ONE row tile
  // and its A fragment serves all column tiles; step 2: row tile wv & 3 of Fp2 in REGISTERS, column tiles (wv >> 2) + 2 j
  const int mt1n = Mx / 32, nt1n = N1 / 32, nt2n = N2 / 32, mt2n = pl.M2 / 32;
  const int ks1n = Wk / 16, ks2n = 2 * Pp / 16;
  const int mt2 = wv & 3;
  constexpr int KR2 = KS2 ? KS2 : DFT_KREG;      // KS2 > 0: exactly the k-steps of step 2 (ks2n == KS2)
  half8 fp2h[KR2], fp2l[KR2];
  if (!ALDS) {
#pragma unroll
    for (int ks = 0; ks < KR2; ++ks) {
      const int kc = ks < ks2n ? ks : 0, mc = mt2 < mt2n ? mt2 : 0;
      fp2h[ks] = dft_frag(Fp2 + ((size_t)((2 * kc + hw) * 2 + 0)) * pl.M2 + mc * 32 + l31);
      fp2l[ks] = dft_frag(Fp2 + ((size_t)((2 * kc + hw) * 2 + 1)) * pl.M2 + mc * 32 + l31);
    }
  }
  // The fragments must have LANDED, as far as the compiler's wait-count pass is concerned, on EVERY path into the loop (round 5).
  // Every wait on a global load below is conditional (a position slot beyond the window, the prefetch of an iteration that does
  // not exist), so there is a path from these loads to step 2 on which nothing was waited for and nothing was issued after them -
  // and since the pass merges paths pessimistically, it guarded the matrix instructions of k-step ks with vmcnt(15 - 2 ks) ..
  // vmcnt(0) for good: in the steady state that made step 2 wait for the NEXT window's loads, issued right in front of it, one
  // by one (step 2 measured 5.3 us for 1.9 us of matrix time).  An unconditional use right here settles it.
  if (!ALDS) {
#pragma unroll
    for (int ks = 0; ks < KR2; ++ks) {
      DFT_LANDED(fp2h[ks]);
      DFT_LANDED(fp2l[ks]);
    }
  }

  // ---- register prefetch of the next iteration's window: position slot s of a thread = (row r, 4 columns c4) of the window,
  // all G images; raw values only (any arithmetic here would make the compiler wait for each load where it is issued)
  constexpr int NSLOT = ALDS ? 2 : 3 * 512 / THR;              // ceil(Pp * Wk / 4 / THR) <= 64 * 24 / THR (G = 8: <= 48 * 16 / 512)
  const int npos = Pp * (Wk / 4);
  f32x4v pc[NSLOT][G], pn[NSLOT];
#define DFT_FWD_ITER(IT)                                                                                     \
  const int pr_ = dft_div((IT), pl.inv_cg), cg_ = (IT)-pr_ * CG;                                             \
  const int nb_ = TILED ? dft_div(pr_, pl.inv_t) : pr_, tile_ = TILED ? pr_ - nb_ * pl.T : 0;               \
  const int ty_ = TILED ? dft_div(tile_, pl.inv_tx) : 0, tx_ = TILED ? tile_ - ty_ * pl.TX : 0;             \
  const int Y0 = TILED ? ty_ * pl.TH - pl.oy : 0, X0 = TILED ? tx_ * pl.TW - pl.ox : 0;                     \
  const int c0_ = cg_ * G;
#define DFT_FWD_POS(TID, S)                                                                                  \
  const int i_ = (TID) + (S)*THR;                                                                        \
  const int r_ = dft_div(i_, pl.inv_c4), c4_ = i_ - r_ * (Wk / 4);                                          \
  const int y_ = Y0 + r_, x_ = X0 + 4 * c4_;
  // (a macro, not a lambda: register arrays captured by a closure end up in scratch memory with this compiler)
#define DFT_FWD_PREFETCH(IT, TID) DFT_FWD_PREFETCH_SLOTS(IT, TID, 0, NSLOT)
#define DFT_FWD_PREFETCH_SLOTS(IT, TID, S0, S1)                                                              \
  {                                                                                                          \
    DFT_FWD_ITER(IT)                                                                                         \
    _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                                                    \
      DFT_FWD_POS(TID, s)                                                                                    \
      if (FAST) {                                                                                            \
        const bool ok = i_ < npos && r_ < LH && 4 * c4_ < LW;                                                \
        const int off = ok ? y_ * W + x_ : 0;                                                                \
        pn[s] = *reinterpret_cast<const f32x4v*>(invn + (size_t)nb_ * HW + off);                             \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                  \
          const int c = c0_ + g < C ? c0_ + g : C - 1;                                                       \
          pc[s][g] = *reinterpret_cast<const f32x4v*>(corr + ((size_t)nb_ * C + c) * HW + off);              \
        }                                                                                                    \
      } else {                                                                                               \
        /* rows of any width / tiles: the slot's 4 columns as ONE 16-byte load with 4-byte alignment (global loads need no   \
           more) wherever the 16 bytes lie inside the image's own plane - columns left of x = 0 / right of x = W - 1 then    \
           come from the neighbouring rows and are zeroed in the W phase like everything outside the window; only the slots  \
           that would leave the plane (first / last row) take the per-element loads, behind a branch that most waves skip.   \
           Round 4 loaded every element on its own: 60 load instructions per thread and iteration instead of 15, and the     \
           levels that take this path (W % 4 != 0, every tiled level) paid 1.8 - 2.1x per location for their forward         \
           transform (profiles/r05/pyramid_levels.txt). */                                                                   \
        const bool rok = i_ < npos && r_ < LH && y_ >= 0 && y_ < H;                                          \
        const int off0 = y_ * W + x_;                                                                        \
        const bool vec = rok && off0 >= 0 && off0 + 3 < HW;                                                  \
        const int offv = vec ? off0 : 0;                                                                     \
        pn[s] = *reinterpret_cast<const f32x4u*>(invn + (size_t)nb_ * HW + offv);                            \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                      \
          const int c = c0_ + g < C ? c0_ + g : C - 1;                                                       \
          pc[s][g] = *reinterpret_cast<const f32x4u*>(corr + ((size_t)nb_ * C + c) * HW + offv);             \
        }                                                                                                    \
        if (rok && !vec) {                                                                                   \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
            const bool ok = 4 * c4_ + e < LW && x_ + e >= 0 && x_ + e < W;                                   \
            const int off = ok ? y_ * W + x_ + e : 0;                                                        \
            pn[s][e] = invn[(size_t)nb_ * HW + off];                                                         \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                  \
              const int c = c0_ + g < C ? c0_ + g : C - 1;                                                   \
              pc[s][g][e] = corr[((size_t)nb_ * C + c) * HW + off];                                          \
            }                                                                                                \
          }                                                                                                  \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  }
  // (round 4, tried: converting the prefetched window to fp16 pairs IN REGISTERS before the store loop, as the inverse kernel now
  // does with its maxima - the W phase fell from 1.6 to 1.0 us but the conversion cost 0.9 us where it went and the stores 0.3 us
  // more: 14.7 against 14.3 us per iteration, profiles/r04/dft_phases_v4_convert_before_stores.txt.  Not kept.)
  // XCD-aware order (work-group L runs on XCD L % 8): every XCD takes a contiguous range of iterations, so work-groups whose
  // outputs share cache lines / 16-byte units (neighbouring channel groups) run on ONE XCD at about the same time
  const int first = (DFT_GRID & 7) == 0 ? (DFT_BID & 7) * (DFT_GRID >> 3) + (DFT_BID >> 3) : DFT_BID;
  if (first < iters) DFT_FWD_PREFETCH(first, tid)
  DFT_BARRIER();     // FqT is in LDS

  DFT_STAMP_BEGIN()
  for (int it = first; it < iters; it += DFT_GRID) {
    int tl = tid;
#ifndef OS2D_HOST_EMU
    asm volatile("" : "+v"(tl));       // per-iteration addresses are recomputed, not hoisted (register pressure)
#endif
    // ... including everything derived from the lane number: the fragment, staging and store addresses of the six phases are
    // loop invariants, and hoisted out of the iteration loop they occupied ~30 registers through all of it (round 5: the
    // spills of round 4's kernels were these, parked around step 1)
    const int lane = tl & 63, l31 = lane & 31, hw = lane >> 5;
    DFT_FWD_ITER(it)
    // ---- W: x = relu(corr) * inv_norm * 2^15 as fp16 hi | lo units [w / 8][hi|lo][m = img * Pp + r]; zero outside the window
    {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        DFT_FWD_POS(tl, s)
        if (i_ < npos) {
          // inverse norm x 2^15, zero outside the window / the map: relu(a) * 0 = 0 for every finite a
          float nsc[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = FAST ? (r_ < LH && 4 * c4_ < LW) : (r_ < LH && y_ >= 0 && y_ < H && 4 * c4_ + e < LW && x_ + e >= 0 && x_ + e < W);
            nsc[e] = ok ? pn[s][e] * 32768.0f : 0.f;
          }
#pragma unroll
          for (int g = 0; g < G; ++g) {
            u32x2v hi = {0u, 0u}, lo = {0u, 0u};
            if (c0_ + g < C)         // (uniform: only the last channel group of a pair has channels beyond C)
              dft_split4(fmaxf(pc[s][g][0], 0.f) * nsc[0], fmaxf(pc[s][g][1], 0.f) * nsc[1], fmaxf(pc[s][g][2], 0.f) * nsc[2],
                         fmaxf(pc[s][g][3], 0.f) * nsc[3], &hi, &lo);
            const int m = g * Pp + r_;
            unsigned char* dst = ldsUb + ((size_t)((c4_ >> 1) * 2) * MxS + m) * 16 + (c4_ & 1) * 8;
            *reinterpret_cast<u32x2v*>(dst) = hi;
            *reinterpret_cast<u32x2v*>(dst + (size_t)MxS * 16) = lo;
          }
        }
      }
      // rows G * Pp .. Mx of the last row tile (only when G * Pp is not a multiple of 32): zeros
      for (int i = tid; i < (Mx - G * Pp) * (Wk / 8) * 2; i += THR) {
        const int m = G * Pp + i % (Mx - G * Pp), kh = i / (Mx - G * Pp);
        ldsU[(size_t)kh * MxS + m] = u32x4v{0u, 0u, 0u, 0u};
      }
    }
    DFT_BARRIER();
    DFT_STAMP(0)

    // ---- step 1: R^T = x . FqT; this wave's tiles t = wv + 8 j of the mt1n x nt1n grid
    f32x16v acc[3];
    int tm[3], tn[3];
    int nt1w = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = wv + NW * j;
      const bool mine = t < mt1n * nt1n;
      tn[j] = mine ? t / mt1n : 0;
      tm[j] = mine ? t - tn[j] * mt1n : -1;
      nt1w += mine ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    dft_product_lds_any<NW, DFT_PIPE_LDS ? 2 : 0>(acc, nt1w, 0, ks1n, ldsU, MxS, ldsF, N1, wv, mt1n, l31, hw);
    DFT_BARRIER();      // every wave is done reading x: the region becomes R2
    DFT_STAMP(1)

    // ---- R: R * 2^8 as units [k = (ri * Pp + h) / 8][hi|lo][n2 = img * V + v]; this lane owns column n = 2 v + ri of its
    // tiles and, per accumulator run, 4 consecutive h of one image: half a unit
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (tm[j] < 0) continue;
      const int n = tn[j] * 32 + l31, v = n >> 1, ri = n & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // rows tm * 32 + 8 q (+ 4 for the upper half-wave): Pp is a multiple of 8, so both half-waves are in the same image -
        // the division is wave-uniform (scalar unit, by multiplication)
        const int mb = tm[j] * 32 + 8 * q;
        const int img = dft_div(mb, pl.inv_pp), h0 = mb - img * Pp + 4 * hw;
        if (v < V && img < G) {
          u32x2v hi, lo;
          const float sc = 1.0f / 2097152.0f;      // 2^-21 = 2^8 / (2^15 * 2^14)
          dft_split4(acc[j][4 * q] * sc, acc[j][4 * q + 1] * sc, acc[j][4 * q + 2] * sc, acc[j][4 * q + 3] * sc, &hi, &lo);
          const int kg = ri * (Pp / 8) + (h0 >> 3);
          unsigned char* dst = ldsUb + ((size_t)(kg * 2) * N2S + img * V + v) * 16 + (h0 & 4) * 2;
          *reinterpret_cast<u32x2v*>(dst) = hi;
          *reinterpret_cast<u32x2v*>(dst + (size_t)N2S * 16) = lo;
        }
      }
    }
    DFT_BARRIER();
    DFT_STAMP(2)

    // ---- step 2: X = Fp2 . R2 (Fp2 fragments in registers); this wave's column tiles (wv >> 2) + 2 j of row tile wv & 3
    // (ALDS: tiles t = wv + NW j of the mt2n x nt2n grid, row tile fastest - as dft_product_lds counts them)
    f32x16v xc[3];
    int un[3], um[3];      // column / row tile of accumulator j, -1: none
    int nt2w = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = wv + NW * j;
      const int nt = ALDS ? t / mt2n : (wv >> 2) + (NW / 4) * j;
      const bool mine = ALDS ? t < mt2n * nt2n : (nt < nt2n && mt2 < mt2n);
      un[j] = mine ? nt : -1;
      um[j] = ALDS ? t - nt * mt2n : mt2;
      nt2w += mine ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) xc[j][r] = 0.f;
    }
    // the next window's loads are issued here - the accumulators of step 1 are dead, the values are needed a whole step 2 +
    // store phase later - so that their registers do not overlap the first product's
    // UNCONDITIONAL (round 5): the last iteration of a work-group requests its own window once more (cache hits, discarded).
    // Behind "if (it + DFT_GRID < iters)" the prefetch registers stayed live through the whole iteration on the path that skips
    // the loads - 60 registers the compiler had to keep next to the 64 of Fp2 in every phase - and the wait-count pass, merging
    // the two paths, could not count what is in flight.  One burst of NSLOT x (G + 1) requests, in front of the product
    // or right behind it (DFT_FWD_PREFETCH_AFTER).
    const int itn = it + DFT_GRID < iters ? it + DFT_GRID : it;
    {
      if (ALDS) {
        dft_product_lds_any<NW, DFT_PIPE_LDS ? 2 : 0>(xc, nt2w, 0, ks2n, ldsP, pl.M2, ldsU, N2S, wv, mt2n, l31, hw);
        DFT_FWD_PREFETCH_SLOTS(itn, tl, 0, NSLOT)
      } else if (DFT_FWD_PREFETCH_AFTER == 1 || (DFT_FWD_PREFETCH_AFTER == 2 && KS2 > 0 && KS2 <= 6)) {      // the burst behind the product
        dft_product_rega_any<KS2>(xc, nt2w, fp2h, fp2l, ks2n, ldsU, N2S, wv >> 2, NW / 4, l31, hw);
        DFT_FWD_PREFETCH_SLOTS(itn, tl, 0, NSLOT)
      } else {
        DFT_FWD_PREFETCH_SLOTS(itn, tl, 0, NSLOT)
        dft_product_rega_any<KS2>(xc, nt2w, fp2h, fp2l, ks2n, ldsU, N2S, wv >> 2, NW / 4, l31, hw);
      }
    }
    DFT_BARRIER();      // every wave is done reading R2: the region becomes the staging buffer of X
    DFT_STAMP(3)

    // ---- XS: X = acc * 2^-22 staged as [v][u / 4][img][u % 4][re|im] (+16 bytes per v: consecutive lanes = consecutive v land
    // in different banks); a lane owns (img, v) and per accumulator run two consecutive u
    const int XS = P * 8 * G + 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (un[j] < 0) continue;
      const int n2 = un[j] * 32 + l31;
      const int img = dft_div(n2, pl.inv_v), v = n2 - img * V;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u0 = (um[j] * 32 + 8 * q + 4 * hw) >> 1;
        if (img < G && u0 < P) {
          const float sc = 1.0f / 4194304.0f;      // 2^-22 = 1 / (2^8 * 2^14)
          f32x4v o;
          o[0] = xc[j][4 * q] * sc;
          o[1] = xc[j][4 * q + 1] * sc;
          o[2] = xc[j][4 * q + 2] * sc;
          o[3] = xc[j][4 * q + 3] * sc;
          *reinterpret_cast<f32x4v*>(ldsUb + (size_t)v * XS + (u0 >> 2) * (G * 32) + img * 32 + (u0 & 3) * 8) = o;
        }
      }
    }
    DFT_BARRIER();
    DFT_STAMP(4)

    // ---- ST: 16-byte pieces, 8 per quad of bins = the 128 contiguous bytes of the 4 channels; then the padding bins
    {
      const int per_v = (P / 4) * (G * 2), npieces = V * per_v;
      size_t qstride;                                            // floats between consecutive quads of bins
      float* dstbase = X + dft_spectra_pair0(pr_, NBT, pl.NBINS / 4, Cpad, &qstride) + (size_t)c0_ * 8;
      for (int i = tl; i < npieces; i += THR) {
        const int v = dft_div(i, pl.inv_pq), rem = i - v * per_v;
        const int uq = rem / (2 * G), jj = rem - uq * (2 * G);
        const f32x4v val = *reinterpret_cast<const f32x4v*>(ldsUb + (size_t)v * XS + uq * (G * 32) + jj * 16);
        DFT_STREAM_STORE(reinterpret_cast<f32x4v*>(dstbase + (size_t)(v * (P / 4) + uq) * qstride + jj * 4), val);
      }
      const int qpad0 = (P * V) / 4, qpad1 = pl.NBINS / 4;
      for (int i = tl; i < (qpad1 - qpad0) * 2 * G; i += THR)
        *reinterpret_cast<f32x4v*>(dstbase + (size_t)(qpad0 + i / (2 * G)) * qstride + (i % (2 * G)) * 4) = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    DFT_BARRIER();      // the staging buffer is free: the next window may be written
    DFT_STAMP(5)
  }
  DFT_STAMP_END(0)
#undef DFT_FWD_ITER
#undef DFT_FWD_POS
#undef DFT_FWD_PREFETCH
#undef DFT_FWD_PREFETCH_SLOTS
}

// ---------------------------------------------------------------------------------------------------- inverse transform
// iteration it -> (pair' = it / OG, output channel group og = it % OG): output channels G og .. G og + G - 1 of pair' = nb * T + tile
// G = 4: E2 in registers, 8-byte halves of the activation units; G = 8 (dft_plan_transform): E2 in LDS, whole 16-byte units
template <bool TILED, int KSA = 0, int G = DFT_G>
DFT_DEV void dft_inverse_body(const float* Y,        // [NBT / 64][NBINS / 4][64][Cout][4][2] (dft_spectra_pair0)
                              const float* bp,       // [3][MTP]: bias | - | 2^out_exp
                              int MTP, unsigned char* out,   // SHB [NB][Cout / 8][2][PLANE] x 16 B
                              const u32x4v* E2, const u32x4v* Gq, const DftPlan& pl, int Cout, int NBT, int PLANE, int Ws,
                              int BASE, int iters, DFT_FLAG bad_flag, int zero_borders) {
  unsigned char* smem = DFT_LDS;
  const int tid = DFT_TID, lane = tid & 63, l31 = lane & 31, hw = lane >> 5;
  const int wv = DFT_UNIFORM(tid >> 6);
  const int P = pl.P, V = pl.V, Pp = pl.Pp, N2 = pl.N2, MB = pl.MB, KB = pl.KB, NBo = pl.NBo, H = pl.H, W = pl.W;
  const int N2S = N2 + 1, MBS = MB + 1;
  const int OG = Cout / G;
  constexpr bool ALDS = G == 8;      // the row operand of step A lives in LDS
  constexpr int LG = G == 8 ? 3 : 2; // log2 G
  u32x4v* ldsG = reinterpret_cast<u32x4v*>(smem);                                   // Gq [KB / 8][2][NBo]
  u32x4v* ldsU = reinterpret_cast<u32x4v*>(smem + pl.lds_const);                    // Y2 | Tt
  unsigned char* ldsUb = smem + pl.lds_const;
  float* smaxw = reinterpret_cast<float*>(smem + pl.lds_const + pl.lds_union);      // [8 waves][G]: |Y| maxima
  const int nG = (KB / 8) * 2 * NBo, NQ = dft_round_up(pl.Q, 32);     // the Gq array has round_up(Q, 32) columns, NBo of them are needed
  for (int i = tid; i < nG; i += DFT_THR) {
    const int kh = i / NBo, n = i - kh * NBo;
    ldsG[i] = Gq[(size_t)kh * NQ + n];
  }
  const int MAfull = dft_round_up(2 * P, 32);                    // row count of the E2 array
  u32x4v* ldsE = ldsG + nG;                                      // ALDS: E2 [2 Pp / 8][2][MAfull]
  if (ALDS)
    for (int i = tid; i < (2 * Pp / 8) * 2 * MAfull; i += DFT_THR) ldsE[i] = E2[i];

  const int mtAn = pl.MA / 32, ntAn = N2 / 32, ksAn = 2 * Pp / 16;
  const int mtBn = MB / 32, ntBn = NBo / 32, ksBn = KB / 16;
  const int mtA = wv & 3;
  constexpr int KRA = KSA ? KSA : DFT_KREG;      // KSA > 0: exactly the k-steps of step A (ksAn == KSA)
  half8 e2h[KRA], e2l[KRA];
  if (!ALDS) {
#pragma unroll
    for (int ks = 0; ks < KRA; ++ks) {
      const int kc = ks < ksAn ? ks : 0, mc = mtA < mtAn ? mtA : 0;
      e2h[ks] = dft_frag(E2 + ((size_t)((2 * kc + hw) * 2 + 0)) * MAfull + mc * 32 + l31);
      e2l[ks] = dft_frag(E2 + ((size_t)((2 * kc + hw) * 2 + 1)) * MAfull + mc * 32 + l31);
    }
#pragma unroll
    for (int ks = 0; ks < KRA; ++ks) {      // landed on every path into the loop: see the Fp2 fragments of the forward kernel
      DFT_LANDED(e2h[ks]);
      DFT_LANDED(e2l[ks]);
    }
  }

  // ---- register prefetch: item = (img, v, octet of u) -> the two quads of bins u0 .. u0 + 7 of column v: 2 x 32 bytes
  constexpr int NITEM = 3;                                      // ceil(G * V * Pp / 8 / 512) <= 4 * 48 * 8 / 512 (G = 8: 8 * 32 * 6 / 512)
  const int uoct = Pp / 8, nitem = G * V * uoct;
  f32x4v py[NITEM][4];
#define DFT_INV_PREFETCH(IT, TID) DFT_INV_PREFETCH_ITEMS(IT, TID, 0, NITEM)
#define DFT_INV_PREFETCH_ITEMS(IT, TID, S0, S1)                                             \
  {                                                                                         \
    const int pr_ = dft_div((IT), pl.inv_og), og_ = (IT)-pr_ * OG;                          \
    size_t qstride;                                                                         \
    const float* src_ = Y + dft_spectra_pair0(pr_, NBT, pl.NBINS / 4, Cout, &qstride) + (size_t)og_ * G * 8; \
    _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                                   \
      const int e_ = (TID) + s * DFT_THR;                                                   \
      const int ec_ = e_ < nitem ? e_ : 0;                                                  \
      const int pimg_ = ec_ & (G - 1), prest_ = ec_ >> LG;                                  \
      const int pv_ = dft_div(prest_, pl.inv_kg), puo_ = prest_ - pv_ * (Pp / 8);           \
      const int q0_ = pv_ * (P / 4) + 2 * puo_;                                             \
      const bool ptwo_ = 8 * puo_ + 4 < P;                                                  \
      const float* p0_ = src_ + (size_t)q0_ * qstride + pimg_ * 8;                          \
      const float* p1_ = src_ + (size_t)(ptwo_ ? q0_ + 1 : q0_) * qstride + pimg_ * 8;      \
      py[s][0] = *reinterpret_cast<const f32x4v*>(p0_);                                     \
      py[s][1] = *reinterpret_cast<const f32x4v*>(p0_ + 4);                                 \
      py[s][2] = *reinterpret_cast<const f32x4v*>(p1_);                                     \
      py[s][3] = *reinterpret_cast<const f32x4v*>(p1_ + 4);                                 \
    }                                                                                       \
  }
#define DFT_INV_ITEM(E)                                                                     \
  const int img_ = (E) & (G - 1), rest_ = (E) >> LG;                                        \
  const int v_ = dft_div(rest_, pl.inv_kg), uo_ = rest_ - v_ * uoct;                        \
  const bool two_ = 8 * uo_ + 4 < P;
  // ---- M: the largest |component| of every image among the spectra in the prefetch registers (a thread's items all belong to
  // image tid % G) -> smaxw.  Run for the NEXT iteration's spectra right after step B, BEFORE the epilogue's stores: the first
  // use of the prefetched registers makes the compiler wait for them, and a wait placed after a loop of stores cannot count
  // what is in flight - it becomes vmcnt(0) and the work-group sat out the write latency of its own epilogue at the top of
  // every iteration (phase stamps: 1.1 us for this handful of instructions).  Up here the loads are the youngest requests.
#define DFT_INV_MAXIMA(TID)                                                                 \
  {                                                                                         \
    float m = 0.f;                                                                          \
    _Pragma("unroll") for (int s = 0; s < NITEM; ++s)                                       \
      if ((TID) + s * DFT_THR < nitem) {                                                    \
        DFT_INV_ITEM((TID) + s * DFT_THR)                                                   \
        (void)img_;                                                                         \
        (void)v_;                                                                           \
        (void)uo_;                                                                          \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                       \
          if (k < 2 || two_) {                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(py[s][k][e])); \
          }                                                                                 \
      }                                                                                     \
    if (G == 4) m = fmaxf(m, DFT_SHFL_XOR(m, 4));                                           \
    m = fmaxf(m, DFT_SHFL_XOR(m, 8));                                                       \
    m = fmaxf(m, DFT_SHFL_XOR(m, 16));                                                      \
    m = fmaxf(m, DFT_SHFL_XOR(m, 32));                                                      \
    if (lane < G) smaxw[wv * G + lane] = m;                                                 \
  }
  bool bad = false;
  const int first = (DFT_GRID & 7) == 0 ? (DFT_BID & 7) * (DFT_GRID >> 3) + (DFT_BID >> 3) : DFT_BID;     // XCD-aware (see the forward kernel)
  if (first < iters) {
    DFT_INV_PREFETCH(first, tid)
    DFT_INV_MAXIMA(tid)
  }
  DFT_BARRIER();

  DFT_STAMP_BEGIN()
  for (int it = first; it < iters; it += DFT_GRID) {
    int tl = tid;
#ifndef OS2D_HOST_EMU
    asm volatile("" : "+v"(tl));
#endif
    const int lane = tl & 63, l31 = lane & 31, hw = lane >> 5;      // (not hoisted: see the forward kernel)
    const int pr = dft_div(it, pl.inv_og), og = it - pr * OG;
    const int nb = TILED ? dft_div(pr, pl.inv_t) : pr, tile = TILED ? pr - nb * pl.T : 0;
    const int ty = TILED ? dft_div(tile, pl.inv_tx) : 0, tx = TILED ? tile - ty * pl.TX : 0;
    const int y0 = TILED ? ty * pl.TH : 0, x0 = TILED ? tx * pl.TW : 0, oy = TILED ? pl.oy : 0, ox = TILED ? pl.ox : 0;
    const int TH_ = TILED ? pl.TH : H, TW_ = TILED ? pl.TW : W;
    const int o0 = og * G;

    // (the maxima of this iteration's spectra are in smaxw: written before the previous iteration's epilogue / in the prologue)
    DFT_STAMP(0)
    float simg = 1.f, cinv[4] = {0.f, 0.f, 0.f, 0.f};      // cinv: of this lane's 4 channels in the epilogue (G = 8: 4 hw .. 4 hw + 3)
    {
      // scale of an image: 2^(13 - E), E = floor(log2(max)): the largest component lands in [2^13, 2^14)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < DFT_WAVES; ++w) m = fmaxf(m, smaxw[w * G + g]);
        int E = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;
        if (!(m > 0.f) || E < -100) E = -100;
        if (E > 100) E = 100;
        const float s = __builtin_bit_cast(float, (unsigned)(13 - E + 127) << 23);
        const float sinv = __builtin_bit_cast(float, (unsigned)(E - 13 + 127) << 23);
        if (g == (tl & (G - 1))) simg = s;
        // output = acc * 2^(eT - 13) / (scale * P * Q)
        const float ci = sinv * __builtin_bit_cast(float, (unsigned)(pl.eT - 13 + 127) << 23) / (float)(P * pl.Q);
        if (G == 4 || (g >> 2) == hw) cinv[g & 3] = ci;      // (compile-time g: a select per value, no indexed register array)
      }
    }
    // ---- WY: units [k = (ri * Pp + u) / 8][hi|lo][n = img * V + v]: this item's 8 u of column (img, v), re and im
#pragma unroll
    for (int s = 0; s < NITEM; ++s) {
      const int e = tl + s * DFT_THR;
      if (e < nitem) {
        DFT_INV_ITEM(e)
        const int img = img_, v = v_, uo = uo_;
        const bool two = two_;
        float re[8], im[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = k < 2 || two;
          re[2 * k] = ok ? py[s][k][0] * simg : 0.f;
          im[2 * k] = ok ? py[s][k][1] * simg : 0.f;
          re[2 * k + 1] = ok ? py[s][k][2] * simg : 0.f;
          im[2 * k + 1] = ok ? py[s][k][3] * simg : 0.f;
        }
        u32x2v h0, l0, h1, l1;
        u32x4v t;
        const size_t n = (size_t)img * V + v;
        dft_split4(re[0], re[1], re[2], re[3], &h0, &l0);
        dft_split4(re[4], re[5], re[6], re[7], &h1, &l1);
        t[0] = h0[0], t[1] = h0[1], t[2] = h1[0], t[3] = h1[1];
        ldsU[(size_t)((uo)*2 + 0) * N2S + n] = t;
        t[0] = l0[0], t[1] = l0[1], t[2] = l1[0], t[3] = l1[1];
        ldsU[(size_t)((uo)*2 + 1) * N2S + n] = t;
        dft_split4(im[0], im[1], im[2], im[3], &h0, &l0);
        dft_split4(im[4], im[5], im[6], im[7], &h1, &l1);
        t[0] = h0[0], t[1] = h0[1], t[2] = h1[0], t[3] = h1[1];
        ldsU[(size_t)((uoct + uo) * 2 + 0) * N2S + n] = t;
        t[0] = l0[0], t[1] = l0[1], t[2] = l1[0], t[3] = l1[1];
        ldsU[(size_t)((uoct + uo) * 2 + 1) * N2S + n] = t;
      }
    }
    DFT_BARRIER();
    DFT_STAMP(1)

    // ---- step A: T = E2 . Y2 (E2 fragments in registers)
    // (ALDS: tiles t = wv + 8 j of the mtAn x ntAn grid, row tile fastest - as dft_product_lds counts them)
    f32x16v ta[3];
    int an[3], am[3];      // column / row tile of accumulator j, -1: none
    int ntaw = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = wv + DFT_WAVES * j;
      const int nt = ALDS ? t / mtAn : (wv >> 2) + 2 * j;
      const bool mine = ALDS ? t < mtAn * ntAn : (nt < ntAn && mtA < mtAn);
      an[j] = mine ? nt : -1;
      am[j] = ALDS ? t - nt * mtAn : mtA;
      ntaw += mine ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) ta[j][r] = 0.f;
    }
    // the next spectra: unconditional, one burst next to the product (see the forward kernel)
    const int itn = it + DFT_GRID < iters ? it + DFT_GRID : it;
    {
      if (ALDS) {
        dft_product_lds_any<DFT_WAVES, DFT_PIPE_LDS ? 1 : 0>(ta, ntaw, 0, ksAn, ldsE, MAfull, ldsU, N2S, wv, mtAn, l31, hw);
        DFT_INV_PREFETCH_ITEMS(itn, tl, 0, NITEM)
      } else if (DFT_INV_PREFETCH_AFTER == 1 || (DFT_INV_PREFETCH_AFTER == 2 && KSA > 0 && KSA <= 6)) {
        dft_product_rega_any<KSA>(ta, ntaw, e2h, e2l, ksAn, ldsU, N2S, wv >> 2, 2, l31, hw);
        DFT_INV_PREFETCH_ITEMS(itn, tl, 0, NITEM)
      } else {
        DFT_INV_PREFETCH_ITEMS(itn, tl, 0, NITEM)
        dft_product_rega_any<KSA>(ta, ntaw, e2h, e2l, ksAn, ldsU, N2S, wv >> 2, 2, l31, hw);
      }
    }
    DFT_BARRIER();      // every wave is done reading Y2: the region becomes Tt
    DFT_STAMP(2)

    // ---- WT: T * 2^-(14 + eT) as units [k = (2 v + ri) / 8][hi|lo][m = h * G + img]; a lane owns (img, v) and per accumulator
    // run (re, im) of two consecutive h: 4 bytes of a unit each.  The k beyond 2 V of the last units are zeroed (Gq has zero
    // rows there, but 0 * NaN-patterned garbage is NaN).
    {
      const float sc = __builtin_bit_cast(float, (unsigned)(127 - 14 - pl.eT) << 23);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (an[j] < 0) continue;
        const int n = an[j] * 32 + l31;
        const int img = dft_div(n, pl.inv_v), v = n - img * V;
        if (img >= G) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int h0 = (am[j] * 32 + 8 * q + 4 * hw) >> 1;
          u32x2v hi, lo;
          dft_split4(ta[j][4 * q] * sc, ta[j][4 * q + 1] * sc, ta[j][4 * q + 2] * sc, ta[j][4 * q + 3] * sc, &hi, &lo);
          unsigned char* d0 = ldsUb + ((size_t)((v >> 2) * 2) * MBS + (size_t)h0 * G + img) * 16 + (v & 3) * 4;
          *reinterpret_cast<unsigned*>(d0) = hi[0];
          *reinterpret_cast<unsigned*>(d0 + (size_t)MBS * 16) = lo[0];
          *reinterpret_cast<unsigned*>(d0 + G * 16) = hi[1];
          *reinterpret_cast<unsigned*>(d0 + G * 16 + (size_t)MBS * 16) = lo[1];
        }
      }
      for (int v = V; v < KB / 2; ++v)       // v slots V .. KB / 2 - 1
        for (int m = tl; m < MB; m += DFT_THR) {
          unsigned char* d0 = ldsUb + ((size_t)((v >> 2) * 2) * MBS + m) * 16 + (v & 3) * 4;
          *reinterpret_cast<unsigned*>(d0) = 0u;
          *reinterpret_cast<unsigned*>(d0 + (size_t)MBS * 16) = 0u;
        }
    }
    DFT_BARRIER();
    DFT_STAMP(3)

    // ---- step B: y = Tt . Gq; this wave's tiles t = wv + 8 j of the mtBn x ntBn grid
    f32x16v yc[3];
    int bm[3], bn[3];
    int ntbw = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = wv + DFT_WAVES * j;
      const bool mine = t < mtBn * ntBn;
      bn[j] = mine ? t / mtBn : 0;
      bm[j] = mine ? t - bn[j] * mtBn : -1;
      ntbw += mine ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) yc[j][r] = 0.f;
    }
    dft_product_lds_any<DFT_WAVES, DFT_PIPE_LDS ? 1 : 0>(yc, ntbw, 0, ksBn, ldsU, MBS, ldsG, NBo, wv, mtBn, l31, hw);
    DFT_INV_MAXIMA(tl)      // of the NEXT iteration's spectra (requested before step A; the last iteration's result is not used)
    DFT_STAMP(4)

    // ---- epilogue: a lane owns window column w and, per accumulator run, 4 channels of one row: + bias, ReLU, channel
    // scale, fp16 hi | lo.  G = 4: 8 + 8 bytes of the two 16-byte units of the cell (the other half of a unit comes from the
    // work-group of the neighbouring channel group).  G = 8: rows are (h, img) with 8 images per h, so the lower half-wave holds
    // channels 0 - 3 and the upper one channels 4 - 7 of the SAME cell: they swap halves and each stores one whole unit.
    {
      float bias[4], osc[4];     // of this lane's 4 channels: o0 + (G = 8: 4 hw) + k
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bias[g] = bp[o0 + (G == 8 ? 4 * hw : 0) + g];
        osc[g] = bp[2 * MTP + o0 + (G == 8 ? 4 * hw : 0) + g];
      }
      const int grp = o0 >> 3, slot = o0 & 7;
      unsigned char* hi_unit = out + (((size_t)nb * ((Cout + 7) >> 3) + grp) * 2 + 0) * (size_t)PLANE * 16 + slot * 2;
      unsigned char* lo_unit = out + (((size_t)nb * ((Cout + 7) >> 3) + grp) * 2 + 1) * (size_t)PLANE * 16 + slot * 2;
      // the zero border of the two planes (cells in front of the map, the 3 pad cells after every row, the tail), this
      // iteration's 8-byte half of every unit: ~3 stores per thread instead of a launch that writes all borders on its own
      // (9 us per step at 64 classes, 0.2 ms at 1024).  Once per (pair, channel group): by the first tile of a tiled map.
      if (zero_borders && (!TILED || tile == 0)) {
        const int rows = H * (Ws - W), tail0 = BASE + H * Ws, npad = BASE + rows + (PLANE - tail0);
        for (int k = tl; k < npad; k += DFT_THR) {
          int cell;
          if (k < BASE) cell = k;
          else if (k < BASE + rows) {
            const int jj = k - BASE, hr = jj / (Ws - W);
            cell = BASE + hr * Ws + W + (jj - hr * (Ws - W));
          } else cell = tail0 + (k - BASE - rows);
          if (G == 8) {
            *reinterpret_cast<u32x4v*>(hi_unit + (size_t)cell * 16) = u32x4v{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4v*>(lo_unit + (size_t)cell * 16) = u32x4v{0u, 0u, 0u, 0u};
          } else {
            *reinterpret_cast<u32x2v*>(hi_unit + (size_t)cell * 16) = u32x2v{0u, 0u};
            *reinterpret_cast<u32x2v*>(lo_unit + (size_t)cell * 16) = u32x2v{0u, 0u};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (bm[j] < 0) continue;
        const int wwin = bn[j] * 32 + l31, tw = wwin - ox;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // row m = h G + img of the window: G = 4: 4 rows = the 4 images of h = m / 4; G = 8: img = 4 hw .. 4 hw + 3 of h = m / 8
          const int hwin = G == 8 ? bm[j] * 4 + q : (bm[j] * 32 + 8 * q + 4 * hw) >> 2, th = hwin - oy;
          const int h = y0 + th, w = x0 + tw;
          const bool inside = th >= 0 && th < TH_ && tw >= 0 && tw < TW_ && h < H && w < W;      // (G = 8: the same for lanes l, l + 32)
          if (G == 8 || inside) {
            float t[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float pre = yc[j][4 * q + g] * cinv[g] + bias[g];
              t[g] = fmaxf(pre, 0.f) * osc[g];
              // out of the fp16 range, or NaN: the pre-activation is tested too, fmaxf(NaN, 0) = 0 hid a NaN spectrum
              if (inside && (!(fabsf(t[g]) <= 65504.f) || pre != pre)) bad = true;
            }
            u32x2v hi, lo;
            dft_split4(t[0], t[1], t[2], t[3], &hi, &lo);
            const size_t off = ((size_t)BASE + (size_t)h * Ws + w) * 16;
            if (G == 8) {
              // the lower lane sends its lo half and receives the partner's hi half (-> the complete hi unit), the upper lane sends its
              // hi half and receives the partner's lo half (-> the complete lo unit); every lane of the wave takes part
              const u32x2v send = hw ? hi : lo, keep = hw ? lo : hi;
              u32x2v recv;
              recv[0] = DFT_SHFL_XOR_U32(send[0], 32);
              recv[1] = DFT_SHFL_XOR_U32(send[1], 32);
              const u32x4v unit = hw ? u32x4v{recv[0], recv[1], keep[0], keep[1]} : u32x4v{keep[0], keep[1], recv[0], recv[1]};
              if (inside) *reinterpret_cast<u32x4v*>((hw ? lo_unit : hi_unit) + off) = unit;
            } else {
              *reinterpret_cast<u32x2v*>(hi_unit + off) = hi;
              *reinterpret_cast<u32x2v*>(lo_unit + off) = lo;
            }
          }
        }
      }
    }
    DFT_BARRIER();      // every wave is done reading Tt (and the maxima): the next spectra may be written
    DFT_STAMP(5)
  }
  DFT_STAMP_END(8)
  if (DFT_FLAG_SET(bad_flag) && DFT_BALLOT(bad) != 0ull) {
    if ((tid & 63) == 0) DFT_RAISE(bad_flag);
  }
#undef DFT_INV_MAXIMA
#undef DFT_INV_ITEM
#undef DFT_INV_PREFETCH
#undef DFT_INV_PREFETCH_ITEMS
}

}  // namespace os2d_dft

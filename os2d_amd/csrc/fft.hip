// Two-dimensional real FFT pair for the frequency-domain form of the 7x7 TransformNet layer (gfx950) - building blocks,
// see spectral.hip.  Everything of one image lives in LDS; one work-group per image at a time.
//
//   fft_forward   x[h][w] = relu(corr[nb][c][h][w]) * inv_norm[nb][h][w]  (the TransformNet input normalisation of
//                 reference head.py:650 folded into the load), zero-padded to P x Q  ->  X[c][nb][u * V + v], V = Q/2 + 1
//   fft_inverse   Y[nb][o][u * V + v]  ->  y[h][w] = the first H x W samples of the inverse transform / (P * Q), then the
//                 layer's epilogue: + bias, ReLU, per-channel power-of-two scale, fp16 hi|lo split into the split-half
//                 blocked activation buffer of conv_f16x3.hip (BatchNorm is folded into the weight spectra and the bias)
//
// Rows: two REAL rows are packed into one complex FFT of length Q (z = x_2p + i x_2p+1) and untangled afterwards
// (X_2p[v] = (Z[v] + conj Z[Q-v]) / 2, X_2p+1[v] = (Z[v] - conj Z[Q-v]) / 2i); columns: V complex FFTs of length P.
// Each 1-D FFT is a sequence of Stockham (auto-sort, out-of-place) passes of radix 4 / 3 / 2 over ping-pong buffers in LDS
// (P and Q are products of 2s and 3s: 64 x 96 for a 60 x 80 map with its 3-cell halo); twiddles come from exact tables
// (float64 on the host, rounded once).  The kernel is centred on the map (the weight spectra carry the -3 shift), so only
// P >= H + 3 and Q >= W + 3 are needed and the result is cropped at the origin.
//
// Maps that do not fit the in-LDS transform (P x Q beyond ~96 x 128: the 96 x 128 level of BASELINE.json configs[4], anything up
// to the 209-column limit of the other kernels) are cut into TY x TX TILES (overlap-save): tile (ty, tx) produces the outputs
// of rows ty*TH .. ty*TH+TH-1 / columns tx*TW .. from an input window that starts 3 cells earlier and is 6 cells longer along
// every tiled axis (cells outside the map are zero), transformed at P >= TH + 6, Q >= TW + 6; the wanted outputs then sit at
// offset (3, 3) of the inverse transform and no wrap-around reaches them.  A tile is just one more "image" for all three
// kernels: X is [C][NB * T][NBINS], Y is [NB * T][Cout][NBINS], pair' = nb * T + tile; the weight spectra depend on (P, Q)
// only, so every map that tiles to the same transform size shares them.
#include "os2d_common.h"
#include <map>
#include <mutex>
#include <utility>
#include "fft_regs.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int FFT_THR = 512;
constexpr int MAXPASS = 8;
constexpr int FFT_EPT_MAX = 14;  // elements per thread the register prefetch can hold at most (bounds the map size, see
                                 // make_plan); the kernels are instantiated for 6 / 10 / 14 so that the 60 x 80 map of the
                                 // benchmark keeps 4 waves per SIMD (two work-groups per CU)

struct FftPlan {
  int P, Q, V;               // padded sizes, V = Q/2 + 1
  int np_row, np_col;        // number of passes
  int rad_row[MAXPASS], rad_col[MAXPASS];
  int PS;                    // padded stride of a column in LDS (P + 1: keeps the transposed accesses off one bank)
  int QS;                    // padded stride of a row pair in LDS (Q + 1: the row -> column transposition reads one
                             // element of every row pair per lane; with stride Q = 96 complex they all sit in ONE bank)
  int row_r1, row_r2;        // two-stage register factorisation Q = r1 * r2 of fft_regs.h (0: Stockham passes)
  int col_r1, col_r2;        // ... of P
  int zs_row, zs_col;        // odd strides of a sequence in the exchange buffer of the two-stage form
  unsigned inv_hp, inv_v;    // ceil(2^32 / HP), ceil(2^32 / V) (0 for a single sequence): item -> (index, sequence)
  unsigned inv_q;            // ceil(2^32 / Q)
  int AB;                    // complex numbers of the A | B region = max(2 * ceil(H/2) * Q, V * PS): the second column
                             // buffer D aliases it
  // tiling (T = TY * TX = 1: the whole map in one transform, offsets 0)
  int T, TY, TX;             // tiles per map
  int TH, TW;                // output rows / columns of a tile (the last tile of an axis may reach beyond the map)
  int oy, ox;                // 3 along a tiled axis (input window starts oy rows above the tile, outputs sit at row oy), else 0
  int LH, LW;                // rows / columns of the input window a tile loads (TH + 6 | H, TW + 6 | W)
  int RH;                    // rows of the inverse transform that are needed (oy + TH | H)
  unsigned inv_t, inv_tx, inv_tw;   // ceil(2^32 / T), ceil(2^32 / TX), ceil(2^32 / TW) (0 where the divisor is 1)
  unsigned inv_hpr;          // ceil(2^32 / ceil(RH / 2)): the inverse kernel transforms only the row pairs it needs
};

__host__ __device__ __forceinline__ unsigned magic_div(unsigned d) { return d > 1 ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u; }
__device__ __forceinline__ int div_magic(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }

// Work-group barrier for data exchanged through LDS only: waits for this wave's LDS operations, NOT for its global loads
// and stores.  __syncthreads() carries a full fence (s_waitcnt vmcnt(0)): inside the per-image loop it would wait for the
// prefetch of the next image and for the stores of the previous one at every one of the ~10 barriers of an image - the
// whole HBM latency serialised per image (measured: 8 us of an image's 17 us).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) { return f32x2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }
__device__ __forceinline__ f32x2 cconj(f32x2 a) { return f32x2{a[0], -a[1]}; }

// One Stockham pass of radix R over `nfft` transforms of length N (Ns = product of the radices already done).
// INV = false: forward (exp(-2 pi i ...)), true: inverse (conjugate twiddles and butterflies, no 1/N).
template <int R, bool INV>
__device__ __forceinline__ void stockham_pass(const f32x2* __restrict__ in, f32x2* __restrict__ out, int N, int Ns, int nfft,
                                              int stride, const f32x2* __restrict__ tw, int tid) {
  const int nb = N / R;          // butterflies per transform
  const int step = N / (Ns * R);
  // thread -> (transform f, butterfly j) with shifts only: 2^lg >= nb threads per transform (no integer division in the
  // loop; Ns is a power of two except in the last pass of a size with two factors 3)
  const int lg = 32 - __builtin_clz(nb - 1 > 0 ? nb - 1 : 1) - (nb == 1 ? 1 : 0);
  const int tpf = 1 << lg;
  const bool pow2 = (Ns & (Ns - 1)) == 0;
  for (int i = tid; i < (nfft << lg); i += FFT_THR) {
    const int f = i >> lg, j = i & (tpf - 1);
    if (j >= nb) continue;
    const int k = pow2 ? (j & (Ns - 1)) : (j % Ns);
    f32x2 v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      v[q] = in[f * stride + j + q * nb];
      if (q > 0 && Ns > 1) {      // the first pass (Ns = 1) has k = 0: all twiddles are 1
        f32x2 w = tw[q * k * step];
        if (INV) w = cconj(w);
        v[q] = cmul(v[q], w);
      }
    }
    if (R == 2) {
      const f32x2 a = v[0] + v[1], b = v[0] - v[1];
      v[0] = a;
      v[1] = b;
    } else if (R == 3) {
      const float s = 0.86602540378443864676f;   // sin(pi/3)
      const f32x2 t1 = v[1] + v[2];
      const f32x2 t2 = v[0] - 0.5f * t1;
      const f32x2 d = v[1] - v[2];
      // forward: -i s d = (s d.im, -s d.re); inverse: +i s d
      const f32x2 t3 = INV ? f32x2{-s * d[1], s * d[0]} : f32x2{s * d[1], -s * d[0]};
      v[0] = v[0] + t1;
      v[1] = t2 + t3;
      v[2] = t2 - t3;
    } else {  // R == 4
      const f32x2 a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], e = v[1] - v[3];
      const f32x2 d = INV ? f32x2{-e[1], e[0]} : f32x2{e[1], -e[0]};   // forward: -i e; inverse: +i e
      v[0] = a + c;
      v[1] = b + d;
      v[2] = a - c;
      v[3] = b - d;
    }
    const int j0 = (j - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) out[f * stride + j0 + q * Ns] = v[q];
  }
}

// all passes of one direction over ping-pong buffers; returns the buffer holding the result
template <bool INV>
__device__ __forceinline__ f32x2* fft_batch(f32x2* a, f32x2* b, int N, int nfft, int stride, int npass, const int* rad,
                                            const f32x2* tw, int tid) {
  int Ns = 1;
  for (int p = 0; p < npass; ++p) {
    const int R = rad[p];
    if (R == 4) stockham_pass<4, INV>(a, b, N, Ns, nfft, stride, tw, tid);
    else if (R == 3) stockham_pass<3, INV>(a, b, N, Ns, nfft, stride, tw, tid);
    else stockham_pass<2, INV>(a, b, N, Ns, nfft, stride, tw, tid);
    Ns *= R;
    lds_barrier();
    f32x2* t = a;
    a = b;
    b = t;
  }
  return a;
}

// N = r1 * r2 in the two-stage register form (fft_regs.h): a -> Z (= b) -> a; falls back to the Stockham passes for
// factorisations that are not instantiated.  Returns the buffer holding the result.
template <int R1, int R2, bool INV>
__device__ __forceinline__ f32x2* fft_two_stage(f32x2* a, f32x2* z, int nfft, int stride, int zstride, unsigned inv_nfft,
                                                const f32x2* tw, int tid) {
  os2d_fft::two_stage_first<R1, R2, INV, FFT_THR>(a, stride, z, zstride, nfft, inv_nfft, tw, tid);
  lds_barrier();
  os2d_fft::two_stage_second<R1, R2, INV, FFT_THR>(z, zstride, a, stride, nfft, inv_nfft, tid);
  lds_barrier();
  return a;
}

template <bool INV>
__device__ __forceinline__ f32x2* fft_any(int r1, int r2, f32x2* a, f32x2* b, int N, int nfft, int stride, int zstride,
                                          unsigned inv_nfft, int npass, const int* rad, const f32x2* tw, int tid) {
  switch (r1 * 32 + r2) {
#define OS2D_FFT_CASE(A_, B_) \
  case A_ * 32 + B_:          \
    return fft_two_stage<A_, B_, INV>(a, b, nfft, stride, zstride, inv_nfft, tw, tid);
    OS2D_FFT_CASE(6, 6)     // 36
    OS2D_FFT_CASE(6, 7)     // 42
    OS2D_FFT_CASE(8, 6)     // 48
    OS2D_FFT_CASE(9, 6)     // 54
    OS2D_FFT_CASE(8, 8)     // 64
    OS2D_FFT_CASE(9, 8)     // 72
    OS2D_FFT_CASE(12, 7)    // 84
    OS2D_FFT_CASE(12, 8)    // 96
    OS2D_FFT_CASE(12, 9)    // 108
    OS2D_FFT_CASE(16, 8)    // 128
#undef OS2D_FFT_CASE
    default:
      return fft_batch<INV>(a, b, N, nfft, stride, npass, rad, tw, tid);
  }
}

// LDS (in complex numbers): twiddles Q + P | rows A, B: 2 x (HP x Q), HP = ceil(H/2) row pairs | columns C: V x PS.
// The second column buffer D aliases A|B (the row stage is finished by then).
// TILED = false: the whole map in one transform (offsets 0, one image per (pair, channel)) - the address arithmetic of the
// tiles is compiled out (it cost the 60 x 80 benchmark map 10 % of this kernel: 0.25 -> 0.28 ms per 64 classes)
template <int FFT_EPT, bool TILED>
__global__ __launch_bounds__(FFT_THR, FFT_EPT <= 6 ? 4 : 2) void fft_forward_kernel(const float* __restrict__ corr,   // [NB][C][H*W]
                                                             const float* __restrict__ inv,    // [NB][H*W]
                                                             f32x2* __restrict__ X,            // [C][NB][NBINS]
                                                             const f32x2* __restrict__ twQ, const f32x2* __restrict__ twP,
                                                             FftPlan pl, int C, int H, int W, int NBINS, int images) {
  // images = NB * C * T: iteration `it` -> map it / T (= nb * C + c), tile it % T; spectra written as X[c][pair'][bin] with
  // pair' = nb * T + tile.  (Round 3 also tried X in quads of bins like Y below: the split-half GEMM's loads got 0.03 ms
  // cheaper per 64 pairs and this kernel's stores - 688 pieces of 32 bytes 460 KB apart per image - 0.04 ms dearer.)
  const int NBT = images / C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = pl.P, Q = pl.Q, V = pl.V, PS = pl.PS, QS = pl.QS, HP = (pl.LH + 1) >> 1, HW = H * W, LH = pl.LH, LW = pl.LW;
  f32x2* tQ = reinterpret_cast<f32x2*>(smem);
  f32x2* tP = tQ + Q;
  f32x2* A = tP + P;
  f32x2* Bf = A + HP * QS;
  f32x2* Cc = A + pl.AB;
  f32x2* D = A;                     // aliases A | B (the row stage is over when the columns start)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NWV = FFT_THR / 64;
  for (int i = tid; i < Q; i += FFT_THR) tQ[i] = twQ[i];
  for (int i = tid; i < P; i += FFT_THR) tP[i] = twP[i];
  // Every global load of an image is issued up front into registers (EPT elements per thread, their (row pair, column)
  // computed once), and the loads of the NEXT image are in flight while this one is transformed: the serial chain of a
  // work-group has no global-memory latency in it.
  const int nelem = HP * Q;
  // element k of a thread: (row pair p, column w) = (i / Q, i % Q) of i = tid + k * 512 -> source offset of its even row
  // (-1: zero padding), whether the odd row exists, its place in A.  Recomputed per image from an opaque copy of the thread
  // index (a multiply-high each): held across the image loop they cost 3 registers per element and spill
  // the tile's window starts at map row Y0 / column X0 (negative along a tiled axis: the halo above / left of the map is zero)
#define FFT_TILE(IT)                                                                                              \
  const int m_ = TILED ? div_magic((IT), pl.inv_t) : (IT), t_ = TILED ? (IT)-m_ * pl.T : 0;                       \
  const int ty_ = TILED ? div_magic(t_, pl.inv_tx) : 0, tx_ = TILED ? t_ - ty_ * pl.TX : 0;                       \
  const int Y0 = TILED ? ty_ * pl.TH - pl.oy : 0, X0 = TILED ? tx_ * pl.TW - pl.ox : 0;
#define FFT_ELEM(TID, K)                                                                                          \
  const int i_ = (TID) + (K)*FFT_THR;                                                                             \
  const int p_ = (int)__umulhi((unsigned)i_, pl.inv_q), w_ = i_ - p_ * Q;                                         \
  const int r_ = Y0 + 2 * p_, c_ = X0 + w_;                                                                       \
  const bool in_ = i_ < nelem && (TILED ? (w_ < LW && c_ >= 0 && c_ < W) : (w_ < W));                             \
  const bool e0_ = in_ && (TILED ? (r_ >= 0 && r_ < H && 2 * p_ < LH) : true);                                    \
  const bool e1_ = in_ && (TILED ? (r_ + 1 >= 0 && r_ + 1 < H && 2 * p_ + 1 < LH) : (r_ + 1 < H));                \
  const int edst_ = p_ * QS + w_;
  // RAW values only are held (correlation + inverse norm of the even and the odd row): any arithmetic here would make the
  // compiler wait for each load right where it is issued; addresses are clamped instead of predicated (no branches)
  float pa0[FFT_EPT], pn0[FFT_EPT], pa1[FFT_EPT], pn1[FFT_EPT];
#define FFT_PREFETCH(IT, TID)                                                                                     \
  {                                                                                                               \
    FFT_TILE(IT)                                                                                                  \
    const float* src_ = corr + (size_t)m_ * HW;                                                                   \
    const float* nv_ = inv + (size_t)(m_ / C) * HW;                                                               \
    _Pragma("unroll") for (int k = 0; k < FFT_EPT; ++k) {                                                         \
      FFT_ELEM(TID, k)                                                                                            \
      (void)edst_;                                                                                                \
      const int o0_ = e0_ ? r_ * W + c_ : 0, o1_ = e1_ ? (r_ + 1) * W + c_ : o0_;                                 \
      pa0[k] = src_[o0_];                                                                                         \
      pn0[k] = nv_[o0_];                                                                                          \
      pa1[k] = src_[o1_];                                                                                         \
      pn1[k] = nv_[o1_];                                                                                          \
    }                                                                                                             \
  }
  if (blockIdx.x < images) FFT_PREFETCH(blockIdx.x, tid)
  for (int img = blockIdx.x; img < images; img += gridDim.x) {
    int tl = tid;
    asm volatile("" : "+v"(tl));
    lds_barrier();
    // ---- row pair p, column w -> A[p][w] = (x[2p][w], x[2p+1][w]) with x = relu(corr) * inv_norm; zero beyond the map
    FFT_TILE(img)
#pragma unroll
    for (int k = 0; k < FFT_EPT; ++k) {
      FFT_ELEM(tl, k)
      if (i_ < nelem)
        A[edst_] = f32x2{e0_ ? fmaxf(pa0[k], 0.f) * pn0[k] : 0.f, e1_ ? fmaxf(pa1[k], 0.f) * pn1[k] : 0.f};
    }
    lds_barrier();
    if (img + (int)gridDim.x < images) FFT_PREFETCH(img + gridDim.x, tl)
    f32x2* R = fft_any<false>(pl.row_r1, pl.row_r2, A, Bf, Q, HP, QS, pl.zs_row, pl.inv_hp, pl.np_row, pl.rad_row, tQ, tid);
    // ---- untangle the two real rows of every pair and transpose into the column buffer C[v][u]; rows >= H are zero
    f32x2* Cb = (R == A) ? Cc : Cc;   // C is separate from A | B
    for (int v = wv; v < V; v += NWV)
      for (int u = lane; u < P; u += 64) {
        f32x2 o = f32x2{0.f, 0.f};
        if (u < LH) {
          const int p = u >> 1;
          const f32x2 z = R[p * QS + v], zc = cconj(R[p * QS + (v == 0 ? 0 : Q - v)]);
          if ((u & 1) == 0) o = 0.5f * (z + zc);
          else {
            const f32x2 d = z - zc;            // (Z - conj Z') / (2 i) = -i d / 2
            o = f32x2{0.5f * d[1], -0.5f * d[0]};
          }
        }
        Cb[v * PS + u] = o;
      }
    lds_barrier();
    f32x2* Rc = fft_any<false>(pl.col_r1, pl.col_r2, Cb, D, P, V, PS, pl.zs_col, pl.inv_v, pl.np_col, pl.rad_col, tP, tid);
    // ---- store the half spectrum (bin = u * V + v, v fastest) + zero padding bins
    const int nb_ = m_ / C, ch_ = m_ - nb_ * C;
    const size_t pair_ = TILED ? (size_t)nb_ * pl.T + t_ : (size_t)nb_;
    {
      // X[c][pair'][bin]: the 64 pairs a GEMM work-group reads for one channel lie in ONE 1.4 MB stretch (22 KB apart), not 5 MB
      // apart - its load instructions then need one address translation instead of one per pair
      f32x2* dst = X + ((size_t)ch_ * NBT + pair_) * NBINS;
      for (int u = wv; u < P; u += NWV)
        for (int v = lane; v < V; v += 64) dst[u * V + v] = Rc[v * PS + u];
      for (int i = P * V + tid; i < NBINS; i += FFT_THR) dst[i] = f32x2{0.f, 0.f};
    }
  }
}

// Inverse: Y -> columns (inverse FFT of length P over u for every v) -> re-tangle row pairs -> inverse complex FFT of
// length Q -> real rows; epilogue of the layer.
// CPT > 0: one work-group takes GRPT (2 or 4) consecutive output channels of an SHB unit in turn and keeps their fp16
// hi | lo results of its (at most CPT) cells per thread in shift registers, so that the SHB is written with 4- / 8-byte
// stores instead of 2-byte ones (measured at 64 pairs, 60x80: 0.377 ms with 2-byte stores, 0.318 ms with GRPT = 4, which
// spills 20 registers at the 128-VGPR budget of two work-groups per CU, 0.288 ms with GRPT = 2; 0.282 ms without stores);
// CPT == 0 (maps with more than 512 * 10 cells): one channel per iteration, 2-byte stores.
template <int FFT_EPT, int CPT, int GRPT, bool TILED>
__global__ __launch_bounds__(FFT_THR, FFT_EPT <= 10 ? 4 : 2) void fft_inverse_kernel(const f32x2* __restrict__ Y,      // [NB][Cout][NBINS]
                                                             const float* __restrict__ bp,     // [3][MTP]: bias | - | 2^out_exp
                                                             int MTP, char* __restrict__ out,  // SHB [NB][Cout/8][2][PLANE] x 16 B
                                                             const f32x2* __restrict__ twQ, const f32x2* __restrict__ twP,
                                                             FftPlan pl, int Cout, int H, int W, int NBINS, int PLANE,
                                                             int images, unsigned inv_v, Os2dRangeFlag status, int quad,
                                                             int out32 /* CPT == 0 only: out = fp32 planes [NB][Cout][PLANE] */) {
  // images = NB * T * Cout: image -> (pair' = nb * T + tile, output channel); RH = rows of the inverse that are needed.
  // quad == 0: Y[pair'][o][bin]; quad == 1 (what os2d_spectral_gemm_f16 writes): Y[bin / 4][pair'][o][bin % 4] - the GEMM
  // then stores 1 KB runs (32 lanes = 32 consecutive output channels x 32 bytes) and this kernel gathers 32-byte pieces
  // (its loads are prefetched a whole image ahead: measured 0.198 ms per 64 pairs in either layout)
  const int NBT = images / Cout;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = pl.P, Q = pl.Q, V = pl.V, PS = pl.PS, QS = pl.QS, RH = pl.RH, HP = (RH + 1) >> 1;
  f32x2* tQ = reinterpret_cast<f32x2*>(smem);
  f32x2* tP = tQ + Q;
  f32x2* A = tP + P;
  f32x2* Bf = A + ((pl.LH + 1) >> 1) * QS;          // the plan sizes A | B for the forward kernel's row pairs (>= HP)
  f32x2* Cc = A + pl.AB;
  f32x2* D = A;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NWV = FFT_THR / 64;
  const int Ws = os2d_ws(W), BASE = os2d_base(W);
  const float norm = 1.0f / (float)(P * Q);
  for (int i = tid; i < Q; i += FFT_THR) tQ[i] = twQ[i];
  for (int i = tid; i < P; i += FFT_THR) tP[i] = twP[i];
  bool bad = false;
  // register prefetch of the spectra (see fft_forward_kernel): element i of the image (v fastest) goes to Cc[v][u]
  const int nelem = P * V;
  f32x2 pf[FFT_EPT];
#define FFT_PREFETCH_Y(IMG, TID)                                                                                  \
  {                                                                                                               \
    const int pp_ = (IMG) / Cout, oo_ = (IMG)-pp_ * Cout;                                                         \
    const f32x2* src_ = quad ? Y + ((size_t)pp_ * Cout + oo_) * 4 : Y + (size_t)(IMG)*NBINS;                      \
    const size_t qs_ = (size_t)NBT * Cout * 4;                                                                    \
    _Pragma("unroll") for (int k = 0; k < FFT_EPT; ++k)                                                           \
      if (k * FFT_THR < nelem) {                                                                                  \
        const int e_ = min((TID) + k * FFT_THR, nelem - 1);                                                       \
        pf[k] = quad ? src_[(size_t)(e_ >> 2) * qs_ + (e_ & 3)] : src_[e_];                                       \
      }                                                                                                           \
  }
  constexpr int GRP = CPT > 0 ? GRPT : 1;                 // consecutive channels per work-group iteration (4 or 2)
  constexpr int NACC = CPT > 0 ? CPT : 1, NR = GRP == 4 ? 2 : 1;
  // GRP == 2 keeps the first channel of a pair as ONE register per cell (hi | lo << 16) and recombines at the second
  unsigned hreg[NACC][NR] = {}, lreg[GRP == 4 ? NACC : 1][NR] = {};
  const int ngroups = images / GRP, cells = TILED ? pl.TH * pl.TW : H * W;
  // XCD-aware order (work-group L runs on XCD L % 8, one L2 per XCD): every XCD takes a contiguous range of channel
  // groups, so the 8 / GRP work-groups that fill the 16-byte units of one (class, 8-channel group) with their 4- / 8-byte
  // pieces run on ONE XCD at about the same time and the pieces merge in its L2 (with the round-robin order each piece
  // left a different L2 as a masked 32-byte write: 632 MB written for 183 MB of activations)
  const int per_xcd = gridDim.x >> 3;                      // the launcher rounds the grid to a multiple of 8
  const int first = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (first < ngroups) FFT_PREFETCH_Y(first * GRP, tid)
  for (int it = first * GRP; it < images; it = ((it + 1) % GRP) ? it + 1 : (it / GRP + (int)gridDim.x) * GRP) {
    const int img = it;
    const int nxt = ((it + 1) % GRP) ? it + 1 : (it / GRP + (int)gridDim.x) * GRP;
    const int pr = img / Cout, o = img - pr * Cout;               // pair' = nb * T + tile
    const int nb = TILED ? div_magic(pr, pl.inv_t) : pr, tile = TILED ? pr - nb * pl.T : 0;
    const int ty = TILED ? div_magic(tile, pl.inv_tx) : 0, tx = TILED ? tile - ty * pl.TX : 0;
    const int y0 = TILED ? ty * pl.TH : 0, x0 = TILED ? tx * pl.TW : 0, oy = TILED ? pl.oy : 0, ox = TILED ? pl.ox : 0;
    const int TW_ = TILED ? pl.TW : W, TH_ = TILED ? pl.TH : H;
    // the per-thread addresses below are cheap to recompute per image; an opaque copy of the thread index keeps the
    // compiler from hoisting ~50 registers of them out of the image loop (and spilling them at the 128-register budget)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    lds_barrier();
#pragma unroll
    for (int k = 0; k < FFT_EPT; ++k) {
      const int i = tl + k * FFT_THR;
      if (i < nelem) {
        const int u = (int)__umulhi((unsigned)i, inv_v);     // i / V (inv_v = ceil(2^32 / V), V >= 2)
        Cc[(i - u * V) * PS + u] = pf[k];
      }
    }
    lds_barrier();
    if (nxt < images) FFT_PREFETCH_Y(nxt, tl)
    f32x2* Rc = fft_any<true>(pl.col_r1, pl.col_r2, Cc, D, P, V, PS, pl.zs_col, pl.inv_v, pl.np_col, pl.rad_col, tP, tid);
    // ---- rows 2p, 2p+1 (only h < H are needed) as one complex spectrum Z[v] = X_2p[v] + i X_2p+1[v], v < Q, with the
    // Hermitian halves of the two real rows: X[Q - v] = conj X[v].  Rc may be D = A | B: stage through registers per element
    // into the row buffer that does not overlap what is still to be read - rows go to Bf (second half of A | B) only after
    // the column results were copied out, so copy the needed H x V block to Cc's region first when Rc aliases A | B.
    f32x2* S = Rc;
    if (Rc == D) {   // move the H x V block that is still needed out of A | B (Cc is free now)
      for (int v = wv; v < V; v += NWV)
        for (int u = lane; u < RH; u += 64) Cc[v * PS + u] = Rc[v * PS + u];
      lds_barrier();
      S = Cc;
    }
    for (int p = wv; p < HP; p += NWV)
      for (int v = lane; v < Q; v += 64) {
        const int vv = v < V ? v : Q - v;                 // Hermitian mirror
        f32x2 x0 = S[vv * PS + 2 * p];
        f32x2 x1 = (2 * p + 1 < RH) ? S[vv * PS + 2 * p + 1] : f32x2{0.f, 0.f};
        if (v >= V) {
          x0 = cconj(x0);
          x1 = cconj(x1);
        }
        A[p * QS + v] = f32x2{x0[0] - x1[1], x0[1] + x1[0]};      // x0 + i x1
      }
    lds_barrier();
    f32x2* R = fft_any<true>(pl.row_r1, pl.row_r2, A, Bf, Q, HP, QS, pl.zs_row, pl.inv_hpr, pl.np_row, pl.rad_row, tQ, tid);
    // ---- epilogue: y = re / im of R (rows 2p / 2p+1), + bias, ReLU, channel scale, fp16 hi | lo into the SHB unit of
    // (nb, o / 8) at slot o % 8 (2-byte stores: the 8 channels of a unit come from 8 different images)
    const float bias = bp[o], osc = bp[2 * MTP + o];
    const int grp = o >> 3, slot = o & 7;
    char* hi_unit = out + (((size_t)nb * ((Cout + 7) >> 3) + grp) * 2 + 0) * (size_t)PLANE * 16;
    char* lo_unit = out + (((size_t)nb * ((Cout + 7) >> 3) + grp) * 2 + 1) * (size_t)PLANE * 16;
    if constexpr (CPT > 0) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        __builtin_amdgcn_sched_barrier(0);                 // keep the unrolled iterations apart: register pressure
        const int i = tl + k * FFT_THR;
        const int th = div_magic(i, pl.inv_tw), tw = i - th * TW_;        // cell of the tile: i / TW, i % TW
        const int h = y0 + th, w = x0 + tw;                               // ... of the map
        if (i < cells && (!TILED || (h < H && w < W))) {
          const f32x2 z = R[((th + oy) >> 1) * QS + tw + ox];
          float t = (((th + oy) & 1) ? z[1] : z[0]) * norm + bias;
          if (t != t) bad = true;          // fmaxf(NaN, 0) = 0 would hide a NaN spectrum from the range test below
          t = fmaxf(t, 0.f) * osc;
          if (!(fabsf(t) <= 65504.f)) bad = true;
          const _Float16 hv = (_Float16)t;
          const _Float16 lv = (_Float16)(t - (float)hv);
          const unsigned hb = __builtin_bit_cast(unsigned short, hv), lb = __builtin_bit_cast(unsigned short, lv);
          if constexpr (GRP == 4) {
            hreg[k][0] = (hreg[k][0] >> 16) | (hreg[k][1] << 16);           // 64-bit shift register: channel slot & 3 ends up
            hreg[k][1] = (hreg[k][1] >> 16) | (hb << 16);                   // at halfword slot & 3 after the 4th push
            lreg[k][0] = (lreg[k][0] >> 16) | (lreg[k][1] << 16);
            lreg[k][1] = (lreg[k][1] >> 16) | (lb << 16);
          } else {
            if ((slot & 1) == 0) hreg[k][0] = hb | (lb << 16);
          }
          if ((slot & (GRP - 1)) == GRP - 1) {
            const size_t off = ((size_t)BASE + (size_t)h * Ws + w) * 16 + (slot & (8 - GRP)) * 2;
            {
              if constexpr (GRP == 4) {
                *reinterpret_cast<uint2*>(hi_unit + off) = uint2{hreg[k][0], hreg[k][NR - 1]};
                *reinterpret_cast<uint2*>(lo_unit + off) = uint2{lreg[k][0], lreg[k][NR - 1]};
              } else {
                *reinterpret_cast<unsigned*>(hi_unit + off) = (hreg[k][0] & 0xffffu) | (hb << 16);
                *reinterpret_cast<unsigned*>(lo_unit + off) = (hreg[k][0] >> 16) | (lb << 16);
              }
            }
          }
        }
      }
    } else if (out32) {
      // all-fp32 mode: the layer's output as fp32 zero-bordered planes (what the fp32 5x5 kernel reads), bias + ReLU only
      float* plane = reinterpret_cast<float*>(out) + ((size_t)nb * Cout + o) * (size_t)PLANE + BASE;
      for (int th = wv; th < TH_ && y0 + th < H; th += NWV)
        for (int tw = lane; tw < TW_ && x0 + tw < W; tw += 64) {
          const f32x2 z = R[((th + oy) >> 1) * QS + tw + ox];
          const float t = (((th + oy) & 1) ? z[1] : z[0]) * norm + bias;
          plane[(size_t)(y0 + th) * Ws + x0 + tw] = os2d_relu(t);
        }
    } else {
      _Float16* hi = reinterpret_cast<_Float16*>(hi_unit) + slot;
      _Float16* lo = reinterpret_cast<_Float16*>(lo_unit) + slot;
      for (int th = wv; th < TH_ && y0 + th < H; th += NWV)
        for (int tw = lane; tw < TW_ && x0 + tw < W; tw += 64) {
          const int h = y0 + th, w = x0 + tw;
          const f32x2 z = R[((th + oy) >> 1) * QS + tw + ox];
          float t = (((th + oy) & 1) ? z[1] : z[0]) * norm + bias;
          if (t != t) bad = true;          // fmaxf(NaN, 0) = 0 would hide a NaN spectrum from the range test below
          t = fmaxf(t, 0.f) * osc;
          if (!(fabsf(t) <= 65504.f)) bad = true;
          const _Float16 hv = (_Float16)t;
          const size_t cell = (size_t)BASE + (size_t)h * Ws + w;
          {
            hi[cell * 8] = hv;
            lo[cell * 8] = (_Float16)(t - (float)hv);
          }
        }
    }
  }
  if (status.word != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0ull) {
    if ((tid & 63) == 0) os2d_raise(status);
  }
}

// radices (4 first, then 2, then 3s) of a size that is a product of 2s and 3s; returns the number of passes or 0
int factor(int N, int* rad) {
  int n = 0, m = N;
  while (m % 4 == 0 && n < MAXPASS) {
    rad[n++] = 4;
    m /= 4;
  }
  while (m % 2 == 0 && n < MAXPASS) {
    rad[n++] = 2;
    m /= 2;
  }
  while (m % 3 == 0 && n < MAXPASS) {
    rad[n++] = 3;
    m /= 3;
  }
  return m == 1 ? n : 0;
}

// N = r1 * r2 in the two-stage register form, for the transform sizes of the benchmark configurations (fft_any
// instantiates exactly these); sizes with a factor 7 exist in this form only
bool split_size(int N, int* r1, int* r2) {
  static const int table[][3] = {{36, 6, 6}, {42, 6, 7}, {48, 8, 6}, {54, 9, 6}, {64, 8, 8},  {72, 9, 8},
                                 {84, 12, 7}, {96, 12, 8}, {108, 12, 9}, {128, 16, 8}};
  *r1 = *r2 = 0;
  for (const auto& e : table)
    if (e[0] == N) {
      *r1 = e[1];
      *r2 = e[2];
      return true;
    }
  return false;
}

int next_size(int n) {  // smallest even transform size >= n: 2^a 3^b (Stockham passes), or one with a two-stage form (42, 84)
  for (int s = n;; ++s) {
    int m = s, r1, r2;
    if (m & 1) continue;
    if (split_size(s, &r1, &r2)) return s;
    while (m % 2 == 0) m /= 2;
    while (m % 3 == 0) m /= 3;
    if (m == 1) return s;
  }
}

// plan of ONE transform whose forward kernel loads an LH x LW window and whose inverse needs RH rows; P >= minP, Q >= minQ
bool plan_transform(int LH, int LW, int RH, int minP, int minQ, FftPlan* pl, size_t* lds) {
  pl->P = next_size(minP);
  pl->Q = next_size(minQ);
  pl->V = pl->Q / 2 + 1;
  pl->PS = pl->P + 1;
  pl->QS = pl->Q + 1;
  pl->np_row = factor(pl->Q, pl->rad_row);
  pl->np_col = factor(pl->P, pl->rad_col);
  const int HP = (LH + 1) / 2, HPR = (RH + 1) / 2;
  const bool row_fast = split_size(pl->Q, &pl->row_r1, &pl->row_r2), col_fast = split_size(pl->P, &pl->col_r1, &pl->col_r2);
  if ((!pl->np_row && !row_fast) || (!pl->np_col && !col_fast)) return false;
  pl->zs_row = pl->row_r1 ? ((pl->row_r1 * (pl->row_r2 | 1)) | 1) : 0;
  pl->zs_col = pl->col_r1 ? ((pl->col_r1 * (pl->col_r2 | 1)) | 1) : 0;
  pl->inv_hp = magic_div((unsigned)HP);
  pl->inv_hpr = magic_div((unsigned)HPR);
  pl->inv_v = (unsigned)(((1ull << 32) + pl->V - 1) / pl->V);
  pl->inv_q = (unsigned)(((1ull << 32) + pl->Q - 1) / pl->Q);
  size_t rows = (size_t)2 * HP * pl->QS, cc = (size_t)pl->V * pl->PS;
  if ((size_t)HP * (pl->QS + pl->zs_row) > rows) rows = (size_t)HP * (pl->QS + pl->zs_row);   // A | exchange buffer
  const size_t dd = (size_t)pl->V * pl->zs_col > cc ? (size_t)pl->V * pl->zs_col : cc;         // exchange buffer in D = A | B
  const size_t ab = rows > dd ? rows : dd;
  pl->AB = (int)ab;
  pl->LH = LH;
  pl->LW = LW;
  pl->RH = RH;
  *lds = (size_t)(pl->Q + pl->P + ab + cc) * 8;
  if ((size_t)HP * pl->Q > (size_t)FFT_EPT_MAX * FFT_THR || (size_t)pl->P * pl->V > (size_t)FFT_EPT_MAX * FFT_THR) return false;
  return *lds <= 160 * 1024;
}

void set_tiles(FftPlan* pl, int TY, int TX, int TH, int TW) {
  pl->TY = TY;
  pl->TX = TX;
  pl->T = TY * TX;
  pl->TH = TH;
  pl->TW = TW;
  pl->oy = TY > 1 ? 3 : 0;
  pl->ox = TX > 1 ? 3 : 0;
  pl->inv_t = magic_div((unsigned)pl->T);
  pl->inv_tx = magic_div((unsigned)TX);
  pl->inv_tw = magic_div((unsigned)TW);
}

// The whole map in one transform when it fits the LDS and the register prefetch (P >= H + 3, Q >= W + 3: the zero padding is
// the halo); otherwise the cheapest tiling.  An axis is either untiled (window = the whole axis, size >= n + 3) or cut into
// >= 2 tiles of ceil(n / k) outputs whose window is 6 longer (size >= tile + 6).  Cost = bins in total, x 1.5 per axis whose
// size has no two-stage register form (Stockham passes: ~2x the LDS traffic), x 1.15 when only one work-group fits a CU (no
// second group to hide the barriers behind): 96 x 128 -> 2 x 2 tiles at 54 x 72 (7992 bins, 48 KB, the transform the 48 x 64
// level uses - one set of weight spectra for both) rather than 2 x 1 at 54 x 144 (7888 bins, 96 KB, Stockham rows).
bool make_plan_search(int H, int W, FftPlan* pl, size_t* lds);

// Plans are pure functions of (H, W) and a head call asks for the same one several times (workspace size, route, forward,
// inverse): memoised per process (the search below runs plan_transform for up to 48 x 48 tilings).
bool make_plan(int H, int W, FftPlan* pl, size_t* lds) {
  struct Entry {
    bool ok;
    FftPlan pl;
    size_t lds;
  };
  static std::mutex mu;
  static std::map<std::pair<int, int>, Entry> cache;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({H, W});
    if (it != cache.end()) {
      if (it->second.ok) {
        *pl = it->second.pl;
        *lds = it->second.lds;
      }
      return it->second.ok;
    }
  }
  Entry e = {};
  e.ok = make_plan_search(H, W, &e.pl, &e.lds);
  if (e.ok) {
    *pl = e.pl;
    *lds = e.lds;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() >= 4096) cache.clear();      // a dataset of arbitrary map sizes must not grow the table without bound
  cache[{H, W}] = e;
  return e.ok;
}

bool make_plan_search(int H, int W, FftPlan* pl, size_t* lds) {
  if (plan_transform(H, W, H, H + 3, W + 3, pl, lds)) {
    set_tiles(pl, 1, 1, H, W);
    return true;
  }
  bool found = false;
  double best_cost = 0.0;
  size_t best_lds = 0;
  FftPlan best;
  for (int TY = 1; TY <= 48; ++TY)          // (up to 48 tiles per axis: any width the head accepts, OS2D_MAX_W)
    for (int TX = 1; TX <= 48; ++TX) {
      if (TY * TX == 1) continue;
      const int TH = (H + TY - 1) / TY, TW = (W + TX - 1) / TX;
      if ((TY > 1 && (TY - 1) * TH >= H) || (TX > 1 && (TX - 1) * TW >= W)) continue;     // an empty last tile
      const int LH = TY > 1 ? TH + 6 : H, LW = TX > 1 ? TW + 6 : W;
      FftPlan c;
      size_t l;
      if (!plan_transform(LH, LW, TY > 1 ? TH + 3 : H, TY > 1 ? TH + 6 : H + 3, TX > 1 ? TW + 6 : W + 3, &c, &l)) continue;
      set_tiles(&c, TY, TX, TH, TW);
      double cost = (double)c.T * os2d_round_up(c.P * c.V, 8);
      if (!c.row_r1) cost *= 1.5;
      if (!c.col_r1) cost *= 1.5;
      if (2 * l > 160 * 1024) cost *= 1.15;
      if (!found || cost < best_cost || (cost == best_cost && l < best_lds)) {
        found = true;
        best = c;
        best_cost = cost;
        best_lds = l;
      }
    }
  if (!found) return false;
  *pl = best;
  *lds = best_lds;
  return true;
}

int check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("%s launch: %s", what, hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

// P, Q and the padded number of bins (multiple of 8) of the transform of an H x W map (of ONE tile of it when the map is
// tiled); tiles[6] (optional) = TY, TX, TH, TW, window rows, window columns.  0 if no plan exists.
int os2d_fft_plan(int H, int W, int* P, int* Q, int* nbins, int* tiles) {
  FftPlan pl;
  size_t lds;
  if (H < 1 || W < 1 || !make_plan(H, W, &pl, &lds)) return 0;
  if (P) *P = pl.P;
  if (Q) *Q = pl.Q;
  if (nbins) *nbins = os2d_round_up(pl.P * pl.V, 8);
  if (tiles) {
    tiles[0] = pl.TY;
    tiles[1] = pl.TX;
    tiles[2] = pl.TH;
    tiles[3] = pl.TW;
    tiles[4] = pl.LH;
    tiles[5] = pl.LW;
  }
  return 1;
}

int os2d_launch_fft_forward(const float* corr, const float* inv, float* X, const float* twQ, const float* twP, int NB, int C,
                            int H, int W, hipStream_t stream) {
  FftPlan pl;
  size_t lds;
  if (!make_plan(H, W, &pl, &lds)) {
    os2d_set_error("fft_forward: a %dx%d map does not fit the in-LDS transform", H, W);
    return -3;
  }
  const int ept = (((pl.LH + 1) / 2) * pl.Q + FFT_THR - 1) / FFT_THR;
  auto kern = pl.T > 1 ? (ept <= 6 ? fft_forward_kernel<6, true> : ept <= 10 ? fft_forward_kernel<10, true> : fft_forward_kernel<14, true>)
                       : (ept <= 6 ? fft_forward_kernel<6, false> : ept <= 10 ? fft_forward_kernel<10, false> : fft_forward_kernel<14, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(fft_forward): %s", hipGetErrorString(e));
    return -4;
  }
  const int images = NB * C * pl.T;
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  const int grid = images < 256 * per_cu * 4 ? images : 256 * per_cu * 4;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FFT_THR), lds, stream, corr, inv, reinterpret_cast<f32x2*>(X),
                     reinterpret_cast<const f32x2*>(twQ), reinterpret_cast<const f32x2*>(twP), pl, C, H, W,
                     os2d_round_up(pl.P * pl.V, 8), images);
  return check("fft_forward");
}

int os2d_launch_fft_inverse(const float* Y, const float* bp, int MTP, void* out, const float* twQ, const float* twP, int NB,
                            int Cout, int H, int W, Os2dRangeFlag status, int layout, int out_fp32, hipStream_t stream) {
  FftPlan pl;
  size_t lds;
  if (!make_plan(H, W, &pl, &lds)) {
    os2d_set_error("fft_inverse: a %dx%d map does not fit the in-LDS transform", H, W);
    return -3;
  }
  const int ept = (pl.P * pl.V + FFT_THR - 1) / FFT_THR;
#ifndef OS2D_FFT_GRP
#define OS2D_FFT_GRP 2
#endif
  constexpr int CPT = 10, GRP = OS2D_FFT_GRP;
  const bool grouped = !out_fp32 && Cout % GRP == 0 && pl.TH * pl.TW <= CPT * FFT_THR && ept <= 10;
#define OS2D_INV_KERN(T_)                                                                        \
  (grouped ? (ept <= 6   ? fft_inverse_kernel<6, CPT, GRP, T_>                                   \
              : ept <= 8 ? fft_inverse_kernel<8, CPT, GRP, T_>                                   \
                         : fft_inverse_kernel<10, CPT, GRP, T_>)                                 \
           : (ept <= 6    ? fft_inverse_kernel<6, 0, 1, T_>                                      \
              : ept <= 10 ? fft_inverse_kernel<10, 0, 1, T_>                                     \
                          : fft_inverse_kernel<14, 0, 1, T_>))
  auto kern = pl.T > 1 ? OS2D_INV_KERN(true) : OS2D_INV_KERN(false);
#undef OS2D_INV_KERN
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(fft_inverse): %s", hipGetErrorString(e));
    return -4;
  }
  const int images = NB * pl.T * Cout, groups = grouped ? images / GRP : images;
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  const int grid = ((groups < 256 * per_cu * 4 ? groups : 256 * per_cu * 4) + 7) / 8 * 8;   // multiple of 8 (XCD-aware order)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FFT_THR), lds, stream, reinterpret_cast<const f32x2*>(Y), bp, MTP,
                     static_cast<char*>(out), reinterpret_cast<const f32x2*>(twQ), reinterpret_cast<const f32x2*>(twP), pl,
                     Cout, H, W, os2d_round_up(pl.P * pl.V, 8), os2d_plane(H, W), images,
                     (unsigned)(((1ull << 32) + pl.V - 1) / pl.V), status, layout ? 1 : 0, out_fp32 ? 1 : 0);
  return check("fft_inverse");
}

// Two-dimensional real FFT pair for the frequency-domain form of the 7x7 TransformNet layer (gfx950) - building blocks,
// see spectral.hip.  Everything of one image lives in LDS; one work-group per image at a time.
//
//   fft_forward   x[h][w] = relu(corr[nb][c][h][w]) * inv_norm[nb][h][w]  (the TransformNet input normalisation of
//                 reference head.py:650 folded into the load), zero-padded to P x Q  ->  X[nb][c][u * V + v], V = Q/2 + 1
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
#include "os2d_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int FFT_THR = 512;
constexpr int MAXPASS = 8;

struct FftPlan {
  int P, Q, V;               // padded sizes, V = Q/2 + 1
  int np_row, np_col;        // number of passes
  int rad_row[MAXPASS], rad_col[MAXPASS];
  int PS;                    // padded stride of a column in LDS (P + 1: keeps the transposed accesses off one bank)
  int AB;                    // complex numbers of the A | B region = max(2 * ceil(H/2) * Q, V * PS): the second column
                             // buffer D aliases it
};

__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) { return f32x2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }
__device__ __forceinline__ f32x2 cconj(f32x2 a) { return f32x2{a[0], -a[1]}; }

// One Stockham pass of radix R over `nfft` transforms of length N (Ns = product of the radices already done).
// INV = false: forward (exp(-2 pi i ...)), true: inverse (conjugate twiddles and butterflies, no 1/N).
template <int R, bool INV>
__device__ __forceinline__ void stockham_pass(const f32x2* __restrict__ in, f32x2* __restrict__ out, int N, int Ns, int nfft,
                                              int stride, const f32x2* __restrict__ tw, int tid) {
  const int nb = N / R;          // butterflies per transform
  const int step = N / (Ns * R);
  for (int i = tid; i < nfft * nb; i += FFT_THR) {
    const int f = i / nb, j = i - f * nb;
    const int k = j % Ns;
    f32x2 v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      v[q] = in[f * stride + j + q * nb];
      if (q > 0) {
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
    __syncthreads();
    f32x2* t = a;
    a = b;
    b = t;
  }
  return a;
}

// LDS (in complex numbers): twiddles Q + P | rows A, B: 2 x (HP x Q), HP = ceil(H/2) row pairs | columns C: V x PS.
// The second column buffer D aliases A|B (the row stage is finished by then).
__global__ __launch_bounds__(FFT_THR) void fft_forward_kernel(const float* __restrict__ corr,   // [NB][C][H*W]
                                                             const float* __restrict__ inv,    // [NB][H*W]
                                                             f32x2* __restrict__ X,            // [NB][C][NBINS]
                                                             const f32x2* __restrict__ twQ, const f32x2* __restrict__ twP,
                                                             FftPlan pl, int C, int H, int W, int NBINS, int images) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = pl.P, Q = pl.Q, V = pl.V, PS = pl.PS, HP = (H + 1) >> 1, HW = H * W;
  f32x2* tQ = reinterpret_cast<f32x2*>(smem);
  f32x2* tP = tQ + Q;
  f32x2* A = tP + P;
  f32x2* Bf = A + HP * Q;
  f32x2* Cc = A + pl.AB;
  f32x2* D = A;                     // aliases A | B (the row stage is over when the columns start)
  const int tid = threadIdx.x;
  for (int i = tid; i < Q; i += FFT_THR) tQ[i] = twQ[i];
  for (int i = tid; i < P; i += FFT_THR) tP[i] = twP[i];
  for (int img = blockIdx.x; img < images; img += gridDim.x) {
    const int nb = img / C;
    const float* src = corr + (size_t)img * HW;
    const float* nv = inv + (size_t)nb * HW;
    __syncthreads();
    // ---- load: row pair p, column w -> A[p][w] = (x[2p][w], x[2p+1][w]); zero beyond the map
    for (int i = tid; i < HP * Q; i += FFT_THR) {
      const int p = i / Q, w = i - p * Q;
      float re = 0.f, im = 0.f;
      if (w < W) {
        const int n0 = (2 * p) * W + w;
        re = fmaxf(src[n0], 0.f) * nv[n0];
        if (2 * p + 1 < H) im = fmaxf(src[n0 + W], 0.f) * nv[n0 + W];
      }
      A[i] = f32x2{re, im};
    }
    __syncthreads();
    f32x2* R = fft_batch<false>(A, Bf, Q, HP, Q, pl.np_row, pl.rad_row, tQ, tid);
    // ---- untangle the two real rows of every pair and transpose into the column buffer C[v][u]; rows >= H are zero
    f32x2* Cb = (R == A) ? Cc : Cc;   // C is separate from A | B
    for (int i = tid; i < V * P; i += FFT_THR) {
      const int v = i / P, u = i - v * P;
      f32x2 o = f32x2{0.f, 0.f};
      if (u < H) {
        const int p = u >> 1;
        const f32x2 z = R[p * Q + v], zc = cconj(R[p * Q + (v == 0 ? 0 : Q - v)]);
        if ((u & 1) == 0) o = 0.5f * (z + zc);
        else {
          const f32x2 d = z - zc;            // (Z - conj Z') / (2 i) = -i d / 2
          o = f32x2{0.5f * d[1], -0.5f * d[0]};
        }
      }
      Cb[v * PS + u] = o;
    }
    __syncthreads();
    f32x2* Rc = fft_batch<false>(Cb, D, P, V, PS, pl.np_col, pl.rad_col, tP, tid);
    // ---- store X[u * V + v] (v fastest) + zero padding bins
    f32x2* dst = X + (size_t)img * NBINS;
    for (int i = tid; i < NBINS; i += FFT_THR) {
      f32x2 o = f32x2{0.f, 0.f};
      if (i < P * V) {
        const int u = i / V, v = i - u * V;
        o = Rc[v * PS + u];
      }
      dst[i] = o;
    }
  }
}

// Inverse: Y -> columns (inverse FFT of length P over u for every v) -> re-tangle row pairs -> inverse complex FFT of
// length Q -> real rows; epilogue of the layer.
__global__ __launch_bounds__(FFT_THR) void fft_inverse_kernel(const f32x2* __restrict__ Y,      // [NB][Cout][NBINS]
                                                             const float* __restrict__ bp,     // [3][MTP]: bias | - | 2^out_exp
                                                             int MTP, char* __restrict__ out,  // SHB [NB][Cout/8][2][PLANE] x 16 B
                                                             const f32x2* __restrict__ twQ, const f32x2* __restrict__ twP,
                                                             FftPlan pl, int Cout, int H, int W, int NBINS, int PLANE,
                                                             int images, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = pl.P, Q = pl.Q, V = pl.V, PS = pl.PS, HP = (H + 1) >> 1;
  f32x2* tQ = reinterpret_cast<f32x2*>(smem);
  f32x2* tP = tQ + Q;
  f32x2* A = tP + P;
  f32x2* Bf = A + HP * Q;
  f32x2* Cc = A + pl.AB;
  f32x2* D = A;
  const int tid = threadIdx.x;
  const int Ws = os2d_ws(W), BASE = os2d_base(W);
  const float norm = 1.0f / (float)(P * Q);
  for (int i = tid; i < Q; i += FFT_THR) tQ[i] = twQ[i];
  for (int i = tid; i < P; i += FFT_THR) tP[i] = twP[i];
  bool bad = false;
  for (int img = blockIdx.x; img < images; img += gridDim.x) {
    const int nb = img / Cout, o = img - nb * Cout;
    const f32x2* src = Y + (size_t)img * NBINS;
    __syncthreads();
    for (int i = tid; i < V * P; i += FFT_THR) {
      const int u = i / V, v = i - u * V;   // global order: v fastest
      Cc[v * PS + u] = src[i];
    }
    __syncthreads();
    f32x2* Rc = fft_batch<true>(Cc, D, P, V, PS, pl.np_col, pl.rad_col, tP, tid);
    // ---- rows 2p, 2p+1 (only h < H are needed) as one complex spectrum Z[v] = X_2p[v] + i X_2p+1[v], v < Q, with the
    // Hermitian halves of the two real rows: X[Q - v] = conj X[v].  Rc may be D = A | B: stage through registers per element
    // into the row buffer that does not overlap what is still to be read - rows go to Bf (second half of A | B) only after
    // the column results were copied out, so copy the needed H x V block to Cc's region first when Rc aliases A | B.
    f32x2* S = Rc;
    if (Rc == D) {   // move the H x V block that is still needed out of A | B (Cc is free now)
      for (int i = tid; i < V * H; i += FFT_THR) {
        const int v = i / H, u = i - v * H;
        Cc[v * PS + u] = Rc[v * PS + u];
      }
      __syncthreads();
      S = Cc;
    }
    for (int i = tid; i < HP * Q; i += FFT_THR) {
      const int p = i / Q, v = i - p * Q;
      const int vv = v < V ? v : Q - v;                 // Hermitian mirror
      f32x2 x0 = S[vv * PS + 2 * p];
      f32x2 x1 = (2 * p + 1 < H) ? S[vv * PS + 2 * p + 1] : f32x2{0.f, 0.f};
      if (v >= V) {
        x0 = cconj(x0);
        x1 = cconj(x1);
      }
      A[i] = f32x2{x0[0] - x1[1], x0[1] + x1[0]};       // x0 + i x1
    }
    __syncthreads();
    f32x2* R = fft_batch<true>(A, Bf, Q, HP, Q, pl.np_row, pl.rad_row, tQ, tid);
    // ---- epilogue: y = re / im of R (rows 2p / 2p+1), + bias, ReLU, channel scale, fp16 hi | lo into the SHB unit of
    // (nb, o / 8) at slot o % 8 (2-byte stores: the 8 channels of a unit come from 8 different images)
    const float bias = bp[o], osc = bp[2 * MTP + o];
    const int grp = o >> 3, slot = o & 7;
    _Float16* hi = reinterpret_cast<_Float16*>(out + (((size_t)nb * ((Cout + 7) >> 3) + grp) * 2 + 0) * (size_t)PLANE * 16) + slot;
    _Float16* lo = reinterpret_cast<_Float16*>(out + (((size_t)nb * ((Cout + 7) >> 3) + grp) * 2 + 1) * (size_t)PLANE * 16) + slot;
    for (int i = tid; i < H * W; i += FFT_THR) {
      const int h = i / W, w = i - h * W;
      const f32x2 z = R[(h >> 1) * Q + w];
      float t = ((h & 1) ? z[1] : z[0]) * norm + bias;
      t = fmaxf(t, 0.f) * osc;
      if (!(fabsf(t) <= 65504.f)) bad = true;
      const _Float16 hv = (_Float16)t;
      const size_t cell = (size_t)BASE + (size_t)h * Ws + w;
      hi[cell * 8] = hv;
      lo[cell * 8] = (_Float16)(t - (float)hv);
    }
  }
  if (status != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0ull) {
    if ((tid & 63) == 0) __hip_atomic_store(status, OS2D_STATUS_F16_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

int next_size(int n) {  // smallest 2^a 3^b >= n (a >= 1: the row length must be even)
  for (int s = n;; ++s) {
    int m = s;
    if (m & 1) continue;
    while (m % 2 == 0) m /= 2;
    while (m % 3 == 0) m /= 3;
    if (m == 1) return s;
  }
}

bool make_plan(int H, int W, FftPlan* pl, size_t* lds) {
  pl->P = next_size(H + 3);
  pl->Q = next_size(W + 3);
  pl->V = pl->Q / 2 + 1;
  pl->PS = pl->P + 1;
  pl->np_row = factor(pl->Q, pl->rad_row);
  pl->np_col = factor(pl->P, pl->rad_col);
  if (!pl->np_row || !pl->np_col) return false;
  const int HP = (H + 1) / 2;
  const size_t rows = (size_t)2 * HP * pl->Q, cc = (size_t)pl->V * pl->PS;
  const size_t ab = rows > cc ? rows : cc;
  pl->AB = (int)ab;
  *lds = (size_t)(pl->Q + pl->P + ab + cc) * 8;
  return *lds <= 160 * 1024;
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

// P, Q and the padded number of bins (multiple of 8) of the transform of an H x W map; 0 if the map does not fit the LDS plan
int os2d_fft_plan(int H, int W, int* P, int* Q, int* nbins) {
  FftPlan pl;
  size_t lds;
  if (H < 1 || W < 1 || !make_plan(H, W, &pl, &lds)) return 0;
  if (P) *P = pl.P;
  if (Q) *Q = pl.Q;
  if (nbins) *nbins = os2d_round_up(pl.P * pl.V, 8);
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
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(fft_forward): %s", hipGetErrorString(e));
    return -4;
  }
  const int images = NB * C;
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  const int grid = images < 256 * per_cu * 4 ? images : 256 * per_cu * 4;
  hipLaunchKernelGGL(fft_forward_kernel, dim3(grid), dim3(FFT_THR), lds, stream, corr, inv, reinterpret_cast<f32x2*>(X),
                     reinterpret_cast<const f32x2*>(twQ), reinterpret_cast<const f32x2*>(twP), pl, C, H, W,
                     os2d_round_up(pl.P * pl.V, 8), images);
  return check("fft_forward");
}

int os2d_launch_fft_inverse(const float* Y, const float* bp, int MTP, void* out, const float* twQ, const float* twP, int NB,
                            int Cout, int H, int W, int* status, hipStream_t stream) {
  FftPlan pl;
  size_t lds;
  if (!make_plan(H, W, &pl, &lds)) {
    os2d_set_error("fft_inverse: a %dx%d map does not fit the in-LDS transform", H, W);
    return -3;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_inverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(fft_inverse): %s", hipGetErrorString(e));
    return -4;
  }
  const int images = NB * Cout;
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  const int grid = images < 256 * per_cu * 4 ? images : 256 * per_cu * 4;
  hipLaunchKernelGGL(fft_inverse_kernel, dim3(grid), dim3(FFT_THR), lds, stream, reinterpret_cast<const f32x2*>(Y), bp, MTP,
                     static_cast<char*>(out), reinterpret_cast<const f32x2*>(twQ), reinterpret_cast<const f32x2*>(twP), pl,
                     Cout, H, W, os2d_round_up(pl.P * pl.V, 8), os2d_plane(H, W), images, status);
  return check("fft_inverse");
}

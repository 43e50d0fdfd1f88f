// Device build of dft_mfma.h (the transforms of the frequency-domain 7x7 layer as matrix products on v_mfma_f32_32x32x16_f16,
// precision "fftx3"): hardware hooks, kernel entry points, the constant-matrix builder and the launchers.  The kernel bodies
// live in the header so that tests/host/dft_mfma_check.cpp can run the same source on the CPU (tests/host/spmd_emu.h).
#include "os2d_common.h"
#include <map>
#include <mutex>
#include <utility>

extern __shared__ __attribute__((aligned(16))) unsigned char os2d_dft_smem[];

#define DFT_DEV __device__ __forceinline__
#define DFT_HD __host__ __device__ static inline
#define DFT_MEMBER __device__ __forceinline__
#define DFT_TID ((int)threadIdx.x)
#define DFT_BID ((int)blockIdx.x)
#define DFT_GRID ((int)gridDim.x)
#define DFT_LDS os2d_dft_smem
// barrier for data exchanged through LDS only: leaves this wave's global loads (the next iteration's prefetch) and stores in
// flight (see fft.hip: __syncthreads() would wait for them at every barrier)
#define DFT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DFT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DFT_SHFL_XOR(v, m) __shfl_xor(v, m, 64)
#define DFT_SHFL_XOR_U32(v, m) ((unsigned)__shfl_xor((int)(v), m, 64))
#define DFT_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
#define DFT_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define DFT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DFT_LANDED(X) asm volatile("" : "+v"(X))
#define DFT_SPLIT_LO_PAIR(a, b, hi) os2d_split_lo_pair(a, b, hi)
#define DFT_STREAM_STORE(P, V) os2d_stream_store(P, V)
#define DFT_FLAG Os2dRangeFlag
#define DFT_FLAG_SET(f) ((f).word != nullptr)
#define DFT_RAISE(f) os2d_raise(f)
#ifdef OS2D_DIAG_DFT_STAMPS
// diagnostic build: thread 0 of every work-group accumulates the wall-clock ticks (100 MHz) between the phase barriers; the sums
// over all work-groups land in os2d_dft_stamps[0..5] (forward: W | step 1 | R | step 2 | XS | ST) and [8..13] (inverse: max |
// WY | step A | WT | step B | epilogue), [6] / [14] count iterations.  Read with os2d_debug_dft_stamps (tools/time_dft_phases.py).
__device__ unsigned long long os2d_dft_stamps[16];
#define DFT_STAMP_BEGIN() unsigned long long st_acc_[6] = {0, 0, 0, 0, 0, 0}, st_n_ = 0, st_last_ = wall_clock64();
#define DFT_STAMP(K)                                   \
  {                                                    \
    const unsigned long long t_ = wall_clock64();      \
    st_acc_[K] += t_ - st_last_;                       \
    st_last_ = t_;                                     \
    if ((K) == 5) ++st_n_;                             \
  }
#define DFT_STAMP_END(BASE)                                                                   \
  if (threadIdx.x == 0) {                                                                     \
    for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&os2d_dft_stamps[(BASE) + k_], st_acc_[k_]);     \
    atomicAdd(&os2d_dft_stamps[(BASE) + 6], st_n_);                                           \
  }
#endif
#include "dft_mfma.h"

namespace {

using namespace os2d_dft;

// KS = k-steps of the product whose row operand lives in registers (step 2 / step A: 2 Pp / 16): a template parameter for the
// canonical transform sizes (5 .. 8: straight-line products, exactly as many fragment registers as the size needs), 0 = any
// size behind uniform guards ($OS2D_DFT_SIZES=exact)
template <bool TILED, bool FAST, int G, int NW, int KS>
__global__ __launch_bounds__(NW * 64, 8 / NW) void dft_forward_kernel(const float* __restrict__ corr, const float* __restrict__ invn,
                                                                      float* __restrict__ X, const u32x4v* __restrict__ FqT,
                                                                      const u32x4v* __restrict__ Fp2, DftPlan pl, int C, int Cpad, int NBT,
                                                                      int iters) {
  dft_forward_body<TILED, FAST, G, NW, KS>(corr, invn, X, FqT, Fp2, pl, C, Cpad, NBT, iters);
}

template <bool TILED, int KS, int G>
__global__ __launch_bounds__(DFT_THR, 1) void dft_inverse_kernel(const float* __restrict__ Y, const float* __restrict__ bp, int MTP,
                                                                 unsigned char* __restrict__ out, const u32x4v* __restrict__ E2,
                                                                 const u32x4v* __restrict__ Gq, DftPlan pl, int Cout, int NBT, int PLANE,
                                                                 int Ws, int BASE, int iters, Os2dRangeFlag status, int zero_borders) {
  dft_inverse_body<TILED, KS, G>(Y, bp, MTP, out, E2, Gq, pl, Cout, NBT, PLANE, Ws, BASE, iters, status, zero_borders);
}

typedef void (*dft_forward_fn)(const float*, const float*, float*, const u32x4v*, const u32x4v*, DftPlan, int, int, int, int);
typedef void (*dft_inverse_fn)(const float*, const float*, int, unsigned char*, const u32x4v*, const u32x4v*, DftPlan, int, int, int, int, int,
                               int, Os2dRangeFlag, int);
template <int G, int NW, int KS>
dft_forward_fn dft_forward_variant(const DftPlan& pl) {
  return pl.T > 1 ? dft_forward_kernel<true, false, G, NW, KS> : (pl.fast ? dft_forward_kernel<false, true, G, NW, KS> : dft_forward_kernel<false, false, G, NW, KS>);
}
template <int G, int NW>
dft_forward_fn dft_forward_pick(const DftPlan& pl) {
  if constexpr (G == 8) {
    return dft_forward_variant<G, NW, 0>(pl);      // (Fp2 in LDS: no register-resident k-steps to count)
  } else {
    switch (2 * pl.Pp / 16) {      // k-steps of step 2
      case 5: return dft_forward_variant<G, NW, 5>(pl);
      case 6: return dft_forward_variant<G, NW, 6>(pl);
      case 7: return dft_forward_variant<G, NW, 7>(pl);
      case 8: return dft_forward_variant<G, NW, 8>(pl);
      default: return dft_forward_variant<G, NW, 0>(pl);
    }
  }
}
template <int KS, int G = DFT_G>
dft_inverse_fn dft_inverse_variant(const DftPlan& pl) {
  return pl.T > 1 ? dft_inverse_kernel<true, KS, G> : dft_inverse_kernel<false, KS, G>;
}
dft_inverse_fn dft_inverse_pick(const DftPlan& pl) {
  if (pl.G == 8) return dft_inverse_variant<0, 8>(pl);
  switch (2 * pl.Pp / 16) {      // k-steps of step A
    case 5: return dft_inverse_variant<5>(pl);
    case 6: return dft_inverse_variant<6>(pl);
    case 7: return dft_inverse_variant<7>(pl);
    case 8: return dft_inverse_variant<8>(pl);
    default: return dft_inverse_variant<0>(pl);
  }
}

// FqT | Fp2 | E2 | Gq of a (P, Q) transform, one thread per 16-byte unit
__global__ __launch_bounds__(256) void dft_matrices_kernel(const double* __restrict__ twP, const double* __restrict__ twQ, int P, int Q,
                                                           u32x4v* __restrict__ out) {
  const int n0 = dft_units_fqt(P, Q), n1 = n0 + dft_units_fp2(P, Q), n2 = n1 + dft_units_e2(P, Q), n3 = n2 + dft_units_gq(P, Q);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n3) return;
  const int which = i < n0 ? 0 : i < n1 ? 1 : i < n2 ? 2 : 3;
  const int base = which == 0 ? 0 : which == 1 ? n0 : which == 2 ? n1 : n2;
  dft_matrix_unit(which, i - base, P, Q, twP, twQ, out + i);
}

int dft_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("%s launch: %s", what, hipGetErrorString(e));
    return -4;
  }
  return 0;
}

// The planner tries up to 48 x 48 tilings x 6 canonical sizes and a head call needs the plan of its map four times (workspace
// size, route, forward, inverse): memoised per process - a plan is a pure function of (H, W) under the process-wide size policy.
// ``g8``: the plan with 8 images per iteration where the map's transform takes it (dft_plan_g8; $OS2D_DFT_G8=0: never - measurements)
bool dft_plan_cached(int H, int W, DftPlan* out, bool g8 = true) {
  static const bool g8_enabled = [] {
    const char* e = getenv("OS2D_DFT_G8");
    return !(e && e[0] == '0');
  }();
  struct Entry {
    bool ok, ok8;
    DftPlan pl, pl8;
  };
  static std::mutex mu;
  static std::map<std::pair<int, int>, Entry> cache;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({H, W});
    if (it != cache.end()) {
      if (it->second.ok) *out = (g8 && g8_enabled && it->second.ok8) ? it->second.pl8 : it->second.pl;
      return it->second.ok;
    }
  }
  Entry e = {};
  e.ok = dft_make_plan(H, W, &e.pl);
  e.ok8 = e.ok && dft_plan_g8(e.pl, &e.pl8);
  if (e.ok) *out = (g8 && g8_enabled && e.ok8) ? e.pl8 : e.pl;
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() >= 4096) cache.clear();
  cache[{H, W}] = e;
  return e.ok;
}

int dft_grid(int iters, int per_cu = 1) {
  int g = iters < 256 * per_cu ? iters : 256 * per_cu;      // work-groups resident on the chip at once (126 - 137 KB of LDS each, or 2 x 80)
  return (g + 7) / 8 * 8;                       // multiple of 8: XCD-aware iteration order
}

}  // namespace

// plan of an H x W map for the matrix-product transforms: P, Q, padded number of bins (multiple of 8; bin = v * P + u) and
// tiles[6] (optional) = TY, TX, TH, TW, window rows, window columns.  0 if the map has no plan.
int os2d_dft_plan(int H, int W, int* P, int* Q, int* nbins, int* tiles) {
  DftPlan pl;
  if (H < 1 || W < 1 || !dft_plan_cached(H, W, &pl)) return 0;
  if (P) *P = pl.P;
  if (Q) *Q = pl.Q;
  if (nbins) *nbins = pl.NBINS;
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

size_t os2d_dft_matrices_size(int P, int Q) { return dft_matrices_units(P, Q) * 16; }

int os2d_launch_dft_matrices(const double* twP64, const double* twQ64, int P, int Q, void* out, hipStream_t stream) {
  if (P < 4 || (P & 3) || P > DFT_MAXP || Q < 2 || (Q & 1) || Q / 2 + 1 > DFT_MAXV) {
    os2d_set_error("dft_matrices: transform %d x %d is outside the kernels' range (P %% 4 == 0, P <= %d, Q even, Q/2 + 1 <= %d)", P, Q,
                   DFT_MAXP, DFT_MAXV);
    return -3;
  }
  const int n = (int)dft_matrices_units(P, Q);
  hipLaunchKernelGGL(dft_matrices_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, twP64, twQ64, P, Q, static_cast<u32x4v*>(out));
  return dft_check("dft_matrices");
}

int os2d_launch_dft_forward(const float* corr, const float* inv, float* X, const void* matrices, int NB, int C, int Cpad, int H, int W,
                            hipStream_t stream) {
  DftPlan pl;
  if (!dft_plan_cached(H, W, &pl)) {
    os2d_set_error("dft_forward: no transform plan for a %dx%d map", H, W);
    return -3;
  }
  if (pl.G == 8 && Cpad < dft_round_up(C, 8)) dft_plan_cached(H, W, &pl, false);      // a channel stride for 4 images per iteration only
  const int G = pl.G;
  if (Cpad < dft_round_up(C, G)) {
    os2d_set_error("dft_forward: channel stride %d < %d", Cpad, dft_round_up(C, G));
    return -1;
  }
  const int CG = (C + G - 1) / G, NBT = NB * pl.T, iters = NBT * CG;
  pl.inv_cg = dft_magic((unsigned)CG);
  const u32x4v* FqT = static_cast<const u32x4v*>(matrices);
  const u32x4v* Fp2 = FqT + dft_units_fqt(pl.P, pl.Q);
  auto kern = G == 8 ? dft_forward_pick<8, 8>(pl) : dft_forward_pick<4, 8>(pl);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds_total);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(dft_forward): %s", hipGetErrorString(e));
    return -4;
  }
  hipLaunchKernelGGL(kern, dim3(dft_grid(iters)), dim3(DFT_THR), pl.lds_total, stream, corr, inv, X, FqT, Fp2, pl,
                     C, Cpad, NBT, iters);
  return dft_check("dft_forward");
}

// zero_borders != 0: the kernel also writes the zero border cells of the planes it fills (no os2d_launch_border_zero_shb_planes)
int os2d_launch_dft_inverse(const float* Y, const float* bp, int MTP, void* out, const void* matrices, int NB, int Cout, int H, int W,
                            Os2dRangeFlag status, int zero_borders, hipStream_t stream) {
  DftPlan pl;
  if (!dft_plan_cached(H, W, &pl)) {
    os2d_set_error("dft_inverse: no transform plan for a %dx%d map", H, W);
    return -3;
  }
  if (Cout % 8) {
    os2d_set_error("dft_inverse: Cout %d must be a multiple of 8", Cout);
    return -1;
  }
  const int OG = Cout / pl.G, NBT = NB * pl.T, iters = NBT * OG;
  pl.inv_og = dft_magic((unsigned)OG);
  const u32x4v* E2 = static_cast<const u32x4v*>(matrices) + dft_units_fqt(pl.P, pl.Q) + dft_units_fp2(pl.P, pl.Q);
  const u32x4v* Gq = E2 + dft_units_e2(pl.P, pl.Q);
  auto kern = dft_inverse_pick(pl);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds_total);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(dft_inverse): %s", hipGetErrorString(e));
    return -4;
  }
  hipLaunchKernelGGL(kern, dim3(dft_grid(iters)), dim3(DFT_THR), pl.lds_total, stream, Y, bp, MTP, static_cast<unsigned char*>(out), E2, Gq,
                     pl, Cout, NBT, os2d_plane(H, W), os2d_ws(W), os2d_base(W), iters, status, zero_borders);
  return dft_check("dft_inverse");
}

// diagnostic builds only (phase stamps, see above): copy the 16 phase counters to the host and optionally reset them
extern "C" int os2d_debug_dft_stamps(unsigned long long* out16, int reset) {
#ifdef OS2D_DIAG_DFT_STAMPS
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(os2d_dft_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -4;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(os2d_dft_stamps), z, sizeof(z)) != hipSuccess) return -4;
  }
  return 0;
#else
  (void)out16, (void)reset;
  os2d_set_error("os2d_debug_dft_stamps: this library was built without -DOS2D_DIAG_DFT_STAMPS");
  return -1;
#endif
}

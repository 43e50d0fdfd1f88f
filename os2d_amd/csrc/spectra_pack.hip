// Weight spectra of the frequency-domain 7x7 TransformNet layer, built on the device (gfx950) - the host-side counterpart
// of fft.hip / spectral*.hip: once per transform size (P, Q) and parameter version, off the per-step path.
//
// The BatchNorm-folded 7x7 filters w[o][c][t][s] (float64: the fold is done in float64 by the caller) are centred on the
// origin of the P x Q grid, tap (t, s) at ((3 - t) mod P, (3 - s) mod Q), so that the circular convolution IS the zero-padded
// correlation of reference os2d/modeling/head.py:619 for the first H x W samples.  Their half spectrum is a 7-term DFT per
// axis,
//     K[o][c][u][v] = sum_s ( sum_t w[t][s] E_P[u * pos_t mod P] ) E_Q[v * pos_s mod Q],      E_n[m] = exp(-2 pi i m / n),
// evaluated in float64 from exact tables (the host rounds cos / sin of 2 pi m / n once): 126 multiply-adds per value instead
// of a P x Q transform of a map that is 99 % zeros (rounds 2 - 3 used torch.fft.rfft2, then float64 matrix products through
// rocBLAS: 50 / 20 ms per transform size; this kernel pair: ~2 ms, and no vendor library is left under the head).
//
// A work item owns ONE (output channel, input channel) pair - its 49 taps stay in registers - and walks over a range of bins;
// all lanes of a wave are at the same bin, so the 14 twiddles of a bin are wave-uniform.  Two passes: row maxima (the
// split-fp16 layout scales row o by the power of two that puts its largest |Kr|, |Ki| in (16384, 32768]), then pack:
//   OS2D_PRECISION_FFTX3  [bins/8][2 halves of 64 out-ch][ceil(C/8) k-steps][8 bins][2 channel groups][hi|lo][64] units of
//                         8 halves = (Kr, Ki) of 4 channels, followed by 128 floats 2^-wexp[o]   (spectral_f16.hip)
//   OS2D_PRECISION_FFT    [bins/8][2][C][8 bins][64] complex64                                    (spectral.hip)
#include "os2d_common.h"

namespace {

constexpr int SP_THR = 256;
constexpr int SP_KS = 7, SP_TAPS = 49, SP_R = 3;

struct cd {
  double re, im;
};

// K of this work item's (o, c) at bin (u, v): w[49] in registers, tables in LDS
__device__ __forceinline__ cd sp_value(const double (&w)[SP_TAPS], const cd* __restrict__ tP, const cd* __restrict__ tQ, int u,
                                       int v, int P, int Q) {
  cd ep[SP_KS], eq[SP_KS];
#pragma unroll
  for (int t = 0; t < SP_KS; ++t) {
    const int pt = (SP_R - t + P) % P, ps = (SP_R - t + Q) % Q;       // position of tap t on the P / Q grid
    ep[t] = tP[(u * pt) % P];
    eq[t] = tQ[(v * ps) % Q];
  }
  cd k{0.0, 0.0};
#pragma unroll
  for (int s = 0; s < SP_KS; ++s) {
    double rr = 0.0, ri = 0.0;
#pragma unroll
    for (int t = 0; t < SP_KS; ++t) {
      rr = fma(w[t * SP_KS + s], ep[t].re, rr);
      ri = fma(w[t * SP_KS + s], ep[t].im, ri);
    }
    k.re = fma(rr, eq[s].re, fma(-ri, eq[s].im, k.re));
    k.im = fma(rr, eq[s].im, fma(ri, eq[s].re, k.im));
  }
  return k;
}

// MODE 0: row maxima -> amax[o] (bits of a non-negative double, atomicMax);  1: split-fp16 units;  2: complex64 layout
template <int MODE>
__global__ __launch_bounds__(SP_THR) void spectra_pack_kernel(const double* __restrict__ wfold,   // [Cout][C][7][7]
                                                              const double* __restrict__ twP, const double* __restrict__ twQ,
                                                              int C, int Cout, int P, int Q, int NBINS, int bins_per_block,
                                                              unsigned long long* __restrict__ amax, void* __restrict__ out,
                                                              float* __restrict__ wscale, int ufast) {
  // ufast: bin = v * P + u (the bin order of the matrix-product transforms, dft_mfma.hip) instead of u * V + v
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cd* tP = reinterpret_cast<cd*>(smem);
  cd* tQ = tP + P;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < P; i += SP_THR) tP[i] = cd{twP[2 * i], twP[2 * i + 1]};
  for (int i = tid; i < Q; i += SP_THR) tQ[i] = cd{twQ[2 * i], twQ[2 * i + 1]};
  __syncthreads();
  const int V = Q / 2 + 1, KSTEPS = (C + 7) / 8;
  // block -> (half of 64 output channels, unit of 4 input channels); work item -> (o, c) in it.  MODE 1: lanes = 16 output
  // channels x 4 channels of the unit (the 4 lanes of a unit write its 16 bytes), a wave per 16 output channels;
  // MODE 0 / 2: lanes = 64 output channels (contiguous complex64 rows), a wave per channel of the unit
  const int cu = blockIdx.x % (KSTEPS * 2), half = blockIdx.x / (KSTEPS * 2);
  const int c4 = MODE == 1 ? (lane & 3) : wv;
  const int ol = MODE == 1 ? wv * 16 + (lane >> 2) : lane;
  const int c = cu * 4 + c4, o = half * 64 + ol;
  const bool live = c < C && o < Cout;
  double w[SP_TAPS];
#pragma unroll
  for (int k = 0; k < SP_TAPS; ++k) w[k] = live ? wfold[((size_t)o * C + c) * SP_TAPS + k] : 0.0;
  const int bin0 = blockIdx.y * bins_per_block, bin1 = min(bin0 + bins_per_block, NBINS);
  if (MODE == 0) {
    double m = 0.0;
    for (int bin = bin0; bin < min(bin1, P * V); ++bin) {
      const int u = ufast ? bin % P : bin / V, v = ufast ? bin / P : bin - u * V;
      const cd k = sp_value(w, tP, tQ, u, v, P, Q);
      m = fmax(m, fmax(fabs(k.re), fabs(k.im)));
    }
    if (live) atomicMax(amax + o, (unsigned long long)__double_as_longlong(m));    // non-negative doubles order like their bits
    return;
  }
  // exponent of the row: the largest power of two that keeps the row's largest entry <= 32768
  double scale = 1.0;
  if (MODE == 1) {
    const double mx = o < Cout ? __longlong_as_double((long long)amax[o]) : 0.0;
    int e = mx > 0.0 ? (int)floor(log2(32768.0 / mx)) : 0;
    e = max(-100, min(100, e));
    scale = ldexp(1.0, e);
    if (blockIdx.x % (KSTEPS * 2) == 0 && blockIdx.y == 0 && c4 == 0 && o < 128) wscale[o] = (float)ldexp(1.0, -e);
  }
  for (int bin = bin0; bin < bin1; ++bin) {
    cd k{0.0, 0.0};
    if (bin < P * V) {
      const int u = ufast ? bin % P : bin / V, v = ufast ? bin / P : bin - u * V;
      k = sp_value(w, tP, tQ, u, v, P, Q);
    }
    const int g = bin >> 3, j = bin & 7;
    if (MODE == 1) {
      const double vr = k.re * scale, vi = k.im * scale;
      const _Float16 hr = (_Float16)vr, hi_ = (_Float16)vi;
      const _Float16 lr = (_Float16)(vr - (double)hr), li = (_Float16)(vi - (double)hi_);
      const unsigned hbits = (unsigned)__builtin_bit_cast(unsigned short, hr) | ((unsigned)__builtin_bit_cast(unsigned short, hi_) << 16);
      const unsigned lbits = (unsigned)__builtin_bit_cast(unsigned short, lr) | ((unsigned)__builtin_bit_cast(unsigned short, li) << 16);
      // unit [g][half][k-step][j][channel group][hi|lo][o % 64], 16 bytes = (re, im) of 4 channels; this lane owns 4 of them
      const int ks = cu >> 1, grp = cu & 1;
      const size_t unit = ((((((size_t)g * 2 + half) * KSTEPS + ks) * 8 + j) * 2 + grp) * 2) * 64 + ol;
      unsigned* dst = reinterpret_cast<unsigned*>(out) + unit * 4 + c4;
      dst[0] = hbits;                 // written for padded channels / rows too (zeros): the buffer needs no memset
      dst[64 * 4] = lbits;
    } else {
      if (c < C) {
        float2* dst = reinterpret_cast<float2*>(out) + ((((size_t)g * 2 + half) * C + c) * 8 + j) * 64 + ol;
        *dst = make_float2((float)k.re, (float)k.im);
      }
    }
  }
}

int sp_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("%s launch: %s", what, hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

// wfold: DEVICE double [Cout <= 128][C][7][7]; twP64 / twQ64: DEVICE double [P][2] / [Q][2] = (cos, sin) of -2 pi m / n;
// split != 0: out = the split-fp16 layout of os2d_spectral_weight16_bytes (incl. the 128 trailing row scales), workspace =
// 128 x 8 bytes (row maxima); split == 0: out = the complex64 layout of os2d_spectral_weight_bytes
int os2d_launch_spectra_pack(const double* wfold, const double* twP64, const double* twQ64, int C, int Cout, int P, int Q,
                             int NBINS, int split, int u_fastest, void* out, void* workspace, hipStream_t stream) {
  const int V = Q / 2 + 1, KSTEPS = (C + 7) / 8;
  if (P * V > NBINS || (NBINS & 7) || Cout > 128 || P < 1 || Q < 2 || (Q & 1)) {
    os2d_set_error("spectra_pack: bad sizes (P=%d Q=%d NBINS=%d Cout=%d)", P, Q, NBINS, Cout);
    return -3;
  }
  const size_t lds = (size_t)(P + Q) * sizeof(cd);
  const int nb_y = 8, per = os2d_round_up((NBINS + nb_y - 1) / nb_y, 8);
  dim3 grid(2 * KSTEPS * 2, (NBINS + per - 1) / per);
  if (split) {
    unsigned long long* amax = static_cast<unsigned long long*>(workspace);
    hipError_t e = hipMemsetAsync(amax, 0, 128 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) {
      os2d_set_error("hipMemsetAsync: %s", hipGetErrorString(e));
      return -4;
    }
    hipLaunchKernelGGL(spectra_pack_kernel<0>, grid, dim3(SP_THR), lds, stream, wfold, twP64, twQ64, C, Cout, P, Q, NBINS, per, amax,
                       nullptr, nullptr, u_fastest);
    int rc = sp_check("spectra_pack (row maxima)");
    if (rc) return rc;
    float* wscale = reinterpret_cast<float*>(static_cast<char*>(out) + (size_t)(NBINS / 8) * 2 * KSTEPS * 8 * 256 * 16);
    hipLaunchKernelGGL(spectra_pack_kernel<1>, grid, dim3(SP_THR), lds, stream, wfold, twP64, twQ64, C, Cout, P, Q, NBINS, per, amax,
                       out, wscale, u_fastest);
    return sp_check("spectra_pack (split)");
  }
  hipLaunchKernelGGL(spectra_pack_kernel<2>, grid, dim3(SP_THR), lds, stream, wfold, twP64, twQ64, C, Cout, P, Q, NBINS, per, nullptr,
                     out, nullptr, u_fastest);
  return sp_check("spectra_pack (complex64)");
}

// Small HBM-bound helper kernels around the MFMA kernels (gfx950):
//   fm_sumsq       per-position sum of squares over channels of the image feature map (head.py:339)
//   border_zero    zero the border cells (and the 226th pad channel) of the normalised-correlation planes
//   class_prepare  class feature map -> 15x15 bilinear resize + L2 normalisation + GEMM packing
//                  (head.py:241-259, :293; the per-class constants an Os2dHead holds)
//   pack_conv      fold eval-mode BatchNorm into a convolution and repack it for conv_mfma.hip
#include "os2d_common.h"

namespace {

// ---- fm_sumsq: one block per 16 positions (300 blocks at 60x80, so every CU streams), 256 threads = 16 positions x 16
// channel lanes (64-byte row segments), fixed-order LDS tree (deterministic: no atomics, so repeated calls and class
// chunking give bit-identical results)
constexpr int SUMSQ_LANES = 16, SUMSQ_POS = 16;

__global__ __launch_bounds__(SUMSQ_LANES * SUMSQ_POS) void fm_sumsq_kernel(const float* __restrict__ fm,
                                                                           float* __restrict__ sumsq, int C, int HW) {
  __shared__ float red[SUMSQ_LANES][SUMSQ_POS];
  const int col = threadIdx.x % SUMSQ_POS, cl = threadIdx.x / SUMSQ_POS;
  const int n = blockIdx.x * SUMSQ_POS + col;
  const int a = blockIdx.y;
  float s = 0.f;
  if (n < HW) {
    const float* p = fm + (size_t)a * C * HW + n;
    for (int c0 = cl; c0 < C; c0 += SUMSQ_LANES * 8) {  // 8 independent loads in flight, accumulated in channel order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * SUMSQ_LANES;
        v[u] = p[(size_t)min(c, C - 1) * HW];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u * SUMSQ_LANES < C) s += v[u] * v[u];
    }
  }
  red[cl][col] = s;
  __syncthreads();
  if (cl == 0 && n < HW) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < SUMSQ_LANES; ++i) t += red[i][col];
    sumsq[(size_t)a * HW + n] = t;
  }
}

// ---- border_zero: one block per plane
__global__ __launch_bounds__(256) void border_zero_kernel(float* __restrict__ rpad, int H, int W, int PLANE) {
  const int ch = blockIdx.x % OS2D_KP;
  float* p = rpad + (size_t)blockIdx.x * PLANE;
  const bool whole = ch >= OS2D_K;
  for (int i = threadIdx.x; i < PLANE; i += 256)
    if (whole || !os2d_interior(i, H, W)) p[i] = 0.f;
}

// ---- corr_normalize: standalone relu -> L2 over the 225 channels (head.py:650) of an arbitrary correlation
// tensor [NB][225][HW] into the padded-plane layout (the fused path does this in the GEMM epilogue)
__global__ __launch_bounds__(256) void corr_normalize_kernel(const float* __restrict__ corr, float* __restrict__ rpad,
                                                             int H, int W, int PLANE) {
  const int HW = H * W, Ws = os2d_ws(W);
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = blockIdx.y;
  if (n >= HW) return;
  const float* c = corr + (size_t)nb * OS2D_K * HW + n;
  float s = 0.f;
  for (int k = 0; k < OS2D_K; ++k) {
    const float v = os2d_relu(c[(size_t)k * HW]);
    s += v * v;
  }
  const float inv = 1.0f / (sqrtf(s) + 1e-6f);
  const int h = n / W, w = n - h * W;
  float* o = rpad + (size_t)nb * OS2D_KP * PLANE + (size_t)os2d_base(W) + (size_t)h * Ws + w;
  for (int k = 0; k < OS2D_K; ++k) o[(size_t)k * PLANE] = os2d_relu(c[(size_t)k * HW]) * inv;
}

// ---- class_prepare: grid 225 (one block per template cell) x classes, block 256 loops over channels
__device__ __forceinline__ void class_prepare_body(const float* __restrict__ src, int C, int h, int w, int normalize,
                                                   float* __restrict__ q15, float* __restrict__ qp, float* red) {
  const int cell = blockIdx.x;  // i*15 + j  (row i, col j)
  const int i = cell / OS2D_T, j = cell - i * OS2D_T;
  // sampling position of the identity grid, as F.affine_grid(align_corners=True) + F.grid_sample build it
  // (torch.linspace(-1, 1, 15) on the CPU: fused multiply-adds from the start for k < 7, from the end for k >= 7)
  const float step = 2.0f / (OS2D_T - 1);
  const float xu = (j < OS2D_T / 2) ? __fmaf_rn(step, (float)j, -1.0f) : __fmaf_rn(-step, (float)(OS2D_T - 1 - j), 1.0f);
  const float yu = (i < OS2D_T / 2) ? __fmaf_rn(step, (float)i, -1.0f) : __fmaf_rn(-step, (float)(OS2D_T - 1 - i), 1.0f);
  const float ix = ((xu + 1.0f) * 0.5f) * (float)(w - 1);
  const float iy = ((yu + 1.0f) * 0.5f) * (float)(h - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float ax = ix - fx0, ay = iy - fy0;
  // zero padding: corners outside the map contribute 0 (their weight is 0 anyway on the identity grid)
  const bool x0in = x0 >= 0 && x0 < w, x1in = x0 + 1 >= 0 && x0 + 1 < w;
  const bool y0in = y0 >= 0 && y0 < h, y1in = y0 + 1 >= 0 && y0 + 1 < h;
  auto sample = [&](int c) -> float {
    const float* p = src + (size_t)c * h * w;
    const float v00 = (x0in && y0in) ? p[y0 * w + x0] : 0.f;
    const float v01 = (x1in && y0in) ? p[y0 * w + x0 + 1] : 0.f;
    const float v10 = (x0in && y1in) ? p[(y0 + 1) * w + x0] : 0.f;
    const float v11 = (x1in && y1in) ? p[(y0 + 1) * w + x0 + 1] : 0.f;
    return v00 * (1.f - ax) * (1.f - ay) + v01 * ax * (1.f - ay) + v10 * (1.f - ax) * ay + v11 * ax * ay;
  };
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = sample(c);
    s += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = normalize ? 1.0f / (sqrtf(red[0] + red[1] + red[2] + red[3]) + 1e-5f) : 1.0f;  // head.py:293
  const int m = j * OS2D_T + i;  // x-major correlation channel (head.py:342-344)
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = sample(c) * inv;
    q15[(size_t)c * OS2D_K + cell] = v;
    qp[(size_t)c * OS2D_QROWS + m] = v;
  }
  // zero the 31 pad rows of the GEMM operand (done by the block of cell 0)
  if (cell == 0)
    for (int t = threadIdx.x; t < C * (OS2D_QROWS - OS2D_K); t += 256) {
      const int c = t / (OS2D_QROWS - OS2D_K), r = t - c * (OS2D_QROWS - OS2D_K);
      qp[(size_t)c * OS2D_QROWS + OS2D_K + r] = 0.f;
    }
}

__global__ __launch_bounds__(256) void class_prepare_kernel(const float* __restrict__ src, int C, int h, int w, int normalize,
                                                            float* __restrict__ q15, float* __restrict__ qp) {
  __shared__ float red[4];
  class_prepare_body(src, C, h, w, normalize, q15, qp, red);
}

// ---- all classes of a head in TWO launches (class b's map, of its own size h x w, is srcs[b]; sizes = [B][2] (h, w)).
// A block owns 32 channels of one class and walks them one at a time with a thread per template cell, so the 4 bilinear
// taps of neighbouring threads are neighbours in the 0.9 KB source plane and every store is a contiguous 900-byte /
// 1 KB row (the one-block-per-cell kernel above reads and writes 4 bytes per 900-byte stride: fine for one class, 0.66 ms
// for 64).  Pass 1 writes the resized, not yet normalised values and the block's partial sums of squares per cell;
// pass 2 adds the partials of all channel blocks in a fixed order and rescales in place.
constexpr int CPB = 32;  // channels per block

__device__ __forceinline__ float class_resize_sample(const float* __restrict__ p, int h, int w, int cell) {
  const int i = cell / OS2D_T, j = cell - i * OS2D_T;
  const float step = 2.0f / (OS2D_T - 1);
  const float xu = (j < OS2D_T / 2) ? __fmaf_rn(step, (float)j, -1.0f) : __fmaf_rn(-step, (float)(OS2D_T - 1 - j), 1.0f);
  const float yu = (i < OS2D_T / 2) ? __fmaf_rn(step, (float)i, -1.0f) : __fmaf_rn(-step, (float)(OS2D_T - 1 - i), 1.0f);
  const float ix = ((xu + 1.0f) * 0.5f) * (float)(w - 1);
  const float iy = ((yu + 1.0f) * 0.5f) * (float)(h - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float ax = ix - fx0, ay = iy - fy0;
  const bool x0in = x0 >= 0 && x0 < w, x1in = x0 + 1 >= 0 && x0 + 1 < w;
  const bool y0in = y0 >= 0 && y0 < h, y1in = y0 + 1 >= 0 && y0 + 1 < h;
  const float v00 = (x0in && y0in) ? p[y0 * w + x0] : 0.f;
  const float v01 = (x1in && y0in) ? p[y0 * w + x0 + 1] : 0.f;
  const float v10 = (x0in && y1in) ? p[(y0 + 1) * w + x0] : 0.f;
  const float v11 = (x1in && y1in) ? p[(y0 + 1) * w + x0 + 1] : 0.f;
  return v00 * (1.f - ax) * (1.f - ay) + v01 * ax * (1.f - ay) + v10 * (1.f - ax) * ay + v11 * ax * ay;
}

__global__ __launch_bounds__(256) void class_resize_batch_kernel(const float* const* __restrict__ srcs,
                                                                 const int* __restrict__ sizes, int C,
                                                                 float* __restrict__ q15, float* __restrict__ qp,
                                                                 float* __restrict__ partial) {
  const int b = blockIdx.y, cb = blockIdx.x, cell = threadIdx.x;
  const int h = sizes[2 * b], w = sizes[2 * b + 1];
  const float* src = srcs[b];
  const int i = cell / OS2D_T, j = cell - i * OS2D_T;
  const int m = j * OS2D_T + i;  // x-major correlation channel (head.py:342-344)
  float s = 0.f;
  const int c1 = min(C, (cb + 1) * CPB);
  for (int c = cb * CPB; c < c1; ++c) {
    float v = 0.f;
    if (cell < OS2D_K) {
      v = class_resize_sample(src + (size_t)c * h * w, h, w, cell);
      q15[((size_t)b * C + c) * OS2D_K + cell] = v;
      qp[((size_t)b * C + c) * OS2D_QROWS + m] = v;
      s += v * v;
    } else {
      qp[((size_t)b * C + c) * OS2D_QROWS + cell] = 0.f;  // the 31 pad rows of the GEMM operand
    }
  }
  if (cell < OS2D_K) partial[((size_t)b * gridDim.x + cb) * OS2D_K + cell] = s;
}

__global__ __launch_bounds__(256) void class_normalize_batch_kernel(int C, int nblocks, float* __restrict__ q15,
                                                                    float* __restrict__ qp,
                                                                    const float* __restrict__ partial) {
  const int b = blockIdx.y, cb = blockIdx.x, cell = threadIdx.x;
  if (cell >= OS2D_K) return;
  float s = 0.f;
  for (int k = 0; k < nblocks; ++k) s += partial[((size_t)b * nblocks + k) * OS2D_K + cell];  // fixed order: deterministic
  const float inv = 1.0f / (sqrtf(s) + 1e-5f);  // head.py:293
  const int i = cell / OS2D_T, j = cell - i * OS2D_T;
  const int m = j * OS2D_T + i;
  const int c1 = min(C, (cb + 1) * CPB);
  for (int c = cb * CPB; c < c1; ++c) {
    const size_t o = ((size_t)b * C + c) * OS2D_K + cell;
    const float v = q15[o] * inv;
    q15[o] = v;
    qp[((size_t)b * C + c) * OS2D_QROWS + m] = v;
  }
}

// ---- corr_normalize_shb: standalone relu -> L2 over the 225 channels (head.py:650) of an arbitrary correlation tensor
// [NB][225][HW] into the split-half blocked layout of the f16x3 convolutions ([NB][29][hi|lo][PLANE] units of 8 halves,
// values scaled by 2^OS2D_RNORM_EXP); the fused path does this in the correlation epilogue.  Border cells are cleared
// by border_zero_shb before.
__global__ __launch_bounds__(256) void corr_normalize_shb_kernel(const float* __restrict__ corr, uint4* __restrict__ rshb,
                                                                 int H, int W, int PLANE) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  const int HW = H * W, Ws = os2d_ws(W);
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = blockIdx.y;
  if (n >= HW) return;
  const float* c = corr + (size_t)nb * OS2D_K * HW + n;
  float s = 0.f;
  for (int k = 0; k < OS2D_K; ++k) {
    const float v = fmaxf(c[(size_t)k * HW], 0.f);
    s += v * v;
  }
  const float inv = 1.0f / (sqrtf(s) + 1e-6f);
  const float scale = ldexpf(1.0f, OS2D_RNORM_EXP);
  const int h = n / W, w = n - h * W;
  uint4* o = rshb + (size_t)nb * OS2D_G * 2 * PLANE + (size_t)os2d_base(W) + (size_t)h * Ws + w;
  for (int g = 0; g < OS2D_G; ++g) {
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 8 + j;
      const float v = k < OS2D_K ? fmaxf(c[(size_t)k * HW], 0.f) * inv * scale : 0.f;
      const _Float16 hv = (_Float16)v;
      hi[j] = hv;
      lo[j] = (_Float16)(v - (float)hv);
    }
    *reinterpret_cast<half8*>(o + (size_t)(2 * g) * PLANE) = hi;
    *reinterpret_cast<half8*>(o + (size_t)(2 * g + 1) * PLANE) = lo;
  }
}

// ---- pack_conv: wp[cp][tap][half][o] = w[o][2cp+half][tap] * bn_scale[o]; bp[o] = folded bias
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                        const float* __restrict__ bn_mean,
                                                        const float* __restrict__ bn_var, float bn_eps, int Cout,
                                                        int Cin, int KS, int MT, float* __restrict__ wp,
                                                        float* __restrict__ bp) {
  const int taps = KS * KS;
  const int CinP = (Cin + 1) & ~1;
  const size_t total = (size_t)(CinP / 2) * taps * 2 * MT;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int o = idx % MT;
    const int half = (idx / MT) % 2;
    const int tap = (idx / (2 * MT)) % taps;
    const int cp = idx / ((size_t)2 * MT * taps);
    const int c = 2 * cp + half;
    float v = 0.f;
    if (o < Cout && c < Cin) {
      const float s = bn_w ? bn_w[o] / sqrtf(bn_var[o] + bn_eps) : 1.0f;
      v = w[((size_t)o * Cin + c) * taps + tap] * s;
    }
    wp[idx] = v;
  }
  if (blockIdx.x == 0)
    for (int o = threadIdx.x; o < MT; o += 256) {
      float v = 0.f;
      if (o < Cout) {
        if (bn_w) {
          const float s = bn_w[o] / sqrtf(bn_var[o] + bn_eps);
          v = (b[o] - bn_mean[o]) * s + bn_b[o];
        } else {
          v = b[o];
        }
      }
      bp[o] = v;
    }
}

// ---- f16x3 path (conv_f16x3.hip): packed weights [G][steps_padded][2 half-wave][2 hi|lo][MT] units of 8 halves.
// Unit (g, ps, hw, part, o) holds channels 8g..8g+7 of output o at tap 2*ps+hw, BN-folded and scaled by 2^(wexp[o] - in_exp[c]) (one exponent per output and per input channel);
// part 0 = rn16(x), part 1 = rn16(x - hi).  Taps / channels / outputs past the real sizes are zero.
__global__ __launch_bounds__(256) void pack_conv_f16_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ bn_w,
                                                            const float* __restrict__ bn_b,
                                                            const float* __restrict__ bn_mean,
                                                            const float* __restrict__ bn_var, float bn_eps, int Cout,
                                                            int Cin, int KS, int MT, int steps_padded,
                                                            const int* __restrict__ wexp,
                                                            const int* __restrict__ in_exp,
                                                            const int* __restrict__ out_exp,
                                                            _Float16* __restrict__ wp, float* __restrict__ bp) {
  const int taps = KS * KS;
  const int G = (Cin + 7) / 8;
  const size_t total = (size_t)G * steps_padded * 2 * 2 * MT * 8;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    size_t t = idx;
    const int j = t % 8;
    t /= 8;
    const int o = t % MT;
    t /= MT;
    const int part = t % 2;
    t /= 2;
    const int hw = t % 2;
    t /= 2;
    const int ps = t % steps_padded;
    const int g = t / steps_padded;
    const int c = g * 8 + j, tap = 2 * ps + hw;
    float v = 0.f;
    if (o < Cout && c < Cin && tap < taps) {
      const float s = bn_w ? bn_w[o] / sqrtf(bn_var[o] + bn_eps) : 1.0f;
      // exact power-of-two scales: 2^wexp[o] per output channel, 2^-in_exp[c] per input channel (the input buffer holds
      // x[c] * 2^in_exp[c]); one ldexpf of the summed exponents, so no intermediate can over- or underflow
      v = ldexpf(w[((size_t)o * Cin + c) * taps + tap] * s, wexp[o] - in_exp[c]);
    }
    const _Float16 hi = (_Float16)v;
    wp[idx] = part ? (_Float16)(v - (float)hi) : hi;
  }
  if (blockIdx.x == 0)
    for (int o = threadIdx.x; o < MT; o += 256) {
      float v = 0.f;
      if (o < Cout) {
        if (bn_w) {
          const float s = bn_w[o] / sqrtf(bn_var[o] + bn_eps);
          v = (b[o] - bn_mean[o]) * s + bn_b[o];
        } else {
          v = b[o];
        }
      }
      bp[o] = v;
      bp[MT + o] = o < Cout ? ldexpf(1.0f, -wexp[o]) : 0.f;                            // undoes the weight scale
      bp[2 * MT + o] = (o < Cout && out_exp) ? ldexpf(1.0f, out_exp[o]) : (o < Cout ? 1.0f : 0.f);  // output buffer scale
    }
}

// border cells of the split-half blocked normalised-correlation buffer [NB][29][2][PLANE] x 16 B
// (the pad cells are enumerated directly: the BASE cells in front, the 3 cells after every row, the tail)
// (blocks beyond `planes`: the inverse norms of the packed correlation kernel, 256 positions each - one launch for both)
__global__ __launch_bounds__(256) void border_zero_shb_kernel(uint4* __restrict__ p, int H, int W, int PLANE, int planes,
                                                              unsigned long long* __restrict__ sumfx, float* __restrict__ invn, size_t n) {
  if ((int)blockIdx.x >= planes) {
    const size_t i = (size_t)(blockIdx.x - planes) * 256 + threadIdx.x;
    if (i < n) os2d_corr_norm_finalize_one(sumfx, invn, i);
    return;
  }
  uint4* q = p + (size_t)blockIdx.x * PLANE;
  const int Ws = os2d_ws(W), BASE = os2d_base(W);
  const int rows = H * OS2D_PAD, tail0 = BASE + H * Ws;
  const int npad = BASE + rows + (PLANE - tail0);
  for (int k = threadIdx.x; k < npad; k += 256) {
    int cell;
    if (k < BASE) cell = k;
    else if (k < BASE + rows) {
      const int j = k - BASE;
      cell = BASE + (j / OS2D_PAD) * Ws + W + j % OS2D_PAD;
    } else cell = tail0 + (k - BASE - rows);
    q[cell] = make_uint4(0u, 0u, 0u, 0u);
  }
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("%s launch: %s", what, hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

int os2d_launch_fm_sumsq(const float* fm, float* sumsq, int A, int C, int HW, hipStream_t stream) {
  hipLaunchKernelGGL(fm_sumsq_kernel, dim3((HW + SUMSQ_POS - 1) / SUMSQ_POS, A), dim3(SUMSQ_LANES * SUMSQ_POS), 0, stream,
                     fm, sumsq, C, HW);
  return check_launch("fm_sumsq");
}

int os2d_launch_border_zero(float* rpad, int planes_total, int H, int W, hipStream_t stream) {
  hipLaunchKernelGGL(border_zero_kernel, dim3(planes_total), dim3(256), 0, stream, rpad, H, W, os2d_plane(H, W));
  return check_launch("border_zero");
}

int os2d_launch_border_zero_shb_planes_norms(void* buf, int planes, int H, int W, void* sumfx, float* invn, size_t n, hipStream_t stream) {
  const size_t extra = sumfx ? (n + 255) / 256 : 0;
  hipLaunchKernelGGL(border_zero_shb_kernel, dim3((unsigned)(planes + extra)), dim3(256), 0, stream, reinterpret_cast<uint4*>(buf), H, W,
                     os2d_plane(H, W), planes, static_cast<unsigned long long*>(sumfx), invn, sumfx ? n : (size_t)0);
  return check_launch("border_zero_shb");
}

int os2d_launch_border_zero_shb(void* rnorm, int NB, int H, int W, hipStream_t stream) {
  return os2d_launch_border_zero_shb_planes_norms(rnorm, NB * OS2D_G * 2, H, W, nullptr, nullptr, 0, stream);
}

int os2d_launch_border_zero_shb_planes(void* buf, int planes, int H, int W, hipStream_t stream) {
  return os2d_launch_border_zero_shb_planes_norms(buf, planes, H, W, nullptr, nullptr, 0, stream);
}

int os2d_launch_pack_conv_f16(const float* w, const float* b, const float* bn_w, const float* bn_b, const float* bn_mean,
                              const float* bn_var, float bn_eps, int Cout, int Cin, int KS, int MT, int steps_padded,
                              const int* wexp, const int* in_exp, const int* out_exp, void* wp, float* bp,
                              hipStream_t stream) {
  hipLaunchKernelGGL(pack_conv_f16_kernel, dim3(1024), dim3(256), 0, stream, w, b, bn_w, bn_b, bn_mean, bn_var, bn_eps,
                     Cout, Cin, KS, MT, steps_padded, wexp, in_exp, out_exp, reinterpret_cast<_Float16*>(wp), bp);
  return check_launch("pack_conv_f16");
}

int os2d_launch_corr_normalize(const float* corr, float* rpad, int NB, int H, int W, hipStream_t stream) {
  int rc = os2d_launch_border_zero(rpad, NB * OS2D_KP, H, W, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(corr_normalize_kernel, dim3((H * W + 255) / 256, NB), dim3(256), 0, stream, corr, rpad, H, W,
                     os2d_plane(H, W));
  return check_launch("corr_normalize");
}

int os2d_launch_class_prepare(const float* src, int C, int h, int w, int normalize, float* q15, float* qp, hipStream_t stream) {
  hipLaunchKernelGGL(class_prepare_kernel, dim3(OS2D_K), dim3(256), 0, stream, src, C, h, w, normalize, q15, qp);
  return check_launch("class_prepare");
}

int os2d_class_prepare_partial_floats(int B, int C) { return B * ((C + CPB - 1) / CPB) * OS2D_K; }

int os2d_launch_class_prepare_batch(const float* const* srcs, const int* sizes, int B, int C, int normalize, float* q15,
                                    float* qp, float* partial, hipStream_t stream) {
  const int nblocks = (C + CPB - 1) / CPB;
  hipLaunchKernelGGL(class_resize_batch_kernel, dim3(nblocks, B), dim3(256), 0, stream, srcs, sizes, C, q15, qp, partial);
  int rc = check_launch("class_resize_batch");
  if (rc || !normalize) return rc;
  hipLaunchKernelGGL(class_normalize_batch_kernel, dim3(nblocks, B), dim3(256), 0, stream, C, nblocks, q15, qp, partial);
  return check_launch("class_normalize_batch");
}

int os2d_launch_corr_normalize_shb(const float* corr, void* rshb, int NB, int H, int W, hipStream_t stream) {
  int rc = os2d_launch_border_zero_shb(rshb, NB, H, W, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(corr_normalize_shb_kernel, dim3((H * W + 255) / 256, NB), dim3(256), 0, stream, corr,
                     reinterpret_cast<uint4*>(rshb), H, W, os2d_plane(H, W));
  return check_launch("corr_normalize_shb");
}

int os2d_launch_pack_conv(const float* w, const float* b, const float* bn_w, const float* bn_b, const float* bn_mean,
                          const float* bn_var, float bn_eps, int Cout, int Cin, int KS, int MT, float* wp, float* bp,
                          hipStream_t stream) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(512), dim3(256), 0, stream, w, b, bn_w, bn_b, bn_mean, bn_var, bn_eps,
                     Cout, Cin, KS, MT, wp, bp);
  return check_launch("pack_conv");
}

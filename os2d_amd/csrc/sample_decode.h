// The per-location work of the alignment epilogue (sample_decode.hip): theta assembly, resample + pool, box / corner extraction and
// loc encoding for ONE (pair, location) - shared by sample_decode_kernel (parameters from HBM) and the fused last-layer kernel of
// conv3_f16x3.hip (parameters straight from the accumulators, through LDS).  One source for both: the same operations in the same
// order, hence the same bits.
#pragma once
#include "os2d_common.h"

namespace {

constexpr int POOL_LO = 2, POOL_HI = OS2D_T - 2;  // head.py:280,296-302: pool_border_width = 2

// Coordinate k of the 15-point template grid exactly as torch.linspace(-1, 1, 15) (the base grid of F.affine_grid with
// align_corners=True, head.py:184) produces it on the CPU: fused multiply-adds from the start for the first half, from
// the end for the second (so the middle element is -4.47e-08, not 0) - it matters once a transform zooms 1000x.
__device__ __forceinline__ float os2d_template_coord(int k) {
  const float step = 2.0f / (OS2D_T - 1);
  return k < OS2D_T / 2 ? __fmaf_rn(step, (float)k, -1.0f) : __fmaf_rn(-step, (float)(OS2D_T - 1 - k), 1.0f);
}

// Transformation parameters of one location -> the 2x3 affine map used for sampling (reference head.py:81-153):
// P = 6 full affine, P = 4 scale + translation; optional inverse of the homogeneous 3x3 matrix.
__device__ __forceinline__ void os2d_theta(const float* __restrict__ pp, int HW /* stride between the parameters */, int P, int inverse, float& t00,
                                           float& t01, float& t02, float& t10, float& t11, float& t12) {
  if (P == 6) {  // head.py:98-100
    t00 = pp[0];
    t01 = pp[HW];
    t02 = pp[2 * (size_t)HW];
    t10 = pp[3 * (size_t)HW];
    t11 = pp[4 * (size_t)HW];
    t12 = pp[5 * (size_t)HW];
  } else {  // head.py:101-107: scale + translation only
    t00 = pp[0];
    t01 = 0.f;
    t02 = pp[HW];
    t10 = 0.f;
    t11 = pp[2 * (size_t)HW];
    t12 = pp[3 * (size_t)HW];
  }
  if (inverse) {  // head.py:111-151: inverse of [[A t],[0 0 1]] = [[A^-1, -A^-1 t],[0 0 1]]
    // Evaluated in fp64 (a dozen operations per location): for ill-conditioned matrices (det ~ 1e-6 with entries ~ 1)
    // the fp32 determinant loses every digit to cancellation, while the reference's pivoted LU (torch.inverse) does
    // not; fp64 keeps the closed form within round-off of the exact inverse of the fp32 parameters.
    double a = t00, b = t01, c = t10, d = t11;
    double det = a * d - b * c;
    double hom = 1.0;
    if (det == 0.0) {
      // torch.inverse raises on an exactly singular matrix and the reference then retries the whole chunk with
      // +1e-5 on the diagonal (head.py:125-134).  We regularise only the singular matrix itself.
      a = (double)(t00 + 1e-5f);
      d = (double)(t11 + 1e-5f);
      hom = (double)(1.0f + 1e-5f);
      det = a * d - b * c;
    }
    const double r = 1.0 / det;
    const double i00 = d * r, i01 = -b * r, i10 = -c * r, i11 = a * r;
    const double i02 = -(i00 * (double)t02 + i01 * (double)t12) / hom;
    const double i12 = -(i10 * (double)t02 + i11 * (double)t12) / hom;
    t00 = (float)i00;
    t01 = (float)i01;
    t02 = (float)i02;
    t10 = (float)i10;
    t11 = (float)i11;
    t12 = (float)i12;
  }
}


// everything after the parameters of location (h, w) of a pair are known: ``pp`` / ``pstride`` as os2d_theta takes them, ``cbase`` = the
// pair's correlation block [225][HW], ``ob`` = its slot in the output tensors
__device__ __forceinline__ void os2d_sample_decode_location(const float* __restrict__ cbase, const float* __restrict__ pp, int pstride, int H,
                                                            int W, int P, int inverse, float stride, float half_box, size_t ob, int n, int h,
                                                            int w, float* __restrict__ loc, float* __restrict__ cls,
                                                            float* __restrict__ corners) {
  const int HW = H * W;
  float t00, t01, t02, t10, t11, t12;
  os2d_theta(pp, pstride, P, inverse, t00, t01, t02, t10, t11, t12);

  // ---- resample + pool: 11x11 inner template points, channel = j*15 + i (x-major)
  const float half_t = 0.5f * OS2D_T;  // feature-map level anchor: box 15, stride 1, centre (w+.5, h+.5)
  const float cx = (float)w + 0.5f, cy = (float)h + 0.5f;
  const float wmax = (float)(W - 1), hmax = (float)(H - 1);
  float sum = 0.f;
  // One template column (11 taps = 44 gathers) at a time, in two passes: ALL 44 loads are requested before the first one is
  // used (round 5).  Written tap by tap the compiler kept 4 loads in flight per thread (24 registers): at 64 classes - 19 waves
  // per CU - the kernel was a chain of 121 cache round trips per thread (0.087 ms against 0.052 ms per 64 classes inside the
  // 1024-class launch, where occupancy hides them).  Same operations in the same order: the sums are bit-identical.
  constexpr int NTAP = POOL_HI - POOL_LO;
  for (int j = POOL_LO; j < POOL_HI; ++j) {
    const float xj = os2d_template_coord(j);
    float axs[NTAP], ays[NTAP], v00[NTAP], v01[NTAP], v10[NTAP], v11[NTAP];
#pragma unroll
    for (int i = POOL_LO; i < POOL_HI; ++i) {
      const float yi = os2d_template_coord(i);
      const float gx = t00 * xj + t01 * yi + t02;
      const float gy = t10 * xj + t11 * yi + t12;
      const float X = fminf(fmaxf(gx * half_t + cx, 0.f), wmax);
      const float Y = fminf(fmaxf(gy * half_t + cy, 0.f), hmax);
      const float fx0 = floorf(X), fy0 = floorf(Y);
      axs[i - POOL_LO] = X - fx0;
      ays[i - POOL_LO] = Y - fy0;
      const int x0 = (int)fx0, y0 = (int)fy0;
      const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
      const float* c = cbase + (size_t)(j * OS2D_T + i) * HW;
      v00[i - POOL_LO] = c[y0 * W + x0];
      v01[i - POOL_LO] = c[y0 * W + x1];
      v10[i - POOL_LO] = c[y1 * W + x0];
      v11[i - POOL_LO] = c[y1 * W + x1];
    }
    __builtin_amdgcn_sched_barrier(0);      // the scheduler keeps the 44 requests in front of the arithmetic
#pragma unroll
    for (int k = 0; k < NTAP; ++k) {
      const float ax = axs[k], ay = ays[k];
      sum += (v00[k] * (1.f - ax) + v01[k] * ax) * (1.f - ay) + (v10[k] * (1.f - ax) + v11[k] * ax) * ay;
    }
  }
  cls[ob * HW + n] = sum * (1.0f / ((POOL_HI - POOL_LO) * (POOL_HI - POOL_LO)));

  // ---- box of the transformed template in image coordinates (the 4 corners bound the affine image)
  const float ecx = stride * cx, ecy = stride * cy;
  float U[4], V[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float yi = (k & 2) ? 1.0f : -1.0f;  // template row 0 / 14
    const float xj = (k & 1) ? 1.0f : -1.0f;  // template col 0 / 14
    U[k] = (t00 * xj + t01 * yi + t02) * half_box + ecx;
    V[k] = (t10 * xj + t11 * yi + t12) * half_box + ecy;
    corners[(ob * 8 + 2 * k) * HW + n] = U[k];
    corners[(ob * 8 + 2 * k + 1) * HW + n] = V[k];
  }
  float x1 = fminf(fminf(U[0], U[1]), fminf(U[2], U[3]));
  float x2 = fmaxf(fmaxf(U[0], U[1]), fmaxf(U[2], U[3]));
  float y1 = fminf(fminf(V[0], V[1]), fminf(V[2], V[3]));
  float y2 = fmaxf(fmaxf(V[0], V[1]), fmaxf(V[2], V[3]));
  if (x1 + 1.0f > x2) x2 = x1 + 1.0f;  // bounding_box.py:267-277
  if (y1 + 1.0f > y2) y2 = y1 + 1.0f;
  const float size = 2.0f * half_box;
  const float bw = x2 - x1, bh = y2 - y1;
  const float gcx = x1 + 0.5f * bw, gcy = y1 + 0.5f * bh;
  // anchors as the reference builds them: xyxy = centre -+ size/2, then centre = x1 + 0.5*w
  const float ax1 = ecx - half_box, ay1 = ecy - half_box;
  const float aw = (ecx + half_box) - ax1, ah = (ecy + half_box) - ay1;
  const float acx = ax1 + 0.5f * aw, acy = ay1 + 0.5f * ah;
  (void)size;
  loc[(ob * 4 + 0) * HW + n] = 10.0f * (gcx - acx) / aw;
  loc[(ob * 4 + 1) * HW + n] = 10.0f * (gcy - acy) / ah;
  loc[(ob * 4 + 2) * HW + n] = 5.0f * logf(bw / aw);
  loc[(ob * 4 + 3) * HW + n] = 5.0f * logf(bh / ah);
}

// the outputs of a location of a flagged image (non-finite input): NaN, as the reference's torch.relu / norm propagate it
__device__ __forceinline__ void os2d_sample_decode_poison(int HW, size_t ob, int n, float* __restrict__ loc, float* __restrict__ cls,
                                                          float* __restrict__ corners) {
  const float qnan = __builtin_nanf("");
  cls[ob * HW + n] = qnan;
#pragma unroll
  for (int k = 0; k < 8; ++k) corners[(ob * 8 + k) * HW + n] = qnan;
#pragma unroll
  for (int k = 0; k < 4; ++k) loc[(ob * 4 + k) * HW + n] = qnan;
}

}  // namespace

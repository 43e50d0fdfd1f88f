// Fused alignment epilogue of the OS2D head (gfx950): per (class, location) it turns the TransformNet
// parameters into the affine map, resamples + pools the correlation tensor along the transformed 15x15
// template grid, and encodes the transformed template box.  Replaces, without materialising any
// [H,W,15,15,2] grid tensor:
//   reference os2d/modeling/head.py:81-153   (theta assembly, optional inverse)
//                                  :184      (F.affine_grid, align_corners=True)
//                                  :371-384  (local -> feature-map coordinates, clamp)
//                                  :439-520  (resample_of_correlation_map_fast + mask pooling; fp64 there,
//                                             fp32 here - agrees to ~1e-7, see tests)
//                                  :405-435  (image-level points, min/max box, corners, build_loc_targets)
//   reference os2d/modeling/box_coder.py:306-317, os2d/structures/bounding_box.py:267-277 (encode, min size)
// plus the per-location decode that follows the head (box_coder.py:319-330 + clip, bounding_box.py:261-265).
//
// One thread per location, consecutive lanes = consecutive w: the 4 bilinear taps of neighbouring lanes
// fall in the same or adjacent cache lines of one correlation channel (the transforms vary smoothly), so
// each of the 121x4 gathers is a near-coalesced L2 read of the 4.3 MB per-class correlation block.
#include "os2d_common.h"
#include "sample_decode.h"

namespace {

__global__ __launch_bounds__(256) void sample_decode_kernel(const float* __restrict__ corr,    // [NB][225][HW]
                                                            const float* __restrict__ params,  // [NB][P][HW]
                                                            int H, int W, int P, int inverse, float stride,
                                                            float half_box, int Bc, int Btot, int b0,
                                                            float* __restrict__ loc, float* __restrict__ cls,
                                                            float* __restrict__ corners,
                                                            const int* __restrict__ flags /* [A + 1] or NULL */, int A, int epoch,
                                                            int* __restrict__ host_status /* or NULL */) {
  const int HW = H * W;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = blockIdx.y;  // index inside the class chunk: a*Bc + b_local
  if (n >= HW) return;
  // output slot in the full [A,Btot,...] tensors
  const int img = nb / Bc;
  const size_t ob = (size_t)img * Btot + b0 + (nb - img * Bc);
  const int h = n / W, w = n - h * W;
  // Non-finite input (reference head.py:339, 650: torch.relu / norm propagate a NaN; the fmaxf ReLUs and the fp16 splits of the
  // split-fp16 kernels do not): an earlier kernel of THIS call raised the range word of the image (split_fm_kernel: a non-finite
  // feature) or of the whole call (an activation outside the fp16 range) by storing the call's epoch there.  The outputs of such
  // an image are NaN - all of them, where the reference has NaN in the neighbourhood of the offending cells - in the call that
  // has the bad input, without a host synchronisation; the sticky host word is raised for ``Os2dHead.range_status``.  Two
  // scalar loads per work-group (the image index is uniform).
  if (flags != nullptr && (flags[img] == epoch || flags[A] == epoch)) {
    os2d_sample_decode_poison(HW, ob, n, loc, cls, corners);
    if (host_status != nullptr && threadIdx.x == 0 && blockIdx.x == 0)
      __hip_atomic_store(host_status, OS2D_STATUS_F16_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }

  os2d_sample_decode_location(corr + (size_t)nb * OS2D_K * HW, params + (size_t)nb * P * HW + n, HW, H, W, P, inverse, stride, half_box, ob, n,
                              h, w, loc, cls, corners);
}

// per-location decode: loc [NB][4][HW] -> boxes [NB][HW][4] xyxy clipped to the level image
__global__ __launch_bounds__(256) void decode_boxes_kernel(const float* __restrict__ loc, int H, int W, float stride,
                                                           float half_box, float img_w, float img_h,
                                                           float* __restrict__ boxes) {
  const int HW = H * W;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = blockIdx.y;
  if (n >= HW) return;
  const float4 o = os2d_decode_box(loc + (size_t)nb * 4 * HW + n, HW, n, W, stride, half_box, img_w, img_h);
  reinterpret_cast<float4*>(boxes)[(size_t)nb * HW + n] = o;
}


// Os2dAlignment.forward as the reference returns it (head.py:155-193): the transformed 15x15 template grid of every
// location in LOCAL coordinates, grids [NB][H][W][15][15][2] (x, y in [-1,1]), i.e. F.affine_grid(theta,
// align_corners=True) - and / or the prepared theta [NB*H*W][2][3] (prepare_transform_parameters_for_grid_sampler,
// head.py:81-153).  The fused head never materialises either; this kernel exists for callers of the class API.
__global__ __launch_bounds__(256) void alignment_grids_kernel(const float* __restrict__ params, int HW, int P, int inverse,
                                                              float* __restrict__ theta, float* __restrict__ grids) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = blockIdx.y;
  if (n >= HW) return;
  float t00, t01, t02, t10, t11, t12;
  os2d_theta(params + (size_t)nb * P * HW + n, HW, P, inverse, t00, t01, t02, t10, t11, t12);
  const size_t loc = (size_t)nb * HW + n;
  if (theta) {
    float* o = theta + loc * 6;
    o[0] = t00, o[1] = t01, o[2] = t02, o[3] = t10, o[4] = t11, o[5] = t12;
  }
  if (grids) {
    float2* g = reinterpret_cast<float2*>(grids) + loc * (OS2D_T * OS2D_T);
    for (int i = 0; i < OS2D_T; ++i) {
      const float yi = os2d_template_coord(i);
#pragma unroll
      for (int j = 0; j < OS2D_T; ++j) {
        const float xj = os2d_template_coord(j);
        g[i * OS2D_T + j] = make_float2(t00 * xj + t01 * yi + t02, t10 * xj + t11 * yi + t12);
      }
    }
  }
}
}  // namespace

int os2d_launch_sample_decode(const float* corr, const float* params, int NB, int H, int W, int P, int inverse,
                              int stride, int rec_field, int Bc, int Btot, int b0, float* loc, float* cls,
                              float* corners, const int* flags, int epoch, int* host_status, hipStream_t stream) {
  const float half_box = 0.5f * (float)(stride * (OS2D_T - 1) + rec_field);  // head.py:236-237: 16*14+16 = 240
  dim3 grid((H * W + 255) / 256, NB);
  hipLaunchKernelGGL(sample_decode_kernel, grid, dim3(256), 0, stream, corr, params, H, W, P, inverse, (float)stride,
                     half_box, Bc, Btot, b0, loc, cls, corners, flags, NB / Bc, epoch, host_status);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("sample_decode launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

int os2d_launch_decode_boxes(const float* loc, int NB, int H, int W, int stride, int rec_field, float img_w,
                             float img_h, float* boxes, hipStream_t stream) {
  const float half_box = 0.5f * (float)(stride * (OS2D_T - 1) + rec_field);
  dim3 grid((H * W + 255) / 256, NB);
  hipLaunchKernelGGL(decode_boxes_kernel, grid, dim3(256), 0, stream, loc, H, W, (float)stride, half_box, img_w,
                     img_h, boxes);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("decode_boxes launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

int os2d_launch_alignment_grids(const float* params, int NB, int H, int W, int P, int inverse, float* theta,
                                float* grids, hipStream_t stream) {
  dim3 grid((H * W + 255) / 256, NB);
  hipLaunchKernelGGL(alignment_grids_kernel, grid, dim3(256), 0, stream, params, H * W, P, inverse, theta, grids);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("alignment_grids launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

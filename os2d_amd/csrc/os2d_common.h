// Shared definitions for the gfx950 (CDNA4) OS2D head kernels.
//
// Padded-plane geometry used by every TransformNet activation buffer:
//   a feature map H x W is stored as (H+2*PAD) rows of WS = W+2*PAD floats with zero borders, PAD = 3
//   (the largest conv radius, 7x7).  A plane is PLANE = round_up((H+2*PAD)*WS, 64) floats, so that every
//   plane and every tile origin is 16-byte aligned.  With zero borders baked into the layout a KSxKS
//   convolution is a pure shift-and-accumulate over the FLAT plane index n = h'*WS + w':
//       out[o][n] = sum_{c,dy,dx} w[o][c][dy][dx] * in[c][n + (dy-R)*WS + (dx-R)],   R = KS/2
//   which is what lets the implicit-GEMM kernel read its B operand from LDS with one runtime row offset
//   and compile-time immediates for dx and the 32-column blocks.  Outputs at border cells are computed
//   as garbage by the MFMAs and replaced by 0 in the epilogue (7.5 % extra columns at 60x80).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OS2D_PAD 3
#define OS2D_T 15            // template side (reference head.py:66-69)
#define OS2D_K 225           // T*T correlation channels
#define OS2D_KP 226          // padded to an even channel count (MFMA 32x32x2 consumes channel pairs)
#define OS2D_QROWS 256       // correlation GEMM M tile: 225 rows padded with zeros

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline __host__ __device__ int os2d_ws(int W) { return W + 2 * OS2D_PAD; }
static inline __host__ __device__ int os2d_hp(int H) { return H + 2 * OS2D_PAD; }
static inline __host__ __device__ int os2d_plane(int H, int W) {
  return ((os2d_hp(H) * os2d_ws(W) + 63) / 64) * 64;
}
static inline __host__ __device__ int os2d_round_up(int x, int m) { return ((x + m - 1) / m) * m; }

// error plumbing (abi.hip)
void os2d_set_error(const char* fmt, ...);

// ---- launchers implemented by the kernel translation units (all asynchronous on `stream`) ----
// prep.hip
int os2d_launch_fm_sumsq(const float* fm, float* sumsq, int A, int C, int HW, hipStream_t stream);
int os2d_launch_border_zero(float* rpad, int planes_total, int H, int W, hipStream_t stream);
int os2d_launch_corr_normalize(const float* corr, float* rpad, int NB, int H, int W, hipStream_t stream);
int os2d_launch_class_prepare(const float* src, int C, int h, int w, int normalize, float* q15, float* qp, hipStream_t stream);
int os2d_launch_pack_conv(const float* w, const float* b, const float* bn_w, const float* bn_b,
                          const float* bn_mean, const float* bn_var, float bn_eps, int Cout, int Cin, int KS,
                          int MT, float* wp, float* bp, hipStream_t stream);
// corr_mfma.hip
int os2d_launch_corr(const float* fm, const float* qp, const float* sumsq, float* corr, float* rpad,
                     int A, int B, int C, int H, int W, hipStream_t stream);
// conv_mfma.hip
int os2d_launch_conv(int layer, const float* in, const float* wp, const float* bp, float* out,
                     int NB, int P, int H, int W, hipStream_t stream);
// sample_decode.hip
int os2d_launch_sample_decode(const float* corr, const float* params, int NB, int H, int W, int P, int inverse,
                              int stride, int rec_field, int Bc, int Btot, int b0, float* loc, float* cls,
                              float* corners, hipStream_t stream);
int os2d_launch_decode_boxes(const float* loc, int NB, int H, int W, int stride, int rec_field, float img_w,
                             float img_h, float* boxes, hipStream_t stream);
// nms.hip
int os2d_launch_nms(const float* boxes, const int* counts, int NC, int N, float thr, unsigned char* keep, int* num_keep,
                    void* workspace, hipStream_t stream);

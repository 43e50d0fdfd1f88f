// Shared definitions for the gfx950 (CDNA4) OS2D head kernels.
//
// Zero-bordered plane layout used by every TransformNet activation buffer.  A feature map H x W is stored as
// rows of WS = W + PAD floats (PAD = 3 = the largest conv radius): W data cells followed by PAD zero cells.  The
// pad cells of row h double as the LEFT border of row h+1, so one gap of 3 zeros between rows serves both sides.
// The map starts at flat offset BASE = round_up(PAD*WS + PAD, 4) (3 zero rows + 3 cells above it) and is followed
// by the same amount of zeros;  PLANE = round_up(BASE + (H+PAD)*WS + PAD, 64) floats per channel, so every plane
// and every tile origin is 16-byte aligned.  cell(h, w) = BASE + h*WS + w.
// With the zeros baked in, a KSxKS convolution is a pure shift-and-accumulate over the FLAT index n:
//       out[o][n] = sum_{c,dy,dx} w[o][c][dy][dx] * in[c][n + (dy-R)*WS + (dx-R)],   R = KS/2
// which lets the implicit-GEMM kernel read its B operand from LDS with one runtime row offset per kernel row and
// compile-time immediates for dx and the 32-column blocks.  Outputs at pad cells are computed as garbage by the
// MFMAs and replaced by exact zeros in the epilogue (3/80 = 3.75 % extra columns at 60x80).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OS2D_PAD 3
#define OS2D_T 15            // template side (reference head.py:66-69)
#define OS2D_K 225           // T*T correlation channels
#define OS2D_KP 226          // padded to an even channel count (MFMA 32x32x2 consumes channel pairs)
#define OS2D_QROWS 256       // correlation GEMM M tile: 225 rows padded with zeros
#define OS2D_MAX_W_DIRECT7 209   // widest map of the DIRECT 7x7 kernels: 256 + 2*(3*(W+3)+3) slab units must fit their prefetch (<= 1536)
#define OS2D_MAX_W_LINEAR5 316   // widest map of the 5x5 kernels' LINEAR slabs, 256 + 2*(2*(W+3)+2) <= 1536 units; wider maps are cut into column
                                 // strips (conv_f16x3.hip: STRIP mode); beyond OS2D_MAX_W_DIRECT7 the 7x7 layer runs in the frequency domain
                                 // (tiled) whatever the batch
#define OS2D_MAX_W 3600      // widest feature map of the head (57,600-px images at stride 16): the transform planner cuts an axis into at most
                             // 48 overlap-save tiles of its largest canonical size (dft_mfma.h); the reference has no limit (head.py:619-629)
#define OS2D_MAX_H 2784      // tallest feature map: 48 tiles of 58 rows (window 64 = the largest canonical transform height)
#define OS2D_XSPEC_CPAD 232  // channel stride of the input spectra of the matrix-product transforms: 225 rounded up to the GEMM's k-steps of 8
#define OS2D_G 29            // 8-channel groups of the 225 correlation channels (f16x3 path)
#define OS2D_RNORM_EXP 12    // the relu+L2-normalised correlation (|x| <= 1) is stored as fp16 hi|lo of x * 2^12
#define OS2D_STATUS_F16_RANGE 1  // sticky status bit: a split-fp16 activation left the fp16 range (non-finite input)

// Where a split-fp16 kernel reports an activation outside the fp16 range / a non-finite input, and WHAT it stores there.
// Per-stage entry points: the caller's word, value OS2D_STATUS_F16_RANGE.  os2d_head_forward_ex: a word of the call's
// workspace and the call's EPOCH - sample_decode_kernel, the last kernel of the call, compares the words with the epoch and
// writes NaN into the outputs of a flagged image (round 6): the words are never cleared, so no kernel has to run before the
// ones that raise them and a stale value of an earlier call never matches.
struct Os2dRangeFlag {
  int* word;      // NULL: nowhere
  int value;
};
#ifdef __HIPCC__
// a plain system-scope store (the word may live in mapped host memory, where PCIe atomics are not a given); every writer of a
// call stores the same value, so racing stores are benign
__device__ __forceinline__ void os2d_raise(Os2dRangeFlag f) {
  __hip_atomic_store(f.word, f.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline __host__ __device__ int os2d_round_up(int x, int m) { return ((x + m - 1) / m) * m; }
static inline __host__ __device__ int os2d_ws(int W) { return W + OS2D_PAD; }
static inline __host__ __device__ int os2d_base(int W) { return os2d_round_up(OS2D_PAD * os2d_ws(W) + OS2D_PAD, 4); }
static inline __host__ __device__ int os2d_plane(int H, int W) {
  return os2d_round_up(os2d_base(W) + (H + OS2D_PAD) * os2d_ws(W) + OS2D_PAD, 64);
}
// Column strips of the 5x5 kernels for maps wider than OS2D_MAX_W_LINEAR5 (conv_f16x3.hip): NS strips of SW = ceil(Ws / NS)
// output columns (the strips cover the W data columns and the 3 pad columns of a row), at most 256 each; *SP = SW + 2 R is the
// row pitch of a strip-plane.
static inline void os2d_conv_strips(int W, int R, int* NS, int* SP) {
  const int Ws = os2d_ws(W);
  *NS = (Ws + 255) / 256;
  *SP = (Ws + *NS - 1) / *NS + 2 * R;
}
// Packed correlation kernel (corr_f16x3.hip, STACK): a position's sum of relu(corr)^2 in 2^-44 fixed point (bit 62: a
// non-finite term) -> 1 / (sqrt(s) + 1e-6) (head.py:650, 597); the word is cleared for the next launch.  acc < 2^53: the
// conversion to double is exact, the one to float rounds once.
__device__ __forceinline__ void os2d_corr_norm_finalize_one(unsigned long long* __restrict__ sumfx, float* __restrict__ invn, size_t i) {
  const unsigned long long v = sumfx[i];
  sumfx[i] = 0ull;
  const float s = (v >> 62) ? __builtin_nanf("") : (float)((double)v * 5.6843418860808015e-14);    // 2^-44
  invn[i] = 1.0f / (sqrtf(s) + 1e-6f);
}
// A NON-TEMPORAL store for a streamed result: the correlation tensor (276 MB per 64 classes) and the input spectra (326 MB), written
// once in whole 16-byte-per-lane runs and read by the NEXT kernel only.  Round 6 measured the hint kernel by kernel (64-class step,
// one box, profiles/r06/stages_nontemporal_stores_*.txt): correlation -16 us, forward transform -15 us (the per-bin GEMM that reads
// its spectra runs 266 -> 238 us), both together -35 us (1.537 -> 1.505 ms); the 5x5 layer's and split_fm's stores: no effect;
// the per-bin GEMM's output spectra +17 us, the resampler's 4-byte stores +13 us, the inverse transform's 8-byte half units +47 us
// (1.06 ms at 1024 classes) - those keep plain stores.  The launch gaps did not change (6 / 10 us): the hint does not shorten the
// end-of-kernel write-back, it keeps the streamed lines from displacing what the kernels share through L2.
template <class T>
__device__ __forceinline__ void os2d_stream_store(T* p, const T& v) {
  __builtin_nontemporal_store(v, p);
}
// (The same hint on LOADS of operands a kernel reads once - the per-bin GEMM's input and weight spectra, the forward transform's
// correlation maps, the inverse transform's spectra - moved the 64-class step by -13 .. +5 us and the 1024-class step by -0.06 ..
// +0.2 ms, inside the run-to-run spread: not adopted, profiles/r06/stages_nontemporal_loads.txt.)
// lo halves of the fp16 hi + lo split of two fp32 values whose hi halves are packed in ``hi`` (x0 -> low 16 bits): rn16(x - hi) as
// ONE mixed-precision instruction per value (v_fma_mixlo_f16 / v_fma_mixhi_f16: fma(x, 1.0, -hi) in fp32, rounded to fp16 into
// the low / high half) instead of v_cvt_f32_f16 + v_sub_f32 per value and a v_cvt_pk_f16_f32 per pair - the split conversions
// are a third of the vector work of the transform kernels' staging phases.  Bit-identical to the long form on 4M values incl.
// subnormal lo parts, out-of-range and non-finite inputs (tools/split_mix_check.hip, profiles/r05/split_mix_check.txt).
__device__ __forceinline__ unsigned os2d_split_lo_pair(float x0, float x1, unsigned hi) {
  unsigned lo;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo)
      : "v"(x0), "v"(x1), "v"(hi));
  return lo;
}
// torch.relu of the fp32 kernels (the reference's own arithmetic, head.py:613-650): a NaN stays a NaN - fmaxf(NaN, 0) would return 0
__device__ __forceinline__ float os2d_relu(float v) { return v < 0.f ? 0.f : v; }
// is flat plane index n an interior (data) cell?
static inline __host__ __device__ bool os2d_interior(int n, int H, int W) {
  const int r = n - os2d_base(W);
  return r >= 0 && r < H * os2d_ws(W) && (r % os2d_ws(W)) < W;
}

// Box decode of ONE location (reference os2d/modeling/box_coder.py:319-330 = torchvision BoxCoder.decode_single with
// weights (10,10,5,5) and the dw/dh clamp log(1000/16), then clip_boxes_to_image): shared by decode_boxes_kernel and
// detect_level_kernel so both produce bit-identical boxes.  ``l`` points at loc[nb][0][n]; channel stride HW.
__device__ __forceinline__ float4 os2d_decode_box(const float* __restrict__ l, int HW, int n, int W, float stride,
                                                  float half_box, float img_w, float img_h) {
  const int h = n / W, w = n - h * W;
  const float ecx = stride * ((float)w + 0.5f), ecy = stride * ((float)h + 0.5f);
  const float ax1 = ecx - half_box, ay1 = ecy - half_box;
  const float aw = (ecx + half_box) - ax1, ah = (ecy + half_box) - ay1;
  const float acx = ax1 + 0.5f * aw, acy = ay1 + 0.5f * ah;
  const float clipv = 4.135166556742356f;  // log(1000/16): torchvision BoxCoder.bbox_xform_clip
  const float dx = l[0] / 10.0f, dy = l[HW] / 10.0f;
  const float dw = fminf(l[2 * (size_t)HW] / 5.0f, clipv), dh = fminf(l[3 * (size_t)HW] / 5.0f, clipv);
  const float pcx = dx * aw + acx, pcy = dy * ah + acy;
  const float pw = expf(dw) * aw, ph = expf(dh) * ah;
  float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
  if (img_w > 0.f && img_h > 0.f) {  // clip_boxes_to_image; a non-positive size means "leave unclipped"
    o.x = fminf(fmaxf(o.x, 0.f), img_w);
    o.y = fminf(fmaxf(o.y, 0.f), img_h);
    o.z = fminf(fmaxf(o.z, 0.f), img_w);
    o.w = fminf(fmaxf(o.w, 0.f), img_h);
  }
  return o;
}

// Chain of axis-aligned box transforms that maps a level's boxes into the output image: what the reference's per-level
// ``TransformList`` of closures amounts to (os2d/structures/transforms.py:12-27, built by os2d/data/dataloader.py:286-336 from
// BoxList.resize / transpose / crop, os2d/structures/bounding_box.py:138-226), applied op by op with the reference's
// roundings - every product / difference rounded on its own (no contraction into fused multiply-adds), so the fused decode
// equals the generic chain (which calls the closures on a BoxList) bit for bit.  Kinds: OS2D_BOX_OP_* of include/os2d_hip.h.
#define OS2D_BOX_MAX_OPS 6
#define OS2D_BOX_MAX_DEFAULT_OPS 12
template <int N>
struct Os2dBoxOpsN {
  int n;
  unsigned char kind[N];
  float ax[N], ay[N];
};
typedef Os2dBoxOpsN<OS2D_BOX_MAX_OPS> Os2dBoxOps;
// the anchors ("default_boxes") go through a chain of their own: in the reference they ride along as a BoxList FIELD of the
// boxes - BoxList.transpose / crop also transform such fields, resize does not (bounding_box.py:162,196-199,222-225) - and the
// level's transform is then applied to the field once more (box_coder.py:515-516); the caller records that whole sequence
typedef Os2dBoxOpsN<OS2D_BOX_MAX_DEFAULT_OPS> Os2dDefaultBoxOps;
template <int N>
__device__ __forceinline__ float4 os2d_apply_box_ops(float4 b, const Os2dBoxOpsN<N>& t) {
#pragma clang fp contract(off)
  for (int k = 0; k < t.n; ++k) {
    const float ax = t.ax[k], ay = t.ay[k];
    switch (t.kind[k]) {
      case 1:   // SCALE: BoxList.resize
        b.x = b.x * ax;
        b.y = b.y * ay;
        b.z = b.z * ax;
        b.w = b.w * ay;
        break;
      case 2: { // HFLIP about the image width ax: (xmin, xmax) = (W - xmax, W - xmin)
        const float lo = ax - b.z, hi = ax - b.x;
        b.x = lo;
        b.z = hi;
        break;
      }
      case 3: { // VFLIP about the image height ay
        const float lo = ay - b.w, hi = ay - b.y;
        b.y = lo;
        b.w = hi;
        break;
      }
      case 4:   // SHIFT: BoxList.crop (x - left, y - top)
        b.x = b.x - ax;
        b.y = b.y - ay;
        b.z = b.z - ax;
        b.w = b.w - ay;
        break;
      default:
        break;
    }
  }
  return b;
}
template <int N>
static inline Os2dBoxOpsN<N> os2d_box_ops_scale(float sx, float sy) {
  Os2dBoxOpsN<N> t = {};
  t.n = 1;
  t.kind[0] = 1;
  t.ax[0] = sx;
  t.ay[0] = sy;
  return t;
}
// ops from the ABI arrays (kinds [nops], args [nops][2]); false on a bad chain
template <int N>
static inline bool os2d_box_ops_from(const int* kinds, const float* args, int nops, Os2dBoxOpsN<N>* t) {
  *t = Os2dBoxOpsN<N>{};
  if (nops < 0 || nops > N || (nops > 0 && (!kinds || !args))) return false;
  t->n = nops;
  for (int k = 0; k < nops; ++k) {
    if (kinds[k] < 1 || kinds[k] > 4) return false;
    t->kind[k] = (unsigned char)kinds[k];
    t->ax[k] = args[2 * k];
    t->ay[k] = args[2 * k + 1];
  }
  return true;
}

// IoU(a, b) > thr with torchvision's arithmetic (inter / (area_a + area_b - inter) in fp32, reference
// os2d/structures/bounding_box.py:367 -> torchvision.ops.nms).  The IEEE division (a dozen VALU instructions) is only
// executed when some lane of the wave is within 1e-5 (relative) of the threshold - everywhere else comparing inter with
// thr * union gives the same answer as the rounded quotient.  The vote makes the branch wave-uniform, so it is a real
// branch and not an if-converted select.
// Every product and sum below is rounded on its own, like the reference's tensor expressions: the compiler must NOT contract
// them into fused multiply-adds (hipcc's default for device code is -ffp-contract=fast, and HIP's __fmul_rn / __fadd_rn are
// plain operators that it fuses just the same: area_a + area_b - w * h became two v_fma_f32).  Whether it did depended on
// unrelated code generation choices - the decisions at the threshold flipped when the library was first built without
// packed-FP32 instructions (tests/test_decode_gpu.py::test_nms_decisions_at_the_iou_threshold).
__device__ __forceinline__ float os2d_box_area(float4 b) {
#pragma clang fp contract(off)
  const float bw = b.z - b.x, bh = b.w - b.y;
  return bw * bh;
}

__device__ __forceinline__ bool os2d_iou_gt(float4 a, float area_a, float4 b, float area_b, float thr) {
#pragma clang fp contract(off)
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  const float sum = area_a + area_b;
  const float uni = sum - inter;
  const float tu = thr * uni;
  bool res = inter > tu;
  const bool near = !(uni > 0.f && fabsf(inter - tu) > 1e-5f * fabsf(tu));
  if (__builtin_amdgcn_ballot_w64(near) != 0ull) {
    if (near) res = inter / uni > thr;
  }
  return res;
}

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
int os2d_launch_border_zero_shb(void* rnorm, int NB, int H, int W, hipStream_t stream);
int os2d_launch_border_zero_shb_planes(void* buf, int planes, int H, int W, hipStream_t stream);
// the same launch + the inverse norms of the packed correlation kernel (corr_f16x3.hip): sumfx [n] -> invn [n], sums cleared
int os2d_launch_border_zero_shb_planes_norms(void* buf, int planes, int H, int W, void* sumfx, float* invn, size_t n, hipStream_t stream);
int os2d_launch_pack_conv_f16(const float* w, const float* b, const float* bn_w, const float* bn_b, const float* bn_mean,
                              const float* bn_var, float bn_eps, int Cout, int Cin, int KS, int MT, int steps_padded,
                              const int* wexp, const int* in_exp, const int* out_exp, void* wp, float* bp,
                              hipStream_t stream);
int os2d_class_prepare_partial_floats(int B, int C);
int os2d_launch_class_prepare_batch(const float* const* srcs, const int* sizes, int B, int C, int normalize, float* q15,
                                    float* qp, float* partial, hipStream_t stream);
int os2d_launch_corr_normalize_shb(const float* corr, void* rshb, int NB, int H, int W, hipStream_t stream);
// corr_mfma.hip (shb != 0: rnorm is written in the split-half blocked layout of conv_f16x3.hip)
int os2d_launch_corr(const float* fm, const float* qp, const float* sumsq, float* corr, void* rnorm /*or NULL*/,
                     float* invn /*or NULL: 1 / (norm + eps) per (pair, location) for the frequency-domain 7x7 layer*/,
                     int A, int B, int C, int H, int W, int shb, hipStream_t stream);
// conv_f16x3.hip
int os2d_launch_conv3_f16x3(const void* in, const void* wp, const float* bp, void* out, int NB, int P, int H, int W,
                            hipStream_t stream);   // conv3_f16x3.hip: layer 3 on the 16-row MFMA
int os2d_launch_conv3_sample_decode(const void* in, const void* wp, const float* bp, const float* corr, int NB, int H, int W, int P,
                                    int inverse, int stride, int rec_field, int Bc, int Btot, int b0, float* loc, float* cls, float* corners,
                                    const int* flags, int epoch, int* host_status, hipStream_t stream);   // + the alignment epilogue (head)
int os2d_launch_conv_f16x3(int layer, const void* in, const void* wp, const float* bp, Os2dRangeFlag status, void* out, int NB,
                           int P, int H, int W, int terms, hipStream_t stream);
// conv_mfma.hip
int os2d_launch_conv(int layer, const float* in, const float* wp, const float* bp, float* out,
                     int NB, int P, int H, int W, hipStream_t stream);
// sample_decode.hip
int os2d_launch_sample_decode(const float* corr, const float* params, int NB, int H, int W, int P, int inverse,
                              int stride, int rec_field, int Bc, int Btot, int b0, float* loc, float* cls,
                              float* corners, const int* flags /* [A + 1] range words of the call or NULL */, int epoch,
                              int* host_status /* or NULL */, hipStream_t stream);
int os2d_launch_decode_boxes(const float* loc, int NB, int H, int W, int stride, int rec_field, float img_w,
                             float img_h, float* boxes, hipStream_t stream);
int os2d_launch_alignment_grids(const float* params, int NB, int H, int W, int P, int inverse, float* theta,
                                float* grids, hipStream_t stream);
// nms.hip
int os2d_launch_nms(const float* boxes, const int* counts, int NC, int N, float thr, unsigned char* keep, int* num_keep,
                    void* workspace, hipStream_t stream);
// detect.hip
size_t os2d_detect_level_lds_bytes(int H, int W);
int os2d_launch_detect_level(const float* loc, const float* cls, int B, int H, int W, int stride, int rec_field,
                             float img_w, float img_h, const Os2dBoxOps& ops, float score_thr, float iou_thr,
                             float* out_boxes, float* out_scores, int* out_index, int* out_count, hipStream_t stream);
// fft.hip
int os2d_fft_plan(int H, int W, int* P, int* Q, int* nbins, int* tiles /* [6]: TY, TX, TH, TW, window rows, window columns */);
int os2d_launch_fft_forward(const float* corr, const float* inv, float* X, const float* twQ, const float* twP, int NB, int C,
                            int H, int W, hipStream_t stream);
int os2d_launch_fft_inverse(const float* Y, const float* bp, int MTP, void* out, const float* twQ, const float* twP, int NB,
                            int Cout, int H, int W, Os2dRangeFlag status, int layout /* OS2D_SPECTRA_ROWS | OS2D_SPECTRA_QUADS */,
                            int out_fp32 /* 1: out = fp32 zero-bordered planes [NB][Cout][PLANE] (all-fp32 mode), no scale / split */,
                            hipStream_t stream);
// spectra_pack.hip
int os2d_launch_spectra_pack(const double* wfold, const double* twP64, const double* twQ64, int C, int Cout, int P, int Q,
                             int NBINS, int split, int u_fastest /* bin = v * P + u (dft_mfma.hip) instead of u * V + v */,
                             void* out, void* workspace, hipStream_t stream);
// spectral.hip
size_t os2d_spectral_weight_floats(int C, int Cout, int NBINS);
int os2d_launch_spectral_gemm(const float* wspec, const float* X, float* Y, int NB, int C, int Cout, int NBINS,
                              hipStream_t stream);
// spectral_f16.hip
size_t os2d_spectral_weight16_size(int C, int NBINS);
float os2d_spectral_xscale_for(int H, int W);
int os2d_launch_spectral_gemm_f16(const void* w16, const float* X, float* Y, int NB, int C, int Cout, int NBINS, float xscale,
                                  int x_quads /* 1: X [NBINS/4][NB][Cpad][4] (dft_mfma.hip), 0: X [C][NB][NBINS] */, int Cpad,
                                  hipStream_t stream);
// dft_mfma.hip: the transforms of the frequency-domain 7x7 layer as matrix products on the half-precision matrix cores
int os2d_dft_plan(int H, int W, int* P, int* Q, int* nbins, int* tiles /* [6]: TY, TX, TH, TW, window rows, window columns */);
size_t os2d_dft_matrices_size(int P, int Q);
int os2d_launch_dft_matrices(const double* twP64, const double* twQ64, int P, int Q, void* out, hipStream_t stream);
int os2d_launch_dft_forward(const float* corr, const float* inv, float* X, const void* matrices, int NB, int C, int Cpad, int H, int W,
                            hipStream_t stream);
int os2d_launch_dft_inverse(const float* Y, const float* bp, int MTP, void* out, const void* matrices, int NB, int Cout, int H, int W,
                            Os2dRangeFlag status, int zero_borders, hipStream_t stream);
// corr_f16x3.hip
int os2d_corr_groups(int C);  // 8-channel groups of the split correlation operands, padded to whole K chunks
// clear / clear_words: 64-bit words zeroed by the same launch (the packed correlation kernel's sums; NULL / 0: none)
// status: word[a] is raised for an image a with a non-finite feature (one word per image)
int os2d_launch_split_fm(const float* fm, const float* sumsq, void* fs, int A, int C, int HW, void* clear, size_t clear_words,
                         Os2dRangeFlag status, hipStream_t stream);
int os2d_launch_split_qp(const float* qp, void* qs, int B, int C, hipStream_t stream);
// flags & 1 (packed form only): the sums stay in sumfx; the caller's next launch turns them into invn; & 2: no half tiles at the tail
// (os2d_launch_border_zero_shb_planes_norms) - one launch less on the per-step path
int os2d_launch_corr_f16x3(const void* fs, const void* qs, float* corr, void* rshb, float* invn, void* sumfx, int flags, int A,
                           int B, int C, int H, int W, hipStream_t stream);
int os2d_corr_f16x3_use_packed(int A, int B, int H, int W);     // the head's choice between the two forms (same bits either way)
int os2d_launch_corr_sums_clear(void* sumfx, int A, int B, int H, int W, hipStream_t stream);

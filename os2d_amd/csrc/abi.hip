// C ABI of libos2d_hip.so (include/os2d_hip.h): argument checking, workspace carving, stage sequencing.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/os2d_hip.h"
#include "os2d_common.h"
#include <atomic>
#include <mutex>
#include <random>
#include <vector>

namespace {
thread_local char g_err[512] = {0};

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

struct ConvShape {
  int cout, cin, ks, mt;
};
bool conv_shape(int layer, int P, ConvShape* s) {
  switch (layer) {
    case 1: *s = {128, OS2D_K, 7, 128}; return true;
    case 2: *s = {64, 128, 5, 64}; return true;
    case 3: *s = {P, 64, 5, 32}; return true;
    default: return false;
  }
}
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// k-steps of one channel group of the packed 7x7 weights (49 taps in pairs), = 5 LDS stages of 5 in conv_f16x3.hip
int os2d_conv1_steps_padded() { return 25; }

// workspace carve for a chunk of Bc classes
struct Carve {
  size_t flags, sumsq, fs, corr, rpad, h1, h2, params, invn, sumfx, xspec, yspec, total;
};
Carve carve(int A, int Bc, int C, int H, int W, int P, int fft_bins = 0, int fft_tiles = 1, int xspec_channels = OS2D_K) {
  const size_t HW = (size_t)H * W, PL = os2d_plane(H, W), NB = (size_t)A * Bc;
  Carve c;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off = align_up(off + floats * sizeof(float), 256);
    return o;
  };
  c.flags = take((size_t)A + 1);           // range words of the call: one per image + one for the whole call (Os2dRangeFlag); FIRST
  c.sumsq = take((size_t)A * HW);
  c.fs = take((size_t)A * os2d_corr_groups(C) * 2 * HW * 4);  // f16x3: split image features, 16 B per (group, part, cell)
  c.corr = take(NB * OS2D_K * HW);
  // fp32: 226 planes; f16x3: 29 groups x (hi|lo) x 16 B = 232 floats; not needed by the frequency-domain 7x7 layer
  c.rpad = take(fft_bins > 0 ? 0 : NB * (OS2D_G * 2 * 4) * PL);
  c.h1 = take(NB * 128 * PL);
  c.h2 = take(NB * 64 * PL);
  c.params = take(NB * P * HW);
  c.invn = c.sumfx = c.xspec = c.yspec = 0;
  if (fft_bins > 0) {  // frequency-domain 7x7 layer: inverse norms, input / output spectra (complex64)
    c.invn = take(NB * HW);
    c.sumfx = take(NB * HW * 2);           // 64-bit fixed-point sums of the packed correlation kernel (corr_f16x3.hip, STACK)
    c.xspec = take(NB * fft_tiles * xspec_channels * (size_t)fft_bins * 2);     // a tile of a tiled map is one more "pair" (fft.hip)
    c.yspec = take(NB * fft_tiles * 128 * (size_t)fft_bins * 2);
  }
  c.total = off;
  return c;
}

// ---- debugging aid (tools/diag_pyramid_dump.py): copies of intermediate buffers of os2d_head_forward_ex, per stream.
// Compiled only into DIAGNOSTIC builds (python -m os2d_amd.build --variant dump, with the macro below defined): a registered destination
// is a raw pointer nobody can unregister safely once its tensor is freed, and the table would be consulted by every
// production call.  In the product library os2d_debug_set_dump only reports that it does nothing.
#ifdef OS2D_DIAG_DUMP
struct DumpEntry {
  void* stream;
  int slot;
  void* dst;
  size_t bytes;
};
std::vector<DumpEntry> g_dumps;
std::mutex g_dumps_mutex;               // ctypes releases the GIL: callers may race on the table
bool dumps_active() {
  std::lock_guard<std::mutex> lock(g_dumps_mutex);
  return !g_dumps.empty();
}
void dump_slot(void* stream, int slot, const void* src, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_dumps_mutex);
  for (const DumpEntry& e : g_dumps)
    if (e.stream == stream && e.slot == slot && e.dst)
      (void)hipMemcpyAsync(e.dst, src, bytes < e.bytes ? bytes : e.bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
}
#else
inline bool dumps_active() { return false; }
inline void dump_slot(void*, int, const void*, size_t) {}
#endif

bool head_args_ok(int A, int B, int C, int H, int W, int P) {
  if (A < 1 || B < 1 || C < 1 || H < 1 || W < 1 || (P != 6 && P != 4)) {
    os2d_set_error("bad head shape A=%d B=%d C=%d H=%d W=%d P=%d (need A,B,C,H,W>=1, P in {6,4})", A, B, C, H,
                   W, P);
    return false;
  }
  if (W > OS2D_MAX_W) {
    os2d_set_error("feature map width %d > %d: beyond what the transform planner of the 7x7 layer tiles (images wider than %d px "
                   "at stride 16 are not supported)", W, OS2D_MAX_W, OS2D_MAX_W * 16);
    return false;
  }
  if (H > OS2D_MAX_H) {
    os2d_set_error("feature map height %d > %d: beyond what the transform planner of the 7x7 layer tiles (images taller than %d px "
                   "at stride 16 are not supported)", H, OS2D_MAX_H, OS2D_MAX_H * 16);
    return false;
  }
  return true;
}
// The epoch of a head call: what its kernels store into the range words of the workspace (os2d_common.h: Os2dRangeFlag).  Unique per
// call of the process, never 0 (a zero-filled workspace) and never OS2D_STATUS_F16_RANGE; the base is random so that the stale
// content of a workspace that was NOT zero-filled matches a call's epoch with probability 2^-32.
int next_epoch() {
  static std::atomic<unsigned> counter{[] {
    std::random_device rd;
    return (unsigned)rd() | 0x100u;
  }()};
  unsigned e;
  do e = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
  while (e < 2u);
  return (int)e;
}
bool is_freq(int precision) {
  return precision == OS2D_PRECISION_FFT || precision == OS2D_PRECISION_FFTX3 || precision == OS2D_PRECISION_FFT32;
}
// transform plan of the frequency-domain 7x7 layer: the matrix-product transforms (dft_mfma.hip) under FFTX3, the in-LDS FFTs
// (fft.hip) under FFT / FFT32
int freq_plan(int precision, int H, int W, int* P, int* Q, int* bins, int* tiles) {
  return precision == OS2D_PRECISION_FFTX3 ? os2d_dft_plan(H, W, P, Q, bins, tiles) : os2d_fft_plan(H, W, P, Q, bins, tiles);
}
bool direct7_width_ok(int W) {
  if (W > OS2D_MAX_W_DIRECT7) {
    os2d_set_error("feature map width %d > %d: the direct 7x7 kernels keep 3 halo rows of their input in LDS - wider maps need a "
                   "frequency-domain precision (fftx3 / fft / fft32), which tiles any width up to %d", W, OS2D_MAX_W_DIRECT7, OS2D_MAX_W);
    return false;
  }
  return true;
}
}  // namespace

void os2d_debug_set_dump(void* stream, int slot, void* dst, size_t bytes) {
#ifdef OS2D_DIAG_DUMP
  std::lock_guard<std::mutex> lock(g_dumps_mutex);
  for (size_t i = 0; i < g_dumps.size(); ++i)
    if (g_dumps[i].stream == stream && g_dumps[i].slot == slot) {
      if (dst) {
        g_dumps[i].dst = dst;
        g_dumps[i].bytes = bytes;
      } else {
        g_dumps.erase(g_dumps.begin() + i);          // dst == NULL unregisters the slot
      }
      return;
    }
  if (dst) g_dumps.push_back(DumpEntry{stream, slot, dst, bytes});
#else
  (void)stream, (void)slot, (void)dst, (void)bytes;
  os2d_set_error("os2d_debug_set_dump: this library was built without -DOS2D_DIAG_DUMP (diagnostic builds only)");
#endif
}

void os2d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int os2d_abi_version(void) { return OS2D_ABI_VERSION; }
const char* os2d_last_error(void) { return g_err; }

size_t os2d_packed_conv_floats(int layer) {
  ConvShape s;
  if (!conv_shape(layer, 6, &s)) return 0;
  return (size_t)((s.cin + 1) / 2) * s.ks * s.ks * 2 * s.mt;
}
size_t os2d_packed_bias_floats(int layer) {
  ConvShape s;
  if (!conv_shape(layer, 6, &s)) return 0;
  return (size_t)3 * s.mt;  // folded bias | per-row 2^-weight_exp | per-row 2^out_exp (f16x3 packing; fp32: first third)
}

int os2d_pack_conv(int layer, int P, const float* w, const float* b, const float* bn_weight, const float* bn_bias,
                   const float* bn_running_mean, const float* bn_running_var, float bn_eps, float* packed_w,
                   float* packed_b, void* stream) {
  ConvShape s;
  if (!conv_shape(layer, P, &s) || (layer == 3 && P != 6 && P != 4)) {
    os2d_set_error("os2d_pack_conv: bad layer %d / P %d", layer, P);
    return -1;
  }
  if (!w || !b || !packed_w || !packed_b) {
    os2d_set_error("os2d_pack_conv: null pointer");
    return -1;
  }
  const bool has_bn = bn_weight || bn_bias || bn_running_mean || bn_running_var;
  if (has_bn && !(bn_weight && bn_bias && bn_running_mean && bn_running_var)) {
    os2d_set_error("os2d_pack_conv: BatchNorm needs all four of weight/bias/running_mean/running_var");
    return -1;
  }
  return os2d_launch_pack_conv(w, b, bn_weight, bn_bias, bn_running_mean, bn_running_var, bn_eps, s.cout, s.cin, s.ks,
                               s.mt, packed_w, packed_b, S(stream));
}

int os2d_class_prepare(const float* src, int C, int h, int w, int normalize, float* q15, float* qp, void* stream) {
  if (!src || !q15 || !qp || C < 1 || h < 1 || w < 1) {
    os2d_set_error("os2d_class_prepare: bad arguments (C=%d h=%d w=%d)", C, h, w);
    return -1;
  }
  return os2d_launch_class_prepare(src, C, h, w, normalize, q15, qp, S(stream));
}

size_t os2d_class_prepare_workspace_floats(int B, int C) {
  if (B < 1 || C < 1) return 0;
  return (size_t)os2d_class_prepare_partial_floats(B, C);
}

int os2d_class_prepare_batch(const float* const* srcs, const int* sizes, int B, int C, int normalize, float* q15,
                             float* qp, float* workspace, void* stream) {
  if (!srcs || !sizes || !q15 || !qp || !workspace || B < 1 || C < 1 || B > 65535) {
    os2d_set_error("os2d_class_prepare_batch: bad arguments (B=%d C=%d; at most 65535 classes per call)", B, C);
    return -1;
  }
  return os2d_launch_class_prepare_batch(srcs, sizes, B, C, normalize, q15, qp, workspace, S(stream));
}

size_t os2d_plane_floats(int H, int W) { return (size_t)os2d_plane(H, W); }

size_t os2d_shb_bytes(int channels, int H, int W) {
  if (channels < 1 || H < 1 || W < 1) return 0;
  return (size_t)((channels + 7) / 8) * 2 * os2d_plane(H, W) * 16;
}

int os2d_head_workspace_bytes(int A, int B, int C, int H, int W, int P, size_t* bytes) {
  if (!bytes) {
    os2d_set_error("os2d_head_workspace_bytes: null output");
    return -1;
  }
  if (!head_args_ok(A, B, C, H, W, P)) return -1;
  *bytes = carve(A, B, C, H, W, P).total;
  return 0;
}

int os2d_head_workspace_bytes_ex(int A, int B, int C, int H, int W, int P, int precision, size_t* bytes) {
  if (!bytes) {
    os2d_set_error("os2d_head_workspace_bytes_ex: null output");
    return -1;
  }
  if (!head_args_ok(A, B, C, H, W, P)) return -1;
  int bins = 0, tiles[6] = {1, 1, 0, 0, 0, 0};
  if (is_freq(precision) && !freq_plan(precision, H, W, nullptr, nullptr, &bins, tiles)) {
    os2d_set_error("os2d_head_workspace_bytes_ex: no transform plan for a %dx%d map", H, W);
    return -3;
  }
  *bytes = carve(A, B, C, H, W, P, bins, tiles[0] * tiles[1], precision == OS2D_PRECISION_FFTX3 ? OS2D_XSPEC_CPAD : OS2D_K).total;
  return 0;
}

int os2d_fm_sumsq(const float* fm, float* sumsq, int A, int C, int H, int W, void* stream) {
  if (!fm || !sumsq || A < 1 || C < 1 || H < 1 || W < 1) {
    os2d_set_error("os2d_fm_sumsq: bad arguments");
    return -1;
  }
  return os2d_launch_fm_sumsq(fm, sumsq, A, C, H * W, S(stream));
}

int os2d_corr(const float* fm, const float* qp, const float* sumsq, float* corr, float* rnorm, int A, int B, int C,
              int H, int W, void* stream) {
  if (!fm || !qp || !sumsq || !corr || !rnorm) {
    os2d_set_error("os2d_corr: null pointer");
    return -1;
  }
  if (!head_args_ok(A, B, C, H, W, 6)) return -1;
  int rc = os2d_launch_border_zero(rnorm, A * B * OS2D_KP, H, W, S(stream));
  if (rc) return rc;
  return os2d_launch_corr(fm, qp, sumsq, corr, rnorm, nullptr, A, B, C, H, W, 0, S(stream));
}

size_t os2d_corr_f16x3_workspace_bytes(int A, int C, int H, int W) {
  if (A < 1 || C < 1 || H < 1 || W < 1) return 0;
  return align_up((size_t)A * H * W * sizeof(float), 256) + (size_t)A * os2d_corr_groups(C) * 2 * H * W * 16;
}

int os2d_corr_f16x3(const float* fm, const void* qs, float* corr, void* rshb, int A, int B, int C, int H, int W,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!fm || !qs || !corr || !rshb || !workspace) {
    os2d_set_error("os2d_corr_f16x3: null pointer");
    return -1;
  }
  if (!head_args_ok(A, B, C, H, W, 6)) return -1;
  if (workspace_bytes < os2d_corr_f16x3_workspace_bytes(A, C, H, W) || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
    os2d_set_error("os2d_corr_f16x3: workspace too small or not 256-byte aligned");
    return -2;
  }
  float* sumsq = static_cast<float*>(workspace);
  void* fs = static_cast<char*>(workspace) + align_up((size_t)A * H * W * sizeof(float), 256);
  int rc = os2d_launch_fm_sumsq(fm, sumsq, A, C, H * W, S(stream));
  if (!rc) rc = os2d_launch_split_fm(fm, sumsq, fs, A, C, H * W, nullptr, 0, Os2dRangeFlag{nullptr, 0}, S(stream));
  if (!rc) rc = os2d_launch_border_zero_shb(rshb, A * B, H, W, S(stream));
  if (!rc) rc = os2d_launch_corr_f16x3(fs, qs, corr, rshb, nullptr, nullptr, 0, A, B, C, H, W, S(stream));
  return rc;
}

size_t os2d_corr_f16x3_packed_workspace_bytes(int A, int B, int C, int H, int W) {
  if (A < 1 || B < 1 || C < 1 || H < 1 || W < 1) return 0;
  return align_up(os2d_corr_f16x3_workspace_bytes(A, C, H, W), 256) + (size_t)A * B * H * W * sizeof(unsigned long long);
}

int os2d_corr_f16x3_packed(const float* fm, const void* qs, float* corr, float* inv_norm, int A, int B, int C, int H, int W,
                           int form, void* workspace, size_t workspace_bytes, void* stream) {
  if (!fm || !qs || !corr || !inv_norm || !workspace) {
    os2d_set_error("os2d_corr_f16x3_packed: null pointer");
    return -1;
  }
  if (!head_args_ok(A, B, C, H, W, 6)) return -1;
  if (workspace_bytes < os2d_corr_f16x3_packed_workspace_bytes(A, B, C, H, W) || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
    os2d_set_error("os2d_corr_f16x3_packed: workspace too small or not 256-byte aligned");
    return -2;
  }
  float* sumsq = static_cast<float*>(workspace);
  void* fs = static_cast<char*>(workspace) + align_up((size_t)A * H * W * sizeof(float), 256);
  void* sumfx = static_cast<char*>(workspace) + align_up(os2d_corr_f16x3_workspace_bytes(A, C, H, W), 256);
  int rc = os2d_launch_fm_sumsq(fm, sumsq, A, C, H * W, S(stream));
  if (!rc) rc = os2d_launch_split_fm(fm, sumsq, fs, A, C, H * W, nullptr, 0, Os2dRangeFlag{nullptr, 0}, S(stream));
  if (form < -1 || form > 7 || (form >= 0 && (form & 3) > 1)) {
    os2d_set_error("os2d_corr_f16x3_packed: form %d (0 padded | 1 packed | -1 the head's choice; + 4: no half tiles at the tail)", form);
    return -1;
  }
  const bool packed = form < 0 ? os2d_corr_f16x3_use_packed(A, B, H, W) != 0 : (form & 1) != 0;
  if (!rc && packed) rc = os2d_launch_corr_sums_clear(sumfx, A, B, H, W, S(stream));
  if (!rc) rc = os2d_launch_corr_f16x3(fs, qs, corr, nullptr, inv_norm, packed ? sumfx : nullptr, (form >= 0 && (form & 4)) ? 2 : 0, A, B, C, H, W, S(stream));
  return rc;
}

int os2d_corr_normalize(const float* corr, float* rnorm, int NB, int H, int W, void* stream) {
  if (!corr || !rnorm || NB < 1 || H < 1 || W < 1) {
    os2d_set_error("os2d_corr_normalize: bad arguments");
    return -1;
  }
  return os2d_launch_corr_normalize(corr, rnorm, NB, H, W, S(stream));
}

int os2d_transform_conv(int layer, const float* in, const float* packed_w, const float* packed_b, float* out, int NB,
                        int P, int H, int W, void* stream) {
  if (!in || !packed_w || !packed_b || !out || NB < 1 || H < 1 || W < 1 || (layer == 3 && P != 6 && P != 4)) {
    os2d_set_error("os2d_transform_conv: bad arguments (layer=%d NB=%d P=%d)", layer, NB, P);
    return -1;
  }
  return os2d_launch_conv(layer, in, packed_w, packed_b, out, NB, P, H, W, S(stream));
}

int os2d_sample_decode(const float* corr, const float* params, int NB, int H, int W, int P, int inverse, int stride,
                       int rec_field, float* loc, float* cls, float* corners, void* stream) {
  if (!corr || !params || !loc || !cls || !corners || NB < 1 || H < 1 || W < 1 || (P != 6 && P != 4)) {
    os2d_set_error("os2d_sample_decode: bad arguments");
    return -1;
  }
  return os2d_launch_sample_decode(corr, params, NB, H, W, P, inverse, stride, rec_field, NB, NB, 0, loc, cls,
                                   corners, nullptr, 0, nullptr, S(stream));
}

int os2d_decode_boxes(const float* loc, int NB, int H, int W, int stride, int rec_field, float img_w, float img_h,
                      float* boxes, void* stream) {
  if (!loc || !boxes || NB < 1 || H < 1 || W < 1) {
    os2d_set_error("os2d_decode_boxes: bad arguments");
    return -1;
  }
  return os2d_launch_decode_boxes(loc, NB, H, W, stride, rec_field, img_w, img_h, boxes, S(stream));
}

int os2d_detect_level_supported(int H, int W) { return os2d_detect_level_lds_bytes(H, W) != 0 ? 1 : 0; }

int os2d_detect_level(const float* loc, const float* cls, int B, int H, int W, int stride, int rec_field, float img_w,
                      float img_h, float scale_x, float scale_y, float score_threshold, float iou_threshold,
                      float* out_boxes, float* out_scores, int* out_index, int* out_count, void* stream) {
  if (!loc || !cls || !out_boxes || !out_scores || !out_index || !out_count || B < 1 || H < 1 || W < 1 || stride < 1 ||
      rec_field < 1) {
    os2d_set_error("os2d_detect_level: bad arguments");
    return -1;
  }
  return os2d_launch_detect_level(loc, cls, B, H, W, stride, rec_field, img_w, img_h, os2d_box_ops_scale<OS2D_BOX_MAX_OPS>(scale_x, scale_y),
                                  score_threshold, iou_threshold, out_boxes, out_scores, out_index, out_count, S(stream));
}

int os2d_detect_level_ops(const float* loc, const float* cls, int B, int H, int W, int stride, int rec_field, float img_w,
                          float img_h, int op_count, const int* op_kinds, const float* op_args, float score_threshold,
                          float iou_threshold, float* out_boxes, float* out_scores, int* out_index, int* out_count,
                          void* stream) {
  if (!loc || !cls || !out_boxes || !out_scores || !out_index || !out_count || B < 1 || H < 1 || W < 1 || stride < 1 ||
      rec_field < 1) {
    os2d_set_error("os2d_detect_level_ops: bad arguments");
    return -1;
  }
  Os2dBoxOps ops;
  if (!os2d_box_ops_from(op_kinds, op_args, op_count, &ops)) {
    os2d_set_error("os2d_detect_level_ops: bad transform chain (at most %d ops of kind 1..4)", OS2D_BOX_MAX_OPS);
    return -1;
  }
  return os2d_launch_detect_level(loc, cls, B, H, W, stride, rec_field, img_w, img_h, ops, score_threshold, iou_threshold,
                                  out_boxes, out_scores, out_index, out_count, S(stream));
}

int os2d_nms_workspace_bytes(int NC, int N, size_t* bytes) {
  if (!bytes || NC < 1 || N < 1) {
    os2d_set_error("os2d_nms_workspace_bytes: bad arguments");
    return -1;
  }
  *bytes = (size_t)NC * N * 4 * sizeof(float);
  return 0;
}

int os2d_nms(const float* boxes, const int* counts, int NC, int N, float iou_threshold, unsigned char* keep,
             int* num_keep, void* workspace, size_t workspace_bytes, void* stream) {
  if (!boxes || !counts || !keep || !num_keep || !workspace || NC < 1 || N < 1) {
    os2d_set_error("os2d_nms: bad arguments");
    return -1;
  }
  if ((reinterpret_cast<uintptr_t>(boxes) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
    os2d_set_error("os2d_nms: boxes and workspace must be 16-byte aligned");
    return -1;
  }
  if (workspace_bytes < (size_t)NC * N * 4 * sizeof(float)) {
    os2d_set_error("os2d_nms: workspace too small");
    return -2;
  }
  return os2d_launch_nms(boxes, counts, NC, N, iou_threshold, keep, num_keep, workspace, S(stream));
}

int os2d_head_forward_ex(const float* fm, const float* qp, const void* w1, const float* b1, const void* w2,
                         const float* b2, const void* w3, const float* b3, int A, int B, int C, int H, int W, int P,
                         int inverse, int stride, int rec_field, float* loc, float* cls, float* corners,
                         void* workspace, size_t workspace_bytes, void* stream, int precision, const void* qs,
                         void** stage_events, int* chunk_classes, int* status, const float* wspec, const float* twQ,
                         const float* twP) {
  if (!fm || !qp || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !loc || !cls || !corners || !workspace) {
    os2d_set_error("os2d_head_forward: null pointer");
    return -1;
  }
  if (precision != OS2D_PRECISION_F32 && precision != OS2D_PRECISION_F16X3 && precision != OS2D_PRECISION_F16X2 &&
      precision != OS2D_PRECISION_FFT && precision != OS2D_PRECISION_FFTX3 && precision != OS2D_PRECISION_FFT32) {
    os2d_set_error("os2d_head_forward: unknown precision %d", precision);
    return -1;
  }
  int fft_bins = 0, tiles[6] = {1, 1, 0, 0, 0, 0};
  const bool dft = precision == OS2D_PRECISION_FFTX3;           // transforms as matrix products (dft_mfma.hip): twQ = the matrices
  if (is_freq(precision)) {
    if (!wspec || !twQ || (!dft && !twP)) {
      os2d_set_error("os2d_head_forward: the frequency-domain modes need the weight spectra and the transform tables "
                     "(fft / fft32: twQ, twP; fftx3: the matrices of os2d_dft_matrices_build as twQ)");
      return -1;
    }
    if (!freq_plan(precision, H, W, nullptr, nullptr, &fft_bins, tiles)) {
      os2d_set_error("os2d_head_forward: no transform plan for a %dx%d map", H, W);
      return -3;
    }
  } else if (!direct7_width_ok(W)) {
    return -1;
  }
  const int fft_T = tiles[0] * tiles[1], xch = dft ? OS2D_XSPEC_CPAD : OS2D_K;
  const bool fp32_ops = precision == OS2D_PRECISION_F32 || precision == OS2D_PRECISION_FFT32;   // fp32 MFMA correlation / 5x5 layers
  if (!fp32_ops && !qs) {
    os2d_set_error("os2d_head_forward: precision f16x3 / f16x2 needs the split class operand (os2d_class_split)");
    return -1;
  }
  if (!head_args_ok(A, B, C, H, W, P)) return -1;
  if (stride < 1 || rec_field < 1) {
    os2d_set_error("os2d_head_forward: bad stride/rec_field %d/%d", stride, rec_field);
    return -1;
  }
  if (reinterpret_cast<uintptr_t>(workspace) & 255) {
    os2d_set_error("os2d_head_forward: workspace must be 256-byte aligned");
    return -1;
  }
  // largest class chunk that fits the workspace (footprint is affine in Bc)
  const size_t one = carve(A, 1, C, H, W, P, fft_bins, fft_T, xch).total;
  if (workspace_bytes < one) {
    os2d_set_error("os2d_head_forward: workspace too small (%zu B, need >= %zu B for one class)", workspace_bytes, one);
    return -2;
  }
  int Bc = B;
  while (Bc > 1 && carve(A, Bc, C, H, W, P, fft_bins, fft_T, xch).total > workspace_bytes) {
    const size_t two = carve(A, 2, C, H, W, P, fft_bins, fft_T, xch).total;
    const size_t per = two - one;
    int guess = per ? (int)((workspace_bytes - one) / per) + 1 : 1;
    if (guess >= Bc) guess = Bc - 1;
    if (guess < 1) guess = 1;
    Bc = guess;
  }
  hipStream_t st = S(stream);
  char* ws = static_cast<char*>(workspace);
  const Carve c = carve(A, Bc, C, H, W, P, fft_bins, fft_T, xch);
  float* invn = fft_bins ? reinterpret_cast<float*>(ws + c.invn) : nullptr;
  // half-precision correlation on the frequency-domain route: classes packed along M (no padded rows 225 .. 255 per class)
  // - when that saves a round of the chip (os2d_corr_f16x3_use_packed: both forms give the same bits; decided on the chunk size
  // so that every chunk but a shorter last one takes the same form)
  void* sumfx = (fft_bins && !fp32_ops && os2d_corr_f16x3_use_packed(A, Bc, H, W)) ? ws + c.sumfx : nullptr;
  float* xspec = fft_bins ? reinterpret_cast<float*>(ws + c.xspec) : nullptr;
  float* yspec = fft_bins ? reinterpret_cast<float*>(ws + c.yspec) : nullptr;
  float* sumsq = reinterpret_cast<float*>(ws + c.sumsq);
  void* fsplit = ws + c.fs;
  float* corr = reinterpret_cast<float*>(ws + c.corr);
  float* rpad = reinterpret_cast<float*>(ws + c.rpad);
  float* h1 = reinterpret_cast<float*>(ws + c.h1);
  float* h2 = reinterpret_cast<float*>(ws + c.h2);
  float* params = reinterpret_cast<float*>(ws + c.params);
  // range words (one per image, one for the whole call) and the value this call's kernels store there; the last kernel of every class
  // chunk - the resampler - turns a raised word into NaN outputs and raises the caller's sticky host word
  int* flags = reinterpret_cast<int*>(ws + c.flags);
  const int epoch = next_epoch();
  const Os2dRangeFlag per_image = {fp32_ops ? nullptr : flags, epoch}, whole_call = {fp32_ops ? nullptr : flags + A, epoch};

  if (chunk_classes) *chunk_classes = Bc;
  // optional per-stage events (first class chunk only): stage_events[2*s] / [2*s+1] bracket stage s
  auto mark = [&](int b0, int idx) {
    if (stage_events && b0 == 0 && stage_events[idx]) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(stage_events[idx]), st);
  };
  int rc = os2d_launch_fm_sumsq(fm, sumsq, A, C, H * W, st);
  if (rc) return rc;
  // (the packed correlation kernel's sums are cleared by the same launch; every chunk's norms pass leaves them cleared again)
  if (!fp32_ops && (rc = os2d_launch_split_fm(fm, sumsq, fsplit, A, C, H * W, sumfx, sumfx ? (size_t)A * Bc * H * W : 0, per_image, st))) return rc;
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int bc = (B - b0 < Bc) ? (B - b0) : Bc;
    const int NB = A * bc;
    const bool f16 = !fp32_ops;
    const int terms1 = precision == OS2D_PRECISION_F16X2 ? 2 : 3;  // 7x7 layer: weights as fp16 roundings only under f16x2
    mark(b0, 0);
    if (f16) {
      // the frequency-domain 7x7 layer takes corr + invn; the split / blocked copy of the normalised maps is not written
      if (!fft_bins && (rc = os2d_launch_border_zero_shb(rpad, NB, H, W, st))) return rc;
    } else if (!fft_bins) {
      if ((rc = os2d_launch_border_zero(rpad, NB * OS2D_KP, H, W, st))) return rc;
    }
    if (f16) {
      const char* qsb = static_cast<const char*>(qs) + (size_t)b0 * os2d_corr_groups(C) * 2 * 256 * 16;
      // packed form: the sums become inverse norms in the border launch below (before the forward transform reads them)
      if ((rc = os2d_launch_corr_f16x3(fsplit, qsb, corr, fft_bins ? nullptr : rpad, invn, sumfx, 1, A, bc, C, H, W, st))) return rc;
    } else {
      if ((rc = os2d_launch_corr(fm, qp + (size_t)b0 * C * OS2D_QROWS, sumsq, corr, fft_bins ? nullptr : rpad, invn, A, bc, C, H, W,
                                 0, st)))
        return rc;
    }
    mark(b0, 1);
    mark(b0, 2);
    if (fft_bins) {
      bool inv_borders = false;
      // the 7x7 layer in the frequency domain (fft.hip, spectral.hip): fp32 FFT of relu(corr) / norm -> one complex GEMM
      // per bin on the fp32 matrix cores -> inverse FFT + bias + ReLU + split into the activation buffer of the 5x5 layer
      if (f16) {
        // the matrix-product inverse transform writes the zero borders of the planes it fills; what is left for this launch on
        // that route is the norms pass of the packed correlation (none for the padded form: no launch at all)
        // ($OS2D_BORDERS_IN_INVERSE=0: the separate launch as before, for measurements)
        static const bool borders_in_inverse = [] {
          const char* e = getenv("OS2D_BORDERS_IN_INVERSE");
          return !(e && e[0] == '0');
        }();
        inv_borders = dft && borders_in_inverse;
        const int planes = inv_borders ? 0 : NB * 16 * 2;
        if ((planes || sumfx) &&
            (rc = os2d_launch_border_zero_shb_planes_norms(h1, planes, H, W, sumfx, invn, (size_t)NB * H * W, st)))
          return rc;
      } else if ((rc = os2d_launch_border_zero(h1, NB * 128, H, W, st))) {     // all-fp32 mode: fp32 planes for the fp32 5x5 kernel
        return rc;
      }
      mark(b0, 10);
      // the split-half GEMM writes its output spectra in quads of bins (include/os2d_hip.h, OS2D_SPECTRA_QUADS)
      const int layout = precision == OS2D_PRECISION_FFTX3 ? OS2D_SPECTRA_QUADS : OS2D_SPECTRA_ROWS;
      if (dft) {
        // the transforms as matrix products on the half-precision matrix cores, spectra in quads of bins on both sides of the
        // per-bin GEMM (dft_mfma.hip); |X| <= number of samples of a window (every sample of the normalised maps is <= 1)
        const void* mats = twQ;
        if ((rc = os2d_launch_dft_forward(corr, invn, xspec, mats, NB, OS2D_K, xch, H, W, st))) return rc;
        mark(b0, 11);
        if ((rc = os2d_launch_spectral_gemm_f16(wspec, xspec, yspec, NB * fft_T, OS2D_K, 128, fft_bins,
                                                os2d_spectral_xscale_for(tiles[4], tiles[5]), 1, xch, st)))
          return rc;
        mark(b0, 12);
        if ((rc = os2d_launch_dft_inverse(yspec, b1, 128, h1, mats, NB, 128, H, W, whole_call, inv_borders ? 1 : 0, st))) return rc;
      } else {
        if ((rc = os2d_launch_fft_forward(corr, invn, xspec, twQ, twP, NB, OS2D_K, H, W, st))) return rc;
        mark(b0, 11);
        if ((rc = os2d_launch_spectral_gemm(wspec, xspec, yspec, NB * fft_T, OS2D_K, 128, fft_bins, st))) return rc;
        mark(b0, 12);
        if ((rc = os2d_launch_fft_inverse(yspec, b1, 128, h1, twQ, twP, NB, 128, H, W, whole_call, layout, f16 ? 0 : 1, st))) return rc;
      }
    } else if (f16) {
      if ((rc = os2d_launch_conv_f16x3(1, rpad, w1, b1, whole_call, h1, NB, P, H, W, terms1, st))) return rc;
    } else {
      if ((rc = os2d_launch_conv(1, rpad, static_cast<const float*>(w1), b1, h1, NB, P, H, W, st))) return rc;
    }
    mark(b0, 3);
    mark(b0, 4);
    if (dumps_active() && b0 == 0) {   // slots: 0 corr, 1 inverse norms, 2 input spectra, 3 output spectra, 4 h1, 5 h2, 6 params
      const size_t PLb = os2d_plane(H, W);
      dump_slot(stream, 0, corr, (size_t)NB * OS2D_K * H * W * 4);
      if (fft_bins) {
        dump_slot(stream, 1, invn, (size_t)NB * H * W * 4);
        dump_slot(stream, 2, xspec, (size_t)NB * fft_T * OS2D_K * fft_bins * 8);
        dump_slot(stream, 3, yspec, (size_t)NB * fft_T * 128 * fft_bins * 8);
      }
      dump_slot(stream, 4, h1, (size_t)NB * 128 * PLb * 4);
    }
    if (f16) {
      if ((rc = os2d_launch_conv_f16x3(2, h1, w2, b2, whole_call, h2, NB, P, H, W, 3, st))) return rc;
    } else {
      if ((rc = os2d_launch_conv(2, h1, static_cast<const float*>(w2), b2, h2, NB, P, H, W, st))) return rc;
    }
    mark(b0, 5);
    mark(b0, 6);
    if (dumps_active() && b0 == 0) dump_slot(stream, 5, h2, (size_t)NB * 64 * os2d_plane(H, W) * 4);
    // split-fp16 route: the last layer and the alignment epilogue in ONE launch (conv3_f16x3.hip, FUSE: the parameters go from the
    // accumulators through LDS to the resampler, never to HBM).  $OS2D_FUSED_TAIL=0: the two launches of rounds 1 - 5 (measurements;
    // same bits).  Diagnostic dumps of the parameters need the separate launches.
    static const bool fused_tail_env = [] {
      const char* e = getenv("OS2D_FUSED_TAIL");
      return !(e && e[0] == '0');
    }();
    const bool fused_tail = f16 && fused_tail_env && !dumps_active();
    if (fused_tail) {
      if ((rc = os2d_launch_conv3_sample_decode(h2, w3, b3, corr, NB, H, W, P, inverse, stride, rec_field, bc, B, b0, loc, cls, corners,
                                                flags, epoch, status, st)))
        return rc;
      mark(b0, 7);
      mark(b0, 8);
      mark(b0, 9);
      continue;
    }
    if (f16) {
      if ((rc = os2d_launch_conv_f16x3(3, h2, w3, b3, whole_call, params, NB, P, H, W, 3, st))) return rc;
    } else {
      if ((rc = os2d_launch_conv(3, h2, static_cast<const float*>(w3), b3, params, NB, P, H, W, st))) return rc;
    }
    mark(b0, 7);
    mark(b0, 8);
    if (dumps_active() && b0 == 0) dump_slot(stream, 6, params, (size_t)NB * P * H * W * 4);
    if ((rc = os2d_launch_sample_decode(corr, params, NB, H, W, P, inverse, stride, rec_field, bc, B, b0, loc, cls,
                                        corners, fp32_ops ? nullptr : flags, epoch, status, st)))
      return rc;
    mark(b0, 9);
  }
  return 0;
}

int os2d_head_forward(const float* fm, const float* qp, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, int A, int B, int C, int H, int W, int P,
                      int inverse, int stride, int rec_field, float* loc, float* cls, float* corners, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return os2d_head_forward_ex(fm, qp, w1, b1, w2, b2, w3, b3, A, B, C, H, W, P, inverse, stride, rec_field, loc, cls,
                              corners, workspace, workspace_bytes, stream, OS2D_PRECISION_F32, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr);
}

size_t os2d_class_split_bytes(int B, int C) {
  if (B < 1 || C < 1) return 0;
  return (size_t)B * os2d_corr_groups(C) * 2 * 256 * 16;
}

int os2d_class_split(const float* qp, void* qs, int B, int C, void* stream) {
  if (!qp || !qs || B < 1 || C < 1) {
    os2d_set_error("os2d_class_split: bad arguments");
    return -1;
  }
  return os2d_launch_split_qp(qp, qs, B, C, S(stream));
}

size_t os2d_packed_conv_bytes(int layer, int precision) {
  if (precision == OS2D_PRECISION_F32) return os2d_packed_conv_floats(layer) * sizeof(float);
  if (precision != OS2D_PRECISION_F16X3 && precision != OS2D_PRECISION_F16X2) return 0;
  // [G][steps padded to whole stages][2][2][MT] units of 16 B (conv_f16x3.hip: layer 1 SS=5, layer 2 SS=7)
  if (layer == 1) return (size_t)OS2D_G * os2d_conv1_steps_padded() * 4 * 128 * 16;
  if (layer == 2) return (size_t)16 * 14 * 4 * 64 * 16;
  if (layer == 3) return (size_t)8 * 14 * 4 * 32 * 16;
  return 0;
}

int os2d_pack_conv_f16x3(int layer, int P, const float* w, const float* b, const float* bn_weight, const float* bn_bias,
                         const float* bn_running_mean, const float* bn_running_var, float bn_eps, const int* weight_exp,
                         const int* in_exp, const int* out_exp, void* packed_w, float* packed_b, void* stream) {
  const bool has_bn = bn_weight || bn_bias || bn_running_mean || bn_running_var;
  if (layer < 1 || layer > 3 || !w || !b || !packed_w || !packed_b || !weight_exp || !in_exp ||
      (layer != 3 && !out_exp) || (has_bn && !(bn_weight && bn_bias && bn_running_mean && bn_running_var)) ||
      (layer == 3 && P != 6 && P != 4)) {
    os2d_set_error("os2d_pack_conv_f16x3: bad arguments (layer %d, P %d)", layer, P);
    return -1;
  }
  if (layer == 1)
    return os2d_launch_pack_conv_f16(w, b, bn_weight, bn_bias, bn_running_mean, bn_running_var, bn_eps, 128, OS2D_K, 7,
                                     128, os2d_conv1_steps_padded(), weight_exp, in_exp, out_exp, packed_w, packed_b,
                                     S(stream));
  if (layer == 2)
    return os2d_launch_pack_conv_f16(w, b, bn_weight, bn_bias, bn_running_mean, bn_running_var, bn_eps, 64, 128, 5, 64,
                                     14, weight_exp, in_exp, out_exp, packed_w, packed_b, S(stream));
  return os2d_launch_pack_conv_f16(w, b, bn_weight, bn_bias, bn_running_mean, bn_running_var, bn_eps, P, 64, 5, 32, 14,
                                   weight_exp, in_exp, nullptr, packed_w, packed_b, S(stream));
}

int os2d_rnorm_exp(void) { return OS2D_RNORM_EXP; }

int os2d_corr_normalize_f16x3(const float* corr, void* rshb, int NB, int H, int W, void* stream) {
  if (!corr || !rshb || NB < 1 || H < 1 || W < 1) {
    os2d_set_error("os2d_corr_normalize_f16x3: bad arguments");
    return -1;
  }
  return os2d_launch_corr_normalize_shb(corr, rshb, NB, H, W, S(stream));
}

int os2d_transform_conv_f16x3(int layer, const void* in, const void* packed_w, const float* packed_b, void* out, int NB,
                              int P, int H, int W, int terms, int* status, void* stream) {
  if (!in || !packed_w || !packed_b || !out || NB < 1 || H < 1 || W < 1 || layer < 1 || layer > 3 ||
      (layer == 3 && P != 6 && P != 4) || (terms != 3 && !(terms == 2 && layer == 1))) {
    os2d_set_error("os2d_transform_conv_f16x3: bad arguments (layer=%d NB=%d P=%d terms=%d)", layer, NB, P, terms);
    return -1;
  }
  if (W > (layer == 1 ? OS2D_MAX_W_DIRECT7 : OS2D_MAX_W)) {
    os2d_set_error("os2d_transform_conv_f16x3: feature map width %d > %d", W, layer == 1 ? OS2D_MAX_W_DIRECT7 : OS2D_MAX_W);
    return -1;
  }
  return os2d_launch_conv_f16x3(layer, in, packed_w, packed_b, Os2dRangeFlag{status, OS2D_STATUS_F16_RANGE}, out, NB, P, H, W, terms, S(stream));
}

int os2d_fft_sizes(int H, int W, int* P, int* Q, int* nbins) {
  if (!P || !Q || !nbins) {
    os2d_set_error("os2d_fft_sizes: null output");
    return -1;
  }
  if (!os2d_fft_plan(H, W, P, Q, nbins, nullptr)) {
    os2d_set_error("os2d_fft_sizes: no transform plan for a %dx%d map", H, W);
    return -3;
  }
  return 0;
}

int os2d_fft_tiles(int H, int W, int* tiles_y, int* tiles_x, int* tile_h, int* tile_w) {
  int t[6];
  if (!tiles_y || !tiles_x || !tile_h || !tile_w) {
    os2d_set_error("os2d_fft_tiles: null output");
    return -1;
  }
  if (!os2d_fft_plan(H, W, nullptr, nullptr, nullptr, t)) {
    os2d_set_error("os2d_fft_tiles: no transform plan for a %dx%d map", H, W);
    return -3;
  }
  *tiles_y = t[0];
  *tiles_x = t[1];
  *tile_h = t[2];
  *tile_w = t[3];
  return 0;
}

int os2d_fft_forward(const float* corr, const float* inv_norm, float* X, const float* twQ, const float* twP, int NB, int C,
                     int H, int W, void* stream) {
  if (!corr || !inv_norm || !X || !twQ || !twP || NB < 1 || C < 1 || H < 1 || W < 1) {
    os2d_set_error("os2d_fft_forward: bad arguments");
    return -1;
  }
  return os2d_launch_fft_forward(corr, inv_norm, X, twQ, twP, NB, C, H, W, S(stream));
}

int os2d_fft_inverse_ex(const float* Y, const float* packed_b, void* out, const float* twQ, const float* twP, int NB, int Cout,
                        int H, int W, int* status, int layout, void* stream) {
  if (!Y || !packed_b || !out || !twQ || !twP || NB < 1 || Cout != 128 || H < 1 || W < 1 ||
      (layout != OS2D_SPECTRA_ROWS && layout != OS2D_SPECTRA_QUADS)) {
    os2d_set_error("os2d_fft_inverse: bad arguments (Cout must be 128: the 7x7 layer)");
    return -1;
  }
  int rc = os2d_launch_border_zero_shb_planes(out, NB * (Cout / 8) * 2, H, W, S(stream));
  if (rc) return rc;
  return os2d_launch_fft_inverse(Y, packed_b, 128, out, twQ, twP, NB, Cout, H, W, Os2dRangeFlag{status, OS2D_STATUS_F16_RANGE}, layout, 0, S(stream));
}

int os2d_fft_inverse(const float* Y, const float* packed_b, void* out, const float* twQ, const float* twP, int NB, int Cout,
                     int H, int W, int* status, void* stream) {
  return os2d_fft_inverse_ex(Y, packed_b, out, twQ, twP, NB, Cout, H, W, status, OS2D_SPECTRA_ROWS, stream);
}

size_t os2d_spectral_weight_bytes(int C, int Cout, int nbins) {
  if (C < 1 || Cout < 1 || Cout > 128 || nbins < 8 || (nbins & 7)) return 0;
  return os2d_spectral_weight_floats(C, Cout, nbins) * sizeof(float);
}

int os2d_spectral_gemm(const float* wspec, const float* X, float* Y, int NB, int C, int Cout, int nbins, void* stream) {
  if (!wspec || !X || !Y || NB < 1 || C < 1 || Cout < 1 || Cout > 128 || nbins < 8 || (nbins & 7)) {
    os2d_set_error("os2d_spectral_gemm: bad arguments (NB=%d C=%d Cout=%d nbins=%d; nbins must be a multiple of 8)", NB, C,
                   Cout, nbins);
    return -1;
  }
  if ((reinterpret_cast<uintptr_t>(wspec) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) {
    os2d_set_error("os2d_spectral_gemm: buffers must be 16-byte aligned");
    return -1;
  }
  return os2d_launch_spectral_gemm(wspec, X, Y, NB, C, Cout, nbins, S(stream));
}

int os2d_spectral_weights_build(const double* wfold, const double* twP64, const double* twQ64, int C, int Cout, int P, int Q,
                                int nbins, int split, void* out, void* workspace, void* stream) {
  if (!wfold || !twP64 || !twQ64 || !out || (split && !workspace) || C < 1 || Cout < 1 || Cout > 128 || nbins < 8 || (nbins & 7)) {
    os2d_set_error("os2d_spectral_weights_build: bad arguments (C=%d Cout=%d nbins=%d)", C, Cout, nbins);
    return -1;
  }
  if ((reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
    os2d_set_error("os2d_spectral_weights_build: out must be 16-byte, workspace 8-byte aligned");
    return -1;
  }
  return os2d_launch_spectra_pack(wfold, twP64, twQ64, C, Cout, P, Q, nbins, split, 0, out, workspace, S(stream));
}

size_t os2d_spectral_weight16_bytes(int C, int nbins) {
  if (C < 1 || nbins < 8 || (nbins & 7)) return 0;
  return os2d_spectral_weight16_size(C, nbins);
}

float os2d_spectral_xscale(int H, int W) {
  int t[6];
  if (H < 1 || W < 1 || !os2d_fft_plan(H, W, nullptr, nullptr, nullptr, t)) return 0.f;
  return os2d_spectral_xscale_for(t[4], t[5]);      // samples of one transform window (the whole map, or a tile + its halo)
}

int os2d_spectral_gemm_f16(const void* w16, const float* X, float* Y, int NB, int C, int Cout, int nbins, float xscale,
                           void* stream) {
  if (!w16 || !X || !Y || NB < 1 || C < 1 || Cout < 1 || Cout > 128 || nbins < 8 || (nbins & 7) || !(xscale > 0.f)) {
    os2d_set_error("os2d_spectral_gemm_f16: bad arguments (NB=%d C=%d Cout=%d nbins=%d; nbins must be a multiple of 8)", NB, C,
                   Cout, nbins);
    return -1;
  }
  if ((reinterpret_cast<uintptr_t>(w16) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) {
    os2d_set_error("os2d_spectral_gemm_f16: buffers must be 16-byte aligned");
    return -1;
  }
  return os2d_launch_spectral_gemm_f16(w16, X, Y, NB, C, Cout, nbins, xscale, 0, 0, S(stream));
}

/* ---- the transforms of the frequency-domain 7x7 layer as matrix products (dft_mfma.hip; OS2D_PRECISION_FFTX3) */
int os2d_dft_sizes(int H, int W, int* P, int* Q, int* nbins, int* tiles) {
  if (!P || !Q || !nbins) {
    os2d_set_error("os2d_dft_sizes: null output");
    return -1;
  }
  if (!os2d_dft_plan(H, W, P, Q, nbins, tiles)) {
    os2d_set_error("os2d_dft_sizes: no transform plan for a %dx%d map", H, W);
    return -3;
  }
  return 0;
}

int os2d_dft_channel_stride(int C) { return C == OS2D_K ? OS2D_XSPEC_CPAD : (C + 7) / 8 * 8; }

size_t os2d_dft_matrices_bytes(int P, int Q) {
  if (P < 4 || (P & 3) || Q < 2 || (Q & 1)) return 0;
  return os2d_dft_matrices_size(P, Q);
}

int os2d_dft_matrices_build(const double* twP64, const double* twQ64, int P, int Q, void* out, void* stream) {
  if (!twP64 || !twQ64 || !out || (reinterpret_cast<uintptr_t>(out) & 15)) {
    os2d_set_error("os2d_dft_matrices_build: bad arguments (out must be 16-byte aligned)");
    return -1;
  }
  return os2d_launch_dft_matrices(twP64, twQ64, P, Q, out, S(stream));
}

int os2d_dft_forward(const float* corr, const float* inv_norm, float* X, const void* matrices, int NB, int C, int H, int W,
                     void* stream) {
  if (!corr || !inv_norm || !X || !matrices || NB < 1 || C < 1 || H < 1 || W < 1) {
    os2d_set_error("os2d_dft_forward: bad arguments");
    return -1;
  }
  return os2d_launch_dft_forward(corr, inv_norm, X, matrices, NB, C, os2d_dft_channel_stride(C), H, W, S(stream));
}

int os2d_dft_inverse(const float* Y, const float* packed_b, void* out, const void* matrices, int NB, int Cout, int H, int W,
                     int* status, void* stream) {
  if (!Y || !packed_b || !out || !matrices || NB < 1 || Cout != 128 || H < 1 || W < 1) {
    os2d_set_error("os2d_dft_inverse: bad arguments (Cout must be 128: the 7x7 layer)");
    return -1;
  }
  return os2d_launch_dft_inverse(Y, packed_b, 128, out, matrices, NB, Cout, H, W, Os2dRangeFlag{status, OS2D_STATUS_F16_RANGE}, 1, S(stream));   // incl. the plane borders
}

int os2d_spectral_weights_build_dft(const double* wfold, const double* twP64, const double* twQ64, int C, int Cout, int P, int Q,
                                    int nbins, void* out, void* workspace, void* stream) {
  if (!wfold || !twP64 || !twQ64 || !out || !workspace || C < 1 || Cout < 1 || Cout > 128 || nbins < 8 || (nbins & 7) || (P & 3)) {
    os2d_set_error("os2d_spectral_weights_build_dft: bad arguments (C=%d Cout=%d P=%d nbins=%d)", C, Cout, P, nbins);
    return -1;
  }
  if ((reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
    os2d_set_error("os2d_spectral_weights_build_dft: out must be 16-byte, workspace 8-byte aligned");
    return -1;
  }
  return os2d_launch_spectra_pack(wfold, twP64, twQ64, C, Cout, P, Q, nbins, 1, 1, out, workspace, S(stream));
}

int os2d_spectral_gemm_f16_quads(const void* w16, const float* X, float* Y, int NB, int C, int Cout, int nbins, float xscale,
                                 void* stream) {
  if (!w16 || !X || !Y || NB < 1 || C < 1 || Cout < 1 || Cout > 128 || nbins < 8 || (nbins & 7) || !(xscale > 0.f)) {
    os2d_set_error("os2d_spectral_gemm_f16_quads: bad arguments (NB=%d C=%d Cout=%d nbins=%d)", NB, C, Cout, nbins);
    return -1;
  }
  if ((reinterpret_cast<uintptr_t>(w16) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) {
    os2d_set_error("os2d_spectral_gemm_f16_quads: buffers must be 16-byte aligned");
    return -1;
  }
  return os2d_launch_spectral_gemm_f16(w16, X, Y, NB, C, Cout, nbins, xscale, 1, os2d_dft_channel_stride(C), S(stream));
}

float os2d_dft_xscale(int H, int W) {
  int t[6];
  if (H < 1 || W < 1 || !os2d_dft_plan(H, W, nullptr, nullptr, nullptr, t)) return 0.f;
  return os2d_spectral_xscale_for(t[4], t[5]);
}

int os2d_alignment_grids(const float* params, int NB, int H, int W, int P, int inverse, float* theta, float* grids,
                         void* stream) {
  if (!params || (!theta && !grids) || NB < 1 || H < 1 || W < 1 || (P != 6 && P != 4)) {
    os2d_set_error("os2d_alignment_grids: bad arguments");
    return -1;
  }
  return os2d_launch_alignment_grids(params, NB, H, W, P, inverse, theta, grids, S(stream));
}

int os2d_prof_event_create(void** ev) {
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess || !ev) {
    os2d_set_error("hipEventCreate: %s", hipGetErrorString(rc));
    return -4;
  }
  *ev = e;
  return 0;
}
int os2d_prof_event_destroy(void* ev) { return hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess ? 0 : -4; }
int os2d_prof_event_elapsed_ms(void* a, void* b, float* ms) {
  hipError_t rc = hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(a), reinterpret_cast<hipEvent_t>(b));
  if (rc != hipSuccess) {
    os2d_set_error("hipEventElapsedTime: %s", hipGetErrorString(rc));
    return -4;
  }
  return 0;
}

}  // extern "C"

/* os2d_hip.h -- C ABI of libos2d_hip.so: the MI355X (gfx950) OS2D correlation + alignment head.
 *
 * The reference (aosokin/os2d) is pure Python/PyTorch and has no FFI; the interface it exposes for this path is
 * the Python class API (Os2dHeadCreator.create_os2d_head / Os2dHead.forward / Os2dBoxCoder decode).  This header is
 * the boundary our Python mirror of that API (os2d_amd/modeling) binds with ctypes; each entry point cites the
 * reference code it replaces (paths relative to the reference checkout).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. the torch caching allocator); fp32, row-major,
 *     dense; nothing is allocated or freed inside the library;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream) and
 *     re-entrant: no global mutable state except the thread-local error string;
 *   - return value: 0 on success, negative on error (-1 bad argument, -2 workspace too small, -3 unsupported
 *     shape, -4 HIP runtime error); os2d_last_error() then describes it.  Nothing throws.
 */
#ifndef OS2D_HIP_H
#define OS2D_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OS2D_ABI_VERSION 9

/* arithmetic of the two large TransformNet convolutions (everything else is fp32 in both modes) */
#define OS2D_PRECISION_F32 0   /* v_mfma_f32_32x32x2_f32: exact fp32 (k-ordered fmaf chain)                            */
#define OS2D_PRECISION_F16X3 1 /* operands split into fp16 hi+lo, three v_mfma_f32_32x32x16_f16 per product, fp32     */
                               /* accumulation: fp32-equivalent results at ~5x the matrix rate  */
#define OS2D_PRECISION_F16X2 2 /* as F16X3 (same packed weights, same split operands), except that the 7x7 layer uses  */
                               /* its weights as fp16 roundings only (two MFMAs per product): box regression within   */
                               /* 5e-5, scores within 1e-6 of fp32 - inside the 1e-4 parity bound, 2/3 of the work     */

#define OS2D_PRECISION_FFT 3   /* as F16X3, except that the 7x7 layer runs in the frequency domain in fp32: in-LDS real FFT of the  */
                               /* normalised correlation, one complex 128 x 225 GEMM per bin on the fp32 matrix cores, inverse FFT  */
                               /* (16.7x fewer multiply-adds; agrees with an fp64 convolution to 1e-7, closer than a direct fp32    */
                               /* convolution).  Maps larger than the in-LDS transform are cut into overlap-save tiles (os2d_fft_tiles) */

#define OS2D_PRECISION_FFT32 5 /* strictly fp32 AND fast: fp32-MFMA correlation and 5x5 layers as F32, the 7x7 layer in the frequency  */
                               /* domain as FFT (fp32 transforms, complex GEMM on the fp32 matrix cores, fp32 planes out); no fp16   */
                               /* value anywhere.  w1..w3 / b1..b3 from os2d_pack_conv (w1 unused), wspec as for FFT                  */

#define OS2D_PRECISION_FFTX3 4 /* as FFT, with the per-bin complex GEMM on the half-precision matrix cores: spectra split into fp16   */
                               /* hi + lo (three v_mfma_f32_32x32x16_f16 per product, the arithmetic of F16X3; scales chosen so that  */
                               /* no spectrum value can leave the fp16 range: |X| <= H*W by construction, the weight spectra are     */
                               /* host data); wspec = the split weight spectra of os2d_spectral_weight16_bytes                     */

/* ABI version of the loaded library (compare with OS2D_ABI_VERSION). */
int os2d_abi_version(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* os2d_last_error(void);

/* ---- TransformNet weight packing: reference os2d/modeling/head.py:604-646 (TransformationNet.__init__) holds
 * conv.0/conv.1(BN)/conv.3/conv.4(BN)/linear; eval-mode BatchNorm (head.py:623) is folded here and the filters are
 * re-laid-out for the MFMA implicit-GEMM kernels.  Sizes of the packed buffers (in floats): */
size_t os2d_packed_conv_floats(int layer /*1|2|3*/);       /* weights */
size_t os2d_packed_bias_floats(int layer /*1|2|3*/);       /* bias (+ the per-row scale factors of the f16x3 packing) */
/* layer 1: w[128,225,7,7]; layer 2: w[64,128,5,5]; layer 3: w[P,64,5,5] (bn_* = NULL, P = 6 or 4).
 * bn_eps is BatchNorm2d.eps (1e-5 in the reference). */
int os2d_pack_conv(int layer, int P, const float* w, const float* b, const float* bn_weight, const float* bn_bias,
                   const float* bn_running_mean, const float* bn_running_var, float bn_eps, float* packed_w,
                   float* packed_b, void* stream);

/* ---- class feature map preparation: reference head.py:241-259 (resize_feature_maps_to_reference_size) and
 * head.py:293 (normalize_feature_map_L2, eps 1e-5; skipped when normalize == 0) for ONE class map src[C,h,w]:
 *   q15 [C,15,15]  resized + L2-normalised map (what Os2dHead.class_feature_maps holds)
 *   qp  [C,256]    the same values as the correlation GEMM operand: row m = x_T*15 + y_T, rows 225..255 zero. */
int os2d_class_prepare(const float* src, int C, int h, int w, int normalize, float* q15, float* qp, void* stream);
/* The same for all B classes of a head in two launches (reference head.py:261-268 loops the maps): srcs = DEVICE array of
 * B device pointers to maps [C,h_b,w_b], sizes = DEVICE int array [B][2] = (h_b, w_b); q15 [B,C,15,15], qp [B,C,256];
 * workspace: os2d_class_prepare_workspace_floats(B, C) floats (per-channel-block partial sums of squares).             */
size_t os2d_class_prepare_workspace_floats(int B, int C);
int os2d_class_prepare_batch(const float* const* srcs, const int* sizes, int B, int C, int normalize, float* q15,
                             float* qp, float* workspace, void* stream);

/* ---- the head: reference head.py:308-435 (Os2dHead.forward, eval mode) for B classes on A image feature maps.
 *   fm [A,C,H,W] raw backbone features; qp [B,C,256] from os2d_class_prepare; packed TransformNet from os2d_pack_conv
 *   P = 6 (affine, head.py:98-100) or 4 (simplified affine, head.py:101-107); inverse = use_inverse_geom_model
 *   stride / rec_field: backbone stride and receptive field (16 / 16 for ResNet-C4, feature_extractor.py:115-117)
 *   outputs loc [A,B,4,H,W], cls [A,B,1,H,W], corners [A,B,8,H,W]  (cls_detached aliases cls in eval, head.py:400-402)
 * The workspace may be smaller than os2d_head_workspace_bytes(A,B,...) reports: classes are then processed in
 * chunks; it must hold at least os2d_head_workspace_bytes(A,1,...) bytes.  Any C >= 1 (zero channel groups pad the operands); W <= 3600 and H <= 2784 (the
 * transform planner's 48 tiles per axis), and W <= 209 for the DIRECT 7x7 kernels of os2d_head_forward / the non-frequency precisions
 * (3344-px wide images at stride 16: they keep three halo rows in LDS); all checked before any launch. */
int os2d_head_workspace_bytes(int A, int B, int C, int H, int W, int P, size_t* bytes);
int os2d_head_workspace_bytes_ex(int A, int B, int C, int H, int W, int P, int precision, size_t* bytes);  /* + spectra (FFT mode) */
int os2d_head_forward(const float* fm, const float* qp, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, int A, int B, int C, int H, int W, int P,
                      int inverse, int stride, int rec_field, float* loc, float* cls, float* corners,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- f16x3 weight packing (bn_* = NULL for layer 3, as in os2d_pack_conv).  Range handling, so that no value of the
 * path can leave the fp16 range for finite inputs and every value keeps its 22 bits ("range safety"):
 * all three are DEVICE int arrays of power-of-two exponents, applied exactly:
 *   in_exp  [Cin]   input channel c of the layer is stored as fp16 hi|lo of x[c] * 2^in_exp[c] (layer 1: os2d_rnorm_exp()
 *                   = 12 for all 225 channels; layers 2 / 3: the previous layer's out_exp);
 *   out_exp [Cout]  output channel o is written as fp16 hi|lo of y[o] * 2^out_exp[o] (NULL for layer 3, whose output is
 *                   fp32); the binding derives it from a rigorous bound of |y[o]| so that finite inputs cannot overflow;
 *   weight_exp [Cout]  the folded weights w[o][c] * 2^-in_exp[c] of output channel o are multiplied by 2^weight_exp[o]
 *                   before the hi/lo split; choose the largest exponent that keeps their maximum <= 16384.
 * packed_b [3*MT] receives per output row the folded bias, 2^-weight_exp (applied to the accumulator) and 2^out_exp.
 * Buffer sizes: os2d_packed_conv_bytes, os2d_packed_bias_floats.                                                      */
size_t os2d_packed_conv_bytes(int layer, int precision);
int os2d_rnorm_exp(void);
int os2d_pack_conv_f16x3(int layer /*1|2|3*/, int P, const float* w, const float* b, const float* bn_weight,
                         const float* bn_bias, const float* bn_running_mean, const float* bn_running_var, float bn_eps,
                         const int* weight_exp, const int* in_exp, const int* out_exp, void* packed_w, float* packed_b,
                         void* stream);

/* split class operand for the f16x3 correlation: qp [B,C,256] fp32 (os2d_class_prepare) -> qs [B, G, hi|lo, 256] units
 * of 8 halves, scaled by 2^12; G = ceil(C/8) padded with zero groups to a multiple of 4 (whole 32-channel K chunks of
 * the correlation kernel): os2d_class_split_bytes(B, C) bytes.                                                       */
size_t os2d_class_split_bytes(int B, int C);
int os2d_class_split(const float* qp, void* qs, int B, int C, void* stream);

/* ---- extended head entry point: identical to os2d_head_forward, plus
 *   precision     OS2D_PRECISION_F32 (w1..w3 from os2d_pack_conv; qs ignored) or OS2D_PRECISION_F16X3 / _F16X2
 *                 (w1..w3 from os2d_pack_conv_f16x3, qs [B, C/8, 2, 256, 8] halves from os2d_class_split);
 *   stage_events  NULL, or an array of 13 hipEvent_t (from os2d_prof_event_create; NULL entries are skipped); events
 *                 [2s] / [2s+1] are recorded on `stream` right before / after stage s of the FIRST class chunk
 *                 (s = 0 correlation, 1 conv 7x7, 2 conv 5x5 128->64, 3 conv 5x5 64->P, 4 resample+encode); under
 *                 OS2D_PRECISION_FFT events [10], [11], [12] additionally mark the start of the forward transform, its end
 *                 (= start of the spectral GEMM) and the end of the GEMM (= start of the inverse transform) inside stage 1;
 *   chunk_classes NULL, or receives the number of classes per chunk chosen for the given workspace;
 *   status        NULL, or a device-visible int (device memory or mapped pinned host memory) that receives sticky
 *                 status bits (plain system-scope stores, never cleared by the library): OS2D_STATUS_F16_RANGE when an image
 *                 feature was non-finite or a split-fp16 activation left the fp16 range (only possible with non-finite
 *                 inputs).  The OUTPUTS of such a call are poisoned on the device, by the call itself: the split-fp16 kernels
 *                 store the call's epoch into range words at the start of the workspace (one per image, one for the whole call)
 *                 and the last kernel writes NaN into loc / cls / corners of every flagged image, as the reference's
 *                 torch.relu / norm propagate a NaN (head.py:339, 650) - no host synchronisation, later calls unaffected.  The
 *                 words are never cleared (a stale epoch matches no later call); zero-fill the first 4 KB of a NEW workspace
 *                 once so that its initial content cannot match either.  A caller that wants the reference's exact NaN
 *                 pattern re-runs a flagged call in OS2D_PRECISION_F32, whose kernels keep NaN through their ReLUs;
 *   wspec, twQ, twP  the frequency-domain precisions only (NULL otherwise; w1 / b1..b3 / w2 / w3 as for F16X3, FFT32: as for F32):
 *                 OS2D_PRECISION_FFT / _FFT32: the weight spectra of the 7x7 layer for THIS map's transform size in the layout of
 *                 os2d_spectral_gemm and the two twiddle tables of os2d_fft_forward;
 *                 OS2D_PRECISION_FFTX3: wspec = the split weight spectra of os2d_spectral_weights_build_dft for the transform size
 *                 of os2d_dft_sizes(H, W), twQ = the operand matrices of os2d_dft_matrices_build for it, twP = NULL.
 * Width: maps up to 3600 columns (reference head.py:619-629 has no limit); beyond 209 (the direct 7x7 kernels' limit) only the
 * frequency-domain precisions run (tiled), beyond 316 the 5x5 kernels run in column strips (conv_f16x3.hip).                   */
#define OS2D_STATUS_F16_RANGE 1
int os2d_head_forward_ex(const float* fm, const float* qp, const void* w1, const float* b1, const void* w2,
                         const float* b2, const void* w3, const float* b3, int A, int B, int C, int H, int W, int P,
                         int inverse, int stride, int rec_field, float* loc, float* cls, float* corners,
                         void* workspace, size_t workspace_bytes, void* stream, int precision, const void* qs,
                         void** stage_events, int* chunk_classes, int* status, const float* wspec, const float* twQ,
                         const float* twP);
/* debugging aid, DIAGNOSTIC builds only (-DOS2D_DIAG_DUMP; the product library ignores the call and sets os2d_last_error):
 * from now on os2d_head_forward_ex on `stream` copies an intermediate buffer of its first class chunk to dst (slot 0 corr,
 * 1 inverse norms, 2 input spectra, 3 output spectra, 4 h1, 5 h2, 6 params; at most `bytes`) on that stream; dst == NULL
 * unregisters the slot (do that before freeing the destination).                                                         */
void os2d_debug_set_dump(void* stream, int slot, void* dst, size_t bytes);
int os2d_prof_event_create(void** ev);
int os2d_prof_event_destroy(void* ev);
int os2d_prof_event_elapsed_ms(void* begin, void* end, float* ms);   /* both events must have completed */

/* ---- per-stage entry points (same kernels, exposed for unit parity tests and profiling) ---- */
/* sumsq [A,H*W] = sum_c fm^2 (head.py:339 norm).                                                              */
int os2d_fm_sumsq(const float* fm, float* sumsq, int A, int C, int H, int W, void* stream);
/* correlation head.py:339-350 + TransformNet input normalisation head.py:650:
 *   corr [A*B,225,H*W] raw;  rnorm [A*B,226,PLANE] relu+L2 (eps 1e-6) in the zero-bordered plane layout,
 *   PLANE = os2d_plane_floats(H,W).                                                                            */
size_t os2d_plane_floats(int H, int W);
int os2d_corr(const float* fm, const float* qp, const float* sumsq, float* corr, float* rnorm, int A, int B, int C,
              int H, int W, void* stream);
/* The same stage on the half-precision matrix cores (the kernels of the f16x3 / f16x2 head): fm [A,C,H,W] raw features, qs
 * from os2d_class_split -> corr [A*B,225,H*W] fp32 and the relu + L2-normalised tensor in the split-half blocked layout
 * (A*B * os2d_shb_bytes(225,H,W) bytes, values scaled by 2^os2d_rnorm_exp()).  workspace: os2d_corr_f16x3_workspace_bytes
 * (per-position norms + the split image operand), 256-byte aligned.                                                       */
size_t os2d_corr_f16x3_workspace_bytes(int A, int C, int H, int W);
int os2d_corr_f16x3(const float* fm, const void* qs, float* corr, void* rshb, int A, int B, int C, int H, int W,
                    void* workspace, size_t workspace_bytes, void* stream);
/* That stage as the frequency-domain heads (fft / fftx3) run it: outputs corr [A*B,225,H*W] and inv_norm [A*B,H*W] =
 * 1 / (L2 over the 225 relu'd channels + 1e-6) (head.py:650, 597) instead of the normalised tensor.  Two forms of the kernel:
 *   form 0  one padded 256-row tile per class (as os2d_corr_f16x3);
 *   form 1  the classes PACKED along the matrix rows: class b owns the stacked rows [228 b, 228 b + 225) and a work-group takes
 *           256 consecutive stacked rows across class boundaries - 228 / 256 of the matrix instructions;
 *   form -1 the head's own choice: whichever takes fewer rounds of the 256 CUs;
 *   + 4     no half tiles: by default the tiles of the last, partial round of a launch are cut into two 128-position halves
 *           (they take the chip half as long; same products in the same order).
 * All of them give the SAME BITS: the per-position sums of relu^2 are accumulated as 2^-44 fixed-point integers (an accumulator run
 * of 4 rows in fp32, then integers: LDS / 64-bit atomics), which do not depend on the order of arrival nor on where in a batch
 * a class sits - a class alone and the same class anywhere in a batch give identical correlation values and norms.
 * workspace: os2d_corr_f16x3_packed_workspace_bytes (that of os2d_corr_f16x3 + A*B*H*W 64-bit sums), 256-byte aligned.     */
size_t os2d_corr_f16x3_packed_workspace_bytes(int A, int B, int C, int H, int W);
int os2d_corr_f16x3_packed(const float* fm, const void* qs, float* corr, float* inv_norm, int A, int B, int C, int H, int W,
                           int form, void* workspace, size_t workspace_bytes, void* stream);
/* standalone TransformNet input normalisation head.py:650 (relu, L2 over 225 channels, eps 1e-6) of an arbitrary
 * correlation tensor corr [NB,225,H*W] -> rnorm [NB,226,PLANE]; used by TransformationNet.forward.               */
int os2d_corr_normalize(const float* corr, float* rnorm, int NB, int H, int W, void* stream);
/* TransformNet layer head.py:619-629 (layer 1,2: conv+BN+ReLU, padded-plane in/out; layer 3: conv, compact
 * [NB,P,H*W] out).  NB = A*B.                                                                                  */
int os2d_transform_conv(int layer, const float* in, const float* packed_w, const float* packed_b, float* out,
                        int NB, int P, int H, int W, void* stream);
/* The same two stages on the half-precision matrix cores (the kernels the f16x3 / f16x2 head runs), for stage-level parity
 * tests: activations are "split-half blocked" buffers [NB][ceil(ch/8)][hi|lo][PLANE] of 16-byte units (os2d_shb_bytes).
 *   os2d_corr_normalize_f16x3  corr [NB,225,H*W] fp32 -> relu + L2 (eps 1e-6), scaled by 2^os2d_rnorm_exp(), split
 *   os2d_transform_conv_f16x3  layer 1 / 2: SHB in -> SHB out (channel scales as packed); layer 3: SHB in -> fp32 [NB,P,H*W];
 *                              terms = 3 (fp32-equivalent) or 2 (layer 1 only: weights as fp16 roundings, f16x2 mode)   */
size_t os2d_shb_bytes(int channels, int H, int W);
int os2d_corr_normalize_f16x3(const float* corr, void* rshb, int NB, int H, int W, void* stream);
int os2d_transform_conv_f16x3(int layer, const void* in, const void* packed_w, const float* packed_b, void* out, int NB,
                              int P, int H, int W, int terms, int* status, void* stream);
/* Os2dAlignment.forward / prepare_transform_parameters_for_grid_sampler as the reference returns them (head.py:81-193):
 * params [NB,P,H*W] -> theta [NB*H*W,2,3] (NULL to skip) and the transformed template grids in local coordinates
 * grids [NB,H,W,15,15,2] (NULL to skip; F.affine_grid, align_corners=True).  The fused head materialises neither.     */
int os2d_alignment_grids(const float* params, int NB, int H, int W, int P, int inverse, float* theta, float* grids,
                         void* stream);
/* alignment epilogue head.py:81-153,184,371-435: params [NB,P,H*W] + corr [NB,225,H*W] -> loc/cls/corners
 * laid out [NB,4|1|8,H*W].                                                                                      */
int os2d_sample_decode(const float* corr, const float* params, int NB, int H, int W, int P, int inverse, int stride,
                       int rec_field, float* loc, float* cls, float* corners, void* stream);

/* ---- per-location box decode: reference os2d/modeling/box_coder.py:319-330 (torchvision BoxCoder.decode_single,
 * weights (10,10,5,5), dw/dh clamp log(1000/16)) + clip to the level image (os2d/structures/bounding_box.py:261-265).
 *   loc [NB,4,H*W] -> boxes [NB,H*W,4] xyxy; img_w <= 0 or img_h <= 0 skips the clip.                                                                  */
int os2d_decode_boxes(const float* loc, int NB, int H, int W, int stride, int rec_field, float img_w, float img_h,
                      float* boxes, void* stream);

/* ---- per-class greedy NMS: reference os2d/modeling/box_coder.py:425-437 + os2d/structures/bounding_box.py:344-387
 * (torchvision.ops.nms semantics: suppress when IoU > threshold).  Batched over NC independent lists:
 *   boxes    [NC,N,4] xyxy, each list sorted by DECREASING score, the first counts[c] entries valid
 *   keep     [NC,N]   1 = survives (in the sorted order), 0 otherwise;  num_keep [NC]
 *   workspace os2d_nms_workspace_bytes(NC,N) bytes (kept-box lists), 16-byte aligned.                           */
int os2d_nms_workspace_bytes(int NC, int N, size_t* bytes);
int os2d_nms(const float* boxes, const int* counts, int NC, int N, float iou_threshold, unsigned char* keep,
             int* num_keep, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused single-level detection: everything reference os2d/modeling/box_coder.py:448-536 does per class for ONE
 * pyramid level, in one launch (one work-group per class, all state in LDS):
 *   decode + clip (:319-330, bounding_box.py:261-265), drop empty boxes and scores <= score_threshold (:489-497; pass
 *   -INFINITY to keep all), map to the output image (BoxList.resize: x * scale_x, y * scale_y; 1,1 = no mapping),
 *   greedy NMS at iou_threshold over the boxes in decreasing score (ties in location order), survivors compacted.
 *   loc [B,4,H*W], cls [B,H*W]  ->  out_boxes [B,H*W,4], out_scores [B,H*W], out_index [B,H*W] (source location),
 *   of which the first out_count[b] entries of every class are valid, by decreasing score.
 * os2d_detect_level_supported(H,W) = 1 when the level fits the kernel's LDS budget (6*pow2(H*W) + 18*H*W bytes <= 155 KB:
 * up to about 5900 locations - 60x80 fits, 72x96 does not); otherwise use os2d_decode_boxes + os2d_nms.                                        */
int os2d_detect_level_supported(int H, int W);
int os2d_detect_level(const float* loc, const float* cls, int B, int H, int W, int stride, int rec_field, float img_w,
                      float img_h, float scale_x, float scale_y, float score_threshold, float iou_threshold,
                      float* out_boxes, float* out_scores, int* out_index, int* out_count, void* stream);

/* Box transform chains: what the reference hands to decode_pyramid as ``inverse_box_transforms`` - per level a TransformList
 * of closures (os2d/structures/transforms.py:12-27, appended by transpose :32-52, resize :78-79 and crop :188-191 from
 * os2d/data/dataloader.py:286-336, called at os2d/modeling/box_coder.py:499-503) - is a chain of at most OS2D_BOX_MAX_OPS
 * axis-aligned box operations (os2d/structures/bounding_box.py:138-226), applied in order, each product / difference rounded
 * on its own exactly like the reference's tensor expressions:
 *   OS2D_BOX_OP_SCALE  BoxList.resize:                     x * ax, y * ay            (ax = target_w / image_w, ...)
 *   OS2D_BOX_OP_HFLIP  BoxList.transpose(FLIP_LEFT_RIGHT): (x1, x2) = (ax - x2, ax - x1)   (ax = image width)
 *   OS2D_BOX_OP_VFLIP  BoxList.transpose(FLIP_TOP_BOTTOM): (y1, y2) = (ay - y2, ay - y1)   (ay = image height)
 *   OS2D_BOX_OP_SHIFT  BoxList.crop:                       x - ax, y - ay            (ax, ay = left, top of the crop box)
 * op_kinds [op_count], op_args [op_count][2] = (ax, ay) are HOST arrays.  os2d_detect_level_ops = os2d_detect_level with such a
 * chain instead of the two scale factors (op_count 0 = no mapping).                                                        */
#define OS2D_BOX_OP_SCALE 1
#define OS2D_BOX_OP_HFLIP 2
#define OS2D_BOX_OP_VFLIP 3
#define OS2D_BOX_OP_SHIFT 4
#define OS2D_BOX_MAX_OPS 6
#define OS2D_BOX_MAX_DEFAULT_OPS 12
int os2d_detect_level_ops(const float* loc, const float* cls, int B, int H, int W, int stride, int rec_field, float img_w,
                          float img_h, int op_count, const int* op_kinds, const float* op_args, float score_threshold,
                          float iou_threshold, float* out_boxes, float* out_scores, int* out_index, int* out_count,
                          void* stream);

/* ---- the frequency-domain form of the 7x7 TransformNet layer (reference head.py:619-623; DESIGN.md section 4, "fft"):
 * what os2d_head_forward_ex chains under OS2D_PRECISION_FFT, exported for callers and tests.  For every frequency bin the
 * layer is one complex matrix product
 *     Y[n][o][bin] = sum_c K[o][c][bin] * X[n][c][bin]      n = image x class, c < C input channels, o < Cout <= 128
 * evaluated for all bins in one launch on the fp32 matrix cores (exact fp32 products, fp32 accumulation).
 *   X [C,NB,nbins] (channel-major: the pairs of one channel are neighbours), Y [NB,Cout,nbins] interleaved complex64 (re, im),
 *   nbins a multiple of 8;
 *   wspec: the weight spectra packed [nbins/8][2][C][8][64] complex64 - for bin group g, half h, channel c, bin j and
 *   row r the entry is K[64*h + r][c][8*g + j] (zero for rows >= Cout); os2d_spectral_weight_bytes(C, Cout, nbins) bytes.  */
size_t os2d_spectral_weight_bytes(int C, int Cout, int nbins);
/* The transforms around it (in-LDS real FFT pair, one work-group per image at a time):
 *   os2d_fft_sizes    transform sizes P, Q (even; 2^a 3^b, or 42 / 84; the weight spectra carry the -3 shift of the centred
 *                     kernel) and nbins = P*(Q/2+1) rounded up to a multiple of 8.  A map that fits the in-LDS transform is
 *                     ONE transform with P >= H+3, Q >= W+3 (the zero padding is the halo); a larger one (beyond ~96 x 128:
 *                     the 96 x 128 level of the 7-scale pyramid, reference os2d/config.py:194, os2d/data/dataloader.py:326) is
 *                     cut into TY x TX overlap-save tiles of tile_h x tile_w outputs, each transformed at P >= tile_h + 6,
 *                     Q >= tile_w + 6 along a tiled axis: T = TY*TX transforms per (map, channel), sizes are those of a tile
 *   os2d_fft_tiles    that tiling (1 x 1 and tile = map for an untiled one); a tile is one more "pair" for the kernels below:
 *                     NBT = NB * T, pair' = nb * T + ty * TX + tx
 *   os2d_fft_forward  x = relu(corr [NB,C,H*W]) * inv_norm [NB,H*W] (head.py:650 folded into the load), zero-padded ->
 *                     X [C,NBT,nbins] complex64, bin = u*(Q/2+1) + v
 *   os2d_fft_inverse  Y [NBT,128,nbins] -> the tile's H x W samples / (P*Q), + folded bias, ReLU, per-channel scale (packed_b of
 *                     os2d_pack_conv_f16x3 for layer 1), fp16 hi|lo -> split-half blocked buffer (NB*os2d_shb_bytes(128,H,W))
 *   twQ / twP         exp(-2 pi i m / Q), m < Q  and  exp(-2 pi i m / P), m < P  as complex64 device tables              */
int os2d_fft_sizes(int H, int W, int* P, int* Q, int* nbins);
int os2d_fft_tiles(int H, int W, int* tiles_y, int* tiles_x, int* tile_h, int* tile_w);
int os2d_fft_forward(const float* corr, const float* inv_norm, float* X, const float* twQ, const float* twP, int NB, int C,
                     int H, int W, void* stream);
int os2d_fft_inverse(const float* Y, const float* packed_b, void* out, const float* twQ, const float* twP, int NB, int Cout,
                     int H, int W, int* status, void* stream);
/* Layout of the OUTPUT spectra between the per-bin GEMM and the inverse transform.  The transform wants the bins of one
 * (pair, channel) together, the GEMM the pairs / channels of one bin: one side has to gather.
 *   OS2D_SPECTRA_ROWS   Y [NBT,Cout,nbins] (above): contiguous for the transform; os2d_spectral_gemm writes it
 *   OS2D_SPECTRA_QUADS  Y [nbins/4, NBT, Cout, 4]: the output rows of a pair for 4 bins are 4 KB of consecutive bytes (a 64-lane
 *                       store of the GEMM writes two 1 KB runs instead of 64 pieces 22 KB apart), the inverse transform gathers
 *                       32-byte pieces; os2d_spectral_gemm_f16 writes it (round 3: the scattered stores were 0.12 of the GEMM's
 *                       0.39 ms at 64 pairs and 2.3 of 5.6 ms at 1024; the inverse transform takes 0.198 ms either way).
 * The input spectra X are always [C,NBT,nbins].                                                                           */
#define OS2D_SPECTRA_ROWS 0
#define OS2D_SPECTRA_QUADS 1
int os2d_fft_inverse_ex(const float* Y, const float* packed_b, void* out, const float* twQ, const float* twP, int NB, int Cout,
                        int H, int W, int* status, int layout, void* stream);
int os2d_spectral_gemm(const float* wspec, const float* X, float* Y, int NB, int C, int Cout, int nbins, void* stream);
/* The same product on the half-precision matrix cores (OS2D_PRECISION_FFTX3).  w16: the weight spectra pre-split on the host,
 * [nbins/8][2][KS = ceil(C/8)][8 bins][2 channel groups][hi|lo][64 rows] units of 8 halves = (Kr, Ki) of 4 channels, row o
 * scaled by 2^wexp[o] (largest |Kr|, |Ki| of the row <= 32768), followed by 128 floats 2^-wexp[o]:
 * os2d_spectral_weight16_bytes(C, nbins) bytes.  The input spectra are scaled by os2d_spectral_xscale(H, W) (largest power of
 * two with xscale * H * W <= 65504: |X| <= H * W because every sample of the normalised maps is <= 1) and split on the fly;
 * Y is returned unscaled, with the values os2d_spectral_gemm returns (up to the 2^-22 relative error of a split product).
 * X is [C, NB, nbins] as for os2d_spectral_gemm; Y is written in the OS2D_SPECTRA_QUADS layout ([nbins/4, NB, Cout, 4]
 * complex64): hand it to os2d_fft_inverse_ex(..., OS2D_SPECTRA_QUADS).                                                    */
size_t os2d_spectral_weight16_bytes(int C, int nbins);
/* Builds the weight spectra of the 7x7 layer on the device, once per transform size and parameter version (off the per-step
 * path): wfold = the BatchNorm-folded filters [Cout,C,7,7] as DEVICE float64 (reference head.py:619-623; the fold is the
 * caller's), centred on the origin of the P x Q grid; twP64 / twQ64 = DEVICE float64 tables [P][2] / [Q][2] of (cos, sin) of
 * -2 pi m / n (exact values rounded once).  K[o][c][u][v] is a 7-term DFT per axis evaluated in float64.
 *   split != 0  out = the layout of os2d_spectral_gemm_f16 (os2d_spectral_weight16_bytes(C, nbins) bytes, incl. the 128 row
 *               scales); workspace = 1 KB of device memory (row maxima)
 *   split == 0  out = the layout of os2d_spectral_gemm (os2d_spectral_weight_bytes(C, Cout, nbins) bytes)                  */
int os2d_spectral_weights_build(const double* wfold, const double* twP64, const double* twQ64, int C, int Cout, int P, int Q,
                                int nbins, int split, void* out, void* workspace, void* stream);
float os2d_spectral_xscale(int H, int W);
int os2d_spectral_gemm_f16(const void* w16, const float* X, float* Y, int NB, int C, int Cout, int nbins, float xscale,
                           void* stream);

/* ---- the same layer with the TRANSFORMS as matrix products on the half-precision matrix cores (round 4; what
 * os2d_head_forward_ex chains under OS2D_PRECISION_FFTX3; dft_mfma.hip): per image the row transform is x . FqT, the column
 * transform Fp2 . R (complex as real 2 x 2 blocks), the inverse E2 . Y and T^T . Gq with the Hermitian weights folded into Gq -
 * every operand split into fp16 hi + lo, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation (the arithmetic of the
 * per-bin GEMM above); four images (4 channels of one pair / 4 output channels) per work-group iteration.  Any transform size
 * with P % 4 == 0, P <= 64, even Q <= 94 will do (no factorisation constraint): P = H + 3 rounded up to 4, Q = W + 3 rounded up
 * to 2 for a map that fits, overlap-save tiles otherwise (any map width up to the head's own limit).
 *   os2d_dft_sizes           P, Q, nbins (multiple of 8) and tiles[6] = TY, TX, tile_h, tile_w, window rows, window columns
 *                            (tiles may be NULL); BIN ORDER: bin = v * P + u (u fastest: a quad of 4 bins never straddles v)
 *   os2d_dft_channel_stride  Cpad: channel stride of the input spectra (225 -> 232 = the GEMM's k-steps of 8)
 *   os2d_dft_matrices_build  the four constant operand matrices of a (P, Q) transform from float64 tables (cos, sin of
 *                            -2 pi m / n as for os2d_spectral_weights_build), os2d_dft_matrices_bytes(P, Q) bytes; they depend
 *                            on (P, Q) only
 *   os2d_dft_forward         relu(corr [NB,C,H*W]) * inv_norm [NB,H*W] -> X [nbins/4, NB*T, Cpad, 4] complex64 (quads of bins x
 *                            channels: 128-byte runs per work-group iteration, 256-byte runs per k-step of the GEMM)
 *   os2d_spectral_weights_build_dft   the split weight spectra of os2d_spectral_gemm_f16 in that bin order
 *   os2d_spectral_gemm_f16_quads      the per-bin GEMM reading X in that layout (xscale: os2d_dft_xscale(H, W)); Y
 *                            [nbins/4, NB*T, Cout, 4]
 *   BLOCKS OF 64 PAIRS: with more than 64 pairs (pair' = class x tile) both X and Y are stored block after block,
 *                            [pair' / 64][nbins/4][pair' % 64][channels][4], the last block holding the NB*T % 64 remaining
 *                            pairs without padding (the buffer sizes do not change).  A block is the pair tile of a GEMM
 *                            work-group: its operands are one contiguous slab, and the 688 runs a transform iteration touches
 *                            are 475 / 262 KB apart at any batch size instead of 7.6 / 4 MB at 1024 pairs.  Up to 64 pairs the
 *                            layout is exactly the one written above.
 *   os2d_dft_inverse         Y -> the layer's activations (bias, ReLU, channel scale, fp16 hi | lo) in the split-half blocked
 *                            buffer, as os2d_fft_inverse; the kernel also writes the ZERO BORDERS of the planes it fills (the
 *                            buffer may hold anything before the call)                                                     */
int os2d_dft_sizes(int H, int W, int* P, int* Q, int* nbins, int* tiles);
int os2d_dft_channel_stride(int C);
size_t os2d_dft_matrices_bytes(int P, int Q);
int os2d_dft_matrices_build(const double* twP64, const double* twQ64, int P, int Q, void* out, void* stream);
int os2d_dft_forward(const float* corr, const float* inv_norm, float* X, const void* matrices, int NB, int C, int H, int W,
                     void* stream);
int os2d_dft_inverse(const float* Y, const float* packed_b, void* out, const void* matrices, int NB, int Cout, int H, int W,
                     int* status, void* stream);
int os2d_spectral_weights_build_dft(const double* wfold, const double* twP64, const double* twQ64, int C, int Cout, int P, int Q,
                                    int nbins, void* out, void* workspace, void* stream);
int os2d_spectral_gemm_f16_quads(const void* w16, const float* X, float* Y, int NB, int C, int Cout, int nbins, float xscale,
                                 void* stream);
float os2d_dft_xscale(int H, int W);

/* ---- detection over a whole image pyramid: reference os2d/modeling/box_coder.py:448-536 per label for L levels, incl. the
 * reference's memory-bounded NMS (os2d/structures/bounding_box.py:343-374: lists longer than nms_max_batch are NMS-ed in
 * consecutive chunks of the list, survivors concatenated, repeated until one chunk is left or a pass removes nothing; final
 * sort by score) - decisions identical to the reference's, one launch per pass with a work-group per (chunk, class).
 *   loc / cls   HOST arrays of L device pointers: loc[l] [B,4,H_l*W_l], cls[l] [B,H_l*W_l];  hw HOST int [L][2] = (H_l, W_l)
 *   corners     NULL, or a HOST array of L device pointers [B,8,H_l*W_l] (transform corners, box_coder.py:493-503)
 *   img_wh      HOST float [L][2]: level image size (clip);  scale_xy HOST float [L][2]: level -> output image (BoxList.resize)
 *   passes      NMS passes to launch (3 covers 39,580 -> a few thousand -> done); classes that would need more get
 *               out_count = -1 and are counted in *unfinished (device int): the caller re-runs those through
 *               os2d_decode_boxes + os2d_nms (the Python binding does)
 *   outputs     [B,N,...] with N = sum of H_l*W_l: the first out_count[b] entries of a class are its detections by decreasing
 *               score; out_index = candidate number in list order (level offsets as in hw); out_default [B,N,4] the
 *               detection's anchor and out_corners [B,N,8] (NULL iff corners is) its transform corners, both mapped to the
 *               output image like the box
 * Limits (os2d_detect_pyramid_supported): L <= 16, nms_max_batch <= 12288, at most 64 chunks per class.                   */
#define OS2D_PYRAMID_MAX_LEVELS 16
int os2d_detect_pyramid_supported(int L, int N, int nms_max_batch);
int os2d_detect_pyramid_workspace_bytes(int B, int N, int passes, size_t* bytes);
int os2d_detect_pyramid(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                        const int* hw, int stride, int rec_field, const float* img_wh, const float* scale_xy,
                        float score_threshold, float iou_threshold, int nms_max_batch, int passes, float* out_boxes,
                        float* out_scores, int* out_index, float* out_default, float* out_corners, int* out_count,
                        int* unfinished, void* workspace, size_t workspace_bytes, void* stream);
/* The same with MERGED LABELS - several head rows per label (class-image views: reference os2d/engine/evaluate.py:241-269
 * builds 4 - 8 rows per class, box_coder.py:483-487 merges the rows that carry one class id before NMS, evaluate.py:294).
 *   G labels of at most V rows;  slot_rows DEVICE int [G][V]: head row (< B) of view v of label g, -1 = no such view.
 * A label's candidate list is its rows in slot order, each row level by level (the reference's order); outputs are
 * [G, V*N, ...], out_index = slot * N + candidate number of that row.  Limits as above with V*N candidates per label.   */
int os2d_detect_pyramid_merged(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                               const int* hw, int stride, int rec_field, const float* img_wh, const float* scale_xy,
                               float score_threshold, float iou_threshold, int nms_max_batch, int passes, int G, int V,
                               const int* slot_rows, float* out_boxes, float* out_scores, int* out_index, float* out_default,
                               float* out_corners, int* out_count, int* unfinished, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Both of the above with a box transform chain per level (see OS2D_BOX_OP_*) instead of the scale table: op_counts HOST int [L],
 * op_kinds HOST int [L][OS2D_BOX_MAX_OPS], op_args HOST float [L][OS2D_BOX_MAX_OPS][2]; the transform corners go through the
 * same chain (as the two "boxes" (x0, y0, x1, y1), (x2, y2, x3, y3), like the reference's box_coder.py:439-446).  The anchors
 * (out_default) have a chain of their own, default_op_* with OS2D_BOX_MAX_DEFAULT_OPS entries per level: in the reference the
 * anchors ride along as a BoxList field of the boxes - BoxList.transpose / crop transform such fields as well, resize does not
 * (bounding_box.py:162,196-199,222-225) - and the level's transform is applied to the field once more afterwards
 * (box_coder.py:515-516); for a resize-only chain (the evaluation's) the two chains are equal.
 * slot_rows NULL: every head row its own label (G, V ignored); otherwise as os2d_detect_pyramid_merged.                      */
int os2d_detect_pyramid_ops(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                            const int* hw, int stride, int rec_field, const float* img_wh, const int* op_counts,
                            const int* op_kinds, const float* op_args, const int* default_op_counts,
                            const int* default_op_kinds, const float* default_op_args, float score_threshold, float iou_threshold,
                            int nms_max_batch, int passes, int G, int V, const int* slot_rows, float* out_boxes,
                            float* out_scores, int* out_index, float* out_default, float* out_corners, int* out_count,
                            int* unfinished, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OS2D_HIP_H */

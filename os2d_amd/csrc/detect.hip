// Fused per-class detection of ONE pyramid level (gfx950): box decode + clip + score / empty-box filter + sort by score
// + greedy NMS + compaction in a single launch, one 1024-thread work-group per class (a class fills the LDS of a CU, so
// its 16 waves are the only ones there: four per SIMD hide the latencies of the decode and sort phases).  Replaces, for the single-level
// case, the chain decode_boxes -> where -> stable argsort (37 rocPRIM launches) -> gather -> nms -> second sort of
// Os2dBoxCoder.decode_pyramid (reference os2d/modeling/box_coder.py:448-536: build_boxes_from_loc_scores :319-330,
// clip / remove-empty / score mask :489-497, nms :526-528 with os2d/structures/bounding_box.py:344-387, sort :431-437).
//
// Everything of a class lives in LDS: the sort keys as two arrays (32-bit descending-score key + 16-bit location), the
// decoded boxes in sorted order (16 B each) and the sorted positions of the kept boxes (2 B each); after the sort the
// score-key array is dead and caches the first pow2/4 kept boxes: 6*pow2(HW) + 18*HW bytes = 134 KB at 60x80.
//   Ascending (score key, location) = decreasing score with ties in location order - the order of a STABLE descending
//   sort, which is what the generic path (torch.argsort stable) produces.
// Sort: bitonic network over the next power of two, three compare-exchange levels per LDS round trip (a thread loads
// the 8 elements that differ in 3 index bits, exchanges them in registers and stores them back): 35 round trips
// instead of 91 passes at 8192 keys.
// NMS runs over 64 candidates per step like nms_kernel (nms.hip) but candidates and kept boxes come from LDS and nothing
// is written to global memory inside the loop, so the serial chain of a class has no memory latency in it; the
// survivors are written out by all threads at the end.
// Outputs are compacted per class in decreasing-score order: out_boxes/out_scores/out_index[c][0 .. out_count[c]).
#include "os2d_common.h"

namespace {

typedef unsigned long long u64;

constexpr int NTHR = 1024, NWAVE = NTHR / 64;

__global__ __launch_bounds__(NTHR) void detect_level_kernel(const float* __restrict__ loc,  // [B][4][HW]
                                                           const float* __restrict__ cls,  // [B][HW]
                                                           int H, int W, float stride, float half_box, float img_w,
                                                           float img_h, Os2dBoxOps ops, float score_thr,
                                                           float iou_thr, int NP2, float4* out_boxes,
                                                           float* __restrict__ out_scores, int* __restrict__ out_index,
                                                           int* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HW = H * W;
  unsigned int* skey = reinterpret_cast<unsigned int*>(smem);                               // [NP2] score keys ...
  float4* kbox = reinterpret_cast<float4*>(smem);                                            // ... then kept boxes [NP2/4]
  unsigned short* sidx = reinterpret_cast<unsigned short*>(smem + (size_t)NP2 * 4);         // [NP2] locations
  unsigned short* kpos = sidx + NP2;                                                         // [HW] sorted positions kept
  float4* sbox = reinterpret_cast<float4*>(smem + (size_t)NP2 * 6 + (((size_t)HW * 2 + 15) & ~(size_t)15));  // [HW]
  __shared__ unsigned int vote[NWAVE][64];  // per wave and candidate: killed by that wave's share of the kept list
  __shared__ int n_valid, kept_count;

  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* lc = loc + (size_t)c * 4 * HW;
  const float* sc = cls + (size_t)c * HW;
  if (tid == 0) {
    n_valid = 0;
    kept_count = 0;
  }
  __syncthreads();

  // ---- 1. keys: valid = score > threshold (false for NaN) and the clipped box is not empty
  int mine = 0;
  for (int i = tid; i < NP2; i += NTHR) {
    unsigned int key = 0xffffffffu;  // invalid entries sort to the end (a valid key is ~u with u >= 0x00800000)
    if (i < HW) {
      const float4 b = os2d_decode_box(lc + i, HW, i, W, stride, half_box, img_w, img_h);
      const float s = sc[i];
      const bool empty = (b.w <= b.y) || (b.z <= b.x);
      if (s > score_thr && !empty) {
        unsigned int u = (s == 0.f) ? 0u : __float_as_uint(s);  // -0 and +0 tie in a comparison sort
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // monotone map float -> uint (ascending)
        key = ~u;                                              // ascending key = descending score
        ++mine;
      }
    }
    skey[i] = key;
    sidx[i] = (unsigned short)i;
  }
  if (mine) atomicAdd(&n_valid, mine);
  __syncthreads();
  const int n = n_valid;

  // ---- 2. bitonic sort (ascending in (key, location)).  Stage k = 2^m needs the levels j = 2^(m-1) .. 1; they are
  // taken NB <= 3 at a time: a thread owns the 2^NB elements whose indices differ in bits lo .. lo+NB-1.
#define DET_SORT_CHUNK(NB)                                                                                        \
  {                                                                                                               \
    const int groups_ = NP2 >> (NB);                                                                              \
    for (int g_ = tid; g_ < groups_; g_ += NTHR) {                                                                \
      const int base_ = ((g_ >> lo) << (lo + (NB))) | (g_ & ((1 << lo) - 1));                                     \
      const bool up_ = (base_ & k) == 0;                                                                          \
      unsigned int kk_[1 << (NB)];                                                                                \
      unsigned short ii_[1 << (NB)];                                                                              \
      _Pragma("unroll") for (int e = 0; e < (1 << (NB)); ++e) {                                                   \
        kk_[e] = skey[base_ | (e << lo)];                                                                         \
        ii_[e] = sidx[base_ | (e << lo)];                                                                         \
      }                                                                                                           \
      _Pragma("unroll") for (int b = (NB)-1; b >= 0; --b) {                                                       \
        _Pragma("unroll") for (int e = 0; e < (1 << (NB)); ++e) {                                                 \
          if ((e >> b) & 1) continue;                                                                             \
          const int f = e | (1 << b);                                                                             \
          const bool gt_ = kk_[e] > kk_[f] || (kk_[e] == kk_[f] && ii_[e] > ii_[f]);                              \
          if (gt_ == up_) {                                                                                       \
            const unsigned int tk_ = kk_[e];                                                                      \
            kk_[e] = kk_[f];                                                                                      \
            kk_[f] = tk_;                                                                                         \
            const unsigned short ti_ = ii_[e];                                                                    \
            ii_[e] = ii_[f];                                                                                      \
            ii_[f] = ti_;                                                                                         \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
      _Pragma("unroll") for (int e = 0; e < (1 << (NB)); ++e) {                                                   \
        skey[base_ | (e << lo)] = kk_[e];                                                                         \
        sidx[base_ | (e << lo)] = ii_[e];                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    __syncthreads();                                                                                              \
  }
  for (int m = 1; (1 << m) <= NP2; ++m) {
    const int k = 1 << m;
    for (int hi = m - 1; hi >= 0; hi -= 3) {  // levels hi .. max(hi-2, 0)
      const int nb = min(3, hi + 1);
      const int lo = hi - nb + 1;
      if (nb == 3) DET_SORT_CHUNK(3)
      else if (nb == 2) DET_SORT_CHUNK(2)
      else DET_SORT_CHUNK(1)
    }
  }
#undef DET_SORT_CHUNK

  // ---- 3. boxes in sorted order, mapped to the output image (the level's chain of BoxList.resize / transpose / crop)
  for (int jx = tid; jx < n; jx += NTHR) {
    const int i = sidx[jx];
    sbox[jx] = os2d_apply_box_ops(os2d_decode_box(lc + i, HW, i, W, stride, half_box, img_w, img_h), ops);
  }
  __syncthreads();  // from here on the score keys are dead: their storage holds the kept boxes

  // ---- 4. greedy NMS, 64 candidates per step: all 16 waves test the candidates against their share of the kept list;
  // wave 0 then resolves the candidates that are still alive in order - the first one is kept, every later alive
  // candidate that overlaps it dies (one IoU test over the 64 lanes per box kept in this step), and so on.
  const int kcap = NP2 >> 2;  // kept boxes cached where the score keys were; later ones are found through kpos
  for (int base = 0; base < n; base += 64) {
    const int nk = kept_count;
    const int idx = base + lane;
    const bool valid = idx < n;
    const float4 me = valid ? sbox[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float my_area = os2d_box_area(me);
    unsigned int v = 0u;
    const int nk_lds = min(nk, kcap);
    for (int j = wv; j < nk_lds; j += NWAVE) {
      const float4 kb = kbox[j];
      v |= os2d_iou_gt(kb, os2d_box_area(kb), me, my_area, iou_thr) ? 1u : 0u;
    }
    for (int j = kcap + wv; j < nk; j += NWAVE) {
      const float4 kb = sbox[kpos[j]];
      v |= os2d_iou_gt(kb, os2d_box_area(kb), me, my_area, iou_thr) ? 1u : 0u;
    }
    vote[wv][lane] = v;
    __syncthreads();
    if (wv == 0) {
      unsigned int dead = valid ? 0u : 1u;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) dead |= vote[w][lane];
      u64 alive = ~__ballot(dead != 0u);
      u64 kbits = 0ull;
      while (alive) {
        const int i = __builtin_ctzll(alive);  // best-scoring candidate still alive: kept
        kbits |= 1ull << i;
        float4 kb;
        kb.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.x), i));
        kb.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.y), i));
        kb.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.z), i));
        kb.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.w), i));
        const bool hit = os2d_iou_gt(kb, os2d_box_area(kb), me, my_area, iou_thr);
        alive &= ~(__ballot(hit) | ((2ull << i) - 1ull));  // drop lanes 0..i and everything the new box suppresses
      }
      if ((kbits >> lane) & 1ull) {
        const int slot = nk + __popcll(kbits & ((1ull << lane) - 1ull));
        if (slot < kcap) kbox[slot] = me;
        kpos[slot] = (unsigned short)idx;
      }
      if (lane == 0) kept_count = nk + __popcll(kbits);
    }
    __syncthreads();
  }

  // ---- 5. survivors, compacted in decreasing score
  const int nkept = kept_count;
  for (int slot = tid; slot < nkept; slot += NTHR) {
    const int pos = kpos[slot];
    const int src = sidx[pos];
    out_boxes[(size_t)c * HW + slot] = sbox[pos];
    out_scores[(size_t)c * HW + slot] = sc[src];
    out_index[(size_t)c * HW + slot] = src;
  }
  if (tid == 0) out_count[c] = nkept;
}

int next_pow2(int v) {
  int p = 8;  // >= 8 keeps the LDS sub-arrays 16-byte aligned
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

// dynamic LDS of one class at this level (0 if the level is too large for the fused kernel)
size_t os2d_detect_level_lds_bytes(int H, int W) {
  const long long HW = (long long)H * W;
  if (H < 1 || W < 1 || HW >= 65536) return 0;
  const size_t np2 = (size_t)next_pow2((int)HW);
  const size_t bytes = np2 * 6 + (((size_t)HW * 2 + 15) & ~(size_t)15) + (size_t)HW * 16;
  return bytes + 4608 <= 160 * 1024 ? bytes : 0;  // 4.5 KB: static arrays of the kernel
}

int os2d_launch_detect_level(const float* loc, const float* cls, int B, int H, int W, int stride, int rec_field,
                             float img_w, float img_h, const Os2dBoxOps& ops, float score_thr, float iou_thr,
                             float* out_boxes, float* out_scores, int* out_index, int* out_count, hipStream_t stream) {
  const size_t lds = os2d_detect_level_lds_bytes(H, W);
  if (lds == 0) {
    os2d_set_error("os2d_detect_level: level %dx%d does not fit the fused kernel's LDS (use decode_boxes + nms)", H, W);
    return -3;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(detect_level_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(detect_level): %s", hipGetErrorString(e));
    return -4;
  }
  const float half_box = 0.5f * (float)(stride * (OS2D_T - 1) + rec_field);
  hipLaunchKernelGGL(detect_level_kernel, dim3(B), dim3(NTHR), lds, stream, loc, cls, H, W, (float)stride, half_box, img_w,
                     img_h, ops, score_thr, iou_thr, next_pow2(H * W), reinterpret_cast<float4*>(out_boxes),
                     out_scores, out_index, out_count);
  e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("detect_level launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

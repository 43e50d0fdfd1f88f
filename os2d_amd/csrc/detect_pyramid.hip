// Per-class detection over a whole image PYRAMID on the device (gfx950): everything reference
// os2d/modeling/box_coder.py:448-536 does per label for several levels - decode + clip (:319-330,
// os2d/structures/bounding_box.py:261-265), empty-box / score mask (:489-497), map to the original image (:499-503), merge
// the levels (:425-437) and the reference's MEMORY-BOUNDED greedy NMS (os2d/structures/bounding_box.py:343-374) - without
// the torch sort / gather / mask chain of the generic path (37 rocPRIM launches per sort at 39,580 boxes per class).
//
// The reference's nms() is not a plain greedy NMS once a label has more than nms_max_batch (10000) candidates, which a
// 7-level pyramid of a 1280x960 image (39,580 locations) always has at the default score threshold of -inf:
//   ids = the valid candidates in LIST order (level by level, location by location)
//   repeat: cut ids into consecutive chunks of nms_max_batch; NMS every chunk on its own (sorted by score inside);
//           ids = the survivors, chunk after chunk, each chunk's in score order
//   until the list needed a single chunk or a pass removed nothing;  finally sort the survivors by score.
// The chunks of a pass are independent, so a pass is ONE launch with a work-group per (chunk, class): at 64 classes the
// first pass runs 256 work-groups of 1024 threads (a class alone would use 64 of the 256 CUs).  Passes are chained on the
// stream without host involvement: every work-group re-derives "is this class finished" from the per-chunk survivor
// counts of the previous two passes, and finished classes fall through.  The host launches a fixed number of passes
// (3 by default: 39,580 -> ~4,000 -> done is the typical sequence) and the finalize kernel reports classes that would need
// more (`unfinished`), for which the caller falls back to the generic path - the result is exact either way.
//
//   pyr_decode    grid (N/256, B):     decode, clip, validity, map to the output image; boxes / scores / sort keys of ALL
//                                      candidates [B][N]
//   pyr_compact   grid (B):            the valid candidates compacted in list order (ids of pass 0)
//   pyr_chunk_nms grid (chunks, B):    one chunk of one class: (key, position) into LDS -> bitonic sort -> greedy NMS 64
//                                      candidates per step (boxes gathered from L2 through the sorted ids, one step ahead)
//   pyr_finalize  grid (B):            final order (one stable sort by score when the last pass had several chunks),
//                                      compacted outputs
//
// MERGED LABELS (class-image views: reference os2d/engine/evaluate.py:241-269 builds 4 - 8 head rows per label, box_coder.py:
// 483-487 merges rows with the same class id before NMS): a label's candidate list is its rows in row order, each row level
// by level.  The kernels work on G labels x V slots: slot_rows[g * V + v] names the head row of view v of label g (-1: none),
// candidate j of label g = (slot j / N, location j % N); everything after pyr_decode just sees G "classes" of V * N candidates.
#include "os2d_common.h"
#include "../../include/os2d_hip.h"

namespace {

typedef unsigned long long u64;

constexpr int NTHR = 1024, NWAVE = NTHR / 64;
constexpr int MAX_LEVELS = OS2D_PYRAMID_MAX_LEVELS;
constexpr int MAX_CHUNKS = 64;      // chunks per class and pass that the count tables hold
constexpr int KCAP = 2048;          // kept boxes cached in LDS (32 KB); later ones are re-read through the id lists

struct DefaultOpsTable {        // per level: the chain the anchors ("default_boxes") go through (os2d_common.h)
  Os2dDefaultBoxOps ops[OS2D_PYRAMID_MAX_LEVELS];
};

struct LevelTable {
  const float* loc[MAX_LEVELS];  // [B][4][HW_l]
  const float* cls[MAX_LEVELS];  // [B][HW_l]
  const float* corners[MAX_LEVELS];  // [B][8][HW_l] or all NULL
  int H[MAX_LEVELS], W[MAX_LEVELS], off[MAX_LEVELS + 1];
  float img_w[MAX_LEVELS], img_h[MAX_LEVELS];
  Os2dBoxOps ops[MAX_LEVELS];    // level -> output image
  int L;
};

__device__ __forceinline__ unsigned int score_key(float s) {
  unsigned int u = (s == 0.f) ? 0u : __float_as_uint(s);  // -0 and +0 tie in a comparison sort
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // monotone map float -> uint (ascending)
  return ~u;                                             // ascending key = descending score
}

// ---- 1a. candidates: decode, clip, validity, map to the output image (one thread per candidate)
__global__ __launch_bounds__(256) void pyr_decode_kernel(LevelTable T, int N, float stride, float half_box, float score_thr,
                                                         float4* __restrict__ boxes, float* __restrict__ scores,
                                                         unsigned int* __restrict__ keys, const int* __restrict__ slot_rows) {
  // blockIdx.y = slot (label * V + view); its candidates are written at [slot][N], i.e. label-major [G][V * N]
  const int slot = blockIdx.y;
  const int b = slot_rows ? slot_rows[slot] : slot;      // head row the slot reads
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  if (b < 0) {                                           // a label with fewer views than V: no candidates here
    keys[(size_t)slot * N + g] = 0xffffffffu;
    return;
  }
  int l = 0;
  while (l + 1 < T.L && g >= T.off[l + 1]) ++l;
  const int n = g - T.off[l], HW = T.H[l] * T.W[l];
  float4 bx = os2d_decode_box(T.loc[l] + (size_t)b * 4 * HW + n, HW, n, T.W[l], stride, half_box, T.img_w[l], T.img_h[l]);
  const float s = T.cls[l][(size_t)b * HW + n];
  const bool empty = (bx.w <= bx.y) || (bx.z <= bx.x);
  const bool valid = s > score_thr && !empty;   // false for NaN scores
  bx = os2d_apply_box_ops(bx, T.ops[l]);
  boxes[(size_t)slot * N + g] = bx;
  scores[(size_t)slot * N + g] = s;
  keys[(size_t)slot * N + g] = valid ? score_key(s) : 0xffffffffu;   // a valid key is never 0xffffffff (that would be score -NaN)
}

// ---- 1b. the valid candidates of a class, compacted in list order (ballot + prefix, NTHR candidates per round)
__global__ __launch_bounds__(NTHR) void pyr_compact_kernel(int N, int M, const unsigned int* __restrict__ keys,
                                                          int* __restrict__ ids0, int* __restrict__ counts,
                                                          int* __restrict__ final_pass) {
  __shared__ int wave_total[NWAVE];
  __shared__ int running;
  const int b = blockIdx.x, B = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < N; base += NTHR) {
    const int g = base + tid;
    const bool valid = g < N && keys[(size_t)b * N + g] != 0xffffffffu;
    const u64 m = __ballot(valid);
    if (lane == 0) wave_total[wv] = __popcll(m);
    __syncthreads();
    int before = running;
    for (int w = 0; w < wv; ++w) before += wave_total[w];
    if (valid) ids0[(size_t)b * N + before + __popcll(m & ((1ull << lane) - 1ull))] = g;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < NWAVE; ++w) t += wave_total[w];
      running += t;
    }
    __syncthreads();
  }
  // input of pass 0: the compacted list as consecutive chunks of M
  const int total = running;
  for (int c = tid; c < MAX_CHUNKS; c += NTHR) counts[((size_t)0 * B + b) * MAX_CHUNKS + c] = max(0, min(M, total - c * M));
  if (tid == 0) final_pass[b] = -1;
}

// concatenated list of a pass input: chunk c holds cnt[c] ids at ids[c * M ..]; element j -> id
__device__ __forceinline__ int list_at(const int* __restrict__ ids, const int* cnt_lds, int M, int j) {
  int c = 0;
  while (j >= cnt_lds[c]) {
    j -= cnt_lds[c];
    ++c;
  }
  return ids[(size_t)c * M + j];
}

// ---- 2. one chunk of one class
__global__ __launch_bounds__(NTHR) void pyr_chunk_nms_kernel(int pass, int N, int M, int NP2 /*of min(M, N)*/, float iou_thr,
                                                            const float4* __restrict__ boxes,
                                                            const unsigned int* __restrict__ keys,
                                                            const int* __restrict__ ids_in, int* __restrict__ ids_out,
                                                            int* __restrict__ sorted_ids /*[B][N] scratch*/,
                                                            int* __restrict__ counts, int* __restrict__ final_pass) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* skey = reinterpret_cast<unsigned int*>(smem);                        // [NP2]
  unsigned short* spos = reinterpret_cast<unsigned short*>(smem + (size_t)NP2 * 4);  // [NP2] position inside the chunk
  unsigned short* kpos = spos + NP2;                                                 // [M] sorted positions kept
  float4* kbox = reinterpret_cast<float4*>(smem + (size_t)NP2 * 6 + (((size_t)M * 2 + 15) & ~(size_t)15));  // [KCAP]
  __shared__ unsigned int vote[NWAVE][64];
  __shared__ int cnt_in[MAX_CHUNKS], cnt_prev[MAX_CHUNKS];
  __shared__ int kept_count;
  __shared__ unsigned short surv[NTHR];   // positions (inside the window) of a batch's survivors, in sorted order
  __shared__ int wave_cnt[NWAVE];

  const int ch = blockIdx.x, b = blockIdx.y, B = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int* cnt_out = counts + ((size_t)(pass + 1) * B + b) * MAX_CHUNKS;
  if (final_pass[b] >= 0) return;  // finished in an earlier pass
  for (int c = tid; c < MAX_CHUNKS; c += NTHR) {
    cnt_in[c] = counts[((size_t)pass * B + b) * MAX_CHUNKS + c];
    cnt_prev[c] = pass > 0 ? counts[((size_t)(pass - 1) * B + b) * MAX_CHUNKS + c] : 0;
  }
  if (tid == 0) kept_count = 0;
  __syncthreads();
  int total = 0, total_prev = 0;
  for (int c = 0; c < MAX_CHUNKS; ++c) {
    total += cnt_in[c];
    total_prev += cnt_prev[c];
  }
  if (pass > 0) {
    // did pass-1 finish the class?  (it needed one chunk, or it removed nothing: bounding_box.py:371-372)
    const int nb_prev = (total_prev + M - 1) / M;
    if (nb_prev <= 1 || total == total_prev) {
      if (ch == 0 && tid == 0) final_pass[b] = pass - 1;
      return;
    }
  }
  const int nb = (total + M - 1) / M;
  if (ch >= nb) {
    if (tid == 0 && ch < MAX_CHUNKS) cnt_out[ch] = 0;
    return;
  }
  const int j0 = ch * M, n = min(M, total - j0);
  const unsigned int* kb = keys + (size_t)b * N;
  const float4* bx = boxes + (size_t)b * N;
  const int* in = ids_in + (size_t)b * N;
  int* srt = sorted_ids + (size_t)b * N + j0;

  // ---- keys of the chunk (sorted over the next power of two of ITS size: a pass over a few hundred survivors must not
  // pay for 16384 keys)
  const int NP2full = NP2;
  NP2 = 64;
  while (NP2 < n) NP2 <<= 1;
  for (int i = tid; i < NP2; i += NTHR) {
    unsigned int key = 0xffffffffu;
    if (i < n) key = kb[list_at(in, cnt_in, M, j0 + i)];
    skey[i] = key;
    spos[i] = (unsigned short)i;
  }
  __syncthreads();

  // ---- bitonic sort, ascending in (key, position): decreasing score, ties in list order (a STABLE descending sort)
#define PYR_SORT_CHUNK(NB)                                                                                        \
  {                                                                                                               \
    const int groups_ = NP2 >> (NB);                                                                              \
    for (int g_ = tid; g_ < groups_; g_ += NTHR) {                                                                \
      const int base_ = ((g_ >> lo) << (lo + (NB))) | (g_ & ((1 << lo) - 1));                                     \
      const bool up_ = (base_ & k) == 0;                                                                          \
      unsigned int kk_[1 << (NB)];                                                                                \
      unsigned short ii_[1 << (NB)];                                                                              \
      _Pragma("unroll") for (int e = 0; e < (1 << (NB)); ++e) {                                                   \
        kk_[e] = skey[base_ | (e << lo)];                                                                         \
        ii_[e] = spos[base_ | (e << lo)];                                                                         \
      }                                                                                                           \
      _Pragma("unroll") for (int bb = (NB)-1; bb >= 0; --bb) {                                                    \
        _Pragma("unroll") for (int e = 0; e < (1 << (NB)); ++e) {                                                 \
          if ((e >> bb) & 1) continue;                                                                            \
          const int f = e | (1 << bb);                                                                            \
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
        spos[base_ | (e << lo)] = ii_[e];                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    __syncthreads();                                                                                              \
  }
  for (int m = 1; (1 << m) <= NP2; ++m) {
    const int k = 1 << m;
    for (int hi = m - 1; hi >= 0; hi -= 3) {
      const int nbits = min(3, hi + 1);
      const int lo = hi - nbits + 1;
      if (nbits == 3) PYR_SORT_CHUNK(3)
      else if (nbits == 2) PYR_SORT_CHUNK(2)
      else PYR_SORT_CHUNK(1)
    }
  }
#undef PYR_SORT_CHUNK

  // ---- candidate ids in sorted order (global scratch: read back 64 at a time by the NMS loop)
  for (int i = tid; i < n; i += NTHR) srt[i] = list_at(in, cnt_in, M, j0 + spos[i]);
  __syncthreads();

  // ---- greedy NMS in sorted order.  The candidates' boxes are staged in sorted order through a window in LDS (the storage of
  // the sort keys, dead by now): all threads gather WIN boxes at once (two dependent global loads each, many in flight).
  // Round 3: a window is worked off in BATCHES of 1024 candidates - (1) every thread tests its own candidate against the
  // whole kept list (broadcast reads of the kept boxes: no cross-lane traffic, no barrier per 64 candidates); most
  // candidates die here; (2) the batch's survivors are compacted in order and resolved 64 at a time the way detect.hip does
  // (votes only against the boxes kept SINCE the batch started, in-order resolve in wave 0).  Exactly the greedy decisions
  // of the 64-per-step loop of round 2, which paid two work-group barriers per 64 candidates whether or not any of them was
  // still alive: 157 steps per 10,000-candidate chunk, now ~16 for the first batch (empty kept list) + ~1 per later batch.
  float4* wbox = reinterpret_cast<float4*>(smem);   // [WIN] = NP2full * 4 bytes / 16
  const int WIN = NP2full >> 2;
  for (int w0 = 0; w0 < n; w0 += WIN) {
    const int wn = min(WIN, n - w0);
    __syncthreads();
    for (int i = tid; i < wn; i += NTHR) wbox[i] = bx[srt[w0 + i]];
    __syncthreads();
    for (int base = 0; base < wn; base += NTHR) {
      const int nk0 = kept_count;                       // kept before this batch (uniform: read right after a barrier)
      const int ci = base + tid;
      const bool in_batch = ci < wn;
      const float4 me = in_batch ? wbox[ci] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float my_area = os2d_box_area(me);
      bool dead = !in_batch;
      const int nk0_lds = min(nk0, KCAP);
      for (int j = 0; j < nk0_lds; ++j) {
        const float4 kbx = kbox[j];
        dead |= os2d_iou_gt(kbx, os2d_box_area(kbx), me, my_area, iou_thr);
      }
      for (int j = KCAP; j < nk0; ++j) {
        const float4 kbx = bx[srt[kpos[j]]];
        dead |= os2d_iou_gt(kbx, os2d_box_area(kbx), me, my_area, iou_thr);
      }
      const u64 alive_m = __ballot(!dead);
      if (lane == 0) wave_cnt[wv] = __popcll(alive_m);
      __syncthreads();
      int before = 0, ns = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) {
        const int c = wave_cnt[w];
        before += w < wv ? c : 0;
        ns += c;
      }
      if (!dead) surv[before + __popcll(alive_m & ((1ull << lane) - 1ull))] = (unsigned short)ci;
      __syncthreads();
      for (int s0 = 0; s0 < ns; s0 += 64) {
        const int nk = kept_count;
        const bool valid = s0 + lane < ns;
        const int cj = valid ? (int)surv[s0 + lane] : 0;
        const int idx = w0 + cj;
        const float4 cand = valid ? wbox[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float cand_area = os2d_box_area(cand);
        unsigned int v = 0u;
        const int nk_lds = min(nk, KCAP);
        for (int j = nk0 + wv; j < nk_lds; j += NWAVE) {                 // only the boxes kept since the batch started
          const float4 kbx = kbox[j];
          v |= os2d_iou_gt(kbx, os2d_box_area(kbx), cand, cand_area, iou_thr) ? 1u : 0u;
        }
        for (int j = max(nk0, KCAP) + wv; j < nk; j += NWAVE) {
          const float4 kbx = bx[srt[kpos[j]]];
          v |= os2d_iou_gt(kbx, os2d_box_area(kbx), cand, cand_area, iou_thr) ? 1u : 0u;
        }
        vote[wv][lane] = v;
        __syncthreads();
        if (wv == 0) {
          unsigned int dd = valid ? 0u : 1u;
#pragma unroll
          for (int w = 0; w < NWAVE; ++w) dd |= vote[w][lane];
          u64 alive = ~__ballot(dd != 0u);
          u64 kbits = 0ull;
          while (alive) {
            const int i = __builtin_ctzll(alive);
            kbits |= 1ull << i;
            float4 kbx;
            kbx.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand.x), i));
            kbx.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand.y), i));
            kbx.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand.z), i));
            kbx.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand.w), i));
            const bool hit = os2d_iou_gt(kbx, os2d_box_area(kbx), cand, cand_area, iou_thr);
            alive &= ~(__ballot(hit) | ((2ull << i) - 1ull));
          }
          if ((kbits >> lane) & 1ull) {
            const int slot = nk + __popcll(kbits & ((1ull << lane) - 1ull));
            if (slot < KCAP) kbox[slot] = cand;
            kpos[slot] = (unsigned short)idx;
          }
          if (lane == 0) kept_count = nk + __popcll(kbits);
        }
        __syncthreads();
      }
    }
  }

  // ---- survivors of this chunk in score order
  const int nkept = kept_count;
  int* out = ids_out + (size_t)b * N + j0;
  for (int s = tid; s < nkept; s += NTHR) out[s] = srt[kpos[s]];
  if (tid == 0) cnt_out[ch] = nkept;
}

// ---- 3. final order + compaction
__global__ __launch_bounds__(NTHR) void pyr_finalize_kernel(int passes, int N, int M, int NP2, const float4* __restrict__ boxes,
                                                           const float* __restrict__ scores,
                                                           const unsigned int* __restrict__ keys,
                                                           const int* __restrict__ ids_a, const int* __restrict__ ids_b,
                                                           const int* __restrict__ counts, const int* __restrict__ final_pass,
                                                           float4* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                           int* __restrict__ out_index, int* __restrict__ out_count,
                                                           int* __restrict__ unfinished, LevelTable T, float stride,
                                                           float half_box, float4* __restrict__ out_default,
                                                           float* __restrict__ out_corners, int N1 /*candidates per head row*/,
                                                           int V, const int* __restrict__ slot_rows, DefaultOpsTable D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* skey = reinterpret_cast<unsigned int*>(smem);                        // [NP2]
  unsigned short* spos = reinterpret_cast<unsigned short*>(smem + (size_t)NP2 * 4);  // [NP2]
  __shared__ int cnt_fin[MAX_CHUNKS];
  const int b = blockIdx.x, B = gridDim.x;
  const int tid = threadIdx.x;
  // which pass produced the final list?
  int fp = final_pass[b];
  if (fp < 0) {  // not detected by a later pass: check the last one launched
    int tp = 0, tn = 0;
    for (int c = 0; c < MAX_CHUNKS; ++c) {
      tp += counts[((size_t)(passes - 1) * B + b) * MAX_CHUNKS + c];
      tn += counts[((size_t)passes * B + b) * MAX_CHUNKS + c];
    }
    if ((tp + M - 1) / M <= 1 || tn == tp) fp = passes - 1;
  }
  if (fp < 0) {  // would need more passes than were launched: the caller falls back to the generic path
    if (tid == 0) {
      out_count[b] = -1;
      atomicAdd(unfinished, 1);
    }
    return;
  }
  for (int c = tid; c < MAX_CHUNKS; c += NTHR) cnt_fin[c] = counts[((size_t)(fp + 1) * B + b) * MAX_CHUNKS + c];
  __syncthreads();
  int total = 0, chunks = 0;
  for (int c = 0; c < MAX_CHUNKS; ++c) {
    total += cnt_fin[c];
    chunks += cnt_fin[c] > 0 ? 1 : 0;
  }
  const int* ids = ((fp + 1) & 1) ? ids_b : ids_a;   // pass p reads buffer p&1 and writes (p+1)&1
  ids += (size_t)b * N;
  const bool need_sort = chunks > 1;                 // several chunks, each sorted inside: one stable sort over all
  if (need_sort && total > NP2) {
    if (tid == 0) {
      out_count[b] = -1;
      atomicAdd(unfinished, 1);
    }
    return;
  }
  if (need_sort) {
    for (int i = tid; i < NP2; i += NTHR) {
      skey[i] = i < total ? keys[(size_t)b * N + list_at(ids, cnt_fin, M, i)] : 0xffffffffu;
      spos[i] = (unsigned short)i;
    }
    __syncthreads();
    for (int k = 2; k <= NP2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < NP2; i += NTHR) {
          const int p = i ^ j;
          if (p > i) {
            const bool up = (i & k) == 0;
            const unsigned int ka = skey[i], kb2 = skey[p];
            const unsigned short pa = spos[i], pb = spos[p];
            const bool gt = ka > kb2 || (ka == kb2 && pa > pb);
            if (gt == up) {
              skey[i] = kb2;
              skey[p] = ka;
              spos[i] = pb;
              spos[p] = pa;
            }
          }
        }
        __syncthreads();
      }
  }
  for (int s = tid; s < total; s += NTHR) {
    const int g = list_at(ids, cnt_fin, M, need_sort ? (int)spos[s] : s);
    out_boxes[(size_t)b * N + s] = boxes[(size_t)b * N + g];
    out_scores[(size_t)b * N + s] = scores[(size_t)b * N + g];
    out_index[(size_t)b * N + s] = g;
    // the anchor of the detection (reference box_coder.py:17-59, field "default_boxes") and the corners of its transformed
    // template (field "transform_corners"), mapped to the output image like the box
    const int view = g / N1, g1 = g - view * N1;       // candidate g of the label = (view, candidate of that head row)
    int l = 0;
    while (l + 1 < T.L && g1 >= T.off[l + 1]) ++l;
    const int nloc = g1 - T.off[l], HWl = T.H[l] * T.W[l];
    const int hh = nloc / T.W[l], ww = nloc - hh * T.W[l];
    const float ecx = stride * ((float)ww + 0.5f), ecy = stride * ((float)hh + 0.5f);
    out_default[(size_t)b * N + s] = os2d_apply_box_ops(make_float4(ecx - half_box, ecy - half_box, ecx + half_box, ecy + half_box), D.ops[l]);
    if (out_corners != nullptr) {
      const int row = slot_rows ? slot_rows[b * V + view] : b;
      const float* cp = T.corners[l] + (size_t)row * 8 * HWl + nloc;
      // the reference maps the corners as two "boxes" (x0, y0, x1, y1), (x2, y2, x3, y3) through the same closures
      // (box_coder.py:493-503): a flip therefore exchanges the x (or y) of the two points of a pair - reproduced as is
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 c4 = os2d_apply_box_ops(make_float4(cp[(size_t)(4 * k) * HWl], cp[(size_t)(4 * k + 1) * HWl],
                                                         cp[(size_t)(4 * k + 2) * HWl], cp[(size_t)(4 * k + 3) * HWl]), T.ops[l]);
        reinterpret_cast<float4*>(out_corners + ((size_t)b * N + s) * 8)[k] = c4;
      }
    }
  }
  if (tid == 0) out_count[b] = total;
}

int next_pow2(int v) {
  int p = 8;
  while (p < v) p <<= 1;
  return p;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carve {
  size_t boxes, scores, keys, ids_a, ids_b, sorted, counts, final_pass, total;
};
Carve carve(int B, int N, int passes) {
  Carve c;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align256(off + bytes);
    return o;
  };
  c.boxes = take((size_t)B * N * 16);
  c.scores = take((size_t)B * N * 4);
  c.keys = take((size_t)B * N * 4);
  c.ids_a = take((size_t)B * N * 4);
  c.ids_b = take((size_t)B * N * 4);
  c.sorted = take((size_t)B * N * 4);
  c.counts = take((size_t)(passes + 1) * B * MAX_CHUNKS * 4);
  c.final_pass = take((size_t)B * 4);
  c.total = off;
  return c;
}

int check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("%s launch: %s", what, hipGetErrorString(e));
    return -4;
  }
  return 0;
}

}  // namespace

extern "C" {

int os2d_detect_pyramid_supported(int L, int N, int nms_max_batch) {
  // nms_max_batch <= 12288: (key, position) arrays of the next power of two + kept list + kept-box cache fit the 160 KB LDS
  if (L < 1 || L > MAX_LEVELS || N < 1 || N > (1 << 22) || nms_max_batch < 1 || nms_max_batch > 12288) return 0;
  if ((N + nms_max_batch - 1) / nms_max_batch > MAX_CHUNKS) return 0;
  return 1;
}

int os2d_detect_pyramid_workspace_bytes(int B, int N, int passes, size_t* bytes) {
  if (!bytes || B < 1 || N < 1 || passes < 1 || passes > 16) {
    os2d_set_error("os2d_detect_pyramid_workspace_bytes: bad arguments");
    return -1;
  }
  *bytes = carve(B, N, passes).total;
  return 0;
}

// B head rows in loc / cls / corners; G labels of at most V rows each (slot_rows: DEVICE int [G * V], NULL = every row its
// own label, G = B, V = 1); outputs are [G][V * N]
static int detect_pyramid_impl(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                               const int* hw, int stride, int rec_field, const float* img_wh, const Os2dBoxOps* level_ops,
                               const Os2dDefaultBoxOps* default_ops, float score_threshold, float iou_threshold, int nms_max_batch, int passes, float* out_boxes,
                               float* out_scores, int* out_index, float* out_default, float* out_corners, int* out_count,
                               int* unfinished, void* workspace, size_t workspace_bytes, void* stream, int G, int V,
                               const int* slot_rows) {
  if (!loc || !cls || !hw || !img_wh || !level_ops || !out_boxes || !out_scores || !out_index || !out_default || !out_count ||
      !unfinished || !workspace || B < 1 || G < 1 || V < 1 || stride < 1 || rec_field < 1 || passes < 1 || passes > 16 ||
      ((corners != nullptr) != (out_corners != nullptr)) || (!slot_rows && (G != B || V != 1))) {
    os2d_set_error("os2d_detect_pyramid: bad arguments");
    return -1;
  }
  LevelTable T;
  DefaultOpsTable D;
  int N1 = 0;
  if (L < 1 || L > MAX_LEVELS) {
    os2d_set_error("os2d_detect_pyramid: %d levels (at most %d)", L, MAX_LEVELS);
    return -3;
  }
  for (int l = 0; l < L; ++l) {
    if (!loc[l] || !cls[l] || hw[2 * l] < 1 || hw[2 * l + 1] < 1) {
      os2d_set_error("os2d_detect_pyramid: bad level %d", l);
      return -1;
    }
    T.loc[l] = loc[l];
    T.cls[l] = cls[l];
    T.corners[l] = corners ? corners[l] : nullptr;
    if (corners && !corners[l]) {
      os2d_set_error("os2d_detect_pyramid: corners of level %d missing", l);
      return -1;
    }
    T.H[l] = hw[2 * l];
    T.W[l] = hw[2 * l + 1];
    T.off[l] = N1;
    N1 += hw[2 * l] * hw[2 * l + 1];
    T.img_w[l] = img_wh[2 * l];
    T.img_h[l] = img_wh[2 * l + 1];
    T.ops[l] = level_ops[l];
    D.ops[l] = default_ops[l];
  }
  T.off[L] = N1;
  T.L = L;
  if ((long long)N1 * V > (1 << 22) || !os2d_detect_pyramid_supported(L, N1 * V, nms_max_batch)) {
    os2d_set_error("os2d_detect_pyramid: unsupported size (%d candidates per label, nms_max_batch=%d)", N1 * V, nms_max_batch);
    return -3;
  }
  const int N = N1 * V;                  // candidates per label
  const Carve c = carve(G, N, passes);
  if (workspace_bytes < c.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
    os2d_set_error("os2d_detect_pyramid: workspace too small or not 256-byte aligned (%zu B, need %zu B)", workspace_bytes, c.total);
    return -2;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  float4* boxes = reinterpret_cast<float4*>(ws + c.boxes);
  float* scores = reinterpret_cast<float*>(ws + c.scores);
  unsigned int* keys = reinterpret_cast<unsigned int*>(ws + c.keys);
  int* ids[2] = {reinterpret_cast<int*>(ws + c.ids_a), reinterpret_cast<int*>(ws + c.ids_b)};
  int* sorted = reinterpret_cast<int*>(ws + c.sorted);
  int* counts = reinterpret_cast<int*>(ws + c.counts);
  int* final_pass = reinterpret_cast<int*>(ws + c.final_pass);
  const int M = nms_max_batch;
  const float half_box = 0.5f * (float)(stride * (OS2D_T - 1) + rec_field);

  hipError_t e = hipMemsetAsync(unfinished, 0, sizeof(int), st);
  if (e == hipSuccess) e = hipMemsetAsync(counts, 0, (size_t)(passes + 1) * G * MAX_CHUNKS * sizeof(int), st);
  if (e != hipSuccess) {
    os2d_set_error("hipMemsetAsync: %s", hipGetErrorString(e));
    return -4;
  }
  hipLaunchKernelGGL(pyr_decode_kernel, dim3((N1 + 255) / 256, G * V), dim3(256), 0, st, T, N1, (float)stride, half_box,
                     score_threshold, boxes, scores, keys, slot_rows);
  int rc = check("pyr_decode");
  if (rc) return rc;
  hipLaunchKernelGGL(pyr_compact_kernel, dim3(G), dim3(NTHR), 0, st, N, M, keys, ids[0], counts, final_pass);
  if ((rc = check("pyr_compact"))) return rc;
  const int NP2 = next_pow2(min(M, N));
  const size_t lds = (size_t)NP2 * 6 + (((size_t)M * 2 + 15) & ~(size_t)15) + (size_t)KCAP * 16;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(pyr_chunk_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(pyr_finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)16384 * 6));
  if (e != hipSuccess) {
    os2d_set_error("hipFuncSetAttribute(detect_pyramid): %s", hipGetErrorString(e));
    return -4;
  }
  const int chunks0 = (N + M - 1) / M;
  for (int p = 0; p < passes; ++p) {
    // pass 0 may need every chunk; a later pass works on the survivors: at most as many chunks as the pass before
    hipLaunchKernelGGL(pyr_chunk_nms_kernel, dim3(chunks0, G), dim3(NTHR), lds, st, p, N, M, NP2, iou_threshold, boxes, keys,
                       ids[p & 1], ids[(p + 1) & 1], sorted, counts, final_pass);
    if ((rc = check("pyr_chunk_nms"))) return rc;
  }
  hipLaunchKernelGGL(pyr_finalize_kernel, dim3(G), dim3(NTHR), (size_t)16384 * 6, st, passes, N, M, 16384, boxes, scores, keys,
                     ids[0], ids[1], counts, final_pass, reinterpret_cast<float4*>(out_boxes), out_scores, out_index, out_count,
                     unfinished, T, (float)stride, half_box, reinterpret_cast<float4*>(out_default), out_corners, N1, V,
                     slot_rows, D);
  return check("pyr_finalize");
}

// level -> output image as plain per-axis scales (BoxList.resize): the one-op chains of the original entry points
static bool ops_from_scales(const float* scale_xy, int L, Os2dBoxOps* ops, Os2dDefaultBoxOps* dops) {
  if (!scale_xy || L < 1 || L > MAX_LEVELS) {
    os2d_set_error("os2d_detect_pyramid: bad scale table / level count");
    return false;
  }
  for (int l = 0; l < L; ++l) {
    ops[l] = os2d_box_ops_scale<OS2D_BOX_MAX_OPS>(scale_xy[2 * l], scale_xy[2 * l + 1]);
    dops[l] = os2d_box_ops_scale<OS2D_BOX_MAX_DEFAULT_OPS>(scale_xy[2 * l], scale_xy[2 * l + 1]);
  }
  return true;
}

int os2d_detect_pyramid(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                        const int* hw, int stride, int rec_field, const float* img_wh, const float* scale_xy,
                        float score_threshold, float iou_threshold, int nms_max_batch, int passes, float* out_boxes,
                        float* out_scores, int* out_index, float* out_default, float* out_corners, int* out_count,
                        int* unfinished, void* workspace, size_t workspace_bytes, void* stream) {
  Os2dBoxOps ops[MAX_LEVELS];
  Os2dDefaultBoxOps dops[MAX_LEVELS];
  if (!ops_from_scales(scale_xy, L, ops, dops)) return -1;
  return detect_pyramid_impl(loc, cls, corners, B, L, hw, stride, rec_field, img_wh, ops, dops, score_threshold, iou_threshold,
                             nms_max_batch, passes, out_boxes, out_scores, out_index, out_default, out_corners, out_count,
                             unfinished, workspace, workspace_bytes, stream, B, 1, nullptr);
}

int os2d_detect_pyramid_merged(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                               const int* hw, int stride, int rec_field, const float* img_wh, const float* scale_xy,
                               float score_threshold, float iou_threshold, int nms_max_batch, int passes, int G, int V,
                               const int* slot_rows, float* out_boxes, float* out_scores, int* out_index, float* out_default,
                               float* out_corners, int* out_count, int* unfinished, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!slot_rows) {
    os2d_set_error("os2d_detect_pyramid_merged: slot_rows is required");
    return -1;
  }
  Os2dBoxOps ops[MAX_LEVELS];
  Os2dDefaultBoxOps dops[MAX_LEVELS];
  if (!ops_from_scales(scale_xy, L, ops, dops)) return -1;
  return detect_pyramid_impl(loc, cls, corners, B, L, hw, stride, rec_field, img_wh, ops, dops, score_threshold, iou_threshold,
                             nms_max_batch, passes, out_boxes, out_scores, out_index, out_default, out_corners, out_count,
                             unfinished, workspace, workspace_bytes, stream, G, V, slot_rows);
}

int os2d_detect_pyramid_ops(const float* const* loc, const float* const* cls, const float* const* corners, int B, int L,
                            const int* hw, int stride, int rec_field, const float* img_wh, const int* op_counts,
                            const int* op_kinds, const float* op_args, const int* default_op_counts,
                            const int* default_op_kinds, const float* default_op_args, float score_threshold, float iou_threshold,
                            int nms_max_batch, int passes, int G, int V, const int* slot_rows, float* out_boxes,
                            float* out_scores, int* out_index, float* out_default, float* out_corners, int* out_count,
                            int* unfinished, void* workspace, size_t workspace_bytes, void* stream) {
  if (!op_counts || !default_op_counts || L < 1 || L > MAX_LEVELS) {
    os2d_set_error("os2d_detect_pyramid_ops: bad op tables / level count");
    return -1;
  }
  Os2dBoxOps ops[MAX_LEVELS];
  Os2dDefaultBoxOps dops[MAX_LEVELS];
  for (int l = 0; l < L; ++l)
    if (!os2d_box_ops_from(op_kinds ? op_kinds + (size_t)l * OS2D_BOX_MAX_OPS : nullptr,
                           op_args ? op_args + (size_t)l * OS2D_BOX_MAX_OPS * 2 : nullptr, op_counts[l], &ops[l]) ||
        !os2d_box_ops_from(default_op_kinds ? default_op_kinds + (size_t)l * OS2D_BOX_MAX_DEFAULT_OPS : nullptr,
                           default_op_args ? default_op_args + (size_t)l * OS2D_BOX_MAX_DEFAULT_OPS * 2 : nullptr,
                           default_op_counts[l], &dops[l])) {
      os2d_set_error("os2d_detect_pyramid_ops: bad transform chain of level %d (at most %d / %d ops of kind 1..4)", l,
                     OS2D_BOX_MAX_OPS, OS2D_BOX_MAX_DEFAULT_OPS);
      return -1;
    }
  if (!slot_rows) {      // every head row its own label
    G = B;
    V = 1;
  }
  return detect_pyramid_impl(loc, cls, corners, B, L, hw, stride, rec_field, img_wh, ops, dops, score_threshold, iou_threshold,
                             nms_max_batch, passes, out_boxes, out_scores, out_index, out_default, out_corners, out_count,
                             unfinished, workspace, workspace_bytes, stream, G, V, slot_rows);
}

}  // extern "C"

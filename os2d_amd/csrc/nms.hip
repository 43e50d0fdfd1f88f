// Per-class greedy NMS (gfx950), the step right after the head: replaces the Python loop over classes around
// torchvision.ops.nms in reference os2d/modeling/box_coder.py:425-437,526-528 / os2d/structures/bounding_box.py:344-387.
//
// Semantics = torchvision nms on one score-sorted list: walk boxes by decreasing score, keep a box unless its IoU
// with an already kept box is > thr;  IoU = inter / (area_a + area_b - inter).
//
// One 256-thread work-group per class (classes are independent), no N x N mask matrix:
// candidates are consumed in sorted order 64 at a time;
//   phase 1  all 4 waves test the 64 candidates against the kept list so far (kept boxes stream through L2/LDS),
//   phase 2  the 64 x 64 intra-chunk overlaps are computed by all 4 waves (each lane ends up with the bitmask of EARLIER
//            lanes that overlap it), then wave 0 walks the 64 candidates in order (v_readlane) to decide survivors,
//   phase 3  survivors are appended to the kept list.
// Work ~ N * kept / 256 IoU tests per class instead of N^2 / 2, and nothing but the kept list (<= N boxes) is stored.
#include "os2d_common.h"

namespace {

constexpr int KEPT_LDS = 2048;  // kept boxes cached in LDS per class (32 KB)

__device__ __forceinline__ bool iou_gt(float4 a, float area_a, float4 b, float area_b, float thr) {
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  return inter / (area_a + area_b - inter) > thr;
}

__global__ __launch_bounds__(256) void nms_kernel(const float4* __restrict__ boxes,  // [NC][N] sorted by score desc
                                                  const int* __restrict__ counts,    // [NC]
                                                  int N, float thr, unsigned char* __restrict__ keep,  // [NC][N]
                                                  int* __restrict__ num_keep, float4* __restrict__ kept_ws) {
  __shared__ float4 cand[64];
  __shared__ float4 kept_lds[KEPT_LDS];  // the kept list lives in LDS (broadcast reads); only its tail spills to HBM
  __shared__ int dead[4][64];
  __shared__ int kept_count;

  const int cls = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float4* bx = boxes + (size_t)cls * N;
  float4* kept = kept_ws + (size_t)cls * N;
  unsigned char* kp = keep + (size_t)cls * N;
  const int n = min(counts[cls], N);
  if (tid == 0) kept_count = 0;
  __syncthreads();

  __shared__ unsigned int sup_lo[4][64], sup_hi[4][64];
  float4 nxt = (lane < n) ? bx[lane] : make_float4(0.f, 0.f, 0.f, 0.f);  // candidates of the next chunk, prefetched
  for (int base = 0; base < n; base += 64) {
    const int nk = kept_count;  // kept before this chunk (uniform)
    const int idx = base + lane;
    const bool valid = idx < n;
    const float4 me = nxt;
    if (base + 64 + lane < n) nxt = bx[base + 64 + lane];  // in flight while this chunk is resolved
    const float my_area = (me.z - me.x) * (me.w - me.y);
    if (wv == 0) cand[lane] = me;
    // ---- phase 1: against the kept list, 4 waves take interleaved kept boxes
    int d = 0;
    const int nk_lds = min(nk, KEPT_LDS);
    for (int j = wv; j < nk_lds; j += 4) {  // no early exit: a wave-uniform trip count keeps the loop pipelined
      const float4 k = kept_lds[j];
      d |= iou_gt(k, (k.z - k.x) * (k.w - k.y), me, my_area, thr) ? 1 : 0;
    }
    for (int j = KEPT_LDS + wv; j < nk; j += 4) {
      const float4 k = kept[j];
      d |= iou_gt(k, (k.z - k.x) * (k.w - k.y), me, my_area, thr) ? 1 : 0;
    }
    dead[wv][lane] = d;
    __syncthreads();
    // ---- phase 2a: 64 x 64 intra-chunk overlaps, wave w tests candidates 16w .. 16w+15 against every lane
    {
      unsigned int bits = 0u;
#pragma unroll 4
      for (int t = 0; t < 16; ++t) {
        const int i = wv * 16 + t;
        const float4 o = cand[i];
        const bool hit = (i < lane) && iou_gt(o, (o.z - o.x) * (o.w - o.y), me, my_area, thr);
        bits |= hit ? (1u << t) : 0u;
      }
      // bit t of wave w = candidate 16w+t: waves 0,1 fill the low word, waves 2,3 the high word
      sup_lo[wv][lane] = (wv < 2) ? (bits << (16 * wv)) : 0u;
      sup_hi[wv][lane] = (wv >= 2) ? (bits << (16 * (wv - 2))) : 0u;
    }
    __syncthreads();
    // ---- phase 2b: in-order resolve, wave 0
    if (wv == 0) {
      const bool pre_dead = (dead[0][lane] | dead[1][lane] | dead[2][lane] | dead[3][lane]) != 0 || !valid;
      const unsigned int lo = sup_lo[0][lane] | sup_lo[1][lane], hi = sup_hi[2][lane] | sup_hi[3][lane];
      const unsigned long long alive0 = ~__ballot(pre_dead);  // candidates not killed by the kept list
      unsigned long long kbits = 0ull;
      for (int i = 0; i < 64; ++i) {
        // readlane returns a signed int: go through unsigned int or bit 31 would sign-extend into the high word
        const unsigned long long s = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)hi, i) << 32) |
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)lo, i);
        if (((alive0 >> i) & 1ull) && (s & kbits) == 0ull) kbits |= (1ull << i);
      }
      const bool k = (kbits >> lane) & 1ull;
      if (valid) kp[idx] = k ? 1 : 0;
      // ---- phase 3: append survivors in order
      if (k) {
        const int slot = nk + __popcll(kbits & ((1ull << lane) - 1ull));
        if (slot < KEPT_LDS) kept_lds[slot] = me;
        else kept[slot] = me;
      }
      if (lane == 0) kept_count = nk + __popcll(kbits);
    }
    __syncthreads();
  }
  // entries past the valid count are never kept
  for (int i = n + tid; i < N; i += 256) kp[i] = 0;
  if (tid == 0) num_keep[cls] = kept_count;
}

}  // namespace

int os2d_launch_nms(const float* boxes, const int* counts, int NC, int N, float thr, unsigned char* keep, int* num_keep,
                    void* workspace, hipStream_t stream) {
  hipLaunchKernelGGL(nms_kernel, dim3(NC), dim3(256), 0, stream, reinterpret_cast<const float4*>(boxes), counts, N, thr,
                     keep, num_keep, reinterpret_cast<float4*>(workspace));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    os2d_set_error("nms launch: %s", hipGetErrorString(e));
    return -4;
  }
  return 0;
}

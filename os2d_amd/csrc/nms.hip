// Per-class greedy NMS (gfx950), the step right after the head: replaces the Python loop over classes around
// torchvision.ops.nms in reference os2d/modeling/box_coder.py:425-437,526-528 / os2d/structures/bounding_box.py:344-387.
//
// Semantics = torchvision nms on one score-sorted list: walk boxes by decreasing score, keep a box unless its IoU
// with an already kept box is > thr;  IoU = inter / (area_a + area_b - inter).
//
// One 256-thread work-group per class (classes are independent), no N x N mask matrix:
// candidates are consumed in sorted order 64 at a time;
//   phase 1  all 4 waves test the 64 candidates against the kept list so far (first 2048 kept boxes in LDS, broadcast reads),
//   phase 2  wave 0 resolves the candidates that are still alive in order: the best one is kept, every later alive
//            candidate overlapping it dies (one 64-lane IoU test per box kept in this step), and so on,
//   phase 3  survivors are appended to the kept list.
// Work ~ N * kept / 256 IoU tests per class instead of N^2 / 2, and nothing but the kept list (<= N boxes) is stored.
#include "os2d_common.h"

namespace {

constexpr int KEPT_LDS = 2048;  // kept boxes cached in LDS per class (32 KB)

__global__ __launch_bounds__(256) void nms_kernel(const float4* __restrict__ boxes,  // [NC][N] sorted by score desc
                                                  const int* __restrict__ counts,    // [NC]
                                                  int N, float thr, unsigned char* __restrict__ keep,  // [NC][N]
                                                  int* __restrict__ num_keep, float4* __restrict__ kept_ws) {
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

  float4 nxt = (lane < n) ? bx[lane] : make_float4(0.f, 0.f, 0.f, 0.f);  // candidates of the next chunk, prefetched
  for (int base = 0; base < n; base += 64) {
    const int nk = kept_count;  // kept before this chunk (uniform)
    const int idx = base + lane;
    const bool valid = idx < n;
    const float4 me = nxt;
    if (base + 64 + lane < n) nxt = bx[base + 64 + lane];  // in flight while this chunk is resolved
    const float my_area = os2d_box_area(me);
    // ---- phase 1: against the kept list, 4 waves take interleaved kept boxes
    int d = 0;
    const int nk_lds = min(nk, KEPT_LDS);
    for (int j = wv; j < nk_lds; j += 4) {  // no early exit: a wave-uniform trip count keeps the loop pipelined
      const float4 k = kept_lds[j];
      d |= os2d_iou_gt(k, os2d_box_area(k), me, my_area, thr) ? 1 : 0;
    }
    for (int j = KEPT_LDS + wv; j < nk; j += 4) {
      const volatile float4* kv = kept;  // written by wave 0 in earlier steps: read past this CU's L1
      float4 k;
      k.x = kv[j].x;
      k.y = kv[j].y;
      k.z = kv[j].z;
      k.w = kv[j].w;
      d |= os2d_iou_gt(k, os2d_box_area(k), me, my_area, thr) ? 1 : 0;
    }
    dead[wv][lane] = d;
    __syncthreads();
    // ---- phase 2: in-order resolve, wave 0
    if (wv == 0) {
      const bool pre_dead = (dead[0][lane] | dead[1][lane] | dead[2][lane] | dead[3][lane]) != 0 || !valid;
      unsigned long long alive = ~__ballot(pre_dead);  // candidates not killed by the kept list
      unsigned long long kbits = 0ull;
      while (alive) {
        const int i = __builtin_ctzll(alive);  // best-scoring candidate still alive: kept
        kbits |= 1ull << i;
        float4 kb;
        kb.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.x), i));
        kb.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.y), i));
        kb.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.z), i));
        kb.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.w), i));
        const bool hit = os2d_iou_gt(kb, os2d_box_area(kb), me, my_area, thr);
        alive &= ~(__ballot(hit) | ((2ull << i) - 1ull));  // drop lanes 0..i and everything the new box suppresses
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

// Sliced inference, the merge step as [3P] sahi performs it by default: "GREEDYNMM" (sahi.postprocess.combine
// GreedyNMMPostprocess + greedy_nmm / batched_greedy_nmm + has_match + merge_object_prediction_pair, sahi @ 6455e84 -
// reference requirements.txt:76; reached by pipe.py:186-188, which passes no postprocess_* argument: GREEDYNMM, match metric
// IOS, threshold 0.5, class-aware).  Restated in oracle/post_ref.py (greedy_nmm, greedy_nmm_merge); this kernel makes the
// same decisions with the same arithmetic: the matching pass in fp32 (torch tensors), the absorb test in fp64 (numpy on
// python floats), no FMA contraction (the library is built with -ffp-contract=off).
//
// One workgroup per frame.  Candidates = the per-slice detections (shifted to frame coordinates, clipped to the frame,
// degenerate boxes dropped - sahi's shift + "fix out of image box coords" + "ignore invalid predictions") in sahi's list
// order: slice 0's rows, slice 1's rows, ..., a full-frame pass last.
//   1. sort by (class ascending, score descending, list position ascending)   [64-bit keys, bitonic, LDS]
//   2. matching, per class segment, in score order: a candidate still in the pool becomes a keep and takes out of the pool
//      every later candidate of its class whose inter / min(area) (IOS) or IoU is NOT < threshold  [fp32; parallel over the
//      later candidates, sequential over keeps]
//   3. absorb, one thread per keep: its matched candidates in score order, each only if it still matches the keep's GROWN
//      box (metric > threshold, fp64): box = hull, score = max (the keep's), class = the keep's
//   4. keeps ranked by descending score (ties: sahi's output order) -> out rows.
#pragma once
#include "common.h"
#include "nms.h"

namespace miyolo {

constexpr int kNmmThreads = 1024;
constexpr int kNmmMax = 4096;        // candidates per frame that fit the workgroup's LDS

struct NmmArgs {
  const float* dets;        // [ns][max_det][6] per-slice detections (slice coordinates)
  const int32_t* counts;    // [ns]
  const int32_t* boxes;     // [ns][4] slice origins (x1, y1, ..)
  int32_t ns, max_det, H, W, metric /* 0 IOS, 1 IOU */, agnostic, max_out;
  float thr;
  float* out_dets;          // [max_out][6]
  int32_t* out_count;       // [2]: rows written; merged boxes in total (> rows written when max_out cut them), or -n when
                            //      n > kNmmMax candidates came in (nothing written)
  int32_t* out_index;       // [max_out] candidate slot (slice * max_det + row) of each row's keep, or null
};

// LDS: keys[4096] u64 | box[4096] float4 | score[4096] f32 | slot[4096] i32 | cls[4096] u16 | owner[4096] i16 | misc
constexpr size_t kNmmLds = (size_t)kNmmMax * (8 + 16 + 4 + 4 + 2 + 2) + 1024;

__device__ __forceinline__ bool nmm_load(const NmmArgs& a, int slot, float (&b)[4], float* score, int* cls) {
  const int sl = slot / a.max_det;
  const float* d = a.dets + (size_t)slot * 6;
  const float ox = (float)a.boxes[sl * 4 + 0], oy = (float)a.boxes[sl * 4 + 1];
  b[0] = fmaxf(0.f, d[0] + ox); b[1] = fmaxf(0.f, d[1] + oy);
  b[2] = fminf((float)a.W, d[2] + ox); b[3] = fminf((float)a.H, d[3] + oy);
  *score = d[4]; *cls = (int)d[5];
  return b[0] < b[2] && b[1] < b[3];
}

__global__ __launch_bounds__(kNmmThreads) void greedy_nmm_kernel(const NmmArgs a) {
  extern __shared__ unsigned char nmm_sm[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(nmm_sm);
  float4* box = reinterpret_cast<float4*>(nmm_sm + (size_t)kNmmMax * 8);
  float* score = reinterpret_cast<float*>(nmm_sm + (size_t)kNmmMax * 24);
  int32_t* slot = reinterpret_cast<int32_t*>(nmm_sm + (size_t)kNmmMax * 28);
  unsigned short* cls = reinterpret_cast<unsigned short*>(nmm_sm + (size_t)kNmmMax * 32);
  short* owner = reinterpret_cast<short*>(nmm_sm + (size_t)kNmmMax * 34);
  int32_t* misc = reinterpret_cast<int32_t*>(nmm_sm + (size_t)kNmmMax * 36);      // [0..ns] prefix of counts (ns <= 250), [255] n valid
  const int tid = threadIdx.x;

  if (tid == 0) {
    int acc = 0;
    for (int s = 0; s < a.ns; ++s) { misc[s] = acc; acc += min(max(a.counts[s], 0), a.max_det); }
    misc[a.ns] = acc;
  }
  __syncthreads();
  const int n = misc[a.ns];
  if (n > kNmmMax) {
    if (tid == 0) { a.out_count[0] = 0; a.out_count[1] = -n; }
    return;
  }
  int npad = 1;
  while (npad < n) npad <<= 1;
  // 1. keys in list order.  [63:48] 0xFFFF - class, [47:16] score bits (positive floats order as integers), [15:0] 0xFFFF - position
  for (int i = tid; i < npad; i += kNmmThreads) {
    unsigned long long k = 0ull;
    if (i < n) {
      int s = 0;
      while (misc[s + 1] <= i) ++s;
      const int sl_slot = s * a.max_det + (i - misc[s]);
      float b[4], sc; int c;
      if (nmm_load(a, sl_slot, b, &sc, &c) && sc > 0.f) {
        const unsigned cfield = a.agnostic ? 0xFFFFu : (0xFFFFu - (unsigned)min(max(c, 0), 0xFFFE));
        k = ((unsigned long long)cfield << 48) | ((unsigned long long)__float_as_uint(sc) << 16) | (unsigned long long)(0xFFFFu - (unsigned)i);
      }
    }
    keys[i] = k;
  }
  __syncthreads();
  bitonic_sort_desc(keys, npad, tid, kNmmThreads);
  // sorted arrays; nv = valid candidates (keys != 0 come first)
  if (tid == 0) misc[255] = 0;
  __syncthreads();
  for (int p = tid; p < n; p += kNmmThreads) {
    const unsigned long long k = keys[p];
    if (k != 0ull) {
      const int i = 0xFFFF - (int)(k & 0xFFFFull);
      int s = 0;
      while (misc[s + 1] <= i) ++s;
      const int sl_slot = s * a.max_det + (i - misc[s]);
      float b[4], sc; int c;
      nmm_load(a, sl_slot, b, &sc, &c);
      box[p] = make_float4(b[0], b[1], b[2], b[3]);
      score[p] = sc; slot[p] = sl_slot; cls[p] = (unsigned short)min(max(c, 0), 0xFFFE); owner[p] = -1;
      atomicMax(&misc[255], p + 1);
    }
  }
  __syncthreads();
  const int nv = misc[255];
  // 2. matching (fp32, as torch: w = clamp(min(x2) - max(x1), 0) ...; areas (x2-x1)*(y2-y1); matched iff !(value < thr))
  for (int p = 0; p < nv; ++p) {
    if (owner[p] != -1) continue;                 // uniform: every thread reads the same LDS word after the barrier below
    const float4 S = box[p];
    const float sarea = (S.z - S.x) * (S.w - S.y);
    const unsigned short sc = cls[p];
    for (int q = p + 1 + tid; q < nv; q += kNmmThreads) {
      if (!a.agnostic && cls[q] != sc) break;     // class segments are contiguous
      if (owner[q] != -1) continue;
      const float4 T = box[q];
      const float w = fmaxf(fminf(T.z, S.z) - fmaxf(T.x, S.x), 0.f);
      const float h = fmaxf(fminf(T.w, S.w) - fmaxf(T.y, S.y), 0.f);
      const float inter = w * h;
      const float tarea = (T.z - T.x) * (T.w - T.y);
      const float val = a.metric == 1 ? inter / ((tarea - inter) + sarea) : inter / fminf(tarea, sarea);
      if (!(val < a.thr)) owner[q] = (short)p;
    }
    if (tid == 0) owner[p] = (short)p;            // nobody reads owner[p] inside this iteration (q > p)
    __syncthreads();
  }
  // 3. absorb (fp64, as numpy on python floats: metric > threshold strictly, against the keep's grown box)
  const double thr64 = (double)a.thr;
  for (int p = tid; p < nv; p += kNmmThreads) {
    if (owner[p] != p) continue;
    const float4 S = box[p];
    double bx1 = S.x, by1 = S.y, bx2 = S.z, by2 = S.w;
    const unsigned short sc = cls[p];
    for (int q = p + 1; q < nv; ++q) {
      if (!a.agnostic && cls[q] != sc) break;
      if (owner[q] != p) continue;
      const float4 T = box[q];
      const double tx1 = T.x, ty1 = T.y, tx2 = T.z, ty2 = T.w;
      const double area_k = (bx2 - bx1) * (by2 - by1), area_t = (tx2 - tx1) * (ty2 - ty1);
      const double w = fmax(fmin(bx2, tx2) - fmax(bx1, tx1), 0.0), h = fmax(fmin(by2, ty2) - fmax(by1, ty1), 0.0);
      const double inter = w * h;
      const double val = a.metric == 1 ? inter / (area_k + area_t - inter) : inter / fmin(area_k, area_t);
      if (val > thr64) { bx1 = fmin(bx1, tx1); by1 = fmin(by1, ty1); bx2 = fmax(bx2, tx2); by2 = fmax(by2, ty2); }
    }
    box[p] = make_float4((float)bx1, (float)by1, (float)bx2, (float)by2);   // only this thread touches a keep's entry
  }
  __syncthreads();
  // 4. keeps by descending score; ties in sahi's output order (= sorted position)
  for (int p = tid; p < npad; p += kNmmThreads)
    keys[p] = (p < nv && owner[p] == p) ? (((unsigned long long)__float_as_uint(score[p]) << 16) | (unsigned long long)(0xFFFFu - (unsigned)p)) : 0ull;
  __syncthreads();
  bitonic_sort_desc(keys, npad, tid, kNmmThreads);
  if (tid == 0) misc[254] = 0;
  __syncthreads();
  for (int r = tid; r < nv; r += kNmmThreads) {
    const unsigned long long k = keys[r];
    if (k == 0ull) continue;
    atomicAdd(&misc[254], 1);
    if (r >= a.max_out) continue;
    const int p = 0xFFFF - (int)(k & 0xFFFFull);
    const float4 b = box[p];
    float* o = a.out_dets + (size_t)r * 6;
    o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = score[p]; o[5] = (float)cls[p];
    if (a.out_index) a.out_index[r] = slot[p];
  }
  __syncthreads();
  const int total = misc[254];
  for (int r = total + tid; r < a.max_out; r += kNmmThreads) {       // zero padding behind the rows, as miyolo_detect
    float* o = a.out_dets + (size_t)r * 6;
    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.f;
    if (a.out_index) a.out_index[r] = -1;
  }
  if (tid == 0) { a.out_count[0] = min(total, a.max_out); a.out_count[1] = total; }
}

}  // namespace miyolo

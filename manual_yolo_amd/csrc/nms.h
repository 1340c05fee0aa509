// Score-threshold + IoU NMS on the GPU, bit-compatible with the reference CPU post-process.
//
// Replaces [3P] ultralytics.utils.ops.non_max_suppression (multi_label=False) +
// torchvision.ops.nms (CPU kernel nms_kernel_impl<float>) + scale_boxes/clip_boxes, i.e.
// everything between the Detect output `y` and `Results.boxes` in the reference's
// `model(frame)` (detect.py:541, pipe.py:179, yolo.py:361).
//
// Semantics reproduced exactly (see oracle/post_ref.py, which states them from the source):
//   candidates: max_c y[4+c] > conf (fp32 compare), class = first arg-max
//   order     : score descending, ties by ascending anchor index (stable sort of the
//               anchor-ordered candidate list), capped at max_nms = 30000
//   boxes     : xyxy = (cx - w/2, cy - h/2, cx + w/2, cy + h/2), then + cls*7680 (unless
//               agnostic) BEFORE areas/IoU, all single fp32 operations (no FMA contraction:
//               this translation unit is built with -ffp-contract=off)
//   suppress  : IoU = inter / (area_i + area_j - inter) > thr, strict; thr is passed as the
//               largest float <= the double threshold torchvision compares against
//   output    : first max_det kept boxes in keep order; optional undo of the letterbox.
//
// Structure: (1) prefilter: one thread per anchor scans the nc class scores (coalesced over
// anchors), candidates append a 64-bit key (score bits << 32 | ~anchor) to the image's list
// with one atomic; (2) one workgroup per image sorts its keys (bitonic, in LDS when they fit)
// and wave 0 runs the greedy pass 64 candidates at a time: each lane tests its candidate
// against the kept boxes of its class (per-class chains in LDS), then the 64 survivors are resolved with
// ballot + shuffles; the next chunk's box gathers are issued before the current chunk is resolved.
// The greedy pass stops as soon as max_det boxes are kept.
#pragma once
#include "common.h"

namespace miyolo {

constexpr int kMaxNms = 30000;       // [3P] non_max_suppression(max_nms=30000)
constexpr float kMaxWh = 7680.0f;    // [3P] non_max_suppression(max_wh=7680)
constexpr int kNmsThreads = 1024;
constexpr int kNmsLdsKeys = 16384;   // keys sorted in LDS when the padded count fits (128 KiB)

struct NmsArgs {
  const float* y;                 // [B, 4+nc, A]
  int32_t B, A, nc, max_det, agnostic, P;   // P = key row length (power of two >= A)
  float conf, iou;
  const float* scale;             // [B,5] gain,pad_x,pad_y,orig_w,orig_h or null
  unsigned long long* keys;       // [B, P]
  int32_t* count;                 // [B]  (zeroed before the prefilter)
  int32_t* cls_idx;               // [B, A]
  float* out_dets;                // [B, max_det, 6]
  int32_t* out_counts;            // [B]
  int32_t* out_anchor;            // [B, max_det] or null
  uint32_t cls_mask[8];           // `classes=` filter: bit c set = class c passes (all ones: no filter); nc <= 256 when used
  int32_t use_mask;
};

__global__ __launch_bounds__(256) void nms_prefilter_kernel(const NmsArgs a) {
  const int an = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (an >= a.A) return;
  const float* yb = a.y + (long)b * (4 + a.nc) * a.A + 4L * a.A + an;
  float best = yb[0];
  int j = 0;
  for (int c = 1; c < a.nc; ++c) {
    const float s = yb[(long)c * a.A];
    if (s > best) { best = s; j = c; }
  }
  // [3P] non_max_suppression: `x = x[(x[:, 5:6] == classes).any(1)]` - the best class must be a wanted one, BEFORE
  // the sort / max_nms / NMS / max_det steps (oracle/post_ref.py:130-132)
  if (a.use_mask && !((a.cls_mask[j >> 5] >> (j & 31)) & 1u)) return;
  if (best > a.conf) {
    const int slot = atomicAdd(a.count + b, 1);
    if (slot < a.P)           // always true behind a zeroed counter; see zero_i32_kernel (kernels_misc.h)
      a.keys[(long)b * a.P + slot] =
          ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)an);
    a.cls_idx[(long)b * a.A + an] = j;
  }
}

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_desc(KeyPtr k, int npad, int tid, int nthreads) {
  for (int size = 2; size <= npad; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = k[i], y = k[ixj];
          const bool desc = (i & size) == 0;
          if (desc ? (x < y) : (x > y)) { k[i] = y; k[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea,
                                       float bx1, float by1, float bx2, float by2, float barea, float thr) {
  const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1);
  const float xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (aarea + barea - inter);
  return ovr > thr;
}

constexpr int kNmsHeads = 256;       // per-class kept lists for up to this many classes (more: one list, as agnostic)

// LDS layout of nms_sort_greedy_kernel: kept[max_det][5] floats | next[max_det] | head[kNmsHeads] | keys[kNmsLdsKeys]
__host__ __device__ inline int nms_keys_offset(int max_det) { return ((max_det * 5 * 4 + 15) & ~15) + ((max_det * 4 + 15) & ~15) + kNmsHeads * 4; }

__global__ __launch_bounds__(kNmsThreads) void nms_sort_greedy_kernel(const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nsm[];
  float* kept = reinterpret_cast<float*>(nsm);
  int* nextk = reinterpret_cast<int*>(nsm + ((a.max_det * 5 * 4 + 15) & ~15));
  int* head = nextk + ((a.max_det * 4 + 15) & ~15) / 4;
  unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(nsm + nms_keys_offset(a.max_det));
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = a.count[b];
  if (n > a.A) n = a.A;
  unsigned long long* gkeys = a.keys + (long)b * a.P;
  int npad = 1;
  while (npad < n) npad <<= 1;
  const bool in_lds = npad <= kNmsLdsKeys;
  if (tid < kNmsHeads) head[tid] = -1;
  if (n > 1) {
    if (in_lds) {
      for (int i = tid; i < npad; i += kNmsThreads) lkeys[i] = (i < n) ? gkeys[i] : 0ull;
      __syncthreads();
      bitonic_sort_desc(lkeys, npad, tid, kNmsThreads);
    } else {
      for (int i = n + tid; i < npad; i += kNmsThreads) gkeys[i] = 0ull;
      __threadfence_block();
      __syncthreads();
      bitonic_sort_desc(gkeys, npad, tid, kNmsThreads);
    }
  } else {
    if (n == 1 && tid == 0) lkeys[0] = gkeys[0];
    __syncthreads();
  }
  if (tid >= 64) return;          // greedy pass: wave 0 only
  const int lane = tid;
  if (n > kMaxNms) n = kMaxNms;
  const float* yb = a.y + (long)b * (4 + a.nc) * a.A;
  float* od = a.out_dets + (long)b * a.max_det * 6;
  int32_t* oa = a.out_anchor ? a.out_anchor + (long)b * a.max_det : nullptr;
  float g = 1.f, px = 0.f, py = 0.f, ow = 0.f, oh = 0.f;
  if (a.scale) {
    g = a.scale[b * 5 + 0]; px = a.scale[b * 5 + 1]; py = a.scale[b * 5 + 2];
    ow = a.scale[b * 5 + 3]; oh = a.scale[b * 5 + 4];
  }
  // Kept boxes are chained per class (head[class] -> newest kept box of the class, nextk[] -> the one before it): boxes of
  // different classes are 7680 apart after the class offset, their IoU is 0 and can never exceed the threshold, so a
  // candidate only has to be tested against the kept boxes of ITS class - the same decisions as testing against all of
  // them (oracle/post_ref.py does), at 1/10th of the work on crowded frames.  Agnostic mode (no offset): one chain.
  const bool one_list = a.agnostic || a.nc > kNmsHeads;
  struct Cand { unsigned an; float score, cx, cy, w, h; int ci; bool has; };
  auto fetch = [&](int s) {
    Cand c;
    const int idx = s + lane;
    c.has = idx < n;
    unsigned long long key = 0ull;
    if (c.has) key = (in_lds || n == 1) ? lkeys[idx] : gkeys[idx];
    c.an = c.has ? (0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0u;
    c.score = __uint_as_float((unsigned)(key >> 32));
    c.cx = c.cy = c.w = c.h = 0.f; c.ci = 0;
    if (c.has) {
      c.cx = yb[0L * a.A + c.an]; c.cy = yb[1L * a.A + c.an];
      c.w = yb[2L * a.A + c.an]; c.h = yb[3L * a.A + c.an];
      c.ci = a.cls_idx[(long)b * a.A + c.an];
    }
    return c;
  };
  int nk = 0;
  // the gathers (scattered 4-byte reads of y: an L2 / HBM latency each) of the next three chunks fly while one is resolved
  Cand q0 = fetch(0), q1 = fetch(64), q2 = fetch(128);
  for (int s = 0; s < n && nk < a.max_det; s += 64) {
    const Cand cur = q0;
    q0 = q1; q1 = q2;
    q2 = fetch(s + 192);
    const bool has = cur.has;
    const unsigned an = cur.an;
    const float score = cur.score;
    const float hw = cur.w / 2.0f, hh = cur.h / 2.0f;
    const float x1 = cur.cx - hw, y1 = cur.cy - hh, x2 = cur.cx + hw, y2 = cur.cy + hh;
    const float cf = (float)cur.ci;
    const int li = one_list ? 0 : cur.ci;
    const float c = a.agnostic ? cf * 0.0f : cf * kMaxWh;
    const float ox1 = x1 + c, oy1 = y1 + c, ox2 = x2 + c, oy2 = y2 + c;
    const float area = (ox2 - ox1) * (oy2 - oy1);
    bool alive = has;
    if (one_list) {
      // one chain = every kept box: walk the array instead (wave-uniform addresses: broadcast reads, no pointer chase)
      for (int k = 0; k < nk; ++k) {
        const float* kb = kept + k * 5;
        if (alive && iou_gt(kb[0], kb[1], kb[2], kb[3], kb[4], ox1, oy1, ox2, oy2, area, a.iou)) alive = false;
      }
    } else {
      int k = has ? head[li] : -1;
      while (__any(k >= 0)) {
        if (k >= 0) {
          const float* kb = kept + k * 5;
          if (iou_gt(kb[0], kb[1], kb[2], kb[3], kb[4], ox1, oy1, ox2, oy2, area, a.iou)) { alive = false; k = -1; }
          else k = nextk[k];
        }
      }
    }
    unsigned long long mask = __ballot(alive);
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      const float bx1 = __shfl(ox1, j, 64), by1 = __shfl(oy1, j, 64);
      const float bx2 = __shfl(ox2, j, 64), by2 = __shfl(oy2, j, 64), ba = __shfl(area, j, 64);
      if (lane == j) {
        float* kb = kept + nk * 5;
        kb[0] = ox1; kb[1] = oy1; kb[2] = ox2; kb[3] = oy2; kb[4] = area;
        nextk[nk] = head[li];
        head[li] = nk;
        float rx1 = x1, ry1 = y1, rx2 = x2, ry2 = y2;
        if (a.scale) {
          rx1 = fminf(fmaxf((rx1 - px) / g, 0.f), ow); ry1 = fminf(fmaxf((ry1 - py) / g, 0.f), oh);
          rx2 = fminf(fmaxf((rx2 - px) / g, 0.f), ow); ry2 = fminf(fmaxf((ry2 - py) / g, 0.f), oh);
        }
        float* o = od + nk * 6;
        o[0] = rx1; o[1] = ry1; o[2] = rx2; o[3] = ry2; o[4] = score; o[5] = cf;
        if (oa) oa[nk] = (int32_t)an;
      }
      if (alive && lane > j && iou_gt(bx1, by1, bx2, by2, ba, ox1, oy1, ox2, oy2, area, a.iou)) alive = false;
      ++nk;
      if (nk >= a.max_det) break;
      const unsigned long long above = (j >= 63) ? 0ull : (~0ull << (j + 1));
      mask = __ballot(alive) & above;
    }
  }
  for (int i = nk * 6 + lane; i < a.max_det * 6; i += 64) od[i] = 0.f;
  if (oa) for (int i = nk + lane; i < a.max_det; i += 64) oa[i] = -1;
  if (lane == 0) a.out_counts[b] = nk;
}

}  // namespace miyolo

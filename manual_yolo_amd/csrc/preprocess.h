// Detect pre-processing on the GPU (reference detect.py:541 -> [3P] LetterBox + cv2.resize).
//
// letterbox_kernel: uint8 HWC frame(s) -> aspect-preserving INTER_LINEAR resize + constant pad, uint8 HWC.
// Integer/byte work, bit-exact against the CPU restatement (oracle/pre_ref.py resize_linear_u8, which follows
// OpenCV's 8-bit linear resize - restated from its source, unverified against cv2): per axis
// f = (float)((d + 0.5) * (1 / (dn / sn)) - 0.5),  s = floor(f),  frac = f - s (float), clamped at both ends, coefficients cvRound(frac * 2048) / cvRound((1 - frac) * 2048);
// horizontal pass in int32, vertical pass  ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// One thread per output pixel (3 channels); HBM-bound, every source byte is read ~ (scale^-2) times through L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miyolo {

struct LetterboxArgs {
  const uint8_t* src;     // [B][sh][sw][3]
  uint8_t* dst;           // [B][dh][dw][3]
  int32_t B, sh, sw, dh, dw, top, left, nh, nw, pad;
  double scale_x, scale_y;   // sw / nw, sh / nh (host doubles: the same division the reference does)
};

__device__ __forceinline__ void lb_taps(int d, double scale, int sn, int* s0, int* s1, int* c0, int* c1) {
  const float f = (float)(((double)d + 0.5) * scale - 0.5);     // OpenCV: fx = (float)((dx + 0.5) * scale_x - 0.5)
  int s = (int)floorf(f);                                       //         sx = cvFloor(fx); fx -= sx
  float fr = f - (float)s;
  if (s < 0) { fr = 0.0f; s = 0; }
  if (s >= sn - 1) { fr = 0.0f; s = sn - 1; }
  *c1 = (int)rintf(fr * 2048.0f);
  *c0 = (int)rintf((1.0f - fr) * 2048.0f);
  *s0 = s;
  *s1 = min(s + 1, sn - 1);
}

__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= a.dw || y >= a.dh) return;
  uint8_t* o = a.dst + (((size_t)b * a.dh + y) * a.dw + x) * 3;
  const int xi = x - a.left, yi = y - a.top;
  if (xi < 0 || xi >= a.nw || yi < 0 || yi >= a.nh) {
    o[0] = (uint8_t)a.pad; o[1] = (uint8_t)a.pad; o[2] = (uint8_t)a.pad;
    return;
  }
  int x0, x1, ax0, ax1, y0, y1, by0, by1;
  lb_taps(xi, a.scale_x, a.sw, &x0, &x1, &ax0, &ax1);
  lb_taps(yi, a.scale_y, a.sh, &y0, &y1, &by0, &by1);
  const uint8_t* s = a.src + (size_t)b * a.sh * a.sw * 3;
  const uint8_t* p00 = s + ((size_t)y0 * a.sw + x0) * 3;
  const uint8_t* p01 = s + ((size_t)y0 * a.sw + x1) * 3;
  const uint8_t* p10 = s + ((size_t)y1 * a.sw + x0) * 3;
  const uint8_t* p11 = s + ((size_t)y1 * a.sw + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = (int)p00[c] * ax0 + (int)p01[c] * ax1;
    const int r1 = (int)p10[c] * ax0 + (int)p11[c] * ax1;
    int v = (((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[c] = (uint8_t)v;
  }
}

}  // namespace miyolo

// ----------------------------------------------------------------------------------------------------------------
// crop_resize_kernel: the classifier's input transform on the device, fused with the crop (SURVEY.md 8f rank 1
// second half + rank 2; reference detect.py:100-113 safe_crop, detect.py:121 rank_model(crop) -> the pickled
// torchvision Compose[Resize(64, bilinear, antialias), CenterCrop(64)], i.e. Pillow's 8-bit resample).
// One workgroup per box: out[n] = centre_crop(PIL_resize(frame[y1:y2, x1:x2], short side -> S), S).
// Pillow's algorithm (src/libImaging/Resample.c; restated and checked against PIL in oracle/pre_ref.py):
//   window [int(c - s + .5), int(c + s + .5)) around c = (xx + .5) * scale, s = max(scale, 1), triangle weights
//   normalised in double, 22-bit fixed point (round half away from zero), horizontal pass -> 8-bit intermediate ->
//   vertical pass, each  clip8((2^21 + sum px * k) >> 22).
// Only the 64 output columns / rows that survive the centre crop are computed.  Byte-exact.
namespace miyolo {

constexpr int kCrKMax = 24;          // taps per output sample: ceil(max(scale,1)) * 2 + 1 <= 21 for scale <= 10
constexpr int kCrMaxShort = 640;     // short side of a crop (scale <= 10 at S = 64)

struct CropResizeArgs {
  const uint8_t* frame;   // [H][W][3]
  const int32_t* boxes;   // [n][4] x1, y1, x2, y2 (already clamped: 0 <= x1 < x2 <= W, 0 <= y1 < y2 <= H)
  uint8_t* out;           // [n][S][S][3]
  int32_t H, W, n, S, tmp_rows;
};

// window and fixed-point coefficients of output sample xx of a Pillow bilinear resample in_size -> out_size
__device__ __forceinline__ void pil_coeffs(int in_size, int out_size, int xx, int* xmin_o, int* cnt_o, int* kk) {
  const double scale = (double)((float)in_size - 0.0f) / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = 0.0 + ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > kCrKMax) xmax = kCrKMax;          // cannot happen for scale <= 10 (host checks); keeps the table in bounds
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = ((double)(x + xmin) - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    ww += (a < 1.0) ? 1.0 - a : 0.0;
  }
  for (int x = 0; x < xmax; ++x) {
    double a = ((double)(x + xmin) - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    double w = (a < 1.0) ? 1.0 - a : 0.0;
    if (ww != 0.0) w = w / ww;
    kk[x] = (w < 0.0) ? (int)(-0.5 + w * 4194304.0) : (int)(0.5 + w * 4194304.0);
  }
  *xmin_o = xmin;
  *cnt_o = xmax;
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const CropResizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char crs[];
  const int S = a.S, tid = threadIdx.x, b = blockIdx.x;
  int* hk = reinterpret_cast<int*>(crs);                 // [S][kCrKMax] horizontal coefficients
  int* vk = hk + S * kCrKMax;                            // [S][kCrKMax] vertical
  int* hb = vk + S * kCrKMax;                            // [S][2] xmin, count
  int* vb = hb + S * 2;
  uint8_t* tmp = reinterpret_cast<uint8_t*>(vb + S * 2); // [tmp_rows][S][3] horizontal-pass image

  const int x1 = a.boxes[b * 4 + 0], y1 = a.boxes[b * 4 + 1], x2 = a.boxes[b * 4 + 2], y2 = a.boxes[b * 4 + 3];
  const int cw = x2 - x1, ch = y2 - y1;
  int rw, rh;                                            // size after the short-side resize
  if ((cw <= ch && cw == S) || (ch <= cw && ch == S)) { rw = cw; rh = ch; }
  else if (cw < ch) { rw = S; rh = (int)((double)(S * ch) / (double)cw); }
  else { rw = (int)((double)(S * cw) / (double)ch); rh = S; }
  const int top = (int)rint((double)(rh - S) / 2.0), left = (int)rint((double)(rw - S) / 2.0);   // Python round(): half to even
  const bool need_h = rw != cw, need_v = rh != ch;

  if (tid < S) {
    if (need_h) pil_coeffs(cw, rw, left + tid, &hb[tid * 2], &hb[tid * 2 + 1], hk + tid * kCrKMax);
    else { hb[tid * 2] = left + tid; hb[tid * 2 + 1] = 1; hk[tid * kCrKMax] = 1 << 22; }
  } else if (tid < 2 * S) {
    const int t = tid - S;
    if (need_v) pil_coeffs(ch, rh, top + t, &vb[t * 2], &vb[t * 2 + 1], vk + t * kCrKMax);
    else { vb[t * 2] = top + t; vb[t * 2 + 1] = 1; vk[t * kCrKMax] = 1 << 22; }
  }
  __syncthreads();
  const int rmin = vb[0];                                               // source rows the vertical pass touches
  const int rmax = vb[(S - 1) * 2] + vb[(S - 1) * 2 + 1];
  const int nrows = min(rmax - rmin, a.tmp_rows);

  // horizontal pass: tmp[r][c] for r in [rmin, rmax), c in [0, S)
  for (int it = tid; it < nrows * S; it += 256) {
    const int r = it / S, c = it - r * S;
    const int xmin = hb[c * 2], cnt = hb[c * 2 + 1];
    const uint8_t* sp = a.frame + ((size_t)(y1 + rmin + r) * a.W + (x1 + xmin)) * 3;
    const int* k = hk + c * kCrKMax;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < cnt; ++x) {
      const int kv = k[x];
      s0 += (int)sp[x * 3 + 0] * kv; s1 += (int)sp[x * 3 + 1] * kv; s2 += (int)sp[x * 3 + 2] * kv;
    }
    s0 >>= 22; s1 >>= 22; s2 >>= 22;
    uint8_t* t = tmp + ((size_t)r * S + c) * 3;
    t[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
    t[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
    t[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
  }
  __syncthreads();
  // vertical pass
  uint8_t* o = a.out + (size_t)b * S * S * 3;
  for (int it = tid; it < S * S; it += 256) {
    const int oy = it / S, ox = it - oy * S;
    const int ymin = vb[oy * 2] - rmin, cnt = vb[oy * 2 + 1];
    const int* k = vk + oy * kCrKMax;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int y = 0; y < cnt; ++y) {
      const int kv = k[y];
      const uint8_t* t = tmp + ((size_t)(ymin + y) * S + ox) * 3;
      s0 += (int)t[0] * kv; s1 += (int)t[1] * kv; s2 += (int)t[2] * kv;
    }
    s0 >>= 22; s1 >>= 22; s2 >>= 22;
    o[it * 3 + 0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
    o[it * 3 + 1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
    o[it * 3 + 2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
  }
}

inline size_t crop_resize_lds_bytes(int S, int tmp_rows) {
  return (size_t)S * kCrKMax * 4 * 2 + (size_t)S * 2 * 4 * 2 + (size_t)tmp_rows * S * 3;
}

// ------------------------------------------------------------------------------------
// Sliced ("SAHI-style") inference, SURVEY.md 8f rank 4 (reference pipe.py:43-45,183-194): cut one frame into a batch of
// slices on the device.  out[i] = frame[y1:y2, x1:x2] placed at the top-left of an sh x sw canvas, the rest = pad.
struct SliceArgs {
  const uint8_t* frame;     // [H][W][3]
  const int32_t* boxes;     // [n][4] x1,y1,x2,y2 (inside the frame)
  uint8_t* out;             // [n][sh][sw][3]
  int32_t H, W, n, sh, sw, pad;
};

__global__ __launch_bounds__(256) void slice_batch_kernel(const SliceArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int i = blockIdx.z;
  if (x >= a.sw || y >= a.sh) return;
  const int x1 = a.boxes[i * 4 + 0], y1 = a.boxes[i * 4 + 1], x2 = a.boxes[i * 4 + 2], y2 = a.boxes[i * 4 + 3];
  uint8_t* o = a.out + (((size_t)i * a.sh + y) * a.sw + x) * 3;
  const int fx = x1 + x, fy = y1 + y;
  if (fx < x2 && fy < y2 && fx < a.W && fy < a.H) {
    const uint8_t* p = a.frame + ((size_t)fy * a.W + fx) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  } else {
    o[0] = (uint8_t)a.pad; o[1] = (uint8_t)a.pad; o[2] = (uint8_t)a.pad;
  }
}

// Candidate list -> the [1, 4+nc, n] layout miyolo_nms consumes: the boxes of all slices, shifted into frame coordinates,
// become "anchors" whose only non-zero class score is the detection's confidence, so that the merge step is the SAME
// class-aware NMS (sort, cls*7680 offset, IoU > thr) as the per-slice post-process.
struct MergeArgs {
  const float* dets;        // [ns][max_det][6] per-slice detections (slice coordinates)
  const int32_t* counts;    // [ns]
  const int32_t* boxes;     // [ns][4] slice origins (x1,y1,..)
  float* y;                 // [1][4+nc][cap]
  int32_t ns, max_det, nc, cap;
};

__global__ __launch_bounds__(256) void merge_pack_kernel(const MergeArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;           // candidate slot = slice * max_det + row
  if (idx >= a.cap) return;
  const int sl = idx / a.max_det, r = idx - sl * a.max_det;
  const bool ok = sl < a.ns && r < a.counts[sl];
  float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, cf = 0.f;
  int cls = 0;
  if (ok) {
    const float* d = a.dets + ((size_t)sl * a.max_det + r) * 6;
    const float ox = (float)a.boxes[sl * 4 + 0], oy = (float)a.boxes[sl * 4 + 1];
    x1 = d[0] + ox; y1 = d[1] + oy; x2 = d[2] + ox; y2 = d[3] + oy; cf = d[4]; cls = (int)d[5];
  }
  // xywh such that the NMS kernel's xy -+ wh/2 reproduces the corners: (x1+x2)/2 and x2-x1 are exact halves / differences
  a.y[0 * (size_t)a.cap + idx] = (x1 + x2) / 2.0f;
  a.y[1 * (size_t)a.cap + idx] = (y1 + y2) / 2.0f;
  a.y[2 * (size_t)a.cap + idx] = x2 - x1;
  a.y[3 * (size_t)a.cap + idx] = y2 - y1;
  for (int c = 0; c < a.nc; ++c) a.y[(size_t)(4 + c) * a.cap + idx] = (ok && c == cls) ? cf : 0.f;
}

}  // namespace miyolo

// Detect pre-processing on the GPU (reference detect.py:541 -> [3P] LetterBox + cv2.resize).
//
// letterbox_kernel: uint8 HWC frame(s) -> aspect-preserving INTER_LINEAR resize + constant pad, uint8 HWC.
// Integer/byte work, bit-exact against the CPU restatement (oracle/pre_ref.py resize_linear_u8, which follows
// cv2's 8-bit linear resize): per axis  f = (d + 0.5) * (sn / dn) - 0.5 in double,  s = floor(f),
// frac = float(f - s), clamped at both ends, coefficients cvRound(frac * 2048) / cvRound((1 - frac) * 2048);
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
  const double f = ((double)d + 0.5) * scale - 0.5;
  int s = (int)floor(f);
  float fr = (float)(f - (double)s);
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

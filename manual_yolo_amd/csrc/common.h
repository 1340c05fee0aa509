// Shared device/host helpers for the miyolo HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miyolo {

typedef _Float16 half_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;  // CDNA wavefront

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int CE = 4; static constexpr int id = 0; };   // elems / 16 B chunk
template <> struct DT<half_t> { static constexpr int CE = 8; static constexpr int id = 1; };

// SiLU exactly as torch computes it in fp32: x / (1 + exp(-x)).
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }

// A channel-slice view resolved to a device pointer for one launch.
struct SrcDesc {
  const void* ptr;   // buffer base (element type = activation dtype)
  int32_t ld;        // channels of the underlying buffer (row stride in elements)
  int32_t ch_off;    // first channel of the view
  int32_t ch_cnt;    // channels in the view
  int32_t up;        // 1: buffer is at half the consumer's resolution (nearest x2 upsample)
  int32_t h, w;      // spatial dims of the underlying buffer
  uint32_t bytes;    // size of the underlying buffer in bytes (< 2 GiB)
  int32_t pad;
};

struct ConvArgs {
  SrcDesc src[2];
  int32_t nsrc;
  const void* w;      // [cout][kpad] activation dtype
  uint32_t wbytes;    // cout * kpad * sizeof(T)
  int32_t pad0;
  const float* bias;  // [cout]
  void* dst;
  int32_t dst_ld, dst_choff, out_f32;
  const void* res;    // nullptr: none
  int32_t res_ld, res_choff;
  int32_t B, Hin, Win, Hout, Wout;  // Hin/Win: input grid as the conv sees it (after upsample)
  int32_t cin, cout, ksize, stride, act;
  int32_t M, kpad, nk;              // M = B*Hout*Wout; kpad = padded K per weight row; nk = kpad / BK
  int32_t vec_ok;                   // epilogue may use 4-channel vector stores
  int32_t exact;                    // 1: accurate expf in SiLU (fp32 parity mode)
};

}  // namespace miyolo

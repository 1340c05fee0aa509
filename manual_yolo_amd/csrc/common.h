// Shared device/host helpers for the miyolo HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing-experiment switches (profiles/r01_conv_dma_ablation.md) are compiled in only with
// -DMIYOLO_ABLATE=1 (csrc/build.sh ablate); -DMIYOLO_ABLATE=2 (csrc/build.sh stamps) compiles the in-kernel
// cycle stamps only (the ablation branches change register allocation); the shipped kernels carry neither.
#ifndef MIYOLO_ABLATE
#define MIYOLO_ABLATE 0
#endif
#define ABL(bit) (MIYOLO_ABLATE == 1 && (a.ablate & (bit)))
// cache policy of the conv kernels' output stores (aux of raw_buffer_store: 0 default, 2 nt, 16 sc1, 17 sc0 sc1)
#ifndef MIYOLO_ST_AUX
#define MIYOLO_ST_AUX 0
#endif
#if MIYOLO_ABLATE
#define STAMP(var) do { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); var = _t; } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

namespace miyolo {

// A wave-uniform pointer pinned to SGPRs: under register pressure the compiler may keep a uniform value in VGPRs and then
// hand an inline-asm "s" operand a VGPR pair (assembler error); readfirstlane of a value already in SGPRs folds away.
template <typename P>
__device__ __forceinline__ P* sgpr_ptr(P* p) {
  const unsigned long long v = (unsigned long long)p;
  return reinterpret_cast<P*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                              (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

typedef _Float16 half_t;
// fp8 e4m3 (OCP "fn": gfx950's native fp8, NOT MI300X's fnuz) activations / weights of the config-5 path: one byte per
// element; conversions go through v_cvt_pk_fp8_f32 / v_cvt_f32_fp8, the arithmetic through the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the f16 MFMA rate).
struct fp8_t { unsigned char v; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;  // CDNA wavefront

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int CE = 4; static constexpr int id = 0; };   // elems / 16 B chunk
template <> struct DT<half_t> { static constexpr int CE = 8; static constexpr int id = 1; };
template <> struct DT<fp8_t> { static constexpr int CE = 16; static constexpr int id = 2; };
template <typename T> struct is_fp8 { static constexpr bool value = false; };
template <> struct is_fp8<fp8_t> { static constexpr bool value = true; };

// 4 floats -> 4 fp8 bytes (saturating at +-448: e4m3fn has no infinity, an overflow would become NaN) and back
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.f, 448.f), __builtin_amdgcn_fmed3f(b, -448.f, 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.f, 448.f), __builtin_amdgcn_fmed3f(d, -448.f, 448.f), w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ void unpack_fp8x4(uint32_t w, float (&v)[4]) {
  v[0] = __builtin_amdgcn_cvt_f32_fp8((int)w, 0); v[1] = __builtin_amdgcn_cvt_f32_fp8((int)w, 1);
  v[2] = __builtin_amdgcn_cvt_f32_fp8((int)w, 2); v[3] = __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
}

// SiLU exactly as torch computes it in fp32: x / (1 + exp(-x)).
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }
// fp16 path: v_exp_f32 + v_rcp_f32 (about 1 ulp each) instead of an IEEE divide - the result is
// rounded to fp16 right after, and the divide was a third of the conv epilogue's instructions.
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

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
  uint32_t dst_bytes, res_bytes;    // sizes of the destination / residual buffers (raw-buffer bounds)
  int32_t dst_ld, dst_choff, out_f32;
  const void* res;    // nullptr: none
  int32_t res_ld, res_choff;
  int32_t B, Hin, Win, Hout, Wout;  // Hin/Win: input grid as the conv sees it (after upsample)
  int32_t cin, cout, ksize, stride, act;
  int32_t M, kpad, nk;              // M = B*Hout*Wout; kpad = padded K per weight row; nk = kpad / BK
  int32_t vec_ok;                   // epilogue may use 4-channel vector stores
  int32_t res_vec;                  // residual view is 4-channel aligned (vector loads)
  uint32_t mg_hw_mul, mg_hw_shift;  // magic division by Hout*Wout (conv_dmap.h host_magic)
  uint32_t mg_w_mul, mg_w_shift;    // magic division by Wout
  unsigned long long* dbg;          // MIYOLO_ABLATE builds: per-workgroup cycle stamps (conv_dmap.h)
  int32_t ablate;                   // timing experiments only (results wrong): 1 no tile DMA in the loop, 2 no MFMA, 4 no LDS reads
  int32_t exact;                    // 1: accurate expf in SiLU (fp32 parity mode)
  // fp8 path: conv value = acc * qscale[n] (per-output-channel weight scale x the folded input scales) + bias[n];
  // the stored activation is value(after SiLU, + res * res_scale) * out_inv_scale, rounded to e4m3.  bias_init =
  // bias[n] / qscale[n], for kernels that start their accumulators at the bias (conv_h2.h).
  const float* qscale;
  const float* bias_init;
  float out_inv_scale, res_scale;
  int32_t pair8;                    // conv_dmap.h: 8-channel (16-byte) f16 stores over channel-tile pairs (set by launch_conv_dmap)
};

// Shared conv epilogue tail: residual add (after the activation, as Bottleneck does) and the store
// of 4 consecutive output channels n..n+3 of pixel m.  v[] already holds bias + activation.
template <typename T>
__device__ __forceinline__ void epilogue_store(const ConvArgs& a, int m, int n, float (&v)[4]) {
  if (a.res) {
    const T* rp = reinterpret_cast<const T*>(a.res) + ((size_t)m * a.res_ld + a.res_choff + n);
    if (a.res_vec) {
      if constexpr (sizeof(T) == 4) {
        const float4 r = *reinterpret_cast<const float4*>(rp);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      } else {
        const f16x4 r = *reinterpret_cast<const f16x4*>(rp);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.cout) v[r] += (float)rp[r];
    }
  }
  if (a.out_f32) {
    float* dp = reinterpret_cast<float*>(a.dst) + ((size_t)m * a.dst_ld + a.dst_choff + n);
    if (a.vec_ok) {
      *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.cout) dp[r] = v[r];
    }
  } else {
    T* dp = reinterpret_cast<T*>(a.dst) + ((size_t)m * a.dst_ld + a.dst_choff + n);
    if (a.vec_ok) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<f16x4*>(dp) = hv;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.cout) dp[r] = (T)v[r];
    }
  }
}

// Branch-free epilogue for one (16-channel tile, 16-pixel tile) pair of a wave: 4 consecutive
// channels n..n+3 of pixel m.  Validity (m < M, n < cout) is folded into the byte offset of raw
// buffer stores/loads (out of range = dropped / zero), so no exec-mask branch is taken per
// element; the activation kind is fixed by the element type (f32 = parity mode: expf + IEEE
// divide; f16 = v_exp + v_rcp).  Requires a.vec_ok (4-channel aligned views).
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef int v4ie_t __attribute__((ext_vector_type(4)));
// The residual operand of one (m, n..n+3) group, loaded ahead of use: callers issue the loads of a whole row of
// pixel tiles first and then run the activations, so the loads' latency is paid once per row instead of once per
// 16x16 tile (the compiler waits vmcnt(0) at the first use).  f16: components 0-1 hold the 4 halves.
template <typename T>
__device__ __forceinline__ v4ie_t epilogue_res_load(const ConvArgs& a, const __amdgpu_buffer_rsrc_t& rres, int m, int n) {
  const bool ok = (m < a.M) && (n < a.cout);
  const uint32_t ro = ok ? (uint32_t)((m * a.res_ld + a.res_choff + n) * (int)sizeof(T)) : 0x80000000u;
  if constexpr (sizeof(T) == 4) {
    return __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
  } else if constexpr (sizeof(T) == 2) {
    const v2i_t r = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0);
    return (v4ie_t){r[0], r[1], 0, 0};
  } else {
    return (v4ie_t){(int)__builtin_amdgcn_raw_buffer_load_b32(rres, ro, 0, 0), 0, 0, 0};
  }
}

template <typename T, bool OUTF32>
__device__ __forceinline__ void epilogue_fast(const ConvArgs& a, const __amdgpu_buffer_rsrc_t& rdst, int m, int n,
                                              const f32x4& acc, const float (&bv)[4], const v4ie_t& res, const float (&sv)[4]) {
  const bool ok = (m < a.M) && (n < a.cout);
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = is_fp8<T>::value ? fmaf(acc[r], sv[r], bv[r]) : acc[r] + bv[r];
    if (a.act) x = (sizeof(T) == 4) ? silu_exact(x) : silu_fast(x);
    v[r] = x;
  }
  if (a.res) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += __int_as_float(res[k]);
    } else if constexpr (sizeof(T) == 2) {
      const v2i_t r2 = {res[0], res[1]};
      const f16x4 h = *reinterpret_cast<const f16x4*>(&r2);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += (float)h[k];
    } else {
      float rr[4];
      unpack_fp8x4((uint32_t)res[0], rr);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaf(rr[k], a.res_scale, v[k]);
    }
  }
  constexpr int OS = OUTF32 ? 4 : (int)sizeof(T);
  const uint32_t so = ok ? (uint32_t)((m * a.dst_ld + a.dst_choff + n) * OS) : 0x80000000u;
  if constexpr (OS == 4) {
    v4ie_t o = {__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(o, rdst, so, 0, MIYOLO_ST_AUX);
  } else if constexpr (OS == 2) {
    f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2i_t*>(&hv), rdst, so, 0, MIYOLO_ST_AUX);
  } else {
    const float q = a.out_inv_scale;
    __builtin_amdgcn_raw_buffer_store_b32((int)pack_fp8x4(v[0] * q, v[1] * q, v[2] * q, v[3] * q), rdst, so, 0, MIYOLO_ST_AUX);
  }
}

template <typename T, bool OUTF32>
__device__ __forceinline__ void epilogue_fast(const ConvArgs& a, const __amdgpu_buffer_rsrc_t& rdst, int m, int n,
                                              const f32x4& acc, const float (&bv)[4], const v4ie_t& res) {
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  epilogue_fast<T, OUTF32>(a, rdst, m, n, acc, bv, res, one);
}

}  // namespace miyolo

// Implicit-GEMM convolution on the CDNA4 matrix cores (gfx950), NHWC.
//
// Replaces the fused Conv2d+BatchNorm2d+SiLU module ([3P] ultralytics.nn.modules.conv.Conv
// after model.fuse()) that dominates `model(frame)` (reference detect.py:541): 82 of the 83
// convolutions of yolov8m and 25 of the 26 of yolov8n-cls run through this kernel.
//
//   D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]      k = (ky, kx, cin) flattened
//
// * MFMA operand roles: A = weights (rows = output channels), B = activations (cols = pixels).
//   With the 16x16 C/D map (col = lane&15, row = 4*(lane>>4)+reg) every lane ends up holding
//   4 CONSECUTIVE output channels of one pixel, so the NHWC epilogue stores 8 B (f16) or
//   16 B (f32) per lane instead of scalars.
// * K is walked in steps of 128 B per row (64 f16 / 32 f32): one LDS row holds 8 chunks of
//   16 B; chunk c of row r lives at r*128 + ((c ^ ((r>>1)&7))<<4), which makes both the
//   ds_read_b128 fragment reads (16 rows x 4 chunks per wave) and the ds_write_b128 staging
//   writes conflict-free on the 64-bank LDS (checked in tests/test_lds_layout.py).
// * K is flattened over (tap, channel) / (concat segment, channel): a 16 B chunk never
//   straddles a tap because every view has a multiple of 8 (f16) / 4 (f32) channels, so
//   odd widths (48, 96, 288) waste nothing except the zero tail of the last step.
// * 3x3 taps, stride 2, zero padding, nearest-x2 upsample and channel concat are all
//   resolved in the staging address computation: no im2col buffer, no cat, no upsample.
// * f32 uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain - the 1e-4 parity mode);
//   f16 uses v_mfma_f32_16x16x32_f16 with fp32 accumulation.  Both read the same LDS image:
//   a lane's 16 B chunk is 4 f32 (k = 4*(lane>>4)+j feeds MFMA j) or 8 f16.
// * Register-staged, double-buffered pipeline: global loads for step k+1 are issued before
//   the MFMAs of step k and written to the other LDS buffer after them; one barrier / step.
// * Blocks are renumbered so that the blocks sharing an activation tile (same pixel tile,
//   different channel tile) sit on one XCD and hit its L2.
#pragma once
#include "common.h"

namespace miyolo {

constexpr int TP = 4;            // 16-pixel MFMA tiles per wave (64 pixels)
constexpr int ROW_BYTES = 128;   // K bytes per LDS row

__device__ __forceinline__ uint32_t lds_off(int row, int chunk) {
  return (uint32_t)(row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <typename T> struct Mma;
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    const float* af = reinterpret_cast<const float*>(&a);
    const float* bf = reinterpret_cast<const float*>(&b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
  }
};
template <> struct Mma<half_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a),
                                               *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
  }
};

// fp8: ONE block-scaled MFMA covers a whole 128-byte row (K = 128) with unit block scales (e8m0 127 = 2^0).  A lane's
// 32 bytes are chunks q and q+4 of its row (q = lane>>4) for BOTH operands - any assignment of the row's bytes to MFMA
// k indices works as long as A and B use the same one - which keeps the f16 path's conflict-free ds_read_b128 pattern.
typedef int v8i_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma_fp8(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, f32x4& c) {
  const v8i_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
  const v8i_t bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
template <> struct Mma<fp8_t> {     // per-chunk form is not used for fp8 (see mma_fp8); declared so that shared code compiles
  __device__ static __forceinline__ void run(const uint4&, const uint4&, f32x4&) {}
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <typename T, int KS, int WC, int TC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int WP = 4 / WC;
  constexpr int BM = WP * TP * 16;
  constexpr int BN = WC * TC * 16;
  constexpr int XR = BM / 32;          // X rows staged per thread
  constexpr int WR = (BN + 31) / 32;   // W rows staged per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Xs = smem;                        // [2][BM][128]
  unsigned char* const Ws = smem + 2 * BM * ROW_BYTES;   // [2][BN][128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave / WC, wc = wave % WC;

  // ---- XCD-aware block renumbering (bijective for any grid size)
  const int NB = (a.cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mb = bid / NB, nb = bid - mb * NB;
  const int m0 = mb * BM, n0 = nb * BN;

  // ---- staging geometry: thread -> (rows r0 + 32*i, chunk column c8)
  // All global reads are raw buffer loads: a byte offset >= num_records returns zeros, which is
  // how conv zero-padding, the M/N/K tails and masked rows are produced without a branch.
  const int c8 = tid & 7, r0 = tid >> 3;
  constexpr uint32_t kOob = 0x80000000u;   // every buffer is < 2 GiB (engine chunks the batch)
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[0].ptr), 0, a.src[0].bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[1].ptr), 0, a.src[1].bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);

  int32_t xoff0[XR];                    // byte offset of the row's pixel (tap 0 / segment 0)
  int32_t xoff1[KS == 1 ? XR : 1];      // segment 1 (1x1 over a concat)
  uint32_t xmask[XR];                   // 3x3: bit t = tap t in bounds; 1x1: bit 0 = row valid
  const int HWo = a.Hout * a.Wout;
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int b = mm / HWo, rem = mm - b * HWo;
    const int ho = rem / a.Wout, wo = rem - ho * a.Wout;
    if constexpr (KS == 3) {
      const int hi0 = ho * a.stride - 1, wi0 = wo * a.stride - 1;
      xoff0[i] = (((b * a.src[0].h + hi0) * a.src[0].w + wi0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
      uint32_t msk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = hi0 + t / 3, wi = wi0 + t % 3;
        if (vm && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win) msk |= 1u << t;
      }
      xmask[i] = msk;
    } else {
      const int h0 = a.src[0].up ? (ho >> 1) : ho, w0 = a.src[0].up ? (wo >> 1) : wo;
      xoff0[i] = (((b * a.src[0].h + h0) * a.src[0].w + w0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
      const int h1 = a.src[1].up ? (ho >> 1) : ho, w1 = a.src[1].up ? (wo >> 1) : wo;
      xoff1[i] = (((b * a.src[1].h + h1) * a.src[1].w + w1) * a.src[1].ld + a.src[1].ch_off) * (int)sizeof(T);
      xmask[i] = vm ? 1u : 0u;
    }
  }
  // K walker: this thread's 16 B chunk sits at chunk `coff` of tap `tap` (3x3), or at chunk
  // `ks*8 + c8` of the concatenated channel axis (1x1; segment boundaries are K-step aligned,
  // so the segment is uniform per step).
  const int ct0 = a.src[0].ch_cnt / CE;
  int tap = 0, coff = c8;
  if constexpr (KS == 3) {
    tap = c8 / ct0;
    coff = c8 - tap * ct0;
  }
  const int wrow_bytes = a.kpad * (int)sizeof(T);

  uint4 xreg[XR], wreg[WR];

  auto stage_load = [&](int ks) {
    if constexpr (KS == 3) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int32_t toff = ((ky * a.src[0].w + kx) * a.src[0].ld + coff * CE) * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool v = (tap < 9) && ((xmask[i] >> tap) & 1u);
        const uint32_t off = v ? (uint32_t)(xoff0[i] + toff) : kOob;
        const v4i_t r = __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
        xreg[i] = *reinterpret_cast<const uint4*>(&r);
      }
      coff += 8;
      while (coff >= ct0) { coff -= ct0; ++tap; }
    } else {
      const int q = ks * 8 + c8;                       // chunk on the concatenated channel axis
      const bool seg1 = (ks * 8) >= ct0;               // uniform: segment 0 is a multiple of 8 chunks
      const int cq = seg1 ? q - ct0 : q;
      const bool kv = cq < (seg1 ? a.src[1].ch_cnt / CE : ct0) && (!seg1 || a.nsrc > 1);
      const int32_t toff = cq * CE * (int)sizeof(T);
      if (!seg1) {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
          const uint32_t off = (kv && xmask[i]) ? (uint32_t)(xoff0[i] + toff) : kOob;
          const v4i_t r = __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
          xreg[i] = *reinterpret_cast<const uint4*>(&r);
        }
      } else {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
          const uint32_t off = (kv && xmask[i]) ? (uint32_t)(xoff1[i] + toff) : kOob;
          const v4i_t r = __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0);
          xreg[i] = *reinterpret_cast<const uint4*>(&r);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int row = r0 + 32 * j, n = n0 + row;
      const bool v = (row < BN) && (n < a.cout);
      const uint32_t off = v ? (uint32_t)(n * wrow_bytes + (ks * 8 + c8) * 16) : kOob;
      const v4i_t r = __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0);
      wreg[j] = *reinterpret_cast<const uint4*>(&r);
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* xs = Xs + buf * (BM * ROW_BYTES);
    unsigned char* ws = Ws + buf * (BN * ROW_BYTES);
#pragma unroll
    for (int i = 0; i < XR; ++i) *reinterpret_cast<uint4*>(xs + lds_off(r0 + 32 * i, c8)) = xreg[i];
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int row = r0 + 32 * j;
      if (row < BN) *reinterpret_cast<uint4*>(ws + lds_off(row, c8)) = wreg[j];
    }
  };

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const unsigned char* xs = Xs + buf * (BM * ROW_BYTES);
    const unsigned char* ws = Ws + buf * (BN * ROW_BYTES);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[TC], bf[TP];
#pragma unroll
      for (int i = 0; i < TC; ++i)
        af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int j = 0; j < TP; ++j)
        bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TP + j) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
    }
  };

  // ---- main loop
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (int ks = 0; ks < a.nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < a.nk) stage_load(ks + 1);
    compute(cur);
    if (ks + 1 < a.nk) stage_store(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, SiLU, residual, store 4 consecutive channels per lane
  const float* __restrict__ bias = a.bias;
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int n = n0 + (wc * TC + i) * 16 + fq * 4;
    if (n >= a.cout) continue;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (n + r < a.cout) ? bias[n + r] : 0.f;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      const int m = m0 + (wp * TP + j) * 16 + frow;
      if (m >= a.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[i][j][r] + bv[r];
        if (a.act) x = a.exact ? silu_exact(x) : silu_fast(x);
        v[r] = x;
      }
      epilogue_store<T>(a, m, n, v);
    }
  }
}

// ------------------------------------------------------------------ host-side launch
struct ConvCfg { int wc, tc; };

// Pick (waves along channels, 16-channel tiles per wave): least padded work first, then the
// wider channel tile (fewer re-reads of the activation tile through L2).
inline ConvCfg pick_conv_cfg(int cout, long M) {
  static const ConvCfg cands[] = {{2, 4}, {2, 3}, {1, 4}, {1, 3}, {1, 2}, {1, 1}};
  ConvCfg best = {1, 1};
  double best_cost = 1e30;
  for (const ConvCfg& c : cands) {
    const int bn = c.wc * c.tc * 16, bm = 256 / c.wc;
    const long nb = (cout + bn - 1) / bn, mbk = (M + bm - 1) / bm;
    double cost = (double)(nb * bn) * (double)(mbk * bm);   // padded MACs / K
    if (nb * mbk < 256) cost *= 1.0 + 0.25 * (256.0 / (double)(nb * mbk) - 1.0);  // under-filled chip
    cost *= 1.0 + 0.02 * (128.0 / bn);   // mild preference for wide channel tiles
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <typename T, int KS, int WC, int TC>
inline hipError_t launch_conv_cfg(const ConvArgs& a, hipStream_t s) {
  // dynamic-LDS limits are raised once per process in miyolo_create (set_conv_attr)
  constexpr int BM = (4 / WC) * TP * 16, BN = WC * TC * 16;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * ROW_BYTES;
  const long mbk = ((long)a.M + BM - 1) / BM, nb = (a.cout + BN - 1) / BN;
  hipLaunchKernelGGL((conv_igemm_kernel<T, KS, WC, TC>), dim3((unsigned)(mbk * nb)), dim3(256), lds, s, a);
  return hipGetLastError();
}

template <typename T, int KS>
inline hipError_t launch_conv_ks(const ConvArgs& a, ConvCfg c, hipStream_t s) {
  if (c.wc == 2 && c.tc == 4) return launch_conv_cfg<T, KS, 2, 4>(a, s);
  if (c.wc == 2 && c.tc == 3) return launch_conv_cfg<T, KS, 2, 3>(a, s);
  if (c.wc == 1 && c.tc == 4) return launch_conv_cfg<T, KS, 1, 4>(a, s);
  if (c.wc == 1 && c.tc == 3) return launch_conv_cfg<T, KS, 1, 3>(a, s);
  if (c.wc == 1 && c.tc == 2) return launch_conv_cfg<T, KS, 1, 2>(a, s);
  return launch_conv_cfg<T, KS, 1, 1>(a, s);
}

template <typename T>
inline hipError_t launch_conv(const ConvArgs& a, hipStream_t s, int force_wc = 0, int force_tc = 0) {
  ConvCfg c = pick_conv_cfg(a.cout, a.M);
  if (force_wc > 0 && force_tc > 0) c = {force_wc, force_tc};
  if (a.ksize == 3) return launch_conv_ks<T, 3>(a, c, s);
  return launch_conv_ks<T, 1>(a, c, s);
}

}  // namespace miyolo

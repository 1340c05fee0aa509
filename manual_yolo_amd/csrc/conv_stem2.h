// The stem and the first down-sampling conv as ONE launch (f16): uint8 frame -> conv3x3 s2 (3 -> C0) + SiLU -> conv3x3 s2
// (C0 -> C1) + SiLU, yolov8's layers 0 and 1 (reference: `model(frame)`, detect.py:541 -> [3P] DetectionModel layers 0, 1).
//
// Why (profiles/r02_per_layer_f16.md): the stem writes its 320 x 320 x 48 map (629 MB at batch 64) to HBM at 2.6 TB/s and
// layer 1 reads it back through the ring kernel at 3 TB/s: 274 + 316 us for 79 MB of frames in and 315 MB of activations
// out.  Here a persistent workgroup keeps layer 1's weights in LDS (96 x 432 halves = 86 KiB), computes the 17 x 33 patch of
// the stem's output that an 8 x 16 tile of layer 1 needs straight from the frame into LDS (zero outside the stem's map:
// layer 1's padding; 10 % of the stem is computed twice), and runs layer 1 on it.  The stem's map never touches HBM.
// Arithmetic is that of stem_kernel (same K' order, same 1/255 scaling, same f16 rounding of its output) followed by that
// of the ring / 2-D-tile kernels (flattened K = (tap, channel) ascending): bit-identical to the two launches.
// MEASURED: no gain (8 620-8 690 vs 8 650-8 705 frames/s A/B) - with one workgroup per CU the stem phase (frame bytes through
// the texture path, 27 k SiLUs per tile) and layer 1's K loop (one pixel tile per wave: 7 KiB of LDS fragments per 6 MFMAs)
// take turns instead of overlapping, which costs what the saved HBM traffic gains.  Off by default (option stem_fuse).
#pragma once
#include "common.h"
#include "conv_dma.h"
#include "conv_dmap.h"
#include "conv_igemm.h"

namespace miyolo {

struct Stem2Args {
  const uint8_t* in;                        // [B][H][W][3]
  const void* w0; const float* b0;          // stem: [C0][32] f16 in the K' order of stem_kernel, bias
  const void* w1; const float* b1;          // layer 1: [C1][kpad] f16, bias
  void* dst; uint32_t dst_bytes, in_bytes;
  int32_t dst_ld, dst_choff, kpad;
  int32_t B, H, W, act0, act1;
  int32_t tiles_x, tiles_y, ntiles;
  uint32_t mg_img_mul, mg_img_shift, mg_tx_mul, mg_tx_shift;
};

constexpr int kS2Oh = 8, kS2Ow = 16;                        // output tile of layer 1
constexpr int kS2Sh = 2 * kS2Oh + 1, kS2Sw = 2 * kS2Ow + 1;  // the stem pixels it reads: 17 x 33

template <int TCS, int TC1> struct Stem2Geo {
  static constexpr int C0 = TCS * 16, C1 = TC1 * 16, SROW = C0 * 2, CPT = C0 / 8, NCH = 9 * CPT, NG = (NCH + 3) / 4, WROW = NG * 64 + 32;   // pitch = 2 (mod 4) chunks: conflict-free weight fragments (conv_bneck.h)
  static constexpr int NSPX = kS2Sh * kS2Sw, NST = (NSPX + 15) / 16, TPW = (NST + 7) / 8;     // stem pixels, their 16-pixel tiles, tiles per wave
  static constexpr int W_BYTES = C1 * WROW, KOFF_BYTES = NG * 16;
  static constexpr int S_OFF = W_BYTES + KOFF_BYTES, S_BYTES = NSPX * SROW;
  static constexpr int LDS = S_OFF + S_BYTES;
};

template <int TCS, int TC1>
__global__ __launch_bounds__(512) void conv_stem2_kernel(const Stem2Args a) {
  typedef Stem2Geo<TCS, TC1> G;
  constexpr int SROW = G::SROW, CPT = G::CPT, NG = G::NG, WROW = G::WROW, TPW = G::TPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  unsigned char* const wl = smem;
  int32_t* const koff = reinterpret_cast<int32_t*>(smem + G::W_BYTES);
  unsigned char* const sl = smem + G::S_OFF;

  const int Gd = gridDim.x;
  const int first = (blockIdx.x & 7) * (Gd >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < a.ntiles) ? (a.ntiles - first + Gd - 1) / Gd : 0;
  if (my_tiles == 0) return;

  // ---- layer 1's weights and its tap table -> LDS, once per workgroup
  {
    const int cpr = NG * 4;
    const size_t row_bytes = (size_t)a.kpad * 2;
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w1);
    for (int e = tid; e < G::C1 * cpr; e += 512) {
      const int n = e / cpr, c = e - n * cpr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((size_t)(c + 1) * 16 <= row_bytes) v = *reinterpret_cast<const uint4*>(wg + (size_t)n * row_bytes + c * 16);
      *reinterpret_cast<uint4*>(wl + n * WROW + c * 16) = v;
    }
    for (int q = tid; q < NG * 4; q += 512) {
      int v = 0;
      if (q < G::NCH) {
        const int tap = q / CPT, co = q - tap * CPT;
        v = ((tap / 3) * kS2Sw + (tap % 3)) * SROW + co * 16;       // stem-patch coordinate = 2 * output coordinate + tap
      }
      koff[q] = v;
    }
  }
  // stem weight fragments and biases (stem_kernel's layout), layer 1's biases
  uint4 wf[TCS];
  const half_t* wp = reinterpret_cast<const half_t*>(a.w0);
#pragma unroll
  for (int tc = 0; tc < TCS; ++tc) wf[tc] = *reinterpret_cast<const uint4*>(wp + (tc * 16 + frow) * 32 + fq * 8);
  float bs[TCS][4], b1v[TC1][4];
#pragma unroll
  for (int tc = 0; tc < TCS; ++tc)
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[tc][r] = a.b0[tc * 16 + fq * 4 + r];
#pragma unroll
  for (int i = 0; i < TC1; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) b1v[i][r] = a.b1[i * 16 + fq * 4 + r];

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const int Hs = a.H / 2, Ws = a.W / 2, Ho = a.H / 4, Wo = a.W / 4;
  // stem_kernel's per-lane byte pattern: q < 3: row hi0 + q, bytes j of the 9-byte run from column wi0; q == 3: rows hi0 + j, byte 8
  const int W3 = a.W * 3;
  const int startq = (fq < 3) ? fq * W3 : 8, stepq = (fq < 3) ? 1 : W3;
  const uint32_t m_always = (fq < 3) ? 0u : 0xF8u, m_top = (fq < 3) ? (fq == 0 ? 0xFFu : 0u) : 1u, m_left = (fq < 3) ? 7u : 0u;
  // this wave's stem pixels: tiles wave, wave + 8, ...
  int sy[TPW], sx[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int ps = (wave + 8 * j) * 16 + frow;
    sy[j] = (ps < G::NSPX) ? ps / kS2Sw : -10000;
    sx[j] = ps % kS2Sw;
  }
  uint32_t abase[TC1];
#pragma unroll
  for (int i = 0; i < TC1; ++i) abase[i] = (uint32_t)((i * 16 + frow) * WROW + fq * 16);
  const int bbase = ((2 * wave) * kS2Sw + 2 * frow) * SROW;          // output pixel (wave, frow) of the tile, tap (0, 0)

  auto tile_coords = [&](int tile, int* b, int* ty, int* tx) {
    const uint32_t bb = magic_div((uint32_t)tile, a.mg_img_mul, a.mg_img_shift);
    const uint32_t r = (uint32_t)tile - bb * (uint32_t)(a.tiles_x * a.tiles_y);
    *ty = (int)magic_div(r, a.mg_tx_mul, a.mg_tx_shift); *tx = (int)r - *ty * a.tiles_x; *b = (int)bb;
  };
  // the frame bytes of this wave's stem pixels for `tile` (the next tile's are in flight under layer 1's K loop)
  auto fetch = [&](int tile, uint32_t (&u)[TPW][8]) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int gy = 2 * ty * kS2Oh - 1 + sy[j], gx = 2 * tx * kS2Ow - 1 + sx[j];
      const bool vm = gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
      const int rb = ((b * a.H + 2 * gy - 1) * a.W + 2 * gx - 1) * 3;
      const uint32_t inval = (vm ? m_always : 0xFFu) | (gy == 0 ? m_top : 0u) | (gx == 0 ? m_left : 0u);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t off = (uint32_t)(rb + startq + k * stepq) | (((inval >> k) & 1u) << 31);
        u[j][k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rin, off, 0, 0);
      }
    }
  };

  __syncthreads();
  uint32_t un[TPW][8];
  fetch(first, un);
  int tile = first;
  for (int t = 0; t < my_tiles; ++t, tile += Gd) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
    // ---- stem on this wave's pixels -> S (f16, zero outside the stem's map)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int ps = (wave + 8 * j) * 16 + frow;
      const int gy = 2 * ty * kS2Oh - 1 + sy[j], gx = 2 * tx * kS2Ow - 1 + sx[j];
      const bool vm = gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
      f16x8 xb;
#pragma unroll
      for (int k = 0; k < 8; ++k) xb[k] = (half_t)((float)un[j][k] * (1.0f / 255.0f));
#pragma unroll
      for (int tc = 0; tc < TCS; ++tc) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&wf[tc]), xb, acc, 0, 0, 0);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = acc[r] + bs[tc][r];
          if (a.act0) y = silu_fast(y);
          v[r] = vm ? y : 0.0f;
        }
        if (ps < G::NSPX) {
          const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<f16x4*>(sl + ps * SROW + (tc * 16 + fq * 4) * 2) = hv;
        }
      }
    }
    __syncthreads();                                         // S complete
    if (t + 1 < my_tiles) fetch(tile + Gd, un);
    // ---- layer 1 on the patch: one 16-pixel tile (row `wave` of the 8 x 16 output tile) x all channels per wave
    f32x4 acc1[TC1];
#pragma unroll
    for (int i = 0; i < TC1; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kg = 0; kg < NG; ++kg) {
      const int ko = koff[kg * 4 + fq];
      const uint4 bf = *reinterpret_cast<const uint4*>(sl + bbase + ko);
#pragma unroll
      for (int i = 0; i < TC1; ++i) {
        const uint4 af = *reinterpret_cast<const uint4*>(wl + abase[i] + kg * 64);
        Mma<half_t>::run(af, bf, acc1[i]);
      }
    }
    {
      const int m = (b * Ho + ty * kS2Oh + wave) * Wo + tx * kS2Ow + frow;
#pragma unroll
      for (int i = 0; i < TC1; ++i) {
        const int n = i * 16 + fq * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc1[i][r] + b1v[i][r];
          if (a.act1) x = silu_fast(x);
          v[r] = x;
        }
        const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2i_t*>(&hv), rdst, (uint32_t)((m * a.dst_ld + a.dst_choff + n) * 2), 0, 0);
      }
    }
    __syncthreads();                                         // S free again
  }
}

// host side ------------------------------------------------------------------------------------------------------
inline bool stem2_shape_ok(int C0, int C1, int H, int W, size_t* lds) {
  if (H % (4 * kS2Oh) || W % (4 * kS2Ow)) return false;
  if (C0 == 48 && C1 == 96) *lds = Stem2Geo<3, 6>::LDS;
  else if (C0 == 16 && C1 == 32) *lds = Stem2Geo<1, 2>::LDS;
  else if (C0 == 32 && C1 == 64) *lds = Stem2Geo<2, 4>::LDS;
  else return false;
  return *lds <= 160 * 1024;
}

inline hipError_t launch_conv_stem2(Stem2Args a, int C0, int C1, hipStream_t s, int ncu) {
  size_t lds;
  if (!stem2_shape_ok(C0, C1, a.H, a.W, &lds)) return hipErrorInvalidValue;
  a.tiles_x = a.W / 4 / kS2Ow; a.tiles_y = a.H / 4 / kS2Oh; a.ntiles = a.B * a.tiles_x * a.tiles_y;
  host_magic((uint32_t)(a.tiles_x * a.tiles_y), &a.mg_img_mul, &a.mg_img_shift);
  host_magic((uint32_t)a.tiles_x, &a.mg_tx_mul, &a.mg_tx_shift);
  long grid = std::min<long>(a.ntiles, ncu);
  grid = (grid + 7) / 8 * 8;
  if (C0 == 48) hipLaunchKernelGGL((conv_stem2_kernel<3, 6>), dim3((unsigned)grid), dim3(512), lds, s, a);
  else if (C0 == 16) hipLaunchKernelGGL((conv_stem2_kernel<1, 2>), dim3((unsigned)grid), dim3(512), lds, s, a);
  else hipLaunchKernelGGL((conv_stem2_kernel<2, 4>), dim3((unsigned)grid), dim3(512), lds, s, a);
  return hipGetLastError();
}

inline hipError_t set_stem2_attrs() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem2_kernel<3, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem2_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem2_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

}  // namespace miyolo

// A whole narrow Bottleneck - y = x + conv3x3(conv3x3(x)), both with BN + SiLU ([3P] ultralytics Bottleneck.forward inside
// C2f, shortcut=True) - as ONE launch, f16, C = 16 / 32 / 48 channels, on 16 x 16 pixel tiles.
//
// Why (SURVEY.md 7 "fusion matters more than MFMA tuning there", profiles/r02_per_layer_f16.md): the two 48 -> 48 layers
// of a Bottleneck at 160 x 160 take 165 + 200 us as two launches; each is 4-5x off every roof (HBM 40-60 us, MFMA 27 us,
// SiLU 28 us) because a narrow layer has nothing to overlap its own phases with, and between them the intermediate map
// (157 MB at batch 64) goes to HBM and comes back, as does the input a second time for the residual.  Here a persistent
// workgroup keeps BOTH weight matrices in LDS (2 x 43 KiB), fetches the 20 x 20 input halo tile once, computes the 18 x 18
// intermediate tile into LDS (zero outside the image: the second conv's padding), computes the 16 x 16 output tile from
// it, adds the residual (read from the input tile into registers before the next tile's fetch overwrites it), and stores:
// 38 KB in + 24 KB out per tile instead of 135 KB; the next tile's halo tile flies under the second conv.
// Same MFMA, same flattened K order (tap, channel), same f16 rounding of the intermediate, same epilogue arithmetic as
// the two-launch path (conv_t2d.h / conv_dmap.h): results are bit-identical to it (tests/test_gpu_conv.py).
// LDS (C = 48): w1, w2 [48][928 B] | tap tables | X 20 rows x 124 x 16 B (+ the tail of its last DMA) | T 18x18 px x 96 B = 158 KiB.
#pragma once
#include "common.h"
#include "conv_dma.h"
#include "conv_dmap.h"
#include "conv_igemm.h"

namespace miyolo {

struct BneckArgs {
  const void* x; void* dst;
  const void* w1; const void* w2;          // [C][kpad] f16, K = (tap, channel)
  const float* b1; const float* b2;
  uint32_t x_bytes, dst_bytes;
  int32_t x_ld, x_choff, dst_ld, dst_choff, kpad;
  int32_t B, H, W, act1, act2;
  int32_t tiles_x, tiles_y, ntiles;
  uint32_t mg_img_mul, mg_img_shift, mg_tx_mul, mg_tx_shift;
};

constexpr int kBnX = 20, kBnT = 18;        // input halo tile, intermediate tile (pixels per side)

template <int TC> struct BneckGeo {
  static constexpr int C = TC * 16, XROW = C * 2, CPT = C / 8, NCH = 9 * CPT, NG = (NCH + 3) / 4;
  // LDS banking of the fragment reads (ds_read_b128: lane groups {0-3, 12-15, 20-27}, ... = fragment rows {0-3, 12-15} at
  // k-chunk c with rows {4-11} at chunk c + 1; bank = 16-byte slot mod 16 - MI355X_MICROARCH.md LDS table):
  //  * weight rows: a pitch of S chunks is conflict-free iff S = 2 (mod 4) (S r covers the even slots for one row set,
  //    S r + 1 the odd ones for the other); an ODD pitch - round 2's NG*64 + 16 - makes every weight read 2-way;
  //  * the X tile: pixels are CPT chunks apart (6: conflict-free by the same rule), but conv A's 16-pixel fragments run
  //    through the 18-wide intermediate rows and jump 2 halo pixels at a row end; with a halo-row pitch of XPC = 18 CPT
  //    (mod 16) chunks the pixel after a row end sits CPT chunks (mod 16) behind the last one, as if there were no jump.
  // Round 2 measured 45.6 % of this kernel's LDS cycles as bank conflicts; the model gives 2.0 and 1.74 cycles per lane
  // group for the old weight and pixel reads, 1.0 and 1.08 for these (tests/test_conv_emulation.py).
  static constexpr int WROW = NG * 64 + 32;
  static constexpr int XPC = (kBnX * CPT + 15 - (18 * CPT) % 16) / 16 * 16 + (18 * CPT) % 16;   // 124 (C = 48), 88, 52
  static constexpr int NSLOT = kBnX * XPC, NDW = (NSLOT + 511) / 512;             // 16-byte slots of X, DMAs per wave
  static constexpr int W_BYTES = C * WROW, KOFF_BYTES = NG * 16;
  static constexpr int X_OFF = 2 * W_BYTES + 2 * KOFF_BYTES, X_BYTES = NDW * 8 * 1024;
  static constexpr int T_OFF = X_OFF + X_BYTES, T_BYTES = kBnT * kBnT * XROW;
  static constexpr int LDS = T_OFF + T_BYTES;
};

template <int TC>
__global__ __launch_bounds__(512) void conv_bneck_kernel(const BneckArgs a) {
  typedef BneckGeo<TC> G;
  constexpr int XROW = G::XROW, CPT = G::CPT, NG = G::NG, WROW = G::WROW, NDW = G::NDW, XPC = G::XPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  unsigned char* const w1l = smem;
  unsigned char* const w2l = smem + G::W_BYTES;
  int32_t* const koffA = reinterpret_cast<int32_t*>(smem + 2 * G::W_BYTES);
  int32_t* const koffB = koffA + NG * 4;
  unsigned char* const xl = smem + G::X_OFF;
  unsigned char* const tl = smem + G::T_OFF;
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

  const int Gd = gridDim.x;
  const int first = (blockIdx.x & 7) * (Gd >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < a.ntiles) ? (a.ntiles - first + Gd - 1) / Gd : 0;
  if (my_tiles == 0) return;

  // ---- both weight matrices and the two tap tables -> LDS, once per workgroup
  {
    const int cpr = NG * 4;
    const size_t row_bytes = (size_t)a.kpad * 2;
    for (int e = tid; e < 2 * G::C * cpr; e += 512) {
      const int which = e / (G::C * cpr), r = e - which * G::C * cpr;
      const int n = r / cpr, c = r - n * cpr;
      const unsigned char* wg = reinterpret_cast<const unsigned char*>(which ? a.w2 : a.w1);
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((size_t)(c + 1) * 16 <= row_bytes) v = *reinterpret_cast<const uint4*>(wg + (size_t)n * row_bytes + c * 16);
      *reinterpret_cast<uint4*>((which ? w2l : w1l) + n * WROW + c * 16) = v;
    }
    for (int q = tid; q < NG * 4; q += 512) {
      int va = 0, vb = 0;
      if (q < G::NCH) {
        const int tap = q / CPT, co = q - tap * CPT;
        va = ((tap / 3) * XPC + (tap % 3) * CPT + co) * 16;        // X coordinate = T coordinate + tap
        vb = ((tap / 3) * kBnT + (tap % 3)) * XROW + co * 16;      // T coordinate = output coordinate + tap
      }
      koffA[q] = va; koffB[q] = vb;
    }
  }
  // biases of this lane's channels (n = i*16 + fq*4 .. +3)
  float bA[TC][4], bB[TC][4];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bA[i][r] = a.b1[i * 16 + fq * 4 + r]; bB[i][r] = a.b2[i * 16 + fq * 4 + r]; }

  // ---- DMA-side per-lane state: slot s of the X tile -> (halo row, pixel, chunk), XPC slots per halo row of which the
  // last XPC - 20 CPT are padding; the image is dense: slot s at 16 s
  const v4i_t rsx = make_srd(a.x, a.x_bytes);
  const int ldB = a.x_ld * 2;
  int32_t rel[NDW], hy[NDW], hx[NDW];
#pragma unroll
  for (int d = 0; d < NDW; ++d) {
    const int s = (wave * NDW + d) * 64 + lane;
    const int y = s / XPC, rem = s - y * XPC;
    const int px = rem / CPT, c = rem - px * CPT;
    hy[d] = (s < G::NSLOT && px < kBnX) ? y : -1000;
    hx[d] = px;
    rel[d] = ((y - 2) * a.W + (px - 2)) * ldB + c * 16;
  }
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);

  // fragment bases
  uint32_t abase[TC];
#pragma unroll
  for (int i = 0; i < TC; ++i) abase[i] = (uint32_t)((i * 16 + frow) * WROW + fq * 16);
  // conv A: intermediate pixel tiles wave, wave + 8, wave + 16 (21 tiles of 16 cover 324 pixels; tile 21.. do not exist)
  int baseA[3], pA[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int p = (wave + 8 * j) * 16 + frow;
    pA[j] = (wave + 8 * j < 21 && p < kBnT * kBnT) ? p : -1;
    const int pp = pA[j] >= 0 ? p : 0;
    baseA[j] = ((pp / kBnT) * XPC + (pp % kBnT) * CPT) * 16;
  }
  // conv B: output rows 2*wave, 2*wave + 1
  int baseB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) baseB[j] = ((2 * wave + j) * kBnT + frow) * XROW;

  auto tile_coords = [&](int tile, int* b, int* ty, int* tx) {
    const uint32_t bb = magic_div((uint32_t)tile, a.mg_img_mul, a.mg_img_shift);
    const uint32_t r = (uint32_t)tile - bb * (uint32_t)(a.tiles_x * a.tiles_y);
    *ty = (int)magic_div(r, a.mg_tx_mul, a.mg_tx_shift); *tx = (int)r - *ty * a.tiles_x; *b = (int)bb;
  };
  // X: the 20 x 20 halo tile of `tile`, zeros outside the image (out-of-range DMA offsets)
  auto issue_x = [&](int tile) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
    const int32_t origin = ((b * a.H + ty * 16) * a.W + tx * 16) * ldB + a.x_choff * 2;
    const uint32_t st = lds_base + (uint32_t)(G::X_OFF + wave * NDW * 1024);
#pragma unroll
    for (int d = 0; d < NDW; ++d) {
      const int gy = ty * 16 - 2 + hy[d], gx = tx * 16 - 2 + hx[d];
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      lds_dma16(rsx, st + d * 1024, ok ? (uint32_t)(origin + rel[d]) : 0x80000000u);
    }
  };
  __syncthreads();
  issue_x(first);
  int tile = first;
  for (int t = 0; t < my_tiles; ++t, tile += Gd) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- conv A -> T (18 x 18, SiLU, f16, zero outside the image)
    {
      f32x4 acc[TC][3];
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const bool three = wave + 16 < 21;                         // wave-uniform: waves 5..7 have two pixel tiles
      // (Round 3 tried reading K group kg + 1's fragments before kg's MFMAs here and in conv B, as conv_stem2.h does: 247 vs 240 us on a
      // box that ran everything 3 % slower - nothing.  The partner wave on the SIMD already covers these latencies.)
#pragma unroll 2
      for (int kg = 0; kg < NG; ++kg) {
        const int ko = koffA[kg * 4 + fq];
        uint4 af[TC], bf[3];
#pragma unroll
        for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(w1l + abase[i] + kg * 64);
#pragma unroll
        for (int j = 0; j < 3; ++j) if (j < 2 || three) bf[j] = *reinterpret_cast<const uint4*>(xl + baseA[j] + ko);
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) if (j < 2 || three) Mma<half_t>::run(af[i], bf[j], acc[i][j]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (pA[j] < 0) continue;
        const int py = pA[j] / kBnT, px = pA[j] - py * kBnT;
        const int gy = ty * 16 - 1 + py, gx = tx * 16 - 1 + px;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float x = acc[i][j][q] + bA[i][q];
            if (a.act1) x = silu_fast(x);
            v[q] = inside ? x : 0.0f;
          }
          const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<f16x4*>(tl + pA[j] * XROW + (i * 16 + fq * 4) * 2) = hv;
        }
      }
    }
    // the residual operand of this wave's output pixels (the input tile's centre) goes to registers now, so that the X
    // buffer is free behind the barrier below and the NEXT tile's halo tile can fly under conv B and its epilogue
    f16x4 resv[TC][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned char* xr = xl + ((2 * wave + j + 2) * XPC + (frow + 2) * CPT) * 16;
#pragma unroll
      for (int i = 0; i < TC; ++i) resv[i][j] = *reinterpret_cast<const f16x4*>(xr + (i * 16 + fq * 4) * 2);
    }
    __syncthreads();                                         // T complete; nobody reads X any more
    if (t + 1 < my_tiles) issue_x(tile + Gd);
    // ---- conv B -> output tile, + residual, store
    {
      f32x4 acc[TC][2];
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int kg = 0; kg < NG; ++kg) {
        const int ko = koffB[kg * 4 + fq];
        uint4 af[TC], bf[2];
#pragma unroll
        for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(w2l + abase[i] + kg * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const uint4*>(tl + baseB[j] + ko);
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma<half_t>::run(af[i], bf[j], acc[i][j]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int oy = 2 * wave + j;
        const int m = (b * a.H + ty * 16 + oy) * a.W + tx * 16 + frow;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int n = i * 16 + fq * 4;
          const f16x4 h = resv[i][j];
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float x = acc[i][j][q] + bB[i][q];
            if (a.act2) x = silu_fast(x);
            v[q] = x + (float)h[q];
          }
          const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2i_t*>(&hv), rdst, (uint32_t)((m * a.dst_ld + a.dst_choff + n) * 2), 0, MIYOLO_ST_AUX);
        }
      }
    }
    // the barrier at the top of the next tile (behind its X wait) also says that everybody is done with T
  }
}

// host side ------------------------------------------------------------------------------------------------------
inline bool bneck_shape_ok(int C, int H, int W, size_t* lds) {
  if (H % 16 || W % 16) return false;
  switch (C) {
    case 16: *lds = BneckGeo<1>::LDS; break;
    case 32: *lds = BneckGeo<2>::LDS; break;
    case 48: *lds = BneckGeo<3>::LDS; break;
    default: return false;
  }
  return *lds <= 160 * 1024;
}

inline hipError_t launch_conv_bneck(BneckArgs a, int C, hipStream_t s, int ncu) {
  size_t lds;
  if (!bneck_shape_ok(C, a.H, a.W, &lds)) return hipErrorInvalidValue;
  a.tiles_x = a.W / 16; a.tiles_y = a.H / 16; a.ntiles = a.B * a.tiles_x * a.tiles_y;
  host_magic((uint32_t)(a.tiles_x * a.tiles_y), &a.mg_img_mul, &a.mg_img_shift);
  host_magic((uint32_t)a.tiles_x, &a.mg_tx_mul, &a.mg_tx_shift);
  long grid = std::min<long>(a.ntiles, ncu);
  grid = (grid + 7) / 8 * 8;
  switch (C) {
    case 16: hipLaunchKernelGGL((conv_bneck_kernel<1>), dim3((unsigned)grid), dim3(512), lds, s, a); break;
    case 32: hipLaunchKernelGGL((conv_bneck_kernel<2>), dim3((unsigned)grid), dim3(512), lds, s, a); break;
    default: hipLaunchKernelGGL((conv_bneck_kernel<3>), dim3((unsigned)grid), dim3(512), lds, s, a); break;
  }
  return hipGetLastError();
}

inline hipError_t set_bneck_attrs() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bneck_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bneck_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bneck_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

}  // namespace miyolo

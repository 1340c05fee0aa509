// The stem and the first down-sampling conv as ONE launch (f16): uint8 frame -> conv3x3 s2 (3 -> C0) + SiLU -> conv3x3 s2
// (C0 -> C1) + SiLU, yolov8's layers 0 and 1 (reference: `model(frame)`, detect.py:541 -> [3P] DetectionModel layers 0, 1).
//
// Why (profiles/r03_per_layer_f16.md): the stem writes its 320 x 320 x 48 map (629 MB at batch 64) to HBM and layer 1 reads
// it back through the ring kernel at 3 TB/s: 242 + 316 us for 79 MB of frames in and 315 MB of activations out.  Here a
// persistent workgroup keeps layer 1's weights in LDS (96 x 432 halves = 87 KiB), computes the 17 x 17 patch of the stem's
// output that an 8 x 8 tile of layer 1 needs straight from the frame into LDS (zero outside the stem's map: layer 1's
// padding; 13 % of the stem is computed twice), and runs layer 1 on it.  The stem's map never touches HBM.
// Arithmetic is that of stem_kernel (same K' order, same 1/255 scaling, same f16 rounding of its output) followed by that
// of the ring / 2-D-tile kernels (flattened K = (tap, channel) ascending): bit-identical to the two launches.
// Round 2's form (one 8 x 16 tile at a time, all eight waves in the same phase) measured no gain: with one workgroup per CU
// the stem phase (frame bytes through the texture path as single-byte loads, 27 k SiLUs per tile: VALU) and layer 1's K
// loop (LDS fragments + MFMA) took turns - 25 k cycles per tile, 518 us.  Round 3: the workgroup is TWO GROUPS of four waves
// on two tiles, one phase apart - while group 0 computes the stem patch of its tile, group 1 runs layer 1 on the patch
// it finished in the previous phase, and vice versa; one workgroup barrier per phase.  Every SIMD hosts one wave of each
// group, so its VALU (stem: conversions, SiLU) and its LDS / matrix pipes (layer 1) work at the same time.  The frame bytes
// come as one 12-byte load per lane (stem_kernel's round-3 gather) issued a phase ahead.
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
  const void* w2; const float* b2;          // HAS3: the 1x1 conv behind layer 1 (C1 -> C1, a C2f's cv1): [C1][kpad2] f16, bias
  int32_t kpad2, act2;
  void* dst; uint32_t dst_bytes, in_bytes;
  int32_t dst_ld, dst_choff, kpad;
  int32_t B, H, W, act0, act1;
  int32_t tiles_x, tiles_y, ntiles;
  uint32_t mg_img_mul, mg_img_shift, mg_tx_mul, mg_tx_shift;
};

constexpr int kS2Oh = 8, kS2Ow = 8;                         // output tile of layer 1 (per wave group)
constexpr int kS2Sh = 2 * kS2Oh + 1, kS2Sw = 2 * kS2Ow + 1;  // the stem pixels it reads: 17 x 17

template <int TCS, int TC1, bool HAS3> struct Stem2Geo {
  static constexpr int C0 = TCS * 16, C1 = TC1 * 16, SROW = C0 * 2, CPT = C0 / 8, NCH = 9 * CPT, NG = (NCH + 3) / 4;
  // weight-row pitch: the smallest number of 16-byte chunks >= NCH that is 2 (mod 4) - conflict-free fragments (conv_bneck.h).
  // For C0 = 48 that is NCH itself (54): the last K group's chunks 54, 55 then belong to the NEXT row (finite weights; 32 zero
  // bytes behind the last row) and the pixel fragment is zeroed for them - 0 x finite adds nothing.
  static constexpr int WCH = (NCH % 4 == 2) ? NCH : (NCH + 3) / 4 * 4 + 2, WROW = WCH * 16;
  static constexpr int NSPX = kS2Sh * kS2Sw, NST = (NSPX + 15) / 16, TPW = (NST + 3) / 4;     // stem pixels, their 16-pixel tiles, tiles per wave of a group
  static constexpr int W_BYTES = C1 * WROW + 32, KOFF_BYTES = NG * 16;
  static constexpr int S_OFF = W_BYTES + KOFF_BYTES, S_BYTES = (NSPX * SROW + 63) / 64 * 64;   // one patch per group
  // HAS3: the 1x1 conv's weights, [C1 rows][C1 channels], pitch 2 (mod 4) chunks
  static constexpr int C2CH = C1 / 8, W2CH = (C2CH % 4 == 2) ? C2CH : C2CH + 2, W2ROW = W2CH * 16;
  static constexpr int W2_OFF = S_OFF + 2 * S_BYTES, W2_BYTES = HAS3 ? C1 * W2ROW : 0;
  static constexpr int LDS = W2_OFF + W2_BYTES;
  static_assert(TC1 % 2 == 0, "layer 1's channel tiles are dealt in pairs");
};

template <int TCS, int TC1, bool HAS3>
__global__ __launch_bounds__(512) void conv_stem2_kernel(const Stem2Args a) {
  typedef Stem2Geo<TCS, TC1, HAS3> G;
  constexpr int NPAIR = TC1 / 2;
  constexpr int SROW = G::SROW, CPT = G::CPT, NG = G::NG, WROW = G::WROW, TPW = G::TPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wg = wave & 3;                  // wave group (its own tile and patch), wave within it
  const int frow = lane & 15, fq = lane >> 4;
  unsigned char* const wl = smem;
  int32_t* const koff = reinterpret_cast<int32_t*>(smem + G::W_BYTES);
  unsigned char* const sl = smem + G::S_OFF + grp * G::S_BYTES;

  const int Gd = gridDim.x;
  const int first = (blockIdx.x & 7) * (Gd >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < a.ntiles) ? (a.ntiles - first + Gd - 1) / Gd : 0;
  if (my_tiles == 0) return;
  const int nk = (my_tiles - grp + 1) / 2;                   // tiles of this group: the workgroup's tiles grp, grp + 2, ...
  const int phases = max(2 * ((my_tiles + 1) / 2), 2 * (my_tiles / 2) + 1);      // group 0: stem at 2k, conv at 2k+1; group 1 one later

  // ---- layer 1's weights and its tap table -> LDS, once per workgroup
  {
    const int cpr = NG * 4;
    const size_t row_bytes = (size_t)a.kpad * 2;
    const unsigned char* wgp = reinterpret_cast<const unsigned char*>(a.w1);
    // LDS row r holds output channel pi(r) (the pair deal of conv_h2.h: tile 2p row 4q+j -> channel 32p + 8q + j, tile 2p+1 -> + 4):
    // a lane ends up with 8 CONSECUTIVE channels of its pixel over a tile pair - one 16-byte store per pair, and, for HAS3,
    // exactly the MFMA B fragment of the 1x1 conv's K slice p: layer 1's activations feed it without leaving the registers.
    auto pi = [](int r) { const int ti = r >> 4, rho = r & 15; return 32 * (ti >> 1) + 8 * (rho >> 2) + 4 * (ti & 1) + (rho & 3); };
    for (int e = tid; e < G::C1 * G::WCH; e += 512) {
      const int n = e / G::WCH, c = e - n * G::WCH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < cpr && (size_t)(c + 1) * 16 <= row_bytes) v = *reinterpret_cast<const uint4*>(wgp + (size_t)pi(n) * row_bytes + c * 16);
      *reinterpret_cast<uint4*>(wl + n * WROW + c * 16) = v;
    }
    if (tid < 2) *reinterpret_cast<uint4*>(wl + G::C1 * WROW + tid * 16) = make_uint4(0, 0, 0, 0);
    if constexpr (HAS3) {
      const unsigned char* w2p = reinterpret_cast<const unsigned char*>(a.w2);
      const size_t row2 = (size_t)a.kpad2 * 2;
      for (int e = tid; e < G::C1 * G::C2CH; e += 512) {
        const int n = e / G::C2CH, c = e - n * G::C2CH;
        *reinterpret_cast<uint4*>(smem + G::W2_OFF + n * G::W2ROW + c * 16) = *reinterpret_cast<const uint4*>(w2p + (size_t)pi(n) * row2 + c * 16);
      }
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
    for (int r = 0; r < 4; ++r) b1v[i][r] = a.b1[32 * (i >> 1) + 8 * fq + 4 * (i & 1) + r];      // this lane's channels under pi
  float b2v[HAS3 ? TC1 : 1][4];
  if constexpr (HAS3) {
#pragma unroll
    for (int i = 0; i < TC1; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) b2v[i][r] = a.b2[32 * (i >> 1) + 8 * fq + 4 * (i & 1) + r];
  }

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const int Hs = a.H / 2, Ws = a.W / 2, Ho = a.H / 4, Wo = a.W / 4;
  // this wave's stem pixels: 16-pixel tiles wg, wg + 4, ... of the group's 17 x 17 patch
  int sy[TPW], sx[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int ps = (wg + 4 * j) * 16 + frow;
    sy[j] = (ps < G::NSPX) ? ps / kS2Sw : -10000;
    sx[j] = ps % kS2Sw;
  }
  uint32_t abase[TC1];
#pragma unroll
  for (int i = 0; i < TC1; ++i) abase[i] = (uint32_t)((i * 16 + frow) * WROW + fq * 16);
  // layer 1: this wave's 16 output pixels = rows 2 wg, 2 wg + 1 of the 8 x 8 tile; tap (0, 0) of pixel (oy, ox) is patch pixel (2 oy, 2 ox)
  const int oy_l = 2 * wg + (frow >> 3), ox_l = frow & 7;
  const int bbase = ((2 * oy_l) * kS2Sw + 2 * ox_l) * SROW;

  auto tile_coords = [&](int tile, int* b, int* ty, int* tx) {
    const uint32_t bb = magic_div((uint32_t)tile, a.mg_img_mul, a.mg_img_shift);
    const uint32_t r = (uint32_t)tile - bb * (uint32_t)(a.tiles_x * a.tiles_y);
    *ty = (int)magic_div(r, a.mg_tx_mul, a.mg_tx_shift); *tx = (int)r - *ty * a.tiles_x; *b = (int)bb;
  };
  // Frame bytes of this wave's stem pixels (stem_kernel's gather): lane (pixel, q < 3) loads the 12 bytes around the 9-byte run of
  // window row q; cut out and handed to the q = 3 lanes at consumption.  Issued a phase ahead (under layer 1's K loop).
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b96(rin, 0u, 0, 0)) v3u_t;
  auto window_off = [&](int b, int ty, int tx, int j, bool* vm, bool* left, bool* rowok) -> int {
    const int gy = 2 * ty * kS2Oh - 1 + sy[j], gx = 2 * tx * kS2Ow - 1 + sx[j];          // stem-map coordinates
    *vm = gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
    *left = gx == 0;
    *rowok = *vm && fq < 3 && !(gy == 0 && fq == 0);
    return ((b * a.H + 2 * gy - 1 + (fq < 3 ? fq : 0)) * a.W + 2 * gx - 1) * 3 + (*left ? 3 : 0);
  };
  auto fetch = [&](int tile, v3u_t (&u)[TPW]) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      bool vm, left, rowok;
      const int o = window_off(b, ty, tx, j, &vm, &left, &rowok);
      u[j] = __builtin_amdgcn_raw_buffer_load_b96(rin, rowok ? ((uint32_t)o & ~3u) : 0x80000000u, 0, 0);
    }
  };
  auto stem_phase = [&](int tile, const v3u_t (&u)[TPW]) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int ps = (wg + 4 * j) * 16 + frow;
      bool vm, left, rowok;
      const int o = window_off(b, ty, tx, j, &vm, &left, &rowok);
      const uint32_t sh = (uint32_t)o & 3u;
      uint32_t lo = __builtin_amdgcn_alignbyte((uint32_t)u[j][1], (uint32_t)u[j][0], sh);
      uint32_t hi = __builtin_amdgcn_alignbyte((uint32_t)u[j][2], (uint32_t)u[j][1], sh);
      uint32_t b8 = ((uint32_t)u[j][2] >> (8u * sh)) & 0xFFu;
      if (left) { b8 = (hi >> 8) & 0xFFu; hi = (hi << 24) | (lo >> 8); lo = lo << 24; }
      const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(frow * 4, (int)b8);
      const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute((frow + 16) * 4, (int)b8);
      const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute((frow + 32) * 4, (int)b8);
      if (fq == 3) { lo = r0 | (r1 << 8) | (r2 << 16); hi = 0u; }
      f16x8 xb;
#pragma unroll
      for (int k = 0; k < 8; ++k) xb[k] = (half_t)((float)(((k < 4 ? lo : hi) >> (8 * (k & 3))) & 0xFFu) * (1.0f / 255.0f));
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
  };
  auto conv_phase = [&](int tile) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
    f32x4 acc1[TC1];
#pragma unroll
    for (int i = 0; i < TC1; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // software pipeline: K group kg + 1's seven fragments are read before kg's six MFMAs are issued (a wave has ONE pixel tile:
    // left to the compiler each group's reads wait behind the previous group's MFMAs and every group exposes an LDS latency)
    uint4 bfn = *reinterpret_cast<const uint4*>(sl + bbase + koff[fq]), afn[TC1];
#pragma unroll
    for (int i = 0; i < TC1; ++i) afn[i] = *reinterpret_cast<const uint4*>(wl + abase[i]);
#pragma unroll
    for (int kg = 0; kg < NG; ++kg) {
      uint4 bf = bfn;
      if (kg == NG - 1 && (G::NCH & 3) && fq >= (G::NCH & 3)) bf = make_uint4(0, 0, 0, 0);     // chunks beyond the row (see Stem2Geo)
      uint4 af[TC1];
#pragma unroll
      for (int i = 0; i < TC1; ++i) af[i] = afn[i];
      if (kg + 1 < NG) {
        bfn = *reinterpret_cast<const uint4*>(sl + bbase + koff[(kg + 1) * 4 + fq]);
#pragma unroll
        for (int i = 0; i < TC1; ++i) afn[i] = *reinterpret_cast<const uint4*>(wl + abase[i] + (kg + 1) * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TC1; ++i) Mma<half_t>::run(af[i], bf, acc1[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int m = (b * Ho + ty * kS2Oh + oy_l) * Wo + tx * kS2Ow + ox_l;
    // layer 1's activations: per channel-tile pair 8 consecutive channels of this lane's pixel, f16
    uint4 y1[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x0 = acc1[2 * p][r] + b1v[2 * p][r], x1 = acc1[2 * p + 1][r] + b1v[2 * p + 1][r];
        if (a.act1) { x0 = silu_fast(x0); x1 = silu_fast(x1); }
        v[r] = x0; v[4 + r] = x1;
      }
      const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
      y1[p] = *reinterpret_cast<const uint4*>(&hv);
    }
    if constexpr (!HAS3) {
#pragma unroll
      for (int p = 0; p < NPAIR; ++p)
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4ie_t*>(&y1[p]), rdst, (uint32_t)((m * a.dst_ld + a.dst_choff + 32 * p + 8 * fq) * 2), 0, 0);
    } else {
      // the 1x1 conv behind layer 1 (a C2f's cv1, C1 -> C1): y1[p] IS the B fragment of its K slice p (channels 32 p + 8 fq .. + 7
      // of pixel frow); K slices in ascending order, as the ring kernel runs them.  Layer 1's map never touches HBM either.
      const unsigned char* w2l = smem + G::W2_OFF;
      f32x4 acc2[TC1];
#pragma unroll
      for (int i = 0; i < TC1; ++i) acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        uint4 af2[TC1];
#pragma unroll
        for (int i = 0; i < TC1; ++i) af2[i] = *reinterpret_cast<const uint4*>(w2l + (i * 16 + frow) * G::W2ROW + (p * 4 + fq) * 16);
#pragma unroll
        for (int i = 0; i < TC1; ++i) Mma<half_t>::run(af2[i], y1[p], acc2[i]);
      }
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x0 = acc2[2 * p][r] + b2v[2 * p][r], x1 = acc2[2 * p + 1][r] + b2v[2 * p + 1][r];
          if (a.act2) { x0 = silu_fast(x0); x1 = silu_fast(x1); }
          v[r] = x0; v[4 + r] = x1;
        }
        const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4ie_t*>(&hv), rdst, (uint32_t)((m * a.dst_ld + a.dst_choff + 32 * p + 8 * fq) * 2), 0, 0);
      }
    }
  };

  // ---- phases: group g computes the stem patch of its k-th tile in phase 2k + g and runs layer 1 on it in phase 2k + g + 1
  v3u_t un[TPW];
  if (nk > 0) fetch(first + grp * Gd, un);
  __syncthreads();                                           // weights and tap table in place
  for (int p = 0; p < phases; ++p) {
    const int q = p - grp;
    const int k = q >> 1;
    if (q >= 0 && k < nk) {                                  // wave-uniform
      const int tile = first + (2 * k + grp) * Gd;
      if (!(q & 1)) {
        stem_phase(tile, un);
      } else {
        if (k + 1 < nk) fetch(tile + 2 * Gd, un);            // the next patch's frame bytes fly under this K loop
        conv_phase(tile);
      }
    }
    __syncthreads();
  }
}

// host side ------------------------------------------------------------------------------------------------------
inline bool stem2_shape_ok(int C0, int C1, int H, int W, bool has3, size_t* lds) {
  if (H % (4 * kS2Oh) || W % (4 * kS2Ow)) return false;           // whole 8 x 8 tiles of layer 1's map
  if (C0 == 48 && C1 == 96) *lds = has3 ? Stem2Geo<3, 6, true>::LDS : Stem2Geo<3, 6, false>::LDS;
  else if (C0 == 16 && C1 == 32) *lds = has3 ? Stem2Geo<1, 2, true>::LDS : Stem2Geo<1, 2, false>::LDS;
  else if (C0 == 32 && C1 == 64) *lds = has3 ? Stem2Geo<2, 4, true>::LDS : Stem2Geo<2, 4, false>::LDS;
  else return false;
  return *lds <= 160 * 1024;
}

inline hipError_t launch_conv_stem2(Stem2Args a, int C0, int C1, bool has3, hipStream_t s, int ncu) {
  size_t lds;
  if (!stem2_shape_ok(C0, C1, a.H, a.W, has3, &lds)) return hipErrorInvalidValue;
  a.tiles_x = a.W / 4 / kS2Ow; a.tiles_y = a.H / 4 / kS2Oh; a.ntiles = a.B * a.tiles_x * a.tiles_y;
  host_magic((uint32_t)(a.tiles_x * a.tiles_y), &a.mg_img_mul, &a.mg_img_shift);
  host_magic((uint32_t)a.tiles_x, &a.mg_tx_mul, &a.mg_tx_shift);
  long grid = std::min<long>(a.ntiles, ncu);
  grid = (grid + 7) / 8 * 8;
  const dim3 g((unsigned)grid), b(512);
  if (C0 == 48) { if (has3) hipLaunchKernelGGL((conv_stem2_kernel<3, 6, true>), g, b, lds, s, a); else hipLaunchKernelGGL((conv_stem2_kernel<3, 6, false>), g, b, lds, s, a); }
  else if (C0 == 16) { if (has3) hipLaunchKernelGGL((conv_stem2_kernel<1, 2, true>), g, b, lds, s, a); else hipLaunchKernelGGL((conv_stem2_kernel<1, 2, false>), g, b, lds, s, a); }
  else { if (has3) hipLaunchKernelGGL((conv_stem2_kernel<2, 4, true>), g, b, lds, s, a); else hipLaunchKernelGGL((conv_stem2_kernel<2, 4, false>), g, b, lds, s, a); }
  return hipGetLastError();
}

inline hipError_t set_stem2_attrs() {
  hipError_t e = hipSuccess;
  auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
  set(reinterpret_cast<const void*>(conv_stem2_kernel<3, 6, true>)); set(reinterpret_cast<const void*>(conv_stem2_kernel<3, 6, false>));
  set(reinterpret_cast<const void*>(conv_stem2_kernel<1, 2, true>)); set(reinterpret_cast<const void*>(conv_stem2_kernel<1, 2, false>));
  set(reinterpret_cast<const void*>(conv_stem2_kernel<2, 4, true>)); set(reinterpret_cast<const void*>(conv_stem2_kernel<2, 4, false>));
  return e;
}

}  // namespace miyolo

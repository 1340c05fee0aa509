// 3x3 stride-1 convolution, f16: the halo-slab kernel (conv_h2.h) re-cut for THREE workgroups per CU.
//
// Why (round 3, profiles/r03_h2_stamps.log): a conv_h2 workgroup lives ~58 k cycles per 256-pixel x 96-channel tile, of
// which its own MFMAs are 10.4 k; the rest is the cold slab fetch (setup 2 k + blocked DMA issue 3-8 k + landing 2 k), the
// SiLU epilogue (9-12 k beside a partner that holds half the SIMD's issue slots), slab re-issue and loop overheads, and a
// tap loop that runs at 1 690 cycles per 768 cycles of MFMA while the partner workgroup is in ITS tap loop.  Two workgroups
// per CU cover each other's non-MFMA phases only about a third of the time (matrix pipe 34-41 % busy, PMC).  None of the
// phases is bound by a saturated unit, so the lever is occupancy: a third workgroup per CU.  That needs <= 53 KiB of LDS
// and <= 168 VGPRs per workgroup, which the 256-pixel tile cannot have (55 KiB slab + 24 KiB weight ring, 96 accumulator
// registers + fragments), so:
//   * 128-pixel tiles (16 x 8 on the 160- / 80-wide maps, 40 x 3 on the 40-wide ones): four waves x 32 pixels x all BN
//     channels = 48 accumulator registers per lane;
//   * the slab stored LINEARLY with pitch HP = TW + 2 LDS rows per halo row (conv_h2 pads every halo row to a multiple of
//     8 rows because a DMA instruction writes 8 rows; here the 8-row DMA groups simply run across halo-row boundaries - the
//     source address is per lane anyway): 23 KiB instead of 30 KiB for the 10 x 18 halo of a 16 x 8 tile;
//   * the same 2-slot weight ring, tap loop, K order (chunk, tap, channel), swizzle by halo column and 8-channel epilogue
//     stores as conv_h2: results are BIT-IDENTICAL to conv_h2's (tests/test_gpu_conv.py).
// Cost: the weight slab of a tap now feeds half as many MFMAs (L2->LDS weight bytes per FLOP x2, LDS fragment bytes per
// FLOP x1.6).  MEASURED (profiles/r03_h3_vs_h2.md): on the 80 x 80 and 40 x 40 maps every layer is 1-9 % SLOWER than on
// conv_h2 - the occupancy hypothesis above does not hold there.  Where it wins is the 20 x 20 level, whose layers give
// conv_h2 384 tiles for 512 workgroup slots and this kernel 768 for 768 (-7 % per layer, profiles/r03_h3_on_p5.md): the
// engine takes it exactly where its tile count fills the chip better (h3_preferred below), conv_h2 everywhere else.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_h2.h"

namespace miyolo {

template <int GEO> struct H3Geo;
template <> struct H3Geo<0> { static constexpr int TW = 16, TH = 8; };    // 160-, 80-wide maps (and anything else)
template <> struct H3Geo<1> { static constexpr int TW = 40, TH = 3; };    // 40-wide maps
template <> struct H3Geo<2> { static constexpr int TW = 20, TH = 6; };    // 20-wide maps
inline void h3_geo(int geo, int* tw, int* th) { *tw = geo == 1 ? 40 : geo == 2 ? 20 : 16; *th = geo == 1 ? 3 : geo == 2 ? 6 : 8; }

constexpr int kH3LdsMax = 53 * 1024;     // three workgroups per CU (160 KiB / 3 = 53.3 KiB)

template <int TC, int GEO>
__global__ __launch_bounds__(256, 3) void conv_h3_kernel(const ConvArgs a, const H2Geom g) {
  using T = half_t;
  constexpr int TW = H3Geo<GEO>::TW, TH = H3Geo<GEO>::TH, HP = TW + 2;
  static_assert(HP % 2 == 0, "LDS row parity must equal the halo column's parity (bank-conflict swizzle)");
  constexpr int NPX = TW * TH;
  constexpr int SROWS = (TH + 2) * HP;             // slab rows (one halo pixel x 128 bytes of channels each)
  constexpr int NG = (SROWS + 7) / 8;              // 8-row DMA groups
  constexpr int NGW = (NG + 3) / 4;                // groups per wave
  constexpr int SLAB = NG * 1024;
  constexpr int ES = 2, CE = 8, CPR = 64;          // f16: 64 channels per 128-byte row
  constexpr int BN = TC * 16;
  constexpr int BNP = (BN + 31) / 32 * 32;         // weight rows per slot: 4 waves x 8 rows per DMA
  constexpr int NWI = BNP / 32;
  constexpr int WSLOT = BNP * ROW_BYTES;
  constexpr int TPW = 2;                           // pixel tiles (16 pixels) per wave
  constexpr int NPAIR = TC / 2;
  static_assert(NPX <= 128 && SLAB + 2 * WSLOT <= kH3LdsMax, "tile does not fit a third of the CU's LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kOob = 0x80000000u;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  const int H = a.Hin, W = a.Win;

  // tiles: channel tile fastest, dealt to XCDs in contiguous chunks (as conv_h2, non-persistent form)
  const int nblk = (int)gridDim.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
  const int xstart = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xlen = xq + (xcd < xr ? 1 : 0);
  if (slot_ >= xlen) return;
  const int L = xstart + slot_;
  const uint32_t t1 = magic_div((uint32_t)L, g.mg_nb_mul, g.mg_nb_shift);
  const int nb = L - (int)t1 * g.NB;
  const uint32_t t2 = magic_div(t1, g.mg_tx_mul, g.mg_tx_shift);
  const int tx = (int)(t1 - t2 * (uint32_t)g.tiles_x);
  const uint32_t bimg = magic_div(t2, g.mg_ty_mul, g.mg_ty_shift);
  const int ty = (int)(t2 - bimg * (uint32_t)g.tiles_y);
  const int y0 = ty * TH, x0 = tx * TW, n0 = nb * BN;

  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int ldB = a.src[0].ld * ES;
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, g.bias_bytes, 0x00020000);

  // ---- weight DMA rows of this lane (identical to conv_h2: the MFMA-row deal that gives a lane 8 consecutive channels)
  uint32_t woff[NWI];
  const int cgw = (lane & 7) ^ ((((8 * wave + (lane >> 3)) >> 1)) & 7);
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int ti = r >> 4, rho = r & 15;
    const int ch = (ti < 2 * NPAIR) ? 32 * (ti >> 1) + 8 * (rho >> 2) + 4 * (ti & 1) + (rho & 3) : r;
    const int n = n0 + ch;
    woff[i] = (r < BN && n < a.cout) ? (uint32_t)(n * a.kpad * ES + cgw * 16) : kOob;
  }
  auto issue_w = [&](int c, int tap, int slot) {
    const uint32_t st = lds_base + (uint32_t)(SLAB + slot * WSLOT + wave * 1024);
    const uint32_t kofs = (uint32_t)((tap * a.cin + c * CPR) * ES);
    const uint32_t inv = ((c * CPR + cgw * CE) < a.cin) ? 0u : kOob;
#pragma unroll
    for (int i = 0; i < NWI; ++i) lds_dma16(rsw, st + i * 4096, (woff[i] + kofs) | inv);
  };

  // ---- slab DMA: wave w fills the 8-row groups w, w + 4, ...; lane (lane >> 3) of a group holds LDS row r = 8 g + (lane >> 3)
  // = halo pixel (hy, hx) = (r / HP, r % HP); its 16-byte slot s = lane & 7 holds channel chunk s ^ h2_swz(hx) (^ 4 on odd halo
  // rows of the 20-wide geometry): conv_h2.h's conflict-free swizzle, by halo column, so a tap's offset dy * HP + dx moves a
  // fragment read by a constant per dx
  int32_t soff[NGW];         // source byte offset of this lane's pixel relative to the tile's halo origin, chunk column included; < 0: never valid
  int32_t scol[NGW];         // chunk column cg of this lane in group i (for the channel-tail test)
#pragma unroll
  for (int i = 0; i < NGW; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int hy = r / HP, hx = r - hy * HP;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const int cg = (lane & 7) ^ h2_swz(hx) ^ (H2RowFlip<GEO>::value ? 4 * (hy & 1) : 0);
    const bool ok = (wave + 4 * i < NG) && hy < TH + 2 && y >= 0 && y < H && x >= 0 && x < W;
    soff[i] = ok ? (int32_t)((((int)bimg * H + y) * W + x) * ldB + a.src[0].ch_off * ES + cg * 16) : -1;
    scol[i] = cg;
  }
  auto issue_slab = [&](int c) {
#pragma unroll
    for (int i = 0; i < NGW; ++i) {
      if (wave + 4 * i < NG) {                                                   // wave-uniform; false only in the last round
        const bool ok = soff[i] >= 0 && (c * CPR + scol[i] * CE) < a.cin;
        lds_dma16(rs0, lds_base + (uint32_t)((wave + 4 * i) * 1024), ok ? (uint32_t)(soff[i] + c * ROW_BYTES) : kOob);
      }
    }
  };

  // ---- per-lane fragment addresses
  uint32_t baddr[TPW][3];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int p = (wave * TPW + j) * 16 + frow;
    const uint32_t pp = p < NPX ? (uint32_t)p : 0u;
    const int py = (int)(pp / (uint32_t)TW);
    const int px = (int)pp - py * TW;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      baddr[j][dx] = (uint32_t)((py * HP + hx) * ROW_BYTES + ((fq ^ h2_swz(hx) ^ (H2RowFlip<GEO>::value ? 4 * (py & 1) : 0)) << 4));
    }
  }
  const uint32_t aaddr = lds_off(frow, fq);

  issue_slab(0);
  issue_w(0, 0, 0);

  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i) {           // accumulators start at the bias
    const int ch = (i < 2 * NPAIR) ? 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : 16 * i + 4 * fq;
    const v4ie_t bv = __builtin_amdgcn_raw_buffer_load_b128(rbias, (uint32_t)((n0 + ch) * 4), 0, 0);
    const f32x4 bf = {__int_as_float(bv[0]), __int_as_float(bv[1]), __int_as_float(bv[2]), __int_as_float(bv[3])};
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = bf;
  }

  auto compute = [&](int dy, auto dx_tag, int slot, bool full) __attribute__((always_inline)) {
    constexpr int dx = decltype(dx_tag)::value;
    const unsigned char* ws = smem + SLAB + slot * WSLOT;
    const unsigned char* xs = smem + dy * (HP * ROW_BYTES);
    const uint32_t rf = H2RowFlip<GEO>::value ? (uint32_t)((dy & 1) << 6) : 0u;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 1 && !full) break;                     // tail chunk with <= half a row of channels (wave-uniform)
      uint4 af[TC], bf[TPW];
#pragma unroll
      for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + (aaddr ^ (uint32_t)(kk << 6)) + i * 16 * ROW_BYTES);
#pragma unroll
      for (int j = 0; j < TPW; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + (baddr[j][dx] ^ (uint32_t)(kk << 6) ^ rf));
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
    }
  };

  // ---- K loop: chunk outermost; per tap one barrier, the next tap's weights in flight under this tap's MFMAs (dy is a real
  // loop, dx unrolled: code size, see conv_h2.h)
  int slot = 0;
  for (int c = 0; c < g.nchunk; ++c) {
    const bool full = (a.cin - c * CPR) > CPR / 2;
    const bool more = (c + 1 < g.nchunk);
#pragma unroll 1
    for (int dy = 0; dy < 3; ++dy) {
#define MIYOLO_H3_TAP(DX)                                                                           \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                               \
      if (dy * 3 + (DX) < 8) issue_w(c, dy * 3 + (DX) + 1, slot ^ 1);                             \
      else if (more) issue_w(c + 1, 0, slot ^ 1);                                                 \
      compute(dy, std::integral_constant<int, DX>{}, slot, full);                                 \
      slot ^= 1;
      MIYOLO_H3_TAP(0) MIYOLO_H3_TAP(1) MIYOLO_H3_TAP(2)
#undef MIYOLO_H3_TAP
    }
    if (more) {
      asm volatile("s_barrier" ::: "memory");          // every wave is done with the slab
      issue_slab(c + 1);
    }
  }

  // ---- epilogue: SiLU, residual, 8 channels per store (the arithmetic of conv_h2's f16 epilogue, operation for operation)
  int32_t mpix[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int p = (wave * TPW + j) * 16 + frow;
    const int py = (int)((uint32_t)p / (uint32_t)TW);
    const int y = y0 + py, x = x0 + p - py * TW;
    mpix[j] = (p < NPX && y < H && x < W) ? ((int)bimg * H + y) * W + x : -1;
  }
  auto act = [&](float x) -> float { return a.act ? silu_fast(x) : x; };
  v2i_t rlast[TPW];
  if constexpr (TC & 1) {
    if (a.res) {
      const int n = n0 + 16 * (TC - 1) + 4 * fq;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const uint32_t ro = (n < a.cout && mpix[j] >= 0) ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
        rlast[j] = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0);
      }
    }
  }
  auto pair_row = [&](auto ip_tag) __attribute__((always_inline)) {
    constexpr int ip = decltype(ip_tag)::value;
    const int n = n0 + 32 * ip + 8 * fq;
    v4ie_t rrow[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) rrow[j] = (v4ie_t){0, 0, 0, 0};
    if (a.res) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const uint32_t ro = (n < a.cout && mpix[j] >= 0) ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
        rrow[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const bool ok = n < a.cout && mpix[j] >= 0;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = act(acc[2 * ip][j][r] * 1.f); v[4 + r] = act(acc[2 * ip + 1][j][r] * 1.f); }
      if (a.res) {
        const f16x8 hr = *reinterpret_cast<const f16x8*>(&rrow[j]);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)hr[r];
      }
      const uint32_t so = ok ? (uint32_t)((mpix[j] * a.dst_ld + a.dst_choff + n) * ES) : kOob;
      const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4ie_t*>(&hv), rdst, so, 0, MIYOLO_ST_AUX);
    }
  };
  if constexpr (NPAIR > 0) pair_row(std::integral_constant<int, 0>{});
  if constexpr (NPAIR > 1) pair_row(std::integral_constant<int, 1>{});
  if constexpr (NPAIR > 2) pair_row(std::integral_constant<int, 2>{});
  static_assert(NPAIR <= 3, "epilogue rows are written out for up to three channel-tile pairs");
  if constexpr (TC & 1) {                              // unpaired last channel tile: 4 channels per lane
    constexpr int i = TC - 1;
    const int n = n0 + 16 * i + 4 * fq;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act(acc[i][j][r] * 1.f);
      const bool ok = n < a.cout && mpix[j] >= 0;
      if (a.res) {
        const f16x4 hr = *reinterpret_cast<const f16x4*>(&rlast[j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)hr[r];
      }
      const uint32_t so = ok ? (uint32_t)((mpix[j] * a.dst_ld + a.dst_choff + n) * ES) : kOob;
      const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2i_t*>(&hv), rdst, so, 0, MIYOLO_ST_AUX);
    }
  }
}

// host side ------------------------------------------------------------------------------------------------------
inline bool h3_shape(int cin, int cout, int B, int H, int W, int tc, H2Geom* g, size_t* lds, int* geo) {
  const size_t wring = (size_t)2 * ((tc * 16 + 31) / 32 * 32) * ROW_BYTES;
  double best = -1.0;
  int bgeo = -1;
  for (int ge = 0; ge < 3; ++ge) {
    int tw, th;
    h3_geo(ge, &tw, &th);
    const size_t need = (size_t)(((th + 2) * (tw + 2) + 7) / 8) * 1024 + wring;
    if (need > (size_t)kH3LdsMax) continue;
    const double tiles = (double)((W + tw - 1) / tw) * ((H + th - 1) / th);
    const double util = (double)W * H / (tiles * 128.0);
    const double score = util - 1e-3 * ge;
    if (score > best) { best = score; bgeo = ge; }
  }
  if (bgeo < 0) return false;
  h3_geo(bgeo, &g->TW, &g->TH);
  g->HP = g->TW + 2; g->GX = 0;
  g->tiles_x = (W + g->TW - 1) / g->TW; g->tiles_y = (H + g->TH - 1) / g->TH; g->NB = (cout + tc * 16 - 1) / (tc * 16);
  g->ntiles = B * g->tiles_x * g->tiles_y * g->NB;
  g->nchunk = (cin + 63) / 64;
  g->npx = g->TH * g->TW;
  g->slab_bytes = (((g->TH + 2) * g->HP + 7) / 8) * 1024;
  host_magic((uint32_t)g->TW, &g->mg_tw_mul, &g->mg_tw_shift);
  host_magic((uint32_t)g->NB, &g->mg_nb_mul, &g->mg_nb_shift);
  host_magic((uint32_t)g->tiles_x, &g->mg_tx_mul, &g->mg_tx_shift);
  host_magic((uint32_t)g->tiles_y, &g->mg_ty_mul, &g->mg_ty_shift);
  g->bias_bytes = (uint32_t)((cout + 127) / 128 * 128 * 4);
  g->scratch_off = 0; g->warm = 0;
  *lds = (size_t)g->slab_bytes + wring;
  *geo = bgeo;
  return true;
}
inline double h3_util(const H2Geom& g, int H, int W) { return (double)W * H / ((double)g.tiles_x * g.tiles_y * 128.0); }

inline bool h3_geometry(const ConvArgs& a, H2Geom* g, size_t* lds, int* tc, int* geo) {
  H2Geom g2; size_t l2; int geo2;
  if (!h2_geometry<half_t>(a, &g2, &l2, tc, &geo2)) return false;        // the same shape conditions as conv_h2 (f16)
  if (*tc != 6 && *tc != 4 && *tc != 3) return false;
  return h3_shape(a.cin, a.cout, a.B, a.Hout, a.Wout, *tc, g, lds, geo);
}
inline bool h3_eligible(const ConvArgs& a, double min_util) {
  H2Geom g; size_t lds; int tc, geo;
  return h3_geometry(a, &g, &lds, &tc, &geo) && h3_util(g, a.Hout, a.Wout) >= min_util;
}

// When to prefer conv_h3 (measured, profiles/r03_h3_vs_h2.md and r03_h3_on_p5.md): it is 1-9 % slower per tile-round than
// conv_h2, so it only pays where conv_h2's tile count fills its 2 x CUs workgroup slots badly and conv_h3's fills its
// 3 x CUs slots well - the 20 x 20 maps of the P5 level (288 -> 288: 384 tiles on 512 slots against 768 on 768: -7 %).
// Criterion: round efficiency = tiles / (slots x ceil(tiles / slots)) at least 0.15 better.
inline bool h3_preferred(const ConvArgs& a, int ncu) {
  H2Geom g2, g3; size_t l2, l3; int tc, geo2, geo3;
  if (!h2_geometry<half_t>(a, &g2, &l2, &tc, &geo2) || !h3_geometry(a, &g3, &l3, &tc, &geo3)) return false;
  auto eff = [](int tiles, int slots) { return (double)tiles / ((double)slots * ((tiles + slots - 1) / slots)); };
  return eff(g3.ntiles, 3 * ncu) >= eff(g2.ntiles, 2 * ncu) + 0.15;
}

template <int TC>
inline void launch_h3_tc(const ConvArgs& a, const H2Geom& g, int geo, size_t lds, hipStream_t s) {
  const dim3 grid((unsigned)g.ntiles), blk(256);
  switch (geo) {
    case 1: hipLaunchKernelGGL((conv_h3_kernel<TC, 1>), grid, blk, lds, s, a, g); break;
    case 2: hipLaunchKernelGGL((conv_h3_kernel<TC, 2>), grid, blk, lds, s, a, g); break;
    default: hipLaunchKernelGGL((conv_h3_kernel<TC, 0>), grid, blk, lds, s, a, g); break;
  }
}
inline hipError_t launch_conv_h3(const ConvArgs& a, hipStream_t s) {
  H2Geom g; size_t lds; int tc, geo;
  if (!h3_geometry(a, &g, &lds, &tc, &geo)) return hipErrorInvalidValue;
  switch (tc) {
    case 3: launch_h3_tc<3>(a, g, geo, lds, s); break;
    case 4: launch_h3_tc<4>(a, g, geo, lds, s); break;
    default: launch_h3_tc<6>(a, g, geo, lds, s); break;
  }
  return hipGetLastError();
}
inline hipError_t set_h3_attrs() {
  hipError_t e;
#define MIYOLO_H3_ATTR(TC, GEO)                                                                         \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h3_kernel<TC, GEO>),                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, kH3LdsMax)) != hipSuccess) return e;
  MIYOLO_H3_ATTR(3, 0) MIYOLO_H3_ATTR(4, 0) MIYOLO_H3_ATTR(6, 0) MIYOLO_H3_ATTR(3, 1) MIYOLO_H3_ATTR(4, 1) MIYOLO_H3_ATTR(6, 1)
  MIYOLO_H3_ATTR(3, 2) MIYOLO_H3_ATTR(4, 2) MIYOLO_H3_ATTR(6, 2)
#undef MIYOLO_H3_ATTR
  return hipSuccess;
}

}  // namespace miyolo

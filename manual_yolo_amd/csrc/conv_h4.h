// 3x3 stride-1 convolution, f16: the halo-slab kernel (conv_h2.h / conv_h3.h) cut the OTHER way - ONE workgroup per CU, one
// wave per SIMD with the whole register file (512 registers: 192 of them accumulators), 480- / 512-pixel tiles, a
// 128-pixel x 96-channel wave tile.  ROUND-3 EXPERIMENT (engine option "h4", off): see DESIGN.md 8.2 "what the next attempt
// should be".  Against conv_h2 a tap's weight slab feeds twice as many MFMAs (48 per wave and K slice), LDS fragment reads per
// MFMA drop from 0.42 to 0.29, a tile's fixed costs are paid once per 512 pixels - and nothing hides a wave's own latencies
// but its own instruction stream: the next K slice's fragments are read under the current slice's MFMAs (software pipelining,
// the registers are there).  Same K order, swizzle, weight-row deal and epilogue arithmetic as conv_h2 / conv_h3: results
// are BIT-IDENTICAL to theirs (tests/test_gpu_conv.py::test_h4_*).
// MEASURED (profiles/r03_h4_experiment.md): 38-45 % SLOWER than conv_h2 on every 80 x 80 layer (96 -> 96: 125-130 us against
// 88-96).  A 480-pixel tile takes 75 k cycles for 20.7 k cycles of MFMA per wave: the 74 KiB slab is fetched cold twice per
// tile (two channel chunks, ~7 k cycles each at the CU's ~11 B/clk miss rate) and nothing runs beside it; the epilogue (192
// values per lane, 6.7 k) likewise; and a tap step takes ~3 k cycles for 1.5 k of MFMA because hipcc puts a full
// s_waitcnt lgkmcnt(0) in front of the first MFMA of every step (see compute()).  What two workgroups per CU give conv_h2
// for free has to be scheduled by hand here - slab of the next chunk under the current taps, epilogue under the next tile's
// taps - and the K loop written in assembly.  Kept as a tested, documented starting point; off.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_h2.h"
#include "conv_h3.h"

namespace miyolo {

template <int GEO> struct H4Geo;
template <> struct H4Geo<0> { static constexpr int TW = 16, TH = 32; };   // 160-wide maps (and anything else)
template <> struct H4Geo<1> { static constexpr int TW = 40, TH = 12; };   // 80- and 40-wide maps (480 pixels: 30 of the 32 pixel tiles)
inline void h4_geo(int geo, int* tw, int* th) { *tw = geo == 1 ? 40 : 16; *th = geo == 1 ? 12 : 32; }

constexpr int kH4LdsMax = 160 * 1024;    // one workgroup per CU

template <int TC, int GEO>
__global__ __launch_bounds__(256, 1) void conv_h4_kernel(const ConvArgs a, const H2Geom g) {
  using T = half_t;
  constexpr int TW = H4Geo<GEO>::TW, TH = H4Geo<GEO>::TH, HP = TW + 2;
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
  constexpr int TPW = 8;                           // pixel tiles (16 pixels) per wave: 128 pixels x all BN channels
  constexpr int NPAIR = TC / 2;
  static_assert(NPX <= 512 && SLAB + 2 * WSLOT <= kH4LdsMax, "tile does not fit the CU's LDS");
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
    const int cg = (lane & 7) ^ h2_swz(hx) ^ (false ? 4 * (hy & 1) : 0);
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
      baddr[j][dx] = (uint32_t)((py * HP + hx) * ROW_BYTES + ((fq ^ h2_swz(hx) ^ (false ? 4 * (py & 1) : 0)) << 4));
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

  // One tap step, software-pipelined: this step's kk = 0 pixel fragments (bfA) were read under the PREVIOUS step's MFMAs (the slab
  // does not change within a chunk); here the weight fragments of both K slices, the kk = 1 pixel fragments and the NEXT tap's
  // kk = 0 pixel fragments are all issued up front (28 ds_read_b128, 112 registers) and the 96 MFMAs run behind them - only
  // the first six weight reads (behind the barrier: the weights arrived by DMA) are exposed.
  // One tap step, software-pipelined: this step's kk = 0 pixel fragments (bfA) were read under the PREVIOUS step's MFMAs (the slab
  // does not change within a chunk); here the weight fragments of both K slices, the kk = 1 pixel fragments and the NEXT tap's
  // kk = 0 pixel fragments are issued up front (28 ds_read_b128, 112 registers) and the 96 MFMAs run behind them.
  // (hipcc waits for ALL of them - s_waitcnt lgkmcnt(0) - before the first MFMA: the 4-bit counter cannot say "the first six".
  // Dictating the interleave with sched_group_barrier - six reads, then one read per two MFMAs - made the allocator spill
  // 248 registers.  This loop needs to be written in assembly.)
  uint4 bfA[TPW];
  auto compute = [&](int dy, auto dx_tag, int slot, bool full, bool first, bool has_next) __attribute__((always_inline)) {
    constexpr int dx = decltype(dx_tag)::value;
    constexpr int dxn = dx == 2 ? 0 : dx + 1;
    const unsigned char* ws = smem + SLAB + slot * WSLOT;
    const unsigned char* xs = smem + dy * (HP * ROW_BYTES);
    const unsigned char* xsn = smem + (dx == 2 ? dy + 1 : dy) * (HP * ROW_BYTES);
    uint4 af0[TC], af1[TC], bf1[TPW], bfn[TPW];
#pragma unroll
    for (int i = 0; i < TC; ++i) af0[i] = *reinterpret_cast<const uint4*>(ws + aaddr + i * 16 * ROW_BYTES);
    if (first) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) bfA[j] = *reinterpret_cast<const uint4*>(xs + baddr[j][dx]);
    }
    if (full) {
#pragma unroll
      for (int i = 0; i < TC; ++i) af1[i] = *reinterpret_cast<const uint4*>(ws + (aaddr ^ 64u) + i * 16 * ROW_BYTES);
#pragma unroll
      for (int j = 0; j < TPW; ++j) bf1[j] = *reinterpret_cast<const uint4*>(xs + (baddr[j][dx] ^ 64u));
    }
    if (has_next) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) bfn[j] = *reinterpret_cast<const uint4*>(xsn + baddr[j][dxn]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TPW; ++j) Mma<T>::run(af0[i], bfA[j], acc[i][j]);
    if (full) {
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) Mma<T>::run(af1[i], bf1[j], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) bfA[j] = bfn[j];
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
#define MIYOLO_H4_TAP(DX)                                                                           \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                               \
      if (dy * 3 + (DX) < 8) issue_w(c, dy * 3 + (DX) + 1, slot ^ 1);                             \
      else if (more) issue_w(c + 1, 0, slot ^ 1);                                                 \
      compute(dy, std::integral_constant<int, DX>{}, slot, full, dy == 0 && (DX) == 0, !(dy == 2 && (DX) == 2)); \
      slot ^= 1;
      MIYOLO_H4_TAP(0) MIYOLO_H4_TAP(1) MIYOLO_H4_TAP(2)
#undef MIYOLO_H4_TAP
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
inline bool h4_shape(int cin, int cout, int B, int H, int W, int tc, H2Geom* g, size_t* lds, int* geo) {
  const size_t wring = (size_t)2 * ((tc * 16 + 31) / 32 * 32) * ROW_BYTES;
  double best = -1.0;
  int bgeo = -1;
  for (int ge = 0; ge < 2; ++ge) {
    int tw, th;
    h4_geo(ge, &tw, &th);
    const size_t need = (size_t)(((th + 2) * (tw + 2) + 7) / 8) * 1024 + wring;
    if (need > (size_t)kH4LdsMax) continue;
    const double tiles = (double)((W + tw - 1) / tw) * ((H + th - 1) / th);
    const double util = (double)W * H / (tiles * 512.0);
    const double score = util - 1e-3 * ge;
    if (score > best) { best = score; bgeo = ge; }
  }
  if (bgeo < 0) return false;
  h4_geo(bgeo, &g->TW, &g->TH);
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
inline double h4_util(const H2Geom& g, int H, int W) { return (double)W * H / ((double)g.tiles_x * g.tiles_y * 512.0); }

inline bool h4_geometry(const ConvArgs& a, H2Geom* g, size_t* lds, int* tc, int* geo) {
  H2Geom g2; size_t l2; int geo2;
  if (!h2_geometry<half_t>(a, &g2, &l2, tc, &geo2)) return false;        // the same shape conditions as conv_h2 (f16)
  if (*tc != 6 && *tc != 4) return false;
  return h4_shape(a.cin, a.cout, a.B, a.Hout, a.Wout, *tc, g, lds, geo);
}
inline bool h4_eligible(const ConvArgs& a, double min_util) {
  H2Geom g; size_t lds; int tc, geo;
  return h4_geometry(a, &g, &lds, &tc, &geo) && h4_util(g, a.Hout, a.Wout) >= min_util;
}

template <int TC>
inline void launch_h4_tc(const ConvArgs& a, const H2Geom& g, int geo, size_t lds, hipStream_t s) {
  const dim3 grid((unsigned)g.ntiles), blk(256);
  switch (geo) {
    case 1: hipLaunchKernelGGL((conv_h4_kernel<TC, 1>), grid, blk, lds, s, a, g); break;
    default: hipLaunchKernelGGL((conv_h4_kernel<TC, 0>), grid, blk, lds, s, a, g); break;
  }
}
inline hipError_t launch_conv_h4(const ConvArgs& a, hipStream_t s) {
  H2Geom g; size_t lds; int tc, geo;
  if (!h4_geometry(a, &g, &lds, &tc, &geo)) return hipErrorInvalidValue;
  switch (tc) {
    case 4: launch_h4_tc<4>(a, g, geo, lds, s); break;
    default: launch_h4_tc<6>(a, g, geo, lds, s); break;
  }
  return hipGetLastError();
}
inline hipError_t set_h4_attrs() {
  hipError_t e;
#define MIYOLO_H4_ATTR(TC, GEO)                                                                         \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h4_kernel<TC, GEO>),                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, kH4LdsMax)) != hipSuccess) return e;
  MIYOLO_H4_ATTR(4, 0) MIYOLO_H4_ATTR(6, 0) MIYOLO_H4_ATTR(4, 1) MIYOLO_H4_ATTR(6, 1)
#undef MIYOLO_H4_ATTR
  return hipSuccess;
}

}  // namespace miyolo

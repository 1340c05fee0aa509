// 3x3 stride-1 convolution, fourth generation: HALO-SLAB kernel, two 4-wave workgroups per CU.
//
// Why (profiles/r01_conv_stamps.md): on the ring kernel (conv_dmap.h) a K step of a 256 px x 96 ch tile
// costs 768 MFMA cycles, 704 cycles of the CU's 64 B/clk L2->LDS path (the activation tile is re-fetched for every one
// of the nine taps), ~800 LDS cycles (DMA writes + 14 KiB of fragment reads per wave) and an un-overlapped epilogue of
// ~840 cycles per step: four resources of equal size taking turns.  This kernel changes the shape of the work:
//   * the activation HALO SLAB - (TH+2) x (TW+2) input pixels x one 128-byte channel chunk - is fetched ONCE per chunk
//     and serves all nine taps: a tap is a constant row offset into the slab (dy*HP + dx); only the weight slab
//     (BN rows x 128 B) is fetched per tap.  L2->LDS bytes per FLOP drop 2.7x (96 channels) ... 1.9x (192).
//   * four waves per workgroup, each owning 64 pixels x ALL BN channels (4 x TC accumulator tiles, up to 96 VGPRs):
//     LDS fragment bytes per FLOP drop 1.4x against the 64 x 48 wave tiles of the ring kernel.
//   * TWO such workgroups per CU (<= 80 KiB of LDS and <= 256 VGPRs each): while one is in its epilogue, its slab
//     reload or at a barrier, the other one's MFMAs keep the matrix pipe busy - the overlap the one-workgroup kernels
//     never achieved with wave roles (profiles/r01_ws_kernel.md).
//   * epilogue: the bias is the accumulators' initial value; weight rows are dealt to MFMA rows so that a lane ends up
//     with 8 CONSECUTIVE channels of a pixel across two channel tiles -> one 16-byte store per 8 channels (half the
//     store instructions of the 4-channel form; the store tail of these epilogues is issue-bound).
// LDS image: rows of 128 B, chunk c of row r at r*128 + ((c ^ ((r>>1)&7))<<4) as in the ring kernels; the slab pitch HP
// (rows per halo row) is a multiple of 8, so a tap's row offset changes the swizzle by at most bit 2 (HP % 16 == 8, odd
// dy), which is the same as swapping the two 64-byte halves of the row: one XOR on a wave-uniform constant.
// K order: channel chunk outermost, then tap, then channel - NOT the flattened (tap, channel) order of conv_dmap.h, so
// fp32 results differ from the ring kernels in the last bits (tests compare against torch, not bit-for-bit).
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dmap.h"

namespace miyolo {

struct H2Geom {
  int32_t TW, TH, HP, GX;            // tile width / height (pixels), slab pitch (LDS rows per halo row), 8-pixel DMA groups per halo row
  int32_t tiles_x, tiles_y, NB;      // pixel tiles per image row / column, channel tiles
  int32_t ntiles, nchunk, npx;       // workgroups, 128-byte channel chunks of cin, TH*TW (<= 256)
  int32_t slab_bytes, flip;          // (TH+2)*HP*128; 1: HP % 16 == 8
  uint32_t mg_tw_mul, mg_tw_shift;   // magic division by TW
  uint32_t mg_nb_mul, mg_nb_shift;   // by NB
  uint32_t mg_tx_mul, mg_tx_shift;   // by tiles_x
  uint32_t mg_ty_mul, mg_ty_shift;   // by tiles_y
  uint32_t bias_bytes;               // padded bias array size
};

constexpr int kH2LdsMax = 80 * 1024;     // two workgroups per CU

template <typename T, int TC>
__global__ __launch_bounds__(256, 2) void conv_h2_kernel(const ConvArgs a, const H2Geom g) {
  constexpr int ES = (int)sizeof(T);
  constexpr int CE = DT<T>::CE;
  constexpr int CPR = 8 * CE;                    // channels per 128-byte row
  constexpr int BN = TC * 16;
  constexpr int BNP = (BN + 31) / 32 * 32;       // weight rows per slot: 4 waves x 8 rows per DMA
  constexpr int NWI = BNP / 32;
  constexpr int WSLOT = BNP * ROW_BYTES;
  constexpr int TPW = 4;
  constexpr int NPAIR = TC / 2;                  // channel-tile pairs with 8-channel stores
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kOob = 0x80000000u;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;

  // ---- workgroup -> tile: XCD-contiguous renumbering (bijective), channel tile fastest so that the channel tiles
  // sharing a halo run side by side on one XCD
  int L = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = L & 7, slot = L >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  uint32_t t1 = magic_div((uint32_t)L, g.mg_nb_mul, g.mg_nb_shift);
  const int nb = L - (int)t1 * g.NB;
  uint32_t t2 = magic_div(t1, g.mg_tx_mul, g.mg_tx_shift);
  const int tx = (int)(t1 - t2 * (uint32_t)g.tiles_x);
  const uint32_t bimg = magic_div(t2, g.mg_ty_mul, g.mg_ty_shift);
  const int ty = (int)(t2 - bimg * (uint32_t)g.tiles_y);
  const int y0 = ty * g.TH, x0 = tx * g.TW, n0 = nb * BN;
  const int H = a.Hin, W = a.Win;

  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int ldB = a.src[0].ld * ES;

  // ---- weight DMA rows of this lane: LDS row r of a slot holds channel n0 + pi(r), pi = the MFMA-row deal that gives a
  // lane 8 consecutive channels over a pair of channel tiles (tile 2p row 4q+j -> channel 32p + 8q + j, tile 2p+1 -> +4)
  uint32_t woff[NWI];
  const int cgw_x = (lane & 7);
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int r = 8 * (wave + 4 * i) + (lane >> 3);
    const int ti = r >> 4, rho = r & 15;
    const int ch = (ti < 2 * NPAIR) ? 32 * (ti >> 1) + 8 * (rho >> 2) + 4 * (ti & 1) + (rho & 3) : r;
    const int cg = cgw_x ^ ((r >> 1) & 7);
    const int n = n0 + ch;
    woff[i] = (r < BN && n < a.cout) ? (uint32_t)(n * a.kpad * ES + cg * 16) : kOob;
  }
  // chunk validity of this lane's weight chunk column is the same for all i:  (r>>1)&7 varies with i only through
  // 8*(wave+4i) -> (4*(wave+4i))&7 = 4*(wave&1): constant over i
  const int cgw = cgw_x ^ ((((8 * wave + (lane >> 3)) >> 1)) & 7);

  auto issue_w = [&](int c, int tap, int slot) {
    const uint32_t st = lds_base + (uint32_t)(g.slab_bytes + slot * WSLOT + wave * 1024);
    const uint32_t kofs = (uint32_t)((tap * a.cin + c * CPR) * ES);
    const uint32_t inv = ((c * CPR + cgw * CE) < a.cin) ? 0u : kOob;
#pragma unroll
    for (int i = 0; i < NWI; ++i) lds_dma16(rsw, st + i * 4096, (woff[i] + kofs) | inv);
  };

  // ---- halo slab DMA: wave w fills halo rows w, w+4, ...; instruction (hy, gx) = LDS rows hy*HP + 8*gx .. +7
  const int hxl = lane >> 3;
  auto issue_slab = [&](int c) {
    for (int hy = wave; hy < g.TH + 2; hy += 4) {
      const int y = y0 - 1 + hy;
      const bool yok = (y >= 0) && (y < H);
      const int rowb = ((int)bimg * H + y) * W;             // pixel index of (y, 0)
      const int swy = (hy * (g.HP >> 1)) & 7;
      for (int gx = 0; gx < g.GX; ++gx) {
        const int hx = gx * 8 + hxl;
        const int x = x0 - 1 + hx;
        const int cg = (lane & 7) ^ ((swy + (hx >> 1)) & 7);
        const bool ok = yok && (x >= 0) && (x < W) && (hx < g.TW + 2) && ((c * CPR + cg * CE) < a.cin);
        const uint32_t off = ok ? (uint32_t)((rowb + x) * ldB + (a.src[0].ch_off + c * CPR) * ES + cg * 16) : kOob;
        lds_dma16(rs0, lds_base + (uint32_t)((hy * g.HP + gx * 8) * ROW_BYTES), off);
      }
    }
  };

  // ---- per-lane fragment addresses and output pixels
  uint32_t baddr[TPW][3];
  int32_t mpix[TPW];                     // output pixel index (b*H + y)*W + x, or -1
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int p = (wave * TPW + j) * 16 + frow;
    const bool pv = p < g.npx;
    const uint32_t pp = pv ? (uint32_t)p : 0u;
    const int py = (int)magic_div(pp, g.mg_tw_mul, g.mg_tw_shift);
    const int px = (int)pp - py * g.TW;
    const int rb = py * g.HP + px;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int r = rb + dx;
      baddr[j][dx] = (uint32_t)(r * ROW_BYTES + ((fq ^ ((r >> 1) & 7)) << 4));
    }
    const int y = y0 + py, x = x0 + px;
    mpix[j] = (pv && y < H && x < W) ? ((int)bimg * H + y) * W + x : -1;
  }
  const uint32_t aaddr = lds_off(frow, fq);

  // ---- accumulators start at the bias
  f32x4 acc[TC][TPW];
  {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, g.bias_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      const int ch = (i < 2 * NPAIR) ? 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : 16 * i + 4 * fq;
      const v4ie_t bv = __builtin_amdgcn_raw_buffer_load_b128(rb, (uint32_t)((n0 + ch) * 4), 0, 0);
      const f32x4 bf = {__int_as_float(bv[0]), __int_as_float(bv[1]), __int_as_float(bv[2]), __int_as_float(bv[3])};
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[i][j] = bf;
    }
  }

  auto compute = [&](auto dy_tag, auto dx_tag, int slot, bool full) __attribute__((always_inline)) {
    constexpr int dy = decltype(dy_tag)::value, dx = decltype(dx_tag)::value;
    const unsigned char* ws = smem + g.slab_bytes + slot * WSLOT;
    const unsigned char* xs = smem + dy * g.HP * ROW_BYTES;
    const uint32_t fl = (g.flip & dy & 1) ? 64u : 0u;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 1 && !full) break;                     // tail chunk with <= half a row of channels (wave-uniform)
      uint4 af[TC], bf[TPW];
#pragma unroll
      for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + (aaddr ^ (uint32_t)(kk << 6)) + i * 16 * ROW_BYTES);
#pragma unroll
      for (int j = 0; j < TPW; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + (baddr[j][dx] ^ ((uint32_t)(kk << 6) ^ fl)));
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
    }
  };

  // ---- K loop: chunk outermost; per tap one barrier, the next tap's weights in flight under this tap's MFMAs
  issue_slab(0);
  issue_w(0, 0, 0);
  int slot = 0;
  for (int c = 0; c < g.nchunk; ++c) {
    const bool full = (a.cin - c * CPR) > CPR / 2;
    const bool more = (c + 1 < g.nchunk);
#define MIYOLO_H2_TAP(T_, DY, DX)                                                                   \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                   \
    if ((T_) < 8) issue_w(c, (T_) + 1, slot ^ 1); else if (more) issue_w(c + 1, 0, slot ^ 1);       \
    compute(std::integral_constant<int, DY>{}, std::integral_constant<int, DX>{}, slot, full);      \
    slot ^= 1;
    MIYOLO_H2_TAP(0, 0, 0) MIYOLO_H2_TAP(1, 0, 1) MIYOLO_H2_TAP(2, 0, 2)
    MIYOLO_H2_TAP(3, 1, 0) MIYOLO_H2_TAP(4, 1, 1) MIYOLO_H2_TAP(5, 1, 2)
    MIYOLO_H2_TAP(6, 2, 0) MIYOLO_H2_TAP(7, 2, 1) MIYOLO_H2_TAP(8, 2, 2)
#undef MIYOLO_H2_TAP
    if (more) {
      asm volatile("s_barrier" ::: "memory");          // every wave is done with the slab
      issue_slab(c + 1);
    }
  }

  // ---- epilogue: activation, residual, 8 channels per store
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  auto act = [&](float x) -> float {
    if (!a.act) return x;
    if constexpr (ES == 4) return silu_exact(x); else return silu_fast(x);
  };
#pragma unroll
  for (int ip = 0; ip < NPAIR; ++ip) {
    const int n = n0 + 32 * ip + 8 * fq;
    const bool nok = n < a.cout;
    v4ie_t r0[TPW], r1[TPW];
    if (a.res) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const uint32_t ro = (nok && mpix[j] >= 0) ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
        r0[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
        if constexpr (ES == 4) r1[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 16);
      }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = act(acc[2 * ip][j][r]); v[4 + r] = act(acc[2 * ip + 1][j][r]); }
      if (a.res) {
        if constexpr (ES == 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] += __int_as_float(r0[j][r]); v[4 + r] += __int_as_float(r1[j][r]); }
        } else {
          const f16x8 hr = *reinterpret_cast<const f16x8*>(&r0[j]);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)hr[r];
        }
      }
      const uint32_t so = (nok && mpix[j] >= 0) ? (uint32_t)((mpix[j] * a.dst_ld + a.dst_choff + n) * ES) : kOob;
      if constexpr (ES == 4) {
        const v4ie_t o0 = {__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
        const v4ie_t o1 = {__float_as_int(v[4]), __float_as_int(v[5]), __float_as_int(v[6]), __float_as_int(v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(o0, rdst, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, rdst, so, 0, 16);
      } else {
        const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4ie_t*>(&hv), rdst, so, 0, 0);
      }
    }
  }
  if constexpr (TC & 1) {                                // unpaired last channel tile: 4 channels per lane
    constexpr int i = TC - 1;
    const int n = n0 + 16 * i + 4 * fq;
    const bool nok = n < a.cout;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act(acc[i][j][r]);
      const bool ok = nok && mpix[j] >= 0;
      if (a.res) {
        const uint32_t ro = ok ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
        if constexpr (ES == 4) {
          const v4ie_t rr = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += __int_as_float(rr[r]);
        } else {
          const v2i_t rr = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0);
          const f16x4 hr = *reinterpret_cast<const f16x4*>(&rr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)hr[r];
        }
      }
      const uint32_t so = ok ? (uint32_t)((mpix[j] * a.dst_ld + a.dst_choff + n) * ES) : kOob;
      if constexpr (ES == 4) {
        const v4ie_t o = {__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o, rdst, so, 0, 0);
      } else {
        const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2i_t*>(&hv), rdst, so, 0, 0);
      }
    }
  }
}

// host side ------------------------------------------------------------------------------------------------------
template <int TC> constexpr size_t h2_wring_bytes() { return (size_t)2 * ((TC * 16 + 31) / 32 * 32) * ROW_BYTES; }

inline int h2_pick_tc(int cout) {
  // channel tile: 96 where it divides or the tail is small, else 64 / 48
  if (cout <= 48) return 3;
  if (cout <= 64) return 4;
  if (cout % 96 == 0) return 6;
  if (cout % 64 == 0) return 4;
  return 6;
}

// Tile geometry for an H x W map: the TW x TH (TH = 256 / TW) shape with the best pixel utilisation whose slab fits.
inline bool h2_shape(int cin, int cout, int B, int H, int W, int elem, int ce, int tc, H2Geom* g, size_t* lds) {
  static const int cands[] = {16, 32, 20, 40, 8, 12, 24, 48, 64, 80, 10};
  const size_t wring = (size_t)2 * ((tc * 16 + 31) / 32 * 32) * ROW_BYTES;
  double best = -1.0;
  int btw = 0;
  for (int tw : cands) {
    if (tw > ((W + 7) / 8 * 8) && tw != 8) continue;
    const int th = 256 / tw;
    const int hp = (tw + 2 + 7) / 8 * 8;
    const size_t need = (size_t)(th + 2) * hp * ROW_BYTES + wring;
    if (need > (size_t)kH2LdsMax) continue;
    const double tiles = (double)((W + tw - 1) / tw) * ((H + th - 1) / th);
    const double util = (double)W * H / (tiles * 256.0);
    const double fillrows = (double)(th + 2) * ((tw + 2 + 7) / 8) * 8;
    const double score = util - 1e-4 * fillrows / 256.0;          // ties: the smaller halo
    if (score > best) { best = score; btw = tw; }
  }
  if (btw == 0) return false;
  const int cpr = 8 * ce;
  g->TW = btw; g->TH = 256 / btw; g->HP = (btw + 2 + 7) / 8 * 8; g->GX = (btw + 2 + 7) / 8;
  g->tiles_x = (W + g->TW - 1) / g->TW; g->tiles_y = (H + g->TH - 1) / g->TH; g->NB = (cout + tc * 16 - 1) / (tc * 16);
  g->ntiles = B * g->tiles_x * g->tiles_y * g->NB;
  g->nchunk = (cin + cpr - 1) / cpr;
  g->npx = g->TH * g->TW;
  g->slab_bytes = (g->TH + 2) * g->HP * ROW_BYTES;
  g->flip = (g->HP % 16) == 8;
  host_magic((uint32_t)g->TW, &g->mg_tw_mul, &g->mg_tw_shift);
  host_magic((uint32_t)g->NB, &g->mg_nb_mul, &g->mg_nb_shift);
  host_magic((uint32_t)g->tiles_x, &g->mg_tx_mul, &g->mg_tx_shift);
  host_magic((uint32_t)g->tiles_y, &g->mg_ty_mul, &g->mg_ty_shift);
  g->bias_bytes = (uint32_t)((cout + 127) / 128 * 128 * 4);
  *lds = (size_t)g->slab_bytes + wring;
  (void)elem;
  return true;
}

inline double h2_util(const H2Geom& g, int H, int W) {
  return (double)W * H / ((double)g.tiles_x * g.tiles_y * 256.0);
}

template <typename T>
inline bool h2_geometry(const ConvArgs& a, H2Geom* g, size_t* lds, int* tc) {
  constexpr int ES = (int)sizeof(T);
  if (a.ksize != 3 || a.stride != 1 || a.nsrc != 1 || a.src[0].up || a.out_f32) return false;
  if (a.Hin != a.Hout || a.Win != a.Wout) return false;
  if ((a.src[0].ch_off * ES) % 16 || (a.src[0].ld * ES) % 16 || a.cin % DT<T>::CE) return false;
  if (a.cout % 8 || a.dst_ld % 8 || a.dst_choff % 8) return false;                 // 8-channel stores
  if (a.res && (a.res_ld % 8 || a.res_choff % 8)) return false;
  if ((long)a.B * a.Hin * a.Win * a.src[0].ld * ES >= (1l << 31)) return false;
  *tc = h2_pick_tc(a.cout);
  return h2_shape(a.cin, a.cout, a.B, a.Hout, a.Wout, ES, DT<T>::CE, *tc, g, lds);
}

template <typename T>
inline bool h2_eligible(const ConvArgs& a, double min_util) {
  H2Geom g; size_t lds; int tc;
  return h2_geometry<T>(a, &g, &lds, &tc) && h2_util(g, a.Hout, a.Wout) >= min_util;
}

template <typename T>
inline hipError_t launch_conv_h2(const ConvArgs& a, hipStream_t s) {
  H2Geom g; size_t lds; int tc;
  if (!h2_geometry<T>(a, &g, &lds, &tc)) return hipErrorInvalidValue;
  switch (tc) {
    case 3: hipLaunchKernelGGL((conv_h2_kernel<T, 3>), dim3((unsigned)g.ntiles), dim3(256), lds, s, a, g); break;
    case 4: hipLaunchKernelGGL((conv_h2_kernel<T, 4>), dim3((unsigned)g.ntiles), dim3(256), lds, s, a, g); break;
    default: hipLaunchKernelGGL((conv_h2_kernel<T, 6>), dim3((unsigned)g.ntiles), dim3(256), lds, s, a, g); break;
  }
  return hipGetLastError();
}

template <typename T>
inline hipError_t set_h2_attrs() {
  hipError_t e;
#define MIYOLO_H2_ATTR(TC)                                                                              \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<T, TC>),                    \
                               hipFuncAttributeMaxDynamicSharedMemorySize, kH2LdsMax)) != hipSuccess) return e;
  MIYOLO_H2_ATTR(3) MIYOLO_H2_ATTR(4) MIYOLO_H2_ATTR(6)
#undef MIYOLO_H2_ATTR
  return hipSuccess;
}

}  // namespace miyolo

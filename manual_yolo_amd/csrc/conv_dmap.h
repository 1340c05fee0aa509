// Implicit-GEMM convolution, third generation: PERSISTENT workgroups over the LDS-DMA ring.
//
// Same tile math as conv_dma.h.  What the ablation timings of that kernel showed
// (profiles/r01_conv_dma_ablation.md): the MFMAs and LDS reads are already hidden, and 35-65 % of
// a launch is per-workgroup fixed cost that nothing overlaps when one 512-thread workgroup
// owns a CU - launch, index arithmetic, the latency of the first two tile DMAs, the epilogue's
// store drain - multiplied by 6-25 rounds of workgroups per launch.  Here a launch has at most
// one workgroup per CU and each walks a static list of tiles:
//   * the DMA stream (tile, K step) runs two steps ahead of the MFMA stream ACROSS tile
//     boundaries: while tile t is in its epilogue, the first two stages of tile t+1 are
//     already landing;
//   * per-tile index arithmetic is a few multiply-high "magic" divisions done in the shadow of
//     the K loop (host supplies the constants), instead of ~40-instruction integer divides;
//   * the ring slot counter and the vmcnt accounting simply continue across tiles.
// Tiles are dealt so that the tiles sharing an activation tile run on one XCD in the same time
// slot.  Exit condition: a static trip count per workgroup (no queues, no spinning).
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dma.h"

#ifndef MIYOLO_DMAP_INTERLEAVE
#define MIYOLO_DMAP_INTERLEAVE 1
#endif
#ifndef MIYOLO_DMAP_DEFER
#define MIYOLO_DMAP_DEFER 0
#endif
#ifndef MIYOLO_DMAP_BALANCED_GRID
#define MIYOLO_DMAP_BALANCED_GRID 1
#endif
// Timing experiment (-DMIYOLO_HALO_EXPT=1, results wrong): the activation DMAs of the six taps with dx != 0 are sent out
// of range (they still issue and write zeros, but move no data from L2) - an upper bound on what fetching each activation
// once per row of taps instead of once per tap would buy, before any cost of doing so.
#ifndef MIYOLO_HALO_EXPT
#define MIYOLO_HALO_EXPT 0
#endif
// L2 warm-up: when the DMA stream moves on to a tile, touch the activation lines of
// the tile AFTER it with plain loads, so that the ring's DMAs find them in L2 a tile later.
#ifndef MIYOLO_DMAP_L2WARM
#define MIYOLO_DMAP_L2WARM 0
#endif
// The interleaved step issues its DMAs unconditionally - in the last two steps of a workgroup, where there is nothing left
// to fetch, with every offset out of range (zeros into a slot nobody reads) - so that no scalar branch sits between the
// MFMAs (tools/probes/probe_step.hip: interleaving is worth 22 % in the bare skeleton).  Measured: -0.2 % on the step, so off.
#ifndef MIYOLO_DMAP_ALWAYS_ISSUE
#define MIYOLO_DMAP_ALWAYS_ISSUE 0
#endif
#ifndef MIYOLO_DMAP_EXACT_VMCNT
#define MIYOLO_DMAP_EXACT_VMCNT 1
#endif

// The 256 x 192 tile (2-slot ring): activations and weights in SEPARATE rings - three activation slots (two K steps ahead),
// two weight slots (one ahead): 144 KiB instead of 112.  See the kernel.
// 2 (default): as 1, with the step's DMAs spread between its MFMAs (as MIYOLO_DMAP_INTERLEAVE does for the 3-slot shapes):
// -6 % / -9.5 % on the two stride-2 3x3 layers of this shape (model.3, model.16) against the burst form, the 1x1 layers even.
#ifndef MIYOLO_DMAP_SPLIT
#define MIYOLO_DMAP_SPLIT 2
#endif

namespace miyolo {

// Ring depth per tile shape: the 256 x 192 tile (56 KiB per stage) only fits twice; its steps are twice as
// long (48 MFMAs per wave), which is what hides the DMA latency with one stage in flight.
template <int WC, int TC>
constexpr int dmap_stages() { return (WC * TC * 16 > 128) ? 2 : 3; }
template <int WC, int TC>
constexpr size_t dmap_lds_bytes() {
  if (MIYOLO_DMAP_SPLIT && dmap_stages<WC, TC>() == 2)
    return (size_t)(3 * DMA_BM + 2 * ((WC * TC * 16 + 63) / 64 * 64)) * ROW_BYTES;
  return (size_t)dmap_stages<WC, TC>() * (DMA_BM + (WC * TC * 16 + 63) / 64 * 64) * ROW_BYTES;
}

__device__ __forceinline__ uint32_t magic_div(uint32_t x, uint32_t mul, uint32_t shift) {
  // floor(x / d) for x < 2^31; (mul, shift) from host_magic(d); d == 1 is encoded as mul == 0
  return mul ? (__umulhi(x, mul) >> shift) : x;
}

template <typename T, int KS, int WC, int TC>
__global__ __launch_bounds__(512) void conv_dmap_kernel(const ConvArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int WP = 8 / WC;
  constexpr int TPW = DMA_BM / (WP * 16);
  constexpr int BM = DMA_BM;
  constexpr int BN = WC * TC * 16;
  constexpr int BNP = (BN + 63) / 64 * 64;
  constexpr int ROWS = BM + BNP;
  constexpr int NI = ROWS / 64;
  constexpr int XI = BM / 64;
  constexpr int STAGE = ROWS * ROW_BYTES;
  constexpr int NST = dmap_stages<WC, TC>();   // ring slots: 3 (two steps ahead), 2 for the 192-channel tile
  // SPLIT (the 2-slot shape, round 3): with one ring of two (activations + weights) slots ONE K step is in flight while the
  // other is computed on, and a step takes what a fill takes (~5.5 k cycles on the HBM-bound 1x1 layers against 1.5 k of
  // MFMAs).  The weights are L2-resident and need no depth; the activations do.  So: three activation slots (X(c + 2) goes
  // out in step c) and two weight slots (W(c + 1)); per step the weights are issued FIRST, so that waiting for W(c) and
  // X(c) leaves exactly the youngest activation stage (XI DMAs per wave) in flight.
  constexpr bool SPLIT = MIYOLO_DMAP_SPLIT && NST == 2;
  constexpr int XSTAGE = BM * ROW_BYTES, WSTAGE = BNP * ROW_BYTES, NXS = 3, NWS = 2;
  // f16 outputs with 8-channel aligned views (a.pair8, set by launch_conv_dmap): weight rows are dealt to MFMA rows so that a
  // lane holds 8 CONSECUTIVE channels of a pixel over a pair of channel tiles -> one 16-byte store per pair and pixel tile
  // instead of two 8-byte ones: 16 rows x 64 B per store instruction instead of 16 x 32 B.  Same arithmetic per output:
  // results do not change.  In isolation the 32-byte pieces write at 4.4-5.5 TB/s, the 64-byte ones at 5.7-6.4
  // (tools/probes/probe_store_patterns.hip, profiles/r03_probe_store_patterns.log).
  constexpr int NPAIRW = (sizeof(T) <= 2) ? TC / 2 : 0;      // f16: 16-byte stores; fp8: 8-byte stores (instead of 4)
  const bool pair8 = NPAIRW > 0 && a.pair8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC, wc = wave % WC;

  // ---- static tile list of this workgroup: slot k -> tile k*G + (b%8)*(G/8) + b/8 (G = grid,
  // a multiple of 8): the G/8 workgroups of one XCD group take consecutive tiles, and
  // consecutive tiles share the pixel tile (channel tile index runs fastest)
  const int NB = (a.cout + BN - 1) / BN;
  const int MB = (a.M + BM - 1) / BM;
  const int ntiles = MB * NB;
  const int G = gridDim.x;
  const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < ntiles) ? (ntiles - first + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int total_steps = my_tiles * a.nk;

  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rs1 = make_srd(a.src[1].ptr, a.src[1].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr uint32_t kOob = 0x80000000u;

  const int rsub = lane >> 3;
  const int cg = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  const int ct0 = a.src[0].ch_cnt / CE;
  const int ct1 = (a.nsrc > 1) ? a.src[1].ch_cnt / CE : 0;
  const int HWo = a.Hout * a.Wout;

  // ---- K table (built once per workgroup, reused by every tile): for K step ks and chunk column
  // c the activation chunk sits `kofs` bytes after the row's base pointer and belongs to tap /
  // segment `sel`:   ktab[ks*8 + c] = sel << 28 | kofs.   3x3: sel = tap 0..8, 9 = K tail (the
  // row masks carry a permanently set bit 9); 1x1: sel = segment 0/1, tail = bit 31 of kofs.
  // This replaces a per-step walker (divisions by 3, a divergent carry loop) with one ds_read.
  uint32_t* const ktab = reinterpret_cast<uint32_t*>(smem + (SPLIT ? NXS * XSTAGE + NWS * WSTAGE : NST * STAGE));
  for (int e = tid; e < a.nk * 8; e += 512) {
    const int q = e;                       // chunk index on the flattened K axis
    uint32_t v;
    if constexpr (KS == 3) {
#ifdef MIYOLO_KORDER_EXPT     // timing experiment (results wrong: weights stay tap-major): channel-group-major K order, see DESIGN.md
      int tp, co;
      {
        const int gfull = ct0 / 8, rem = ct0 - gfull * 8;
        if (q < gfull * 72) { const int grp = q / 72, r = q - grp * 72; tp = r / 8; co = grp * 8 + (r - tp * 8); }
        else if (rem > 0) { const int q2 = q - gfull * 72; tp = q2 / rem; co = gfull * 8 + (q2 - tp * rem); }
        else { tp = 9; co = 0; }
      }
#else
      const int tp = q / ct0, co = q - tp * ct0;
#endif
      v = (tp < 9) ? ((uint32_t)tp << 28) | (uint32_t)((((tp / 3) * a.src[0].w + tp % 3) * a.src[0].ld + co * CE) * (int)sizeof(T))
                   : (9u << 28);
    } else {
      const bool s1 = q >= ct0;
      const int cq = s1 ? q - ct0 : q;
      const bool ok = cq < (s1 ? ct1 : ct0);
      v = ok ? ((s1 ? 1u : 0u) << 28) | (uint32_t)(cq * CE * (int)sizeof(T)) : kOob;
    }
    ktab[e] = v;
  }
  __syncthreads();

  // ---- DMA-side state: the tile whose stages are being issued
  int32_t xoff0[XI];
  int32_t xoff1[KS == 1 ? XI : 1];
  uint32_t xinv[XI];         // 3x3: bit t = tap t OUTSIDE the image (bit 9 always set); 1x1: 0 / 0x80000000
  uint32_t woff[NI - XI];
  int d_tile = first, d_ks = 0, d_slot = 0, d_issued = 0;
  int w_tile = first, w_ks = 0, w_slot = 0;            // SPLIT: the weight stream (one step behind the activation stream)
  int v_stores = 0;                      // vector memory ops issued since the last DMA of the youngest stage (wave-uniform)

  // Per-lane row state of `tile`.  DMA i of this wave covers rows 8*(wave + 8*i) + (lane >> 3) and the 8 lanes of a
  // row group share a row, so computing per (lane, i) would do every row 8 times over.  Lane L computes ONE row -
  // (i = (L >> 3) & 3, rsub = L & 7) - and the wave transposes with ds_bpermute: DMA i of lane L takes its values
  // from lane 8*i + (L >> 3)  (profiles/r01_ws_kernel.md: the redundant form cost thousands of VALU cycles per tile).
  auto set_w = [&](int tile) {                         // weight-row offsets of `tile`'s channel tile
    const int mb = tile / NB, nb = tile - mb * NB;
    const int n0 = nb * BN;
#pragma unroll
    for (int i = 0; i < NI - XI; ++i) {
      const int row = 8 * (wave + 8 * i) + rsub;
      int nrow = row;
      if (pair8) {                       // the MFMA-row deal of conv_h2.h within each wave's TC channel tiles: a lane ends up with
        const int ti = row >> 4, rho = row & 15, w = ti / TC, iw = ti - w * TC;      // 8 consecutive channels over a tile pair
        if (iw < 2 * NPAIRW) nrow = (w * TC) * 16 + 32 * (iw >> 1) + 8 * (rho >> 2) + 4 * (iw & 1) + (rho & 3);
      }
      const int n = n0 + nrow;
      woff[i] = (row < BN && n < a.cout) ? (uint32_t)(n * a.kpad * (int)sizeof(T) + cg * 16) : kOob;
    }
  };
  const int bp_base = (lane >> 3) * 4;
  auto setup_tile = [&](int tile) {
    const int mb = tile / NB, nb = tile - mb * NB;       // wave-uniform: scalar unit
    const int m0 = mb * BM, n0 = nb * BN;
    int32_t c_off0, c_off1 = 0;
    uint32_t c_inv;
    {
      const int m = m0 + 8 * (wave + 8 * ((lane >> 3) & (XI - 1))) + (lane & 7);
      const bool vm = m < a.M;
      const uint32_t mm = vm ? (uint32_t)m : 0u;
      const int b = (int)magic_div(mm, a.mg_hw_mul, a.mg_hw_shift);
      const uint32_t rem = mm - (uint32_t)b * (uint32_t)HWo;
      const int ho = (int)magic_div(rem, a.mg_w_mul, a.mg_w_shift);
      const int wo = (int)rem - ho * a.Wout;
      if constexpr (KS == 3) {
        const int hi0 = ho * a.stride - 1, wi0 = wo * a.stride - 1;
        c_off0 = (((b * a.src[0].h + hi0) * a.src[0].w + wi0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
        // rows/cols of the 3x3 window inside the image -> 9-bit tap mask as an outer product
        const uint32_t hm = (hi0 >= 0 ? 1u : 0u) | 2u | ((hi0 + 2 < a.Hin) ? 4u : 0u);
        const uint32_t wm = (wi0 >= 0 ? 1u : 0u) | 2u | ((wi0 + 2 < a.Win) ? 4u : 0u);
        const uint32_t msk = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? (wm << 3) : 0u) | ((hm & 4u) ? (wm << 6) : 0u);
        c_inv = (vm ? (~msk & 0x1FFu) : 0x1FFu) | 0x200u;
      } else {
        const int h0 = a.src[0].up ? (ho >> 1) : ho, w0 = a.src[0].up ? (wo >> 1) : wo;
        c_off0 = (((b * a.src[0].h + h0) * a.src[0].w + w0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
        const int h1 = a.src[1].up ? (ho >> 1) : ho, w1 = a.src[1].up ? (wo >> 1) : wo;
        c_off1 = (((b * a.src[1].h + h1) * a.src[1].w + w1) * a.src[1].ld + a.src[1].ch_off) * (int)sizeof(T);
        c_inv = vm ? 0u : kOob;
      }
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      xoff0[i] = __builtin_amdgcn_ds_bpermute(bp_base + i * 32, c_off0);
      xinv[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + i * 32, (int)c_inv);
      if constexpr (KS == 1) xoff1[i] = __builtin_amdgcn_ds_bpermute(bp_base + i * 32, c_off1);
    }
    if constexpr (!SPLIT) set_w(tile);
  };

  // L2 warm-up of `tile` (3x3 stride-1, single source): lanes 0-31 of a wave own 32 of the tile's 256 pixels (as in
  // setup_tile); lanes 32-63 take the same pixels' second 128-byte line.  Plain dword loads through the raw buffer
  // (out of range -> nothing fetched); their results are summed into `warm_acc`, which is only "used" at the very end,
  // so the compiler never waits for them inside the loop.  They are counted in v_stores for the ring's vmcnt.
  float warm_acc = 0.f;
  const __amdgpu_buffer_rsrc_t rwarm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[0].ptr), 0, a.src[0].bytes, 0x00020000);
  auto warm_tile = [&](int tile) -> int {
    if constexpr (!(MIYOLO_DMAP_L2WARM && KS == 3)) { return 0; }
    else {
      if (a.stride != 1) return 0;
      const int mb = tile / NB;
      const int m = mb * BM + 8 * (wave + 8 * ((lane >> 3) & (XI - 1))) + (lane & 7);
      const bool vm = m < a.M;
      const int cinB = a.cin * (int)sizeof(T);
      const int line = (lane >> 5) * 128;
      const uint32_t base = (uint32_t)((m * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T));      // stride 1: input pixel = output pixel
      int n = 0;
#pragma unroll
      for (int l = 0; l < 2; ++l) {                          // lines 0/128 then 256/384 of the pixel's channel slice
        if (l * 256 < cinB) {                                // wave-uniform
          const uint32_t off = (vm && line + l * 256 < cinB) ? base + (uint32_t)(line + l * 256) : 0x80000000u;
          warm_acc += __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwarm, off, 0, 0));
          ++n;
        }
      }
      return n;
    }
  };

  // issue the NI DMAs of the DMA stream's next (tile, K step); branch-free validity (conv_dma.h)
  // timing experiments (a.ablate, results wrong): 1 = no tile DMA, 128 = activation offsets wrapped
  // into a 1 MiB window (everything L2 resident), 256 = activation rows forced to 128-byte alignment
  const uint32_t ab_and = (ABL(128) ? 0x800FFFFFu : 0xFFFFFFFFu) & (ABL(256) ? 0xFFFFFF8Fu : 0xFFFFFFFFu);   // all ones in the shipped build
  auto issue_next = [&]() {
    const uint32_t st = lds_base + (uint32_t)(d_slot * STAGE + wave * 1024);
    const int ks = d_ks;
    if (!ABL(1)) {
    const uint32_t e = ktab[ks * 8 + cg];
    if constexpr (KS == 3) {
      const uint32_t tp = e >> 28, kofs = e & 0x0FFFFFFFu;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const uint32_t off = ((((uint32_t)xoff0[i] + kofs) | (((xinv[i] >> tp) & 1u) << 31)) & ab_and) | ((MIYOLO_HALO_EXPT && (tp % 3) != 1) ? 0x80000000u : 0u);
        lds_dma16(rs0, st + i * 8192, off);
      }
    } else {
      const bool seg1 = (ks * 8) >= ct0;                      // wave-uniform (segment 0 is K-step aligned)
      const uint32_t kofs = e & 0x8FFFFFFFu;                  // bit 31 = K tail
      if (!seg1) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const uint32_t off = (((uint32_t)xoff0[i] + kofs) | xinv[i]) & ab_and;
          lds_dma16(rs0, st + i * 8192, off);
        }
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const uint32_t off = (((uint32_t)xoff1[i] + kofs) | xinv[i]) & ab_and;
          lds_dma16(rs1, st + i * 8192, off);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NI - XI; ++i) {
      const uint32_t off = woff[i] + (uint32_t)(ks * 128);
      lds_dma16(rsw, st + BM * ROW_BYTES + i * 8192, off);
    }
    }   // !(ablate & 1)
    d_slot = (d_slot == NST - 1) ? 0 : d_slot + 1;
    ++d_issued;
    if (++d_ks == a.nk) {                 // DMA stream moves on to this workgroup's next tile
      d_ks = 0;
      d_tile += G;
      if (d_tile < ntiles) { setup_tile(d_tile); if (d_tile + G < ntiles) v_stores += warm_tile(d_tile + G); }
    }
  };

  // SPLIT: the two streams.  issue_x: the activation DMAs of the activation stream's next (tile, K step) into slot d_slot of
  // three; issue_w: the weight DMAs of the weight stream's next step into slot w_slot of two.
  auto issue_x = [&]() {
    const uint32_t st = lds_base + (uint32_t)(d_slot * XSTAGE + wave * 1024);
    const int ks = d_ks;
    const uint32_t e = ktab[ks * 8 + cg];
    if constexpr (KS == 3) {
      const uint32_t tp = e >> 28, kofs = e & 0x0FFFFFFFu;
#pragma unroll
      for (int i = 0; i < XI; ++i) lds_dma16(rs0, st + i * 8192, ((uint32_t)xoff0[i] + kofs) | (((xinv[i] >> tp) & 1u) << 31));
    } else {
      const bool seg1 = (ks * 8) >= ct0;
      const uint32_t kofs = e & 0x8FFFFFFFu;
      if (!seg1) {
#pragma unroll
        for (int i = 0; i < XI; ++i) lds_dma16(rs0, st + i * 8192, ((uint32_t)xoff0[i] + kofs) | xinv[i]);
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) lds_dma16(rs1, st + i * 8192, ((uint32_t)xoff1[KS == 1 ? i : 0] + kofs) | xinv[i]);
      }
    }
    d_slot = (d_slot == NXS - 1) ? 0 : d_slot + 1;
    ++d_issued;
    if (++d_ks == a.nk) {
      d_ks = 0;
      d_tile += G;
      if (d_tile < ntiles) setup_tile(d_tile);
    }
  };
  auto issue_w = [&]() {
    const uint32_t st = lds_base + (uint32_t)(NXS * XSTAGE + w_slot * WSTAGE + wave * 1024);
#pragma unroll
    for (int i = 0; i < NI - XI; ++i) lds_dma16(rsw, st + i * 8192, woff[i] + (uint32_t)(w_ks * 128));
    w_slot ^= 1;
    if (++w_ks == a.nk) {
      w_ks = 0;
      w_tile += G;
      if (w_tile < ntiles) set_w(w_tile);
    }
  };

  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int slot, int wslot = 0) {
    const unsigned char* xs = smem + slot * (SPLIT ? XSTAGE : STAGE);
    const unsigned char* ws = SPLIT ? smem + NXS * XSTAGE + wslot * WSTAGE : xs + BM * ROW_BYTES;
    if constexpr (is_fp8<T>::value) {                  // one K = 128 MFMA per (channel tile, pixel tile): chunks q and q + 4
      uint4 af[TC][2], bf[TPW][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TC; ++i) af[i][kk] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < TPW; ++j) bf[j][kk] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TPW + j) * 16 + frow, kk * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) mma_fp8(af[i][0], af[i][1], bf[j][0], bf[j][1], acc[i][j]);
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[TC], bf[TPW];
#pragma unroll
      for (int i = 0; i < TC; ++i)
        af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TPW + j) * 16 + frow, kk * 4 + fq));
      if (!ABL(2)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j) acc[i][j][0] += __uint_as_float(af[i].x ^ bf[j].y);
      }
    }
  };

  // The same step with the stage's DMAs spread between its MFMAs instead of issued as a burst before them: a burst
  // of NI x 8 DMAs backs up the CU's vector-memory queue and every wave stalls in issue (~900 cycles per step,
  // profiles/r01_conv_stamps.md) while the matrix pipe idles; one DMA every few MFMAs keeps both busy.
  auto compute_issue = [&](int slot, const bool do_issue) {
    const uint32_t st = lds_base + (uint32_t)(d_slot * STAGE + wave * 1024);
    const int ks = d_ks;
    const uint32_t e = ktab[ks * 8 + cg];
    const uint32_t tp = e >> 28;
    const uint32_t k3_expt = (MIYOLO_HALO_EXPT && KS == 3 && (tp % 3) != 1) ? 0x80000000u : 0u;   // timing experiment, see below
    const uint32_t kofs = (KS == 3) ? (e & 0x0FFFFFFFu) : (e & 0x8FFFFFFFu);
    const bool seg1 = (KS == 1) && (ks * 8) >= ct0;
    const uint32_t dead = (MIYOLO_DMAP_ALWAYS_ISSUE && !do_issue) ? 0x80000000u : 0u;
    auto issue_one = [&](int d) {
      if (d < XI) {
        if constexpr (KS == 3) {
          lds_dma16(rs0, st + d * 8192, ((uint32_t)xoff0[d] + kofs) | (((xinv[d] >> tp) & 1u) << 31) | k3_expt | dead);
        } else {
          if (!seg1) lds_dma16(rs0, st + d * 8192, ((uint32_t)xoff0[d] + kofs) | xinv[d] | dead);
          else lds_dma16(rs1, st + d * 8192, ((uint32_t)xoff1[KS == 1 ? d : 0] + kofs) | xinv[d] | dead);
        }
      } else {
        lds_dma16(rsw, st + BM * ROW_BYTES + (d - XI) * 8192, (woff[d - XI] + (uint32_t)(ks * 128)) | dead);
      }
    };
    constexpr int NM = 2 * TC * TPW;                 // MFMAs of the step
    const unsigned char* xs = smem + slot * STAGE;
    const unsigned char* ws = xs + BM * ROW_BYTES;
    if constexpr (is_fp8<T>::value) {                  // K = 128 MFMAs (TC * TPW of them), the stage's DMAs spread between them
      uint4 af[TC][2], bf[TPW][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TC; ++i) af[i][kk] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < TPW; ++j) bf[j][kk] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TPW + j) * 16 + frow, kk * 4 + fq));
      }
      constexpr int NM8 = TC * TPW;
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          mma_fp8(af[i][0], af[i][1], bf[j][0], bf[j][1], acc[i][j]);
          const int q = i * TPW + j;
#pragma unroll
          for (int d = 0; d < NI; ++d)
            if (q == (d * NM8) / NI) { if (MIYOLO_DMAP_ALWAYS_ISSUE || do_issue) issue_one(d); }
        }
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[TC], bf[TPW];
#pragma unroll
      for (int i = 0; i < TC; ++i)
        af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TPW + j) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          Mma<T>::run(af[i], bf[j], acc[i][j]);
          const int q = (kk * TC + i) * TPW + j;     // constant after unrolling
#pragma unroll
          for (int d = 0; d < NI; ++d)
            if (q == (d * NM) / NI) { if (MIYOLO_DMAP_ALWAYS_ISSUE || do_issue) issue_one(d); }
        }
    }
    }
    if (MIYOLO_DMAP_ALWAYS_ISSUE && !do_issue) d_slot = (d_slot == NST - 1) ? 0 : d_slot + 1;
    if (do_issue) {
      d_slot = (d_slot == NST - 1) ? 0 : d_slot + 1;
      ++d_issued;
      if (++d_ks == a.nk) {
        d_ks = 0;
        d_tile += G;
        if (d_tile < ntiles) { setup_tile(d_tile); if (d_tile + G < ntiles) v_stores += warm_tile(d_tile + G); }
      }
    }
  };

  // SPLIT with the step's DMAs spread between its MFMAs (MIYOLO_DMAP_SPLIT == 2): the weight DMAs of W(c + 1) first, then the
  // activation DMAs of X(c + 2) - the issue order the wait at the top of a step relies on.
  auto compute_issue_split = [&](int slot, int wslot, const bool do_w, const bool do_x) {
    const uint32_t stx = lds_base + (uint32_t)(d_slot * XSTAGE + wave * 1024);
    const uint32_t stw = lds_base + (uint32_t)(NXS * XSTAGE + w_slot * WSTAGE + wave * 1024);
    const int ks = d_ks;
    const uint32_t e = ktab[ks * 8 + cg];
    const uint32_t tp = e >> 28;
    const uint32_t kofs = (KS == 3) ? (e & 0x0FFFFFFFu) : (e & 0x8FFFFFFFu);
    const bool seg1 = (KS == 1) && (ks * 8) >= ct0;
    const uint32_t wk = (uint32_t)(w_ks * 128);
    auto issue_one = [&](int d) {                    // d < NI - XI: weight DMA d; else activation DMA d - (NI - XI)
      if (d < NI - XI) {
        if (do_w) lds_dma16(rsw, stw + d * 8192, woff[d] + wk);
      } else if (do_x) {
        const int i = d - (NI - XI);
        if constexpr (KS == 3) lds_dma16(rs0, stx + i * 8192, ((uint32_t)xoff0[i] + kofs) | (((xinv[i] >> tp) & 1u) << 31));
        else if (!seg1) lds_dma16(rs0, stx + i * 8192, ((uint32_t)xoff0[i] + kofs) | xinv[i]);
        else lds_dma16(rs1, stx + i * 8192, ((uint32_t)xoff1[KS == 1 ? i : 0] + kofs) | xinv[i]);
      }
    };
    constexpr int NM = 2 * TC * TPW;
    const unsigned char* xs = smem + slot * XSTAGE;
    const unsigned char* ws = smem + NXS * XSTAGE + wslot * WSTAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[TC], bf[TPW];
#pragma unroll
      for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int j = 0; j < TPW; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TPW + j) * 16 + frow, kk * 4 + fq));
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          Mma<T>::run(af[i], bf[j], acc[i][j]);
          const int q = (kk * TC + i) * TPW + j;     // constant after unrolling
#pragma unroll
          for (int d = 0; d < NI; ++d)
            if (q == (d * NM) / NI) issue_one(d);
        }
    }
    if (do_w) {
      w_slot ^= 1;
      if (++w_ks == a.nk) { w_ks = 0; w_tile += G; if (w_tile < ntiles) set_w(w_tile); }
    }
    if (do_x) {
      d_slot = (d_slot == NXS - 1) ? 0 : d_slot + 1;
      ++d_issued;
      if (++d_ks == a.nk) { d_ks = 0; d_tile += G; if (d_tile < ntiles) setup_tile(d_tile); }
    }
  };

  // ---- deferred epilogue (MIYOLO_DMAP_DEFER, OFF: measured 8 % slower on the whole step, same box, back to back):
  // a finished tile's accumulators move to a second register set and its bias + SiLU + store run one 16x16 piece per K
  // step of the NEXT tile, right after that step's MFMAs have been issued.  The idea was that the matrix pipe works
  // through its queue while the VALU does the activations (the epilogue is 7 k of ~36 k cycles per tile with the matrix
  // pipe idle); in practice every step of every wave gets ~500 VALU cycles longer before its barrier and the step time
  // grows by about that much - the activations do not hide under this kernel's MFMAs.  Kept for the next round's
  // 4-wave kernel.  3-slot shapes only, not for residual layers (their loads would drain the DMAs in flight).
  constexpr bool DEFER_CT = MIYOLO_DMAP_DEFER && NST > 2;
  constexpr int NP = TC * TPW;                               // pieces per tile
  f32x4 eacc[DEFER_CT ? TC : 1][DEFER_CT ? TPW : 1];
  const bool defer = DEFER_CT && a.vec_ok && !a.res && !pair8;   // wave-uniform, constant for the launch
  const int ppk = (NP + a.nk - 1) / a.nk;                    // pieces per K step so that a tile's pieces finish within the next tile
  int e_next = NP, e_m0 = 0, e_n0 = 0;                       // next piece to run (NP = nothing pending)
  auto piece_ij = [&](auto ic, auto jc) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
    if constexpr (DEFER_CT) {
      const int nt = __builtin_amdgcn_readfirstlane(e_n0 + (wc * TC + i) * 16);
      const int n = nt + fq * 4;
      v4i_t s0, s1, s2, s3;
      const float* bp = sgpr_ptr(a.bias + (nt < a.cout ? nt : 0));
      asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                   "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(bp));
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bv[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s1[r] : fq == 2 ? s2[r] : s3[r]);
      const int m = __builtin_amdgcn_readfirstlane(e_m0) + (wp * TPW + j) * 16 + frow;
      const v4ie_t zr = {0, 0, 0, 0};
      if (a.out_f32) epilogue_fast<T, true>(a, rdst, m, n, eacc[i][j], bv, zr);
      else epilogue_fast<T, false>(a, rdst, m, n, eacc[i][j], bv, zr);
    }
  };
  auto run_piece = [&](int p) __attribute__((always_inline)) {   // p is wave-uniform: a jump over NP small blocks; must inline (captures live in registers)
    if constexpr (DEFER_CT) {
#define MIYOLO_PIECE(I, J) case (I) * TPW + (J): if constexpr ((I) < TC && (J) < TPW) piece_ij(std::integral_constant<int, (I)>{}, std::integral_constant<int, (J)>{}); break;
#define MIYOLO_PIECE_ROW(I) MIYOLO_PIECE(I, 0) MIYOLO_PIECE(I, 1) MIYOLO_PIECE(I, 2) MIYOLO_PIECE(I, 3)
      if constexpr (TPW == 4) {
        switch (p) { MIYOLO_PIECE_ROW(0) MIYOLO_PIECE_ROW(1) MIYOLO_PIECE_ROW(2) MIYOLO_PIECE_ROW(3) default: break; }
      } else {                                               // TPW == 2
        switch (p) { MIYOLO_PIECE(0, 0) MIYOLO_PIECE(0, 1) MIYOLO_PIECE(1, 0) MIYOLO_PIECE(1, 1) MIYOLO_PIECE(2, 0) MIYOLO_PIECE(2, 1)
                     MIYOLO_PIECE(3, 0) MIYOLO_PIECE(3, 1) default: break; }
      }
#undef MIYOLO_PIECE_ROW
#undef MIYOLO_PIECE
    }
  };

  // ---- stream: prologue issues two stages, then one barrier + one issue + one compute per step
  setup_tile(d_tile);
  if constexpr (SPLIT) {
    set_w(w_tile);
    issue_x();                                   // X(0)
    issue_w();                                   // W(0)
    if (total_steps > 1) issue_x();              // X(1): the order W(c), X(c + 1) of the loop
  } else {
    issue_next();
    if (NST > 2 && total_steps > 1) issue_next();
  }

  int c_tile = first, c_ks = 0, c_slot = 0, cw_slot = 0;
#if MIYOLO_ABLATE
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, acc_wait = 0, acc_issue = 0, acc_comp = 0, acc_epi = 0, t_begin = 0;
  STAMP(t_begin);
#endif
  for (int c = 0; c < total_steps; ++c) {
    STAMP(t0);
    if (SPLIT && c + 1 < total_steps) {        // X(c + 1) - XI DMAs per wave - and the stores behind it may stay in flight
#define MIYOLO_WAIT(K) case (K): asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(XI + (K)) : "memory"); break;
      switch (MIYOLO_DMAP_EXACT_VMCNT ? v_stores : 0) {
        MIYOLO_WAIT(0) MIYOLO_WAIT(1) MIYOLO_WAIT(2) MIYOLO_WAIT(3) MIYOLO_WAIT(4) MIYOLO_WAIT(5) MIYOLO_WAIT(6) MIYOLO_WAIT(7) MIYOLO_WAIT(8)
        MIYOLO_WAIT(9) MIYOLO_WAIT(10) MIYOLO_WAIT(11) MIYOLO_WAIT(12) MIYOLO_WAIT(13) MIYOLO_WAIT(14) MIYOLO_WAIT(15) MIYOLO_WAIT(16)
        MIYOLO_WAIT(17) MIYOLO_WAIT(18) MIYOLO_WAIT(19) MIYOLO_WAIT(20) MIYOLO_WAIT(21) MIYOLO_WAIT(22) MIYOLO_WAIT(23) MIYOLO_WAIT(24)
        default: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(XI) : "memory"); break;   // uncounted: conservative
      }
#undef MIYOLO_WAIT
    } else if (!SPLIT && NST > 2 && c + 1 < total_steps) {      // one younger stage may stay in flight
      // vmcnt counts loads, stores and LDS-DMAs together, in issue order: what may stay outstanding is the youngest
      // stage's NI DMAs PLUS the epilogue stores issued after them (v_stores).  Waiting for fewer would wait for those
      // DMAs themselves - a full DMA latency per step (measured: 2.3x slower with one store per step unaccounted).
#define MIYOLO_WAIT(K) case (K): asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NI + (K)) : "memory"); break;
      switch (MIYOLO_DMAP_EXACT_VMCNT ? v_stores : 0) {
        MIYOLO_WAIT(0) MIYOLO_WAIT(1) MIYOLO_WAIT(2) MIYOLO_WAIT(3) MIYOLO_WAIT(4) MIYOLO_WAIT(5) MIYOLO_WAIT(6) MIYOLO_WAIT(7) MIYOLO_WAIT(8)
        MIYOLO_WAIT(9) MIYOLO_WAIT(10) MIYOLO_WAIT(11) MIYOLO_WAIT(12) MIYOLO_WAIT(13) MIYOLO_WAIT(14) MIYOLO_WAIT(15) MIYOLO_WAIT(16)
        MIYOLO_WAIT(17) MIYOLO_WAIT(18) MIYOLO_WAIT(19) MIYOLO_WAIT(20)
        default: asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NI) : "memory"); break;   // uncounted: conservative
      }
#undef MIYOLO_WAIT
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    v_stores = 0;
    STAMP(t1);
    // (Tried and rejected, profiles/r01_conv_stamps.md: letting waves 4-7 run MFMAs first and DMAs
    // second, so that the vector-memory path and the matrix pipe overlap across SIMD partners,
    // made every layer 20-25 % slower - the LDS-DMA writes and the partner's ds_reads/MFMA issue
    // interfere.)
    if constexpr (MIYOLO_DMAP_INTERLEAVE && (NST > 2 || MIYOLO_DMAP_INTERLEAVE > 1)) {      // measured: -3...-9 % on the 3-slot 3x3 shapes, +4 % on the 2-slot 256x192 tile
      STAMP(t2);
      compute_issue(c_slot, d_issued < total_steps);
    } else if constexpr (SPLIT && MIYOLO_DMAP_SPLIT == 2 && !is_fp8<T>::value) {
      STAMP(t2);
      compute_issue_split(c_slot, cw_slot, c + 1 < total_steps, d_issued < total_steps);
      cw_slot ^= 1;
    } else if constexpr (SPLIT) {
      if (c + 1 < total_steps) issue_w();          // W(c + 1) first, then X(c + 2): see the wait above
      if (d_issued < total_steps) issue_x();
      STAMP(t2);
      compute(c_slot, cw_slot);
      cw_slot ^= 1;
    } else {
      if (d_issued < total_steps) issue_next();
      STAMP(t2);
      compute(c_slot);
    }
    if constexpr (DEFER_CT) {                                  // pieces of the previous tile, in the shadow of this step's MFMAs
      for (int r = 0; r < ppk && e_next < NP; ++r, ++e_next) { run_piece(e_next); ++v_stores; }
    }
    STAMP(t3);
#if MIYOLO_ABLATE
    acc_wait += t1 - t0; acc_issue += t2 - t1; acc_comp += t3 - t2;
#endif
    c_slot = (c_slot == (SPLIT ? NXS : NST) - 1) ? 0 : c_slot + 1;
    if (++c_ks == a.nk) {
      // ---- epilogue of tile c_tile (the next tile's first stages are already in flight)
      const int mb = c_tile / NB, nb = c_tile - mb * NB;
      const int m0 = mb * BM, n0 = nb * BN;
      // bias through the SCALAR unit (wave-uniform address -> s_load, lgkmcnt): a vector load
      // here would make the compiler wait vmcnt(0) and drain the next tile's DMAs in flight.
      // The bias array is padded to a multiple of 128 floats by the host (weights.py).
      const float* __restrict__ bias = a.bias;
      auto run_epilogue_tiles = [&](auto outf32_tag, auto first_tag) {
        constexpr bool OUTF32 = decltype(outf32_tag)::value;
#pragma unroll
        for (int i = decltype(first_tag)::value; i < TC; ++i) {
          const int nt = __builtin_amdgcn_readfirstlane(n0 + (wc * TC + i) * 16);
          const int n = nt + fq * 4;
          // 16 consecutive biases of this 16-channel tile, in SGPRs.  A channel tile that lies wholly beyond cout (the last
          // workgroup tile of e.g. 256 channels cut in 96s) loads tile 0 instead: scalar loads are not range-checked, and
          // past the 128-float padding of the array they ran off its allocation - an intermittent memory access fault
          // once a 256-channel layer existed (the fused Detect convs) and the array sat at the end of a mapped range
          v4i_t s0, s1, s2, s3;
          const float* bp = sgpr_ptr(bias + (nt < a.cout ? nt : 0));
          asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                       "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(bp));
          float bv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bv[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s1[r] : fq == 2 ? s2[r] : s3[r]);
          float sv[4] = {1.f, 1.f, 1.f, 1.f};
          if constexpr (is_fp8<T>::value) {        // per-output-channel dequantisation scale, same scalar-load route
            const float* qp = sgpr_ptr(a.qscale + (nt < a.cout ? nt : 0));
            asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                         "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(qp));
#pragma unroll
            for (int r = 0; r < 4; ++r)
              sv[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s1[r] : fq == 2 ? s2[r] : s3[r]);
          }
          v4ie_t rv[TPW];
          if (a.res) {                       // wave-uniform: all residual loads of this channel tile in flight together
#pragma unroll
            for (int j = 0; j < TPW; ++j) rv[j] = epilogue_res_load<T>(a, rres, m0 + (wp * TPW + j) * 16 + frow, n);
          } else {
#pragma unroll
            for (int j = 0; j < TPW; ++j) rv[j] = (v4ie_t){0, 0, 0, 0};
          }
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int m = m0 + (wp * TPW + j) * 16 + frow;
            if (!ABL(8)) epilogue_fast<T, OUTF32>(a, rdst, m, n, acc[i][j], bv, rv[j], sv);
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      };
      auto run_epilogue = [&](auto outf32_tag) { run_epilogue_tiles(outf32_tag, std::integral_constant<int, 0>{}); };
      if (defer) {
        if constexpr (DEFER_CT) {
#pragma unroll
          for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TPW; ++j) { eacc[i][j] = acc[i][j]; acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
          e_m0 = m0; e_n0 = n0; e_next = 0;
        }
      } else if (pair8) {
        if constexpr (NPAIRW > 0) {
#pragma unroll
          for (int p = 0; p < NPAIRW; ++p) {
            const int nt = __builtin_amdgcn_readfirstlane(n0 + (wc * TC + 2 * p) * 16);
            const int n = nt + fq * 8;
            // 32 consecutive biases in SGPRs (cout % 16 == 0 and the array is padded to a multiple of 128 floats: nt + 32 never
            // passes the padding; a pair wholly beyond cout loads pair 0 - see the note at the 16-channel form below)
            v4i_t s0, s1, s2, s3, s4, s5, s6, s7;
            const float* bp = sgpr_ptr(bias + (nt < a.cout ? nt : 0));
            asm volatile("s_load_dwordx4 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x10\n\ts_load_dwordx4 %2, %8, 0x20\n\t"
                         "s_load_dwordx4 %3, %8, 0x30\n\ts_load_dwordx4 %4, %8, 0x40\n\ts_load_dwordx4 %5, %8, 0x50\n\t"
                         "s_load_dwordx4 %6, %8, 0x60\n\ts_load_dwordx4 %7, %8, 0x70\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5), "=&s"(s6), "=&s"(s7) : "s"(bp));
            float blo[4], bhi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              blo[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s2[r] : fq == 2 ? s4[r] : s6[r]);
              bhi[r] = __int_as_float(fq == 0 ? s1[r] : fq == 1 ? s3[r] : fq == 2 ? s5[r] : s7[r]);
            }
            float slo[4] = {1.f, 1.f, 1.f, 1.f}, shi[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (is_fp8<T>::value) {        // per-output-channel dequantisation scales, same scalar-load route
              const float* qp = sgpr_ptr(a.qscale + (nt < a.cout ? nt : 0));
              asm volatile("s_load_dwordx4 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x10\n\ts_load_dwordx4 %2, %8, 0x20\n\t"
                           "s_load_dwordx4 %3, %8, 0x30\n\ts_load_dwordx4 %4, %8, 0x40\n\ts_load_dwordx4 %5, %8, 0x50\n\t"
                           "s_load_dwordx4 %6, %8, 0x60\n\ts_load_dwordx4 %7, %8, 0x70\n\ts_waitcnt lgkmcnt(0)"
                           : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5), "=&s"(s6), "=&s"(s7) : "s"(qp));
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                slo[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s2[r] : fq == 2 ? s4[r] : s6[r]);
                shi[r] = __int_as_float(fq == 0 ? s1[r] : fq == 1 ? s3[r] : fq == 2 ? s5[r] : s7[r]);
              }
            }
            constexpr int ESP = (int)sizeof(T);
            v4ie_t rv[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) rv[j] = (v4ie_t){0, 0, 0, 0};
            if (a.res) {                   // wave-uniform: the residual loads of this channel pair in flight together
#pragma unroll
              for (int j = 0; j < TPW; ++j) {
                const int m = m0 + (wp * TPW + j) * 16 + frow;
                const uint32_t ro = (m < a.M && n < a.cout) ? (uint32_t)((m * a.res_ld + a.res_choff + n) * ESP) : kOob;
                if constexpr (ESP == 2) rv[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
                else { const v2i_t r2 = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0); rv[j] = (v4ie_t){r2[0], r2[1], 0, 0}; }
              }
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
              const int m = m0 + (wp * TPW + j) * 16 + frow;
              float v[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float x0 = is_fp8<T>::value ? fmaf(acc[2 * p][j][r], slo[r], blo[r]) : acc[2 * p][j][r] + blo[r];
                float x1 = is_fp8<T>::value ? fmaf(acc[2 * p + 1][j][r], shi[r], bhi[r]) : acc[2 * p + 1][j][r] + bhi[r];
                if (a.act) { x0 = silu_fast(x0); x1 = silu_fast(x1); }
                v[r] = x0; v[4 + r] = x1;
              }
              const uint32_t so = (m < a.M && n < a.cout) ? (uint32_t)((m * a.dst_ld + a.dst_choff + n) * ESP) : kOob;
              if constexpr (ESP == 2) {
                if (a.res) {
                  const f16x8 hr = *reinterpret_cast<const f16x8*>(&rv[j]);
#pragma unroll
                  for (int r = 0; r < 8; ++r) v[r] += (float)hr[r];
                }
                const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
                if (!ABL(8)) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4ie_t*>(&hv), rdst, so, 0, MIYOLO_ST_AUX);
              } else {
                if (a.res) {
                  float ra[4], rb[4];
                  unpack_fp8x4((uint32_t)rv[j][0], ra); unpack_fp8x4((uint32_t)rv[j][1], rb);
#pragma unroll
                  for (int r = 0; r < 4; ++r) { v[r] = fmaf(ra[r], a.res_scale, v[r]); v[4 + r] = fmaf(rb[r], a.res_scale, v[4 + r]); }
                }
                const float q = a.out_inv_scale;
                const v2i_t o = {(int)pack_fp8x4(v[0] * q, v[1] * q, v[2] * q, v[3] * q), (int)pack_fp8x4(v[4] * q, v[5] * q, v[6] * q, v[7] * q)};
                if (!ABL(8)) __builtin_amdgcn_raw_buffer_store_b64(o, rdst, so, 0, MIYOLO_ST_AUX);
              }
              acc[2 * p][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[2 * p + 1][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
          if constexpr (TC & 1) run_epilogue_tiles(std::false_type{}, std::integral_constant<int, TC - 1>{});
          v_stores += (ABL(8) ? 0 : (NPAIRW + (TC & 1)) * TPW);
        }
      } else if (a.vec_ok) {
        if (a.out_f32) run_epilogue(std::true_type{}); else run_epilogue(std::false_type{});
        v_stores += (ABL(8) ? 0 : NP);          // one vector store per 16x16 piece (its residual loads have been consumed)
      } else if constexpr (!is_fp8<T>::value) {      // odd channel counts (e.g. nc = 13): scalar path (fp8 requires vec_ok)
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int n = n0 + (wc * TC + i) * 16 + fq * 4;
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int m = m0 + (wp * TPW + j) * 16 + frow;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = acc[i][j][r] + ((n + r < a.cout) ? bias[n + r] : 0.f);
              if (a.act) x = a.exact ? silu_exact(x) : silu_fast(x);
              v[r] = x;
              acc[i][j][r] = 0.f;
            }
            if (n < a.cout && m < a.M) epilogue_store<T>(a, m, n, v);
          }
        }
      }
      c_ks = 0;
      c_tile += G;
      STAMP(t4);
#if MIYOLO_ABLATE
      acc_epi += t4 - t3;
#endif
    }
  }
  if constexpr (DEFER_CT) {
    for (; e_next < NP; ++e_next) run_piece(e_next);           // the last tile's pieces
  }
  if (MIYOLO_DMAP_ALWAYS_ISSUE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dead DMAs of the last steps
  if (MIYOLO_DMAP_L2WARM && warm_acc == 1.2345e-30f && a.dbg) a.dbg[0] = 1;   // never true: keeps the warm-up loads alive
#if MIYOLO_ABLATE
  if (a.dbg && lane == 0) {          // per wave: total, wait, issue, compute, epilogue cycles + steps
    STAMP(t4);
    unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = t4 - t_begin; d[1] = acc_wait; d[2] = acc_issue; d[3] = acc_comp; d[4] = acc_epi; d[5] = (unsigned long long)total_steps;
    d[6] = (unsigned long long)my_tiles; d[7] = 1;
  }
#endif
}

// host: magic constants for floor(x/d), x < 2^31 (Granlund-Montgomery, 31-bit dividend):
// l = ceil(log2 d), mul = floor(2^(31+l)/d) + 1, q = umulhi(x, mul) >> (l-1); d == 1 -> mul = 0.
inline void host_magic(uint32_t d, uint32_t* mul, uint32_t* shift) {
  if (d <= 1) { *mul = 0; *shift = 0; return; }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  const unsigned long long m = (1ull << (31 + l)) / d + 1;
  *mul = (uint32_t)m;
  *shift = l - 1;
}

template <typename T, int KS, int WC, int TC>
inline hipError_t launch_dmap_cfg(const ConvArgs& a, hipStream_t s, int ncu) {
  constexpr int BN = WC * TC * 16;
  const size_t lds = dmap_lds_bytes<WC, TC>() + (size_t)a.nk * 32;    // ring + K table
  if (lds > 160 * 1024) return hipErrorInvalidValue;                  // K > ~32k: not a YOLO layer
  const long mbk = ((long)a.M + DMA_BM - 1) / DMA_BM, nb = (a.cout + BN - 1) / BN;
  long grid = std::min<long>(mbk * nb, ncu);
#if MIYOLO_DMAP_BALANCED_GRID
  {                                      // same number of rounds, but every workgroup gets the same number of tiles: the CUs
                                         // that would idle through the last round stay idle throughout and the others see less
                                         // contention on the L2 -> LDS path (measured +3.3 % on the whole step, same box)
    const long rounds = (mbk * nb + ncu - 1) / ncu;
    grid = std::min<long>((mbk * nb + rounds - 1) / rounds, ncu);
  }
#endif
  grid = (grid + 7) / 8 * 8;             // the tile dealing assumes a multiple of 8
  hipLaunchKernelGGL((conv_dmap_kernel<T, KS, WC, TC>), dim3((unsigned)grid), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename T, int KS>
inline hipError_t launch_dmap_ks(const ConvArgs& a, ConvCfg c, hipStream_t s, int ncu) {
  if (c.wc == 2 && c.tc == 6) return launch_dmap_cfg<T, KS, 2, 6>(a, s, ncu);
  if (c.wc == 2 && c.tc == 4) return launch_dmap_cfg<T, KS, 2, 4>(a, s, ncu);
  if (c.wc == 2 && c.tc == 3) return launch_dmap_cfg<T, KS, 2, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 4) return launch_dmap_cfg<T, KS, 1, 4>(a, s, ncu);
  if (c.wc == 1 && c.tc == 3) return launch_dmap_cfg<T, KS, 1, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 2) return launch_dmap_cfg<T, KS, 1, 2>(a, s, ncu);
  return launch_dmap_cfg<T, KS, 1, 1>(a, s, ncu);
}

// Tile choice for the persistent kernel.  Cost model fitted to same-box per-layer timings of every forced shape
// (profiles/r01_tile_choice.md): a launch takes  rounds x (BN + c0)  with  rounds = ceil(tiles / CUs), c0 = 100 (15: -0.6 %, 40: -0.4 %, 80-200: equal, same-box)  - what decides
// between shapes is mostly how the tile count divides by the CU count (576 channels at 20x20: 600 tiles of 96 channels
// are three rounds, 500 tiles of 128 are two: -13...-19 %), then the channel padding, with a small penalty for the
// 32-pixel wave tiles of the 8x1 layouts.  The 256 x 192 tile (8 waves of 64 x 96: 30 % fewer LDS fragment bytes and
// 37 % fewer filled bytes per FLOP) competes under the same rule; it needs at least one tile per CU.
inline double dmap_cost(int cout, long M, int ncu, ConvCfg c) {
  const int bn = c.wc * c.tc * 16;
  const long mbk = (M + DMA_BM - 1) / DMA_BM, nb = (cout + bn - 1) / bn, tiles = mbk * nb;
  const long rounds = (tiles + ncu - 1) / ncu;
#ifndef MIYOLO_DMAP_C0
#define MIYOLO_DMAP_C0 100.0
#endif
  double cost = (double)rounds * (bn + MIYOLO_DMAP_C0);
  if (c.wc == 1) cost *= 1.05;
  return cost;
}
inline ConvCfg pick_dmap_cfg(int cout, long M, int ncu, int ksize = 1) {
  static const ConvCfg cands[] = {{2, 6}, {2, 4}, {2, 3}, {1, 4}, {1, 3}, {1, 2}, {1, 1}};
  const long mbk = (M + DMA_BM - 1) / DMA_BM;
  ConvCfg best = {1, 1};
  double best_cost = 1e30;
  for (const ConvCfg& c : cands) {
    // the 256 x 192 tile wanted at least one tile per CU on its 2-slot ring; with the split rings a 3x3 layer whose tiles fill
    // most of ONE round beats two rounds of smaller tiles (model.19, 384 -> 384 stride 2 at 20 x 20: 200 tiles, 103 -> 85 us);
    // the 1x1 layers of that size keep the rule (measured slower)
    if (c.tc == 6 && (cout % 192 || (mbk * ((cout + 191) / 192) < ncu && !(ksize == 3 && mbk * ((cout + 191) / 192) * 4 >= ncu * 3)))) continue;
    const double cost = dmap_cost(cout, M, ncu, c);
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <typename T>
inline hipError_t launch_conv_dmap(const ConvArgs& a, hipStream_t s, int ncu, int force_wc = 0, int force_tc = 0, int pair8 = 1) {
  ConvCfg c = pick_dmap_cfg(a.cout, a.M, ncu, a.ksize);
  if (force_wc > 0 && force_tc > 0) c = {force_wc, force_tc};
  ConvArgs b = a;
  b.pair8 = (pair8 && sizeof(T) <= 2 && !a.out_f32 && a.vec_ok && a.cout % 16 == 0 && a.dst_ld % 8 == 0 && a.dst_choff % 8 == 0 &&
             (!a.res || (a.res_ld % 8 == 0 && a.res_choff % 8 == 0))) ? 1 : 0;
  if (a.ksize == 3) return launch_dmap_ks<T, 3>(b, c, s, ncu);
  return launch_dmap_ks<T, 1>(b, c, s, ncu);
}

}  // namespace miyolo

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
  int32_t slab_bytes, scratch_off;   // (TH+2)*HP*128; LDS offset of the 256-byte landing area of the warm-up DMAs
  uint32_t mg_tw_mul, mg_tw_shift;   // magic division by TW
  uint32_t mg_nb_mul, mg_nb_shift;   // by NB
  uint32_t mg_tx_mul, mg_tx_shift;   // by tiles_x
  uint32_t mg_ty_mul, mg_ty_shift;   // by tiles_y
  uint32_t bias_bytes;               // padded bias array size
  int32_t warm;                      // unused (L2 warm-up experiments: no gain, removed)
};

constexpr int kH2LdsMax = 80 * 1024;     // two workgroups per CU

// Tile geometries (compile-time: scalar setup code is slow on this machine - with run-time geometry the slab issue loop
// cost ~300 cycles per DMA and the per-workgroup prologue >10 k cycles, gpurun stamps): TW x TH pixels, slab pitch HP.
template <int GEO> struct H2Geo;
template <> struct H2Geo<0> { static constexpr int TW = 16, TH = 16, HP = 24; };   // 160-, 80-wide maps (and anything else)
template <> struct H2Geo<1> { static constexpr int TW = 40, TH = 6, HP = 48; };    // 40-wide maps
template <> struct H2Geo<2> { static constexpr int TW = 20, TH = 12, HP = 24; };   // 20-wide maps
inline void h2_geo(int geo, int* tw, int* th, int* hp) {
  *tw = geo == 1 ? 40 : geo == 2 ? 20 : 16; *th = geo == 1 ? 6 : geo == 2 ? 12 : 16; *hp = geo == 1 ? 48 : 24;
}

// Swizzle of the slab (round 3): the 16-byte slot s of halo pixel (hy, hx) holds channel chunk s ^ h2_swz(hx) ^ 4 * (hy & 1 if
// ROWFLIP).  A fragment read takes 16 pixels in the lane groups of ds_read_b128 ({0-3, 12-15} of one K chunk with {4-11} of its
// neighbour, i.e. chunk c and c ^ 1); its pixels are 16 consecutive halo columns starting at the tap's dx = 0, 1 or 2 - or,
// where the tile width is not a multiple of 16, two runs in consecutive halo rows.  The ring kernels' row swizzle
// (hx >> 1) & 7 is conflict-free only for even starts: the dx = 1 and 2 taps paid 2-way conflicts on every pixel-fragment read
// (lane-group model: 21 % of the LDS cycles at 16 x 16, the PMC counter said 21.1 %; 22-27 % on the wrapping geometries).
// ((hx >> 1) & 3) << 1 keeps bit 0 of the chunk position for the c / c ^ 1 pairing and is conflict-free for every start on
// the 16- and 40-wide tiles; the 20-wide tile additionally flips bit 2 with the halo row's parity (two runs per fragment).
// tests/test_conv_emulation.py::test_h2_slab_swizzle_is_conflict_free_for_every_tap checks all fragments x taps x geometries.
__device__ __forceinline__ int h2_swz(int hx) { return ((hx >> 1) & 3) << 1; }
template <int GEO> struct H2RowFlip { static constexpr bool value = (GEO == 2); };

template <typename T, int TC, int GEO, bool PERSIST>
__global__ __launch_bounds__(256, 2) void conv_h2_kernel(const ConvArgs a, const H2Geom g) {
  constexpr int TW = H2Geo<GEO>::TW, TH = H2Geo<GEO>::TH, HP = H2Geo<GEO>::HP;
  constexpr int GX = (TW + 2 + 7) / 8, NPX = TW * TH;
  constexpr int SLAB = (TH + 2) * HP * ROW_BYTES;
  constexpr int RPW = (TH + 2 + 3) / 4;          // halo rows per wave
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
  const int H = a.Hin, W = a.Win;

  // ---- tiles of this workgroup.  Tiles are numbered channel tile fastest (the channel tiles sharing a halo are
  // neighbours) and dealt to XCDs in contiguous chunks (workgroups b, b + 8, ... share an XCD and its L2).
  //   !PERSIST: one tile per workgroup, grid = tiles (the hardware's dispatch balances).
  //   PERSIST : grid = two workgroups per CU; workgroup (xcd, slot) walks its XCD's chunk with stride grid / 8, and the NEXT
  //             tile's first slab is issued piecewise inside the epilogue (a workgroup's first slab costs 3-9 k cycles of
  //             blocked DMA issue + 2.5 k of waiting at the head of a 55-75 k-cycle tile).  Measured (option "h2_warm" = 1):
  //             the prologue shrinks from ~13 k to ~6 k cycles per tile, but the same DMAs cost ~270 cycles each inside the
  //             epilogue - a slab that comes from beyond L2 is admitted at the CU's ~11 B/clk miss rate wherever its
  //             DMAs are placed, and the issuing wave waits - and the static tile lists balance worse than the hardware's
  //             dispatch: 1.4 % slower on the whole step.  Kept selectable and tested, off by default.
  // (Round 3 tried a STAGGER here: the second workgroup a CU receives in the first round - wave slot 1 of its SIMD, read
  // from HW_REG_HW_ID - started 25 ... 150 % of a tile's tap-loop time late, on the idea that two workgroups of equal
  // cost dispatched together stay in step and idle the matrix pipe through both's prologues and epilogues at once.
  // Measured on the whole step: 8 724 / 8 707 / 8 644 / 8 596 / 8 487 / 8 357 frames/s at 0 / 25 / 50 / 75 / 100 / 150 % -
  // the delay is simply added, nothing is won back; removed.  gpurun_out/r3_sg_*.log)
  const int nblk = PERSIST ? g.ntiles : (int)gridDim.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
  const int xstart = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xlen = xq + (xcd < xr ? 1 : 0);
  const int xstride = PERSIST ? (int)(gridDim.x >> 3) : 1;
  if (slot_ >= xlen) return;
  // (Round 3 tried walking every XCD's chunk from its END in alternate / all launches, so that a layer starts on the tiles its
  // producer wrote last, still in that XCD's L2: 8 742 / 8 729 / 8 675 frames/s for off / all / alternate - no effect.  What
  // serves a layer's input is the 256 MB Infinity Cache, which does not care about order: with the producers' stores marked
  // sc1 or nt the step loses 6 % / 18 %, profiles/r03_store_policy_ab.log.)
  int L = xstart + slot_;
  int tleft = PERSIST ? (xlen - slot_ - 1) / xstride : 0;      // tiles after the first

  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int ldB = a.src[0].ld * ES;
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(is_fp8<T>::value ? a.bias_init : a.bias), 0, g.bias_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rqs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(is_fp8<T>::value ? a.qscale : a.bias), 0, g.bias_bytes, 0x00020000);

  struct Tile { int bimg, y0, x0, n0; };
  auto tile_coords = [&](int L) -> Tile {
    const uint32_t t1 = magic_div((uint32_t)L, g.mg_nb_mul, g.mg_nb_shift);
    const int nb = L - (int)t1 * g.NB;
    const uint32_t t2 = magic_div(t1, g.mg_tx_mul, g.mg_tx_shift);
    const int tx = (int)(t1 - t2 * (uint32_t)g.tiles_x);
    const uint32_t bi = magic_div(t2, g.mg_ty_mul, g.mg_ty_shift);
    const int ty = (int)(t2 - bi * (uint32_t)g.tiles_y);
    return Tile{(int)bi, ty * TH, tx * TW, nb * BN};
  };

  // ---- weight DMA rows of this lane: LDS row r of a slot holds channel n0 + pi(r), pi = the MFMA-row deal that gives a
  // lane 8 consecutive channels over a pair of channel tiles (tile 2p row 4q+j -> channel 32p + 8q + j, tile 2p+1 -> +4)
  uint32_t woff[NWI];
  // chunk column of this lane's weight DMAs: (r>>1)&7 with r = 8*(wave+4i) + (lane>>3) does not depend on i
  const int cgw = (lane & 7) ^ ((((8 * wave + (lane >> 3)) >> 1)) & 7);
  auto set_w = [&](int n0) {
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int r = 8 * (wave + 4 * i) + (lane >> 3);
      const int ti = r >> 4, rho = r & 15;
      const int ch = (ti < 2 * NPAIR) ? 32 * (ti >> 1) + 8 * (rho >> 2) + 4 * (ti & 1) + (rho & 3) : r;
      const int n = n0 + ch;
      woff[i] = (r < BN && n < a.cout) ? (uint32_t)(n * a.kpad * ES + cgw * 16) : kOob;
    }
  };
  auto issue_w = [&](int c, int tap, int slot) {
    const uint32_t st = lds_base + (uint32_t)(SLAB + slot * WSLOT + wave * 1024);
    const uint32_t kofs = (uint32_t)((tap * a.cin + c * CPR) * ES);
    const uint32_t inv = ((c * CPR + cgw * CE) < a.cin) ? 0u : kOob;
#pragma unroll
    for (int i = 0; i < NWI; ++i) lds_dma16(rsw, st + i * 4096, (woff[i] + kofs) | inv);
  };

  // ---- halo slab DMA: wave w fills halo rows w, w+4, ...; instruction (hy, gx) = LDS rows hy*HP + 8*gx .. +7, halo
  // pixel hx = 8*gx + (lane>>3); the 16-byte slot s of a row holds channel chunk s ^ f(hx), f(h) = (h>>1)&7 - a swizzle
  // by the COLUMN of the halo pixel, so a tap's row offset (dy*HP + dx) moves the address by a constant per dx
  const int hxl = lane >> 3;
  int32_t lc[GX];                                            // lane part of the source offset per DMA group (its chunk column included)
  uint32_t cvm[GX];                                          // chunk column of this lane in group gx (for the channel-tail test)
#pragma unroll
  for (int gx = 0; gx < GX; ++gx) {
    const int cg = (lane & 7) ^ h2_swz(gx * 8 + hxl) ^ (H2RowFlip<GEO>::value ? 4 * (wave & 1) : 0);   // halo row hy = wave + 4 i: parity of wave
    lc[gx] = hxl * ldB + a.src[0].ch_off * ES + cg * 16;
    cvm[gx] = (uint32_t)(cg * CE);
  }
  // per-lane, tile-dependent part (computed once per workgroup): bit gx of xmask = halo column hx = 8*gx + hxl is
  // inside the image and inside the halo
  uint32_t xmask = 0;
  auto set_x = [&](int x0) {
#pragma unroll
    for (int gx = 0; gx < GX; ++gx) {
      const int hx = gx * 8 + hxl, x = x0 - 1 + hx;
      xmask |= (x >= 0 && x < W && hx < TW + 2) ? (1u << gx) : 0u;
    }
  };
  auto chunk_mask = [&](int c) -> uint32_t {                 // bit gx: this lane's chunk of group gx lies inside cin
    uint32_t m = 0;
#pragma unroll
    for (int gx = 0; gx < GX; ++gx) m |= ((uint32_t)(c * CPR) + cvm[gx] < (uint32_t)a.cin) ? (1u << gx) : 0u;
    return m;
  };
  auto issue_slab = [&](const Tile& t, int c) {
    const uint32_t msk = xmask & chunk_mask(c);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int hy = wave + 4 * i;
      if (hy < TH + 2) {                                                        // wave-uniform; false only in the last round
        const int y = t.y0 - 1 + hy;
        const bool yok = (y >= 0) && (y < H);
        const int32_t sb = ((t.bimg * H + y) * W + t.x0 - 1) * ldB + c * ROW_BYTES;
        const uint32_t st = lds_base + (uint32_t)(hy * HP * ROW_BYTES);
        const uint32_t mrow = yok ? msk : 0u;
#pragma unroll
        for (int gx = 0; gx < GX; ++gx) {
          const uint32_t off = (uint32_t)(lc[gx] + sb + gx * 8 * ldB);
          lds_dma16(rs0, st + gx * 1024, ((mrow >> gx) & 1u) ? off : kOob);
        }
      }
    }
  };

  // one DMA of a slab (piece q = i * GX + gx, compile-time): the next tile's slab goes out between the epilogue's blocks
  auto issue_slab_piece = [&](const Tile& t, uint32_t msk, auto q_tag) __attribute__((always_inline)) {
    constexpr int q = decltype(q_tag)::value, i = q / GX, gx = q % GX;
    const int hy = wave + 4 * i;
    if (hy < TH + 2) {
      const int y = t.y0 - 1 + hy;
      const int32_t sb = ((t.bimg * H + y) * W + t.x0 - 1) * ldB;
      const uint32_t mrow = (y >= 0 && y < H) ? msk : 0u;
      const uint32_t off = (uint32_t)(lc[gx] + sb + gx * 8 * ldB);
      lds_dma16(rs0, lds_base + (uint32_t)(hy * HP * ROW_BYTES) + gx * 1024, ((mrow >> gx) & 1u) ? off : kOob);
    }
  };

  // ---- per-lane fragment addresses (tile independent)
  uint32_t baddr[TPW][3];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int p = (wave * TPW + j) * 16 + frow;
    const bool pv = p < NPX;
    const uint32_t pp = pv ? (uint32_t)p : 0u;
    const int py = (int)(pp / (uint32_t)TW);
    const int px = (int)pp - py * TW;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = px + dx;
      baddr[j][dx] = (uint32_t)((py * HP + hx) * ROW_BYTES + ((fq ^ h2_swz(hx) ^ (H2RowFlip<GEO>::value ? 4 * (py & 1) : 0)) << 4));
    }
  }
  const uint32_t aaddr = lds_off(frow, fq);

  f32x4 acc[TC][TPW];
  auto init_acc = [&](int n0) {          // accumulators start at the bias
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      const int ch = (i < 2 * NPAIR) ? 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : 16 * i + 4 * fq;
      const v4ie_t bv = __builtin_amdgcn_raw_buffer_load_b128(rbias, (uint32_t)((n0 + ch) * 4), 0, 0);
      const f32x4 bf = {__int_as_float(bv[0]), __int_as_float(bv[1]), __int_as_float(bv[2]), __int_as_float(bv[3])};
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[i][j] = bf;
    }
  };

  auto compute = [&](int dy, auto dx_tag, int slot, bool full) __attribute__((always_inline)) {
    constexpr int dx = decltype(dx_tag)::value;
    const unsigned char* ws = smem + SLAB + slot * WSLOT;
    // the tap's row offset into the slab; ROWFLIP: an odd dy moves every pixel to a halo row of the other parity (chunk bit 2)
    const unsigned char* xs = smem + dy * (HP * ROW_BYTES);
    const uint32_t rf = H2RowFlip<GEO>::value ? (uint32_t)((dy & 1) << 6) : 0u;
    if constexpr (is_fp8<T>::value) {                  // one K = 128 MFMA per tile pair: chunks q and q + 4 of the row
      uint4 bf[TPW][2];                                // the pixel fragments stay, the weight fragments stream (register budget)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < TPW; ++j) bf[j][kk] = *reinterpret_cast<const uint4*>(xs + (baddr[j][dx] ^ (uint32_t)(kk << 6) ^ rf));
      uint4 a0 = *reinterpret_cast<const uint4*>(ws + aaddr), a1 = *reinterpret_cast<const uint4*>(ws + (aaddr ^ 64u));
#pragma unroll
      for (int i = 0; i < TC; ++i) {                   // one channel tile ahead, pinned: the scheduler would otherwise hoist all
        uint4 n0 = a0, n1 = a1;                        // twelve reads and spill
        if (i + 1 < TC) {
          n0 = *reinterpret_cast<const uint4*>(ws + aaddr + (i + 1) * 16 * ROW_BYTES);
          n1 = *reinterpret_cast<const uint4*>(ws + (aaddr ^ 64u) + (i + 1) * 16 * ROW_BYTES);
        }
#pragma unroll
        for (int j = 0; j < TPW; ++j) mma_fp8(a0, a1, bf[j][0], bf[j][1], acc[i][j]);
        a0 = n0; a1 = n1;
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    // (Round 3 tried software-pipelining the fragment reads here - the second 64-byte half's pixel fragments up front and
    // each of its weight fragments read into the first half's registers as soon as that one's MFMAs were issued, so that a
    // wave pays one LDS latency per tap instead of two: 8 526 vs 8 663 frames/s over three alternating runs on one box,
    // -1.6 %; TC = 6 then needs all 256 VGPRs and spills 6-7.  The partner workgroup's wave already fills those gaps.)
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

  auto act = [&](float x) -> float {
    if (!a.act) return x;
    if constexpr (ES == 4) return silu_exact(x); else return silu_fast(x);     // f16 / fp8: v_exp + v_rcp
  };

#if MIYOLO_ABLATE
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s_begin = 0, acc_wait = 0, acc_issue = 0, acc_comp = 0, acc_slab = 0, acc_slabbar = 0, acc_epi = 0, n_tiles = 0;
  STAMP(s_begin);
#define H2_ACC acc_wait += s1 - s0; acc_issue += s2 - s1; acc_comp += s3 - s2;
#else
#define H2_ACC
#endif

  // ---- K loop: chunk outermost; per tap one barrier, the next tap's weights in flight under this tap's MFMAs
  Tile cur = tile_coords(L);
  set_w(cur.n0);
  set_x(cur.x0);
#if MIYOLO_ABLATE
  unsigned long long p0 = 0, p1 = 0, p2 = 0;
  STAMP(p0);
#endif
  issue_slab(cur, 0);
  issue_w(0, 0, 0);
  STAMP(p1);
  int slot = 0;
  for (;;) {
    const bool has_next = PERSIST && tleft > 0;
    Tile nxt = cur;
    init_acc(cur.n0);
#if MIYOLO_ABLATE
    if (n_tiles == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(p2); }
#endif
    // The tap loop is a real loop over dy with dx unrolled - NOT nine unrolled taps: the fully unrolled kernel was 60-80 KB
    // of code for an instruction cache of 64 KB shared by two CUs (four resident workgroups at different places of it),
    // and every layer ran 10-15 % slower than with a third of the code.
    for (int c = 0; c < g.nchunk; ++c) {
      const bool full = (a.cin - c * CPR) > CPR / 2;
      const bool more = (c + 1 < g.nchunk);
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) {
#define MIYOLO_H2_TAP(DX)                                                                           \
        STAMP(s0);                                                                                  \
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                               \
        STAMP(s1);                                                                                  \
        if (dy * 3 + (DX) < 8) issue_w(c, dy * 3 + (DX) + 1, slot ^ 1);                             \
        else if (more) issue_w(c + 1, 0, slot ^ 1);                                                 \
        else if (has_next) { nxt = tile_coords(L + xstride); set_w(nxt.n0); issue_w(0, 0, slot ^ 1); } \
        STAMP(s2);                                                                                  \
        compute(dy, std::integral_constant<int, DX>{}, slot, full);                                 \
        STAMP(s3);                                                                                  \
        H2_ACC                                                                                      \
        slot ^= 1;
        MIYOLO_H2_TAP(0) MIYOLO_H2_TAP(1) MIYOLO_H2_TAP(2)
#undef MIYOLO_H2_TAP
      }
      if (more) {
        STAMP(s0);
        asm volatile("s_barrier" ::: "memory");        // every wave is done with the slab
        STAMP(s5);
        if (!ABL(1)) issue_slab(cur, c + 1);
        STAMP(s1);
#if MIYOLO_ABLATE
        acc_slab += s1 - s5; acc_slabbar += s5 - s0;
#endif
      }
    }
    STAMP(s4);

    // ---- epilogue (activation, residual, 8 channels per store).  PERSIST: between its blocks the NEXT tile's first slab
    // goes out one DMA at a time - the queue drains while the VALU works, nobody blocks, and the slab has landed when
    // the next tile starts.  All residual loads are issued BEFORE the first of those DMAs (vmcnt retires in order: a
    // load behind a slab DMA would wait for HBM).
    int32_t mpix[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int p = (wave * TPW + j) * 16 + frow;
      const int py = (int)((uint32_t)p / (uint32_t)TW);
      const int y = cur.y0 + py, x = cur.x0 + p - py * TW;
      mpix[j] = (p < NPX && y < H && x < W) ? (cur.bimg * H + y) * W + x : -1;
    }
    constexpr int NBLK = NPAIR * TPW + ((TC & 1) ? TPW : 0);       // epilogue blocks
    constexpr int NPIECE = RPW * GX;                                  // DMAs of a slab per wave
    uint32_t nmsk = 0;
    if (has_next) {
      asm volatile("s_barrier" ::: "memory");                         // every wave is done with the slab of `cur`
      xmask = 0;
      set_x(nxt.x0);
      nmsk = xmask & chunk_mask(0);
    }
    // Without a residual operand the slab's DMAs are spread over all blocks; with one, over the blocks behind the last
    // residual loads only (a load issued behind a slab DMA would wait for it: vmcnt retires in order).
    constexpr int KRES = (NPAIR > 0 ? (NPAIR - 1) * TPW : 0);          // first block behind the last pair's residual loads
    auto pieces_after_block = [&](auto k_tag) __attribute__((always_inline)) {
      constexpr int k = decltype(k_tag)::value;
      if constexpr (PERSIST) {
        if (has_next) {
          if (!a.res) {
            constexpr int q0 = k * NPIECE / NBLK, q1 = (k + 1) * NPIECE / NBLK;
            static_assert(q1 - q0 <= 3, "more than three slab DMAs per epilogue block");
            if constexpr (q1 > q0) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0>{});
            if constexpr (q1 > q0 + 1) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 1>{});
            if constexpr (q1 > q0 + 2) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 2>{});
          } else if constexpr (k >= KRES) {
            constexpr int nb = NBLK - KRES, kk = k - KRES;
            constexpr int q0 = kk * NPIECE / nb, q1 = (kk + 1) * NPIECE / nb;
            static_assert(q1 - q0 <= 6, "more than six slab DMAs per epilogue block");
            if constexpr (q1 > q0) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0>{});
            if constexpr (q1 > q0 + 1) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 1>{});
            if constexpr (q1 > q0 + 2) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 2>{});
            if constexpr (q1 > q0 + 3) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 3>{});
            if constexpr (q1 > q0 + 4) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 4>{});
            if constexpr (q1 > q0 + 5) issue_slab_piece(nxt, nmsk, std::integral_constant<int, q0 + 5>{});
          }
        }
      }
    };
    v2i_t rlast[TPW];
    if constexpr ((TC & 1) && ES != 4) {                 // the unpaired channel tile's residual: with the first loads
      if (a.res) {
        const int n = cur.n0 + 16 * (TC - 1) + 4 * fq;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const uint32_t ro = (n < a.cout && mpix[j] >= 0) ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
          if constexpr (ES == 2) rlast[j] = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0);
          else rlast[j] = (v2i_t){(int)__builtin_amdgcn_raw_buffer_load_b32(rres, ro, 0, 0), 0};
        }
      }
    }
    auto pair_block = [&](auto ip_tag, auto j_tag, const float (&qm)[8], const v4ie_t (&rrow)[TPW]) __attribute__((always_inline)) {
      constexpr int ip = decltype(ip_tag)::value, j = decltype(j_tag)::value;
      const int n = cur.n0 + 32 * ip + 8 * fq;
      const bool ok = n < a.cout && mpix[j] >= 0;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = act(acc[2 * ip][j][r] * qm[r]); v[4 + r] = act(acc[2 * ip + 1][j][r] * qm[4 + r]); }
      if (a.res) {
        if constexpr (ES == 1) {
          float ra[4], rb[4];
          unpack_fp8x4((uint32_t)rrow[j][0], ra); unpack_fp8x4((uint32_t)rrow[j][1], rb);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = fmaf(ra[r], a.res_scale, v[r]); v[4 + r] = fmaf(rb[r], a.res_scale, v[4 + r]); }
        } else if constexpr (ES == 4) {
          const uint32_t ro = ok ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
          const v4ie_t q0 = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rres, ro + 16u, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] += __int_as_float(q0[r]); v[4 + r] += __int_as_float(q1[r]); }
        } else {
          const f16x8 hr = *reinterpret_cast<const f16x8*>(&rrow[j]);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)hr[r];
        }
      }
      const uint32_t so = ok ? (uint32_t)((mpix[j] * a.dst_ld + a.dst_choff + n) * ES) : kOob;
      if constexpr (ES == 1) {
        const float q = a.out_inv_scale;
        const v2i_t o = {(int)pack_fp8x4(v[0] * q, v[1] * q, v[2] * q, v[3] * q), (int)pack_fp8x4(v[4] * q, v[5] * q, v[6] * q, v[7] * q)};
        __builtin_amdgcn_raw_buffer_store_b64(o, rdst, so, 0, MIYOLO_ST_AUX);
      } else if constexpr (ES == 4) {
        const v4ie_t o0 = {__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
        const v4ie_t o1 = {__float_as_int(v[4]), __float_as_int(v[5]), __float_as_int(v[6]), __float_as_int(v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(o0, rdst, so, 0, MIYOLO_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(o1, rdst, so + 16u, 0, MIYOLO_ST_AUX);
      } else {
        const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4ie_t*>(&hv), rdst, so, 0, MIYOLO_ST_AUX);
      }
      pieces_after_block(std::integral_constant<int, ip * TPW + j>{});
    };
    auto pair_row = [&](auto ip_tag) __attribute__((always_inline)) {
      constexpr int ip = decltype(ip_tag)::value;
      float qm[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if constexpr (ES == 1) {             // fp8: per-channel dequantisation scale of this lane's 8 channels
        const int n = cur.n0 + 32 * ip + 8 * fq;
        const v4ie_t q0 = __builtin_amdgcn_raw_buffer_load_b128(rqs, (uint32_t)(n * 4), 0, 0);
        const v4ie_t q1 = __builtin_amdgcn_raw_buffer_load_b128(rqs, (uint32_t)(n * 4 + 16), 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { qm[r] = __int_as_float(q0[r]); qm[4 + r] = __int_as_float(q1[r]); }
      }
      v4ie_t rrow[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) rrow[j] = (v4ie_t){0, 0, 0, 0};
      if constexpr (ES != 4) {
        if (a.res) {                         // the four residual loads of this channel pair in flight together
          const int n = cur.n0 + 32 * ip + 8 * fq;
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const uint32_t ro = (n < a.cout && mpix[j] >= 0) ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
            if constexpr (ES == 2) rrow[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
            else { const v2i_t rr = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0); rrow[j] = (v4ie_t){rr[0], rr[1], 0, 0}; }
          }
        }
      }
      pair_block(ip_tag, std::integral_constant<int, 0>{}, qm, rrow); pair_block(ip_tag, std::integral_constant<int, 1>{}, qm, rrow);
      pair_block(ip_tag, std::integral_constant<int, 2>{}, qm, rrow); pair_block(ip_tag, std::integral_constant<int, 3>{}, qm, rrow);
    };
    if constexpr (NPAIR > 0) pair_row(std::integral_constant<int, 0>{});
    if constexpr (NPAIR > 1) pair_row(std::integral_constant<int, 1>{});
    if constexpr (NPAIR > 2) pair_row(std::integral_constant<int, 2>{});
    static_assert(NPAIR <= 3 && TPW == 4, "epilogue rows are written out for up to three channel-tile pairs of four pixel tiles");
    if constexpr (TC & 1) {                              // unpaired last channel tile: 4 channels per lane
      constexpr int i = TC - 1;
      const int n = cur.n0 + 16 * i + 4 * fq;
      const bool nok = n < a.cout;
      float qm[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (ES == 1) {
        const v4ie_t q0 = __builtin_amdgcn_raw_buffer_load_b128(rqs, (uint32_t)(n * 4), 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) qm[r] = __int_as_float(q0[r]);
      }
      auto last_block = [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act(acc[i][j][r] * qm[r]);
        const bool ok = nok && mpix[j] >= 0;
        if (a.res) {
          if constexpr (ES == 1) {
            float ra[4];
            unpack_fp8x4((uint32_t)rlast[j][0], ra);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(ra[r], a.res_scale, v[r]);
          } else if constexpr (ES == 4) {
            const uint32_t ro = ok ? (uint32_t)((mpix[j] * a.res_ld + a.res_choff + n) * ES) : kOob;
            const v4ie_t rr = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += __int_as_float(rr[r]);
          } else {
            const f16x4 hr = *reinterpret_cast<const f16x4*>(&rlast[j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)hr[r];
          }
        }
        const uint32_t so = ok ? (uint32_t)((mpix[j] * a.dst_ld + a.dst_choff + n) * ES) : kOob;
        if constexpr (ES == 1) {
          const float q = a.out_inv_scale;
          __builtin_amdgcn_raw_buffer_store_b32((int)pack_fp8x4(v[0] * q, v[1] * q, v[2] * q, v[3] * q), rdst, so, 0, MIYOLO_ST_AUX);
        } else if constexpr (ES == 4) {
          const v4ie_t o = {__float_as_int(v[0]), __float_as_int(v[1]), __float_as_int(v[2]), __float_as_int(v[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rdst, so, 0, MIYOLO_ST_AUX);
        } else {
          const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2i_t*>(&hv), rdst, so, 0, MIYOLO_ST_AUX);
        }
        pieces_after_block(std::integral_constant<int, NPAIR * TPW + j>{});
      };
      last_block(std::integral_constant<int, 0>{}); last_block(std::integral_constant<int, 1>{});
      last_block(std::integral_constant<int, 2>{}); last_block(std::integral_constant<int, 3>{});
    }
    STAMP(s5);
#if MIYOLO_ABLATE
    acc_epi += s5 - s4; ++n_tiles;
#endif
    if (!has_next) break;
    cur = nxt;
    L += xstride;
    --tleft;
  }
#undef H2_ACC
#if MIYOLO_ABLATE
  if (a.dbg && lane == 0 && blockIdx.x < 512) {      // per wave: total, wait, issue, compute, slab, epilogue cycles
    STAMP(s5);
    unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
    d[0] = s5 - s_begin; d[1] = acc_wait; d[2] = acc_issue; d[3] = acc_comp; d[4] = acc_slab; d[5] = acc_epi;
    d[6] = (unsigned long long)g.nchunk * 9 * n_tiles; d[7] = acc_slabbar + 1;
    unsigned long long* e = a.dbg + 16384 + ((size_t)blockIdx.x * 4 + wave) * 4;
    e[0] = p0 - s_begin; e[1] = p1 - p0; e[2] = p2 - p1; e[3] = n_tiles;
  }
#endif
}

// host side ------------------------------------------------------------------------------------------------------
template <int TC> constexpr size_t h2_wring_bytes() { return (size_t)2 * ((TC * 16 + 31) / 32 * 32) * ROW_BYTES; }

inline int h2_pick_tc(int cout, int elem = 2) {
  // channel tile: 96 where it divides or the tail is small, else 64 / 48
  if (elem == 4) return (cout % 48 == 0 || cout <= 48) ? 3 : (cout % 64 == 0 || cout <= 64) ? 4 : 3;   // fp32 (parity mode): 96-channel tiles spill
  if (elem == 1) return (cout % 64 == 0 || (cout > 48 && cout <= 64)) ? 4 : 3;   // fp8: 32-byte fragments, 96-channel tiles spill too
  if (cout <= 48) return 3;
  if (cout <= 64) return 4;
  if (cout % 96 == 0) return 6;
  if (cout % 64 == 0) return 4;
  return 6;
}

// Tile geometry for an H x W map: the compiled TW x TH shape with the best pixel utilisation.
inline bool h2_shape(int cin, int cout, int B, int H, int W, int elem, int ce, int tc, H2Geom* g, size_t* lds, int* geo) {
  const size_t wring = (size_t)2 * ((tc * 16 + 31) / 32 * 32) * ROW_BYTES;
  double best = -1.0;
  int bgeo = -1;
  for (int ge = 0; ge < 3; ++ge) {
    int tw, th, hp;
    h2_geo(ge, &tw, &th, &hp);
    const size_t need = (size_t)(th + 2) * hp * ROW_BYTES + wring;
    if (need > (size_t)kH2LdsMax) continue;
    const double tiles = (double)((W + tw - 1) / tw) * ((H + th - 1) / th);
    const double util = (double)W * H / (tiles * 256.0);
    const double score = util - 1e-3 * ge;                           // ties: the 16 x 16 tile
    if (score > best) { best = score; bgeo = ge; }
  }
  if (bgeo < 0) return false;
  const int cpr = 8 * ce;
  h2_geo(bgeo, &g->TW, &g->TH, &g->HP);
  g->GX = (g->TW + 2 + 7) / 8;
  g->tiles_x = (W + g->TW - 1) / g->TW; g->tiles_y = (H + g->TH - 1) / g->TH; g->NB = (cout + tc * 16 - 1) / (tc * 16);
  g->ntiles = B * g->tiles_x * g->tiles_y * g->NB;
  g->nchunk = (cin + cpr - 1) / cpr;
  g->npx = g->TH * g->TW;
  g->slab_bytes = (g->TH + 2) * g->HP * ROW_BYTES;
  host_magic((uint32_t)g->TW, &g->mg_tw_mul, &g->mg_tw_shift);
  host_magic((uint32_t)g->NB, &g->mg_nb_mul, &g->mg_nb_shift);
  host_magic((uint32_t)g->tiles_x, &g->mg_tx_mul, &g->mg_tx_shift);
  host_magic((uint32_t)g->tiles_y, &g->mg_ty_mul, &g->mg_ty_shift);
  g->bias_bytes = (uint32_t)((cout + 127) / 128 * 128 * 4);
  g->scratch_off = 0;          // unused
  g->warm = 0;                 // unused
  *lds = (size_t)g->slab_bytes + wring;
  *geo = bgeo;
  (void)elem;
  return true;
}

inline double h2_util(const H2Geom& g, int H, int W) {
  return (double)W * H / ((double)g.tiles_x * g.tiles_y * 256.0);
}

template <typename T>
inline bool h2_geometry(const ConvArgs& a, H2Geom* g, size_t* lds, int* tc, int* geo) {
  constexpr int ES = (int)sizeof(T);
  if (a.ksize != 3 || a.stride != 1 || a.nsrc != 1 || a.src[0].up || a.out_f32) return false;
  if (a.Hin != a.Hout || a.Win != a.Wout) return false;
  if ((a.src[0].ch_off * ES) % 16 || (a.src[0].ld * ES) % 16 || a.cin % DT<T>::CE) return false;
  if (a.cout % 8 || a.dst_ld % 8 || a.dst_choff % 8) return false;                 // 8-channel stores
  if (a.res && (a.res_ld % 8 || a.res_choff % 8)) return false;
  if ((long)a.B * a.Hin * a.Win * a.src[0].ld * ES >= (1l << 31)) return false;
  *tc = h2_pick_tc(a.cout, ES);
  return h2_shape(a.cin, a.cout, a.B, a.Hout, a.Wout, ES, DT<T>::CE, *tc, g, lds, geo);
}

template <typename T>
inline bool h2_eligible(const ConvArgs& a, double min_util) {
  H2Geom g; size_t lds; int tc, geo;
  return h2_geometry<T>(a, &g, &lds, &tc, &geo) && h2_util(g, a.Hout, a.Wout) >= min_util;
}

template <typename T, int TC, bool PERSIST>
inline void launch_h2_geo(const ConvArgs& a, const H2Geom& g, int geo, size_t lds, hipStream_t s, unsigned nwg) {
  const dim3 grid(nwg), blk(256);
  switch (geo) {
    case 1: hipLaunchKernelGGL((conv_h2_kernel<T, TC, 1, PERSIST>), grid, blk, lds, s, a, g); break;
    case 2: hipLaunchKernelGGL((conv_h2_kernel<T, TC, 2, PERSIST>), grid, blk, lds, s, a, g); break;
    default: hipLaunchKernelGGL((conv_h2_kernel<T, TC, 0, PERSIST>), grid, blk, lds, s, a, g); break;
  }
}
template <typename T, int TC>
inline void launch_h2_tc(const ConvArgs& a, const H2Geom& g, int geo, size_t lds, hipStream_t s, int ncu, int persist) {
  // persistent form: two workgroups per CU, every XCD chunk walked with stride grid / 8; only worth it (and only built)
  // for the 16 / 8-bit types and when there is more than one tile per workgroup slot
  if constexpr (sizeof(T) != 4) {
    if (persist && g.ntiles > 2 * ncu) {
      const unsigned nwg = (unsigned)((2 * ncu + 7) / 8 * 8);
      launch_h2_geo<T, TC, true>(a, g, geo, lds, s, nwg);
      return;
    }
  }
  launch_h2_geo<T, TC, false>(a, g, geo, lds, s, (unsigned)g.ntiles);
}

template <typename T>
inline hipError_t launch_conv_h2(const ConvArgs& a, hipStream_t s, int ncu, int warm = 1) {     // warm: 1 = persistent form where it applies
  H2Geom g; size_t lds; int tc, geo;
  if (!h2_geometry<T>(a, &g, &lds, &tc, &geo)) return hipErrorInvalidValue;
  switch (tc) {
    case 3: launch_h2_tc<T, 3>(a, g, geo, lds, s, ncu, warm); break;
    case 4: launch_h2_tc<T, 4>(a, g, geo, lds, s, ncu, warm); break;
    default:
      if constexpr (!is_fp8<T>::value) launch_h2_tc<T, 6>(a, g, geo, lds, s, ncu, warm);
      else return hipErrorInvalidValue;
      break;
  }
  return hipGetLastError();
}

template <typename T>
inline hipError_t set_h2_attrs() {
  hipError_t e;
#define MIYOLO_H2_ATTR(TC, GEO)                                                                         \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<T, TC, GEO, false>),        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, kH2LdsMax)) != hipSuccess) return e; \
  if constexpr (sizeof(T) != 4) {                                                                         \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<T, TC, GEO, true>),       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kH2LdsMax)) != hipSuccess) return e; }
  MIYOLO_H2_ATTR(3, 0) MIYOLO_H2_ATTR(4, 0) MIYOLO_H2_ATTR(3, 1) MIYOLO_H2_ATTR(4, 1) MIYOLO_H2_ATTR(3, 2) MIYOLO_H2_ATTR(4, 2)
  if constexpr (!is_fp8<T>::value) { MIYOLO_H2_ATTR(6, 0) MIYOLO_H2_ATTR(6, 1) MIYOLO_H2_ATTR(6, 2) }
#undef MIYOLO_H2_ATTR
  return hipSuccess;
}

}  // namespace miyolo

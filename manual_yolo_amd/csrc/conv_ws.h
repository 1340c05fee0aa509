// Implicit-GEMM convolution, warp-specialised: 4 producer waves feed an LDS ring by LDS-DMA, 4 consumer
// waves (one per SIMD) run the MFMAs.  Persistent workgroups, same tile math and K table as conv_dmap.h.
//
// Why (profiles/r01_conv_stamps.md, r01_ws_kernel.md): when every wave both fetches and computes, a K step is
// [all waves stalled in DMA issue ~900 cycles] + [both SIMD partners queueing on the matrix pipe ~770] +
// [barrier skew ~500]: the vector-memory path, the LDS and the matrix pipe take turns instead of overlapping.
// LDS-DMA issue blocks the issuing WAVE, not the CU - so the stall is moved onto waves with nothing else to do:
//   * waves 4-7 (producers) run the fill stream: wait for a free ring slot, issue its DMAs (a quarter of the
//     rows each), and publish a slot once their quarter has landed (vmcnt is per wave and counts in order).
//   * waves 0-3 (consumers, one per SIMD) own a 128 x (32..64) or 64 x (16..64) channel-by-pixel wave tile (twice
//     the register reuse of the ring kernel's 64 x 48 tiles), with a ROLLING fragment prefetch: as soon as pixel
//     fragment j of the current stage has fed its MFMAs, its registers are reloaded from the next stage; the
//     weight fragments are double-buffered.  One wave per SIMD then keeps the matrix pipe fed without a partner.
// Ring: 6 slots of K = 32 elements (64-byte rows, chunk position XOR-swizzled by the row's quad within its 16-row
// block: conflict-free ds_read_b128, and a 16-row DMA is one contiguous KiB).  Half-size stages and twice the slots, because what limits a flag-synchronised
// ring is the loop  release slot -> poll -> issue -> land -> publish -> poll : it has (slots - 1.5) stage times to
// complete, which 3 slots of K = 64 did not give it (measured: both sides waiting on each other).
// Synchronisation is per slot through LDS counters (full[6], empty[6]), no s_barrier in the main loop.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dmap.h"

namespace miyolo {

constexpr int WS_NST = 6;          // ring slots
constexpr int WS_ROW = 64;         // bytes per LDS row of a stage (4 chunks of 16 B = 32 f16 / 16 f32 of K)
template <int CN, int TC>
constexpr int ws_bnp() { return (CN * TC * 16 + 63) / 64 * 64; }
template <int CN, int TC>
constexpr size_t ws_lds_bytes() { return (size_t)WS_NST * (DMA_BM + ws_bnp<CN, TC>()) * WS_ROW; }

// LDS flag words behind the K table: full[slot] counts producer waves that have landed their quarter of the
// stage in `slot` (4 per use of the slot), empty[slot] counts consumer waves done with it (4 per use).
// A waiting wave polls with s_sleep; the trip counts are static, every flag is eventually raised, and a
// bounded spin (kSpinMax polls) turns a protocol bug into wrong results instead of a hung GPU.
// Flags are read/written with inline-asm LDS ops on raw LDS addresses: a C++ volatile/atomic access makes the
// compiler wait vmcnt(0) first, which would drain the producers' DMAs in flight on every poll.
constexpr int kSpinMax = 1 << 16;
__device__ __forceinline__ uint32_t ws_peek(uint32_t flag_addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr) : "memory");
  return v;
}
__device__ __forceinline__ void ws_wait_ge(uint32_t flag_addr, uint32_t target) {
#ifdef MIYOLO_WS_NOWAIT
  return;                                // timing experiment (results wrong): no handshake at all
#endif
  uint32_t v = ws_peek(flag_addr);
  int spins = 0;
  while ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) < target && ++spins < kSpinMax) {
    __builtin_amdgcn_s_sleep(1);
    v = ws_peek(flag_addr);
  }
}
__device__ __forceinline__ void ws_signal(uint32_t flag_addr, int lane) {
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(flag_addr), "v"(1u) : "memory");
}

template <typename T, int KS, int CN, int TC>
__global__ __launch_bounds__(512) void conv_ws_kernel(const ConvArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int CM = 4 / CN;                    // consumer waves along pixels
  constexpr int TPW = DMA_BM / CM / 16;         // 16-pixel tiles per consumer wave: 8 (CN=2) or 4 (CN=1)
  constexpr int BM = DMA_BM;
  constexpr int BN = CN * TC * 16;
  constexpr int BNP = ws_bnp<CN, TC>();
  constexpr int ROWS = BM + BNP;
  constexpr int XIP = BM / 64;                  // activation DMAs (16 rows each) per producer wave per stage: 4
  constexpr int WIP = BNP / 64;                 // weight DMAs per producer wave per stage: 1 or 2
  constexpr int NIP = XIP + WIP;
  constexpr int STAGE = ROWS * WS_ROW;
  constexpr int NST = WS_NST;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int NB = (a.cout + BN - 1) / BN;
  const int MB = (a.M + BM - 1) / BM;
  const int ntiles = MB * NB;
  const int G = gridDim.x;
  const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < ntiles) ? (ntiles - first + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int nk2 = a.nk * 2;                     // stages per tile (a.nk counts K steps of 8 chunks)
  const int total_stages = my_tiles * nk2;
  constexpr uint32_t kOob = 0x80000000u;
  const int ct0 = a.src[0].ch_cnt / CE;
  const int ct1 = (a.nsrc > 1) ? a.src[1].ch_cnt / CE : 0;

  // ---- K table (see conv_dmap.h: one entry per 16-byte chunk of the flattened K axis), flags zeroed
  uint32_t* const ktab = reinterpret_cast<uint32_t*>(smem + NST * STAGE);
  uint32_t* const flags = ktab + a.nk * 8;      // full[0..5], empty[0..5]
  for (int e = tid; e < a.nk * 8; e += 512) {
    uint32_t v;
    if constexpr (KS == 3) {
      const int tp = e / ct0, co = e - tp * ct0;
      v = (tp < 9) ? ((uint32_t)tp << 28) | (uint32_t)((((tp / 3) * a.src[0].w + tp % 3) * a.src[0].ld + co * CE) * (int)sizeof(T))
                   : (9u << 28);
    } else {
      const bool s1 = e >= ct0;
      const int cq = s1 ? e - ct0 : e;
      const bool ok = cq < (s1 ? ct1 : ct0);
      v = ok ? ((s1 ? 1u : 0u) << 28) | (uint32_t)(cq * CE * (int)sizeof(T)) : kOob;
    }
    ktab[e] = v;
  }
  if (tid < 2 * NST) flags[tid] = 0u;
  __syncthreads();
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t fl_full = lds_base + (uint32_t)(NST * STAGE + a.nk * 32);
  const uint32_t fl_empty = fl_full + NST * 4;

  if (wave >= 4) {
    // =========================================================================== producers
    const int p = wave - 4;
    __builtin_amdgcn_s_setprio(3);       // the few VALU ops of the fill stream go ahead of the consumer's MFMA queue
    const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
    const v4i_t rs1 = make_srd(a.src[1].ptr, a.src[1].bytes);
    const v4i_t rsw = make_srd(a.w, a.wbytes);
    const int rsub = lane >> 2;          // row within a 16-row DMA
    // chunk swizzle: row r of a 16-row block stores logical chunk c at position c ^ ((-(r >> 2)) & 3), which puts the
    // four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS) on 16 distinct 16-byte bank slots
    const int cq = (lane & 3) ^ ((4 - (lane >> 4)) & 3);   // logical K chunk this lane fetches
    const int HWo = a.Hout * a.Wout;
    int32_t xoff0[XIP];
    int32_t xoff1[KS == 1 ? XIP : 1];
    uint32_t xinv[XIP];
    uint32_t woff[WIP];
    int d_tile = first, d_ks = 0, d_slot = 0;

    // Row state of a tile.  DMA i of this wave covers rows 16*(p + 4*i) + (lane >> 2): the 4 lanes of a row group
    // share a row, so lane L computes ONE row - (i = L >> 4, row = L & 15) - and the wave transposes with
    // ds_bpermute: DMA i of lane L takes its values from lane 16*i + (L >> 2).
    const int bp_base = (lane >> 2) * 4;
    auto setup_tile = [&](int tile) {
      const int mb = tile / NB, nb = tile - mb * NB;
      const int m0 = mb * BM, n0 = nb * BN;
      int32_t c_off0, c_off1 = 0;
      uint32_t c_inv;
      {
        const int m = m0 + 16 * (p + 4 * (lane >> 4)) + (lane & 15);
        const bool vm = m < a.M;
        const uint32_t mm = vm ? (uint32_t)m : 0u;
        const int b = (int)magic_div(mm, a.mg_hw_mul, a.mg_hw_shift);
        const uint32_t rem = mm - (uint32_t)b * (uint32_t)HWo;
        const int ho = (int)magic_div(rem, a.mg_w_mul, a.mg_w_shift);
        const int wo = (int)rem - ho * a.Wout;
        if constexpr (KS == 3) {
          const int hi0 = ho * a.stride - 1, wi0 = wo * a.stride - 1;
          c_off0 = (((b * a.src[0].h + hi0) * a.src[0].w + wi0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
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
      for (int i = 0; i < XIP; ++i) {
        xoff0[i] = __builtin_amdgcn_ds_bpermute(bp_base + i * 64, c_off0);
        xinv[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + i * 64, (int)c_inv);
        if constexpr (KS == 1) xoff1[i] = __builtin_amdgcn_ds_bpermute(bp_base + i * 64, c_off1);
      }
#pragma unroll
      for (int i = 0; i < WIP; ++i) {
        const int row = 16 * (p + 4 * i) + rsub, n = n0 + row;
        woff[i] = (row < BN && n < a.cout) ? (uint32_t)(n * a.kpad * (int)sizeof(T) + cq * 16) : kOob;
      }
    };
#ifdef MIYOLO_WS_L2WRAP
    const uint32_t ab_and = 0x800FFFFFu;   // timing experiment (-DMIYOLO_WS_L2WRAP, results wrong): activations wrapped into 1 MiB (L2 resident)
#else
    const uint32_t ab_and = ABL(128) ? 0x800FFFFFu : 0xFFFFFFFFu;
#endif
    auto issue_stage = [&](const uint32_t e) {   // the NIP DMAs of (d_tile, stage d_ks) into slot d_slot; e = its K-table entry
      const uint32_t st = lds_base + (uint32_t)(d_slot * STAGE + p * 1024);
      const int ks = d_ks;
      if (!ABL(1)) {                 // timing experiment (results wrong): no tile DMA
      if constexpr (KS == 3) {
        const uint32_t tp = e >> 28, kofs = e & 0x0FFFFFFFu;
#pragma unroll
        for (int i = 0; i < XIP; ++i) {
          const uint32_t off = (((uint32_t)xoff0[i] + kofs) | (((xinv[i] >> tp) & 1u) << 31)) & ab_and;
          lds_dma16(rs0, st + i * 4096, off);
        }
      } else {
        const bool seg1 = (ks * 4) >= ct0;                      // wave-uniform (segment 0 is K-step aligned)
        const uint32_t kofs = e & 0x8FFFFFFFu;
        if (!seg1) {
#pragma unroll
          for (int i = 0; i < XIP; ++i) lds_dma16(rs0, st + i * 4096, (((uint32_t)xoff0[i] + kofs) | xinv[i]) & ab_and);
        } else {
#pragma unroll
          for (int i = 0; i < XIP; ++i) lds_dma16(rs1, st + i * 4096, (((uint32_t)xoff1[i] + kofs) | xinv[i]) & ab_and);
        }
      }
#pragma unroll
      for (int i = 0; i < WIP; ++i) lds_dma16(rsw, st + BM * WS_ROW + i * 4096, woff[i] + (uint32_t)(ks * 64));
      }
    };

#if MIYOLO_ABLATE
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, pa_issue = 0, pa_vm = 0, pa_empty = 0, pa_setup = 0, t_begin = 0;
    STAMP(t_begin);
#endif
    setup_tile(d_tile);
    // Up to three stages of this wave in flight: after issuing stage s the wave waits until only the two youngest
    // are outstanding (vmcnt counts in order), i.e. stage s-2 has landed, and publishes it.
    int n_pend = 0, pend_slot = 0;       // issued, unpublished stages: slots pend_slot, pend_slot+1, ... (mod NST)
    uint32_t round4 = 0;                 // 4 * (uses of the current slot so far)
    auto publish_oldest = [&]() {
      ws_signal(fl_full + pend_slot * 4, lane);
      pend_slot = (pend_slot == NST - 1) ? 0 : pend_slot + 1;
      --n_pend;
    };
    // Flag reads cost a full LDS round trip behind whatever the wave has queued (measured ~200-400 cycles), so
    // they are issued one stage EARLY as plain loads and only looked at when needed; the polling loop is the
    // fallback when the early look says "not yet".
    uint32_t pk = 0;                                            // early look at empty[d_slot]
    uint32_t e_cur = ktab[cq];                                  // K-table entry of the stage to issue
    for (int s = 0; s < total_stages; ++s) {
      STAMP(t0);
      if (round4 && (uint32_t)__builtin_amdgcn_readfirstlane((int)pk) < round4)
        ws_wait_ge(fl_empty + d_slot * 4, round4);              // slot reuse: the consumers have released stage s-6
      STAMP(t1);
      {
        const int nslot = (d_slot == NST - 1) ? 0 : d_slot + 1;
        const int nks = (d_ks + 1 == nk2) ? 0 : d_ks + 1;
        asm volatile("" ::: "memory");
        pk = flags[NST + nslot];
        const uint32_t e_next = ktab[nks * 4 + cq];
        issue_stage(e_cur);
        e_cur = e_next;
      }
      ++n_pend;
      STAMP(t2);
      if (n_pend == 3) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIP) : "memory");
        publish_oldest();
      }
      STAMP(t3);
      if (d_slot == NST - 1) { d_slot = 0; round4 += 4; } else ++d_slot;
      if (++d_ks == nk2) {
        d_ks = 0;
        d_tile += G;
        if (d_tile < ntiles) {
          // the next tile's row state is index arithmetic on a SIMD shared with a consumer: publish what is in
          // flight first (the consumers would otherwise wait for the tile's last stages until after the setup)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          while (n_pend > 0) publish_oldest();
          setup_tile(d_tile);
        }
      }
      STAMP(t4);
#if MIYOLO_ABLATE
      pa_empty += t1 - t0; pa_issue += t2 - t1; pa_vm += t3 - t2; pa_setup += t4 - t3;
      if (a.dbg && blockIdx.x == 0 && p == 0 && lane == 0 && s < 480) {     // event trace of workgroup 0 (tools/stamp_probe.py --trace)
        unsigned long long* tr = a.dbg + 16384 + (size_t)s * 8;
        tr[0] = t0 - t_begin; tr[1] = t1 - t_begin; tr[2] = t2 - t_begin; tr[3] = t3 - t_begin; tr[4] = t4 - t_begin;
      }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    while (n_pend > 0) publish_oldest();
#if MIYOLO_ABLATE
    if (a.dbg && lane == 0) {          // per producer wave: total, vmcnt wait, issue, wait for a free slot, tile setup
      unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      d[0] = t4 - t_begin; d[1] = pa_vm; d[2] = pa_issue; d[3] = pa_empty; d[4] = pa_setup; d[5] = (unsigned long long)(total_stages / 2);
      d[6] = (unsigned long long)my_tiles; d[7] = 2;
    }
#endif
    return;
  }

  // ============================================================================= consumers
  const int cm = wave / CN, cn = wave % CN;
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  const int frow = lane & 15, fq = lane >> 4;
  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage: 16 rows x 64 B = one contiguous KiB per fragment
  const uint32_t fl = (uint32_t)(frow * WS_ROW + ((fq ^ ((4 - (frow >> 2)) & 3)) * 16));
  auto lda = [&](const unsigned char* st, int i) -> uint4 {
    if (ABL(4)) return make_uint4(i, lane, 1, 2);          // timing experiment: no fragment reads
    return *reinterpret_cast<const uint4*>(st + (BM + (cn * TC + i) * 16) * WS_ROW + fl);
  };
  auto ldb = [&](const unsigned char* st, int j) -> uint4 {
    if (ABL(4)) return make_uint4(j, lane, 3, 4);
    return *reinterpret_cast<const uint4*>(st + ((cm * TPW + j) * 16) * WS_ROW + fl);
  };
  auto mma = [&](const uint4& x, const uint4& y, f32x4& c) {
    if (!ABL(2)) Mma<T>::run(x, y, c); else c[0] += __uint_as_float(x.x ^ y.y);
  };

  uint4 af[2][TC], bf[TPW];
  int c_tile = first, c_ks = 0, c_slot = 0;
  uint32_t c_full = 4;                   // full[] target of the stage in c_slot
#if MIYOLO_ABLATE
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ca_comp = 0, ca_epi = 0, ca_wait = 0, t_begin = 0;
  STAMP(t_begin);
#endif
  ws_wait_ge(fl_full, 4);
#pragma unroll
  for (int i = 0; i < TC; ++i) af[0][i] = lda(smem, i);
#pragma unroll
  for (int j = 0; j < TPW; ++j) bf[j] = ldb(smem, j);

  // two stages (one K step of 8 chunks) per iteration, so that the weight-fragment double buffer has static names
  const int total_steps = total_stages >> 1;
  uint32_t pk1 = 0;                      // early look at full[] of the next iteration's first prefetch stage
  for (int c = 0; c < total_steps; ++c) {
    STAMP(t0);
    const int s1 = (c_slot == NST - 1) ? 0 : c_slot + 1;                 // NST is even: s1 never wraps... kept general
    const uint32_t f1 = (c_slot == NST - 1) ? c_full + 4 : c_full;
    const int s2 = (s1 == NST - 1) ? 0 : s1 + 1;
    const uint32_t f2 = (s1 == NST - 1) ? f1 + 4 : f1;
    const bool has_next = c + 1 < total_steps;
    const bool last = (c_ks + 1 == a.nk);          // last K step of the tile: the epilogue comes before the next stage
    // ---- even stage (fragments already in registers; every read of it is issued: the slot is free once the add runs,
    // LDS executes a wave's ops in order).  pk1/pk2: early looks at full[s1]/full[s2] (see the producer loop).
    ws_signal(fl_empty + c_slot * 4, lane);
    const uint32_t pk2 = flags[s2];
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)pk1) < f1) ws_wait_ge(fl_full + s1 * 4, f1);
    STAMP(t1);
    {
      const unsigned char* sn = smem + s1 * STAGE;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
#pragma unroll
        for (int i = 0; i < TC; ++i) mma(af[0][i], bf[j], acc[i][j]);
        bf[j] = ldb(sn, j);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < TC; ++i) af[1][i] = lda(sn, i);
        }
      }
    }
    // ---- odd stage.  On the last step of a tile the prefetch re-reads this (released) stage and the result is dropped:
    // one code path keeps the accumulators in place; the real fragments are loaded after the epilogue.
    STAMP(t2);
    ws_signal(fl_empty + s1 * 4, lane);
    pk1 = flags[(s2 == NST - 1) ? 0 : s2 + 1];                 // next iteration's s1
    if (!last && (uint32_t)__builtin_amdgcn_readfirstlane((int)pk2) < f2) ws_wait_ge(fl_full + s2 * 4, f2);
    STAMP(t3);
    {
      const unsigned char* sn = smem + (last ? s1 : s2) * STAGE;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
#pragma unroll
        for (int i = 0; i < TC; ++i) mma(af[1][i], bf[j], acc[i][j]);
        bf[j] = ldb(sn, j);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < TC; ++i) af[0][i] = lda(sn, i);
        }
      }
    }
    c_slot = s2;
    c_full = f2;
#if MIYOLO_ABLATE
    ca_wait += (t1 - t0) + (t3 - t2); ca_comp += t2 - t1;
    {
      unsigned long long t5;
      STAMP(t5);
      if (a.dbg && blockIdx.x == 0 && wave == 0 && lane == 0 && c < 240) {
        unsigned long long* tr = a.dbg + 16384 + 480 * 8 + (size_t)c * 8;
        tr[0] = t0 - t_begin; tr[1] = t1 - t_begin; tr[2] = t2 - t_begin; tr[3] = t3 - t_begin; tr[4] = t5 - t_begin;
      }
      ca_comp += t5 - t3;
      t3 = t5;
    }
#endif
    if (++c_ks == a.nk) {
      const int mb = c_tile / NB, nb = c_tile - mb * NB;
      const int m0 = mb * BM, n0 = nb * BN;
      const float* __restrict__ bias = a.bias;
      auto run_epilogue = [&](auto outf32_tag) {
        constexpr bool OUTF32 = decltype(outf32_tag)::value;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int nt = __builtin_amdgcn_readfirstlane(n0 + (cn * TC + i) * 16);
          const int n = nt + fq * 4;
          v4i_t s0, s1, s2, s3;
          const float* bp = sgpr_ptr(bias + (nt < a.cout ? nt : 0));
          asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                       "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(bp));
          float bv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bv[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s1[r] : fq == 2 ? s2[r] : s3[r]);
          v4ie_t rv[TPW];
          if (a.res) {                       // wave-uniform: all residual loads of this channel tile in flight together
#pragma unroll
            for (int j = 0; j < TPW; ++j) rv[j] = epilogue_res_load<T>(a, rres, m0 + (cm * TPW + j) * 16 + frow, n);
          } else {
#pragma unroll
            for (int j = 0; j < TPW; ++j) rv[j] = (v4ie_t){0, 0, 0, 0};
          }
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int m = m0 + (cm * TPW + j) * 16 + frow;
            if (!ABL(8)) epilogue_fast<T, OUTF32>(a, rdst, m, n, acc[i][j], bv, rv[j]);
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      };
      if (a.vec_ok) {
        if (a.out_f32) run_epilogue(std::true_type{}); else run_epilogue(std::false_type{});
      } else {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int n = n0 + (cn * TC + i) * 16 + fq * 4;
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int m = m0 + (cm * TPW + j) * 16 + frow;
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
#if MIYOLO_ABLATE
      STAMP(t0);
      ca_epi += t0 - t3;
#endif
      if (has_next) {                              // first fragments of the next tile (its producers ran during the epilogue)
        ws_wait_ge(fl_full + c_slot * 4, c_full);
        const unsigned char* sn = smem + c_slot * STAGE;
#pragma unroll
        for (int i = 0; i < TC; ++i) af[0][i] = lda(sn, i);
#pragma unroll
        for (int j = 0; j < TPW; ++j) bf[j] = ldb(sn, j);
      }
#if MIYOLO_ABLATE
      STAMP(t1);
      ca_wait += t1 - t0;
#endif
    }
  }
#if MIYOLO_ABLATE
  if (a.dbg && lane == 0) {            // per consumer wave: total, wait for a full stage, -, compute, epilogue
    STAMP(t3);
    unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = t3 - t_begin; d[1] = ca_wait; d[2] = 0; d[3] = ca_comp; d[4] = ca_epi; d[5] = (unsigned long long)total_steps;
    d[6] = (unsigned long long)my_tiles; d[7] = 1;
  }
#endif
}

// consumer grid / channel tiles: {CN, TC}; BN = CN*TC*16
inline ConvCfg pick_ws_cfg(int cout, long M) {
  static const ConvCfg cands[] = {{2, 4}, {2, 3}, {2, 2}, {1, 4}, {1, 3}, {1, 2}, {1, 1}};
  ConvCfg best = {1, 1};
  double best_cost = 1e30;
  for (const ConvCfg& c : cands) {
    const int bn = c.wc * c.tc * 16;
    const long nb = (cout + bn - 1) / bn, mbk = (M + DMA_BM - 1) / DMA_BM;
    double cost = (double)(nb * bn) * (double)(mbk * DMA_BM);
    if (nb * mbk < 256) cost *= 1.0 + 0.25 * (256.0 / (double)(nb * mbk) - 1.0);
    cost *= 1.0 + 0.03 * (128.0 / bn);
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <typename T, int KS, int CN, int TC>
inline hipError_t launch_ws_cfg(const ConvArgs& a, hipStream_t s, int ncu) {
  constexpr int BN = CN * TC * 16;
  const size_t lds = ws_lds_bytes<CN, TC>() + (size_t)a.nk * 32 + 64;   // ring + K table + flags
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const long mbk = ((long)a.M + DMA_BM - 1) / DMA_BM, nb = (a.cout + BN - 1) / BN;
  long grid = std::min<long>(mbk * nb, ncu);
  grid = (grid + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_ws_kernel<T, KS, CN, TC>), dim3((unsigned)grid), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename T, int KS>
inline hipError_t launch_ws_ks(const ConvArgs& a, ConvCfg c, hipStream_t s, int ncu) {
  if (c.wc == 2 && c.tc == 4) return launch_ws_cfg<T, KS, 2, 4>(a, s, ncu);
  if (c.wc == 2 && c.tc == 3) return launch_ws_cfg<T, KS, 2, 3>(a, s, ncu);
  if (c.wc == 2 && c.tc == 2) return launch_ws_cfg<T, KS, 2, 2>(a, s, ncu);
  if (c.wc == 1 && c.tc == 4) return launch_ws_cfg<T, KS, 1, 4>(a, s, ncu);
  if (c.wc == 1 && c.tc == 3) return launch_ws_cfg<T, KS, 1, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 2) return launch_ws_cfg<T, KS, 1, 2>(a, s, ncu);
  return launch_ws_cfg<T, KS, 1, 1>(a, s, ncu);
}

template <typename T>
inline hipError_t launch_conv_ws(const ConvArgs& a, hipStream_t s, int ncu, int force_wc = 0, int force_tc = 0) {
  ConvCfg c = pick_ws_cfg(a.cout, a.M);
  if (force_wc > 0 && force_tc > 0 && force_tc <= 4) c = {force_wc, force_tc};
  if (a.ksize == 3) return launch_ws_ks<T, 3>(a, c, s, ncu);
  return launch_ws_ks<T, 1>(a, c, s, ncu);
}

}  // namespace miyolo

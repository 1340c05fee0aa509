// Implicit-GEMM convolution, persistent ring kernel with HALF-SIZE stages and TWO workgroups per CU.
//
// The hardware counters of conv_dmap.h (profiles/r01_pmc_conv_impl3.md) show no saturated unit - matrix pipe 24 %,
// VALU ~22 %, LDS array 14 %, vector-memory path 30-60 % - and waves waiting 53 % of the time: with ONE 8-wave
// workgroup per CU every phase (DMA issue, MFMA, barrier, epilogue) is the whole CU's phase.  The fix the
// hardware is built for is a second, independent workgroup on the CU.  That needs the ring to fit twice into 160 KiB:
// stages of K = 32 elements (64-byte rows, 24 KiB per stage for 256 px + 128 ch rows), three slots = 72 KiB per
// workgroup, and <= 128 VGPRs per lane.  Same tile math, K table, persistent tile list and epilogue as conv_dmap.h.
// LDS rows are 64 B; the chunk position is XOR-swizzled by the row's quad within its 16-row block, which puts the four
// 16-lane groups of a ds_read_b128 on distinct 16-byte bank slots, and a 16-row DMA is one contiguous KiB.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dmap.h"

namespace miyolo {

constexpr int DMH_ROW = 64;            // bytes per LDS row of a stage: 4 chunks of 16 B
constexpr int DMH_BNP = 128;           // weight rows per stage (padded: 3 DMAs per wave for every tile shape)
constexpr int DMH_NST = 3;
constexpr size_t dmh_lds_bytes() { return (size_t)DMH_NST * (DMA_BM + DMH_BNP) * DMH_ROW; }

template <typename T, int KS, int WC, int TC>
__global__ __launch_bounds__(512, 2) void conv_dmh_kernel(const ConvArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int WP = 8 / WC;
  constexpr int TPW = DMA_BM / (WP * 16);
  constexpr int BM = DMA_BM;
  constexpr int BN = WC * TC * 16;
  constexpr int ROWS = BM + DMH_BNP;
  constexpr int XI = BM / 128;           // activation DMAs (16 rows each) per wave per stage: 2
  constexpr int NI = ROWS / 128;         // 3
  constexpr int STAGE = ROWS * DMH_ROW;
  constexpr int NST = DMH_NST;
  static_assert(BN <= DMH_BNP, "channel tile wider than the padded weight rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC, wc = wave % WC;

  const int NB = (a.cout + BN - 1) / BN;
  const int MB = (a.M + BM - 1) / BM;
  const int ntiles = MB * NB;
  const int G = gridDim.x;
  const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < ntiles) ? (ntiles - first + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int nk2 = a.nk * 2;              // stages per tile
  const int total_steps = my_tiles * nk2;

  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rs1 = make_srd(a.src[1].ptr, a.src[1].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr uint32_t kOob = 0x80000000u;

  const int rsub = lane >> 2;                                  // row within a 16-row DMA
  const int cq = (lane & 3) ^ ((4 - (lane >> 4)) & 3);         // logical K chunk this lane fetches (swizzle, see top)
  const int ct0 = a.src[0].ch_cnt / CE;
  const int ct1 = (a.nsrc > 1) ? a.src[1].ch_cnt / CE : 0;
  const int HWo = a.Hout * a.Wout;

  // ---- K table: one entry per 16-byte chunk of the flattened K axis (conv_dmap.h)
  uint32_t* const ktab = reinterpret_cast<uint32_t*>(smem + NST * STAGE);
  for (int e = tid; e < a.nk * 8; e += 512) {
    uint32_t v;
    if constexpr (KS == 3) {
      const int tp = e / ct0, co = e - tp * ct0;
      v = (tp < 9) ? ((uint32_t)tp << 28) | (uint32_t)((((tp / 3) * a.src[0].w + tp % 3) * a.src[0].ld + co * CE) * (int)sizeof(T))
                   : (9u << 28);
    } else {
      const bool s1 = e >= ct0;
      const int cc = s1 ? e - ct0 : e;
      const bool ok = cc < (s1 ? ct1 : ct0);
      v = ok ? ((s1 ? 1u : 0u) << 28) | (uint32_t)(cc * CE * (int)sizeof(T)) : kOob;
    }
    ktab[e] = v;
  }
  __syncthreads();

  // ---- DMA-side state.  DMA i of this wave covers rows 16*(wave + 8*i) + (lane >> 2); lane L computes ONE row -
  // (i = (L >> 4) & 1, row = L & 15) - and the wave transposes with ds_bpermute (DMA i of lane L <- lane 16*i + (L >> 2)).
  int32_t xoff0[XI];
  int32_t xoff1[KS == 1 ? XI : 1];
  uint32_t xinv[XI];
  uint32_t woff;
  int d_tile = first, d_ks = 0, d_slot = 0, d_issued = 0;
  const int bp_base = (lane >> 2) * 4;
  auto setup_tile = [&](int tile) {
    const int mb = tile / NB, nb = tile - mb * NB;
    const int m0 = mb * BM, n0 = nb * BN;
    int32_t c_off0, c_off1 = 0;
    uint32_t c_inv;
    {
      const int m = m0 + 16 * (wave + 8 * ((lane >> 4) & (XI - 1))) + (lane & 15);
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
    for (int i = 0; i < XI; ++i) {
      xoff0[i] = __builtin_amdgcn_ds_bpermute(bp_base + i * 64, c_off0);
      xinv[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + i * 64, (int)c_inv);
      if constexpr (KS == 1) xoff1[i] = __builtin_amdgcn_ds_bpermute(bp_base + i * 64, c_off1);
    }
    {
      const int row = 16 * wave + rsub, n = n0 + row;        // this wave's one weight DMA: rows 16*wave ..+16 of 128
      woff = (row < BN && n < a.cout) ? (uint32_t)(n * a.kpad * (int)sizeof(T) + cq * 16) : kOob;
    }
  };

  auto issue_next = [&]() {
    const uint32_t st = lds_base + (uint32_t)(d_slot * STAGE + wave * 1024);
    const int ks = d_ks;
    const uint32_t e = ktab[ks * 4 + cq];
    if constexpr (KS == 3) {
      const uint32_t tp = e >> 28, kofs = e & 0x0FFFFFFFu;
#pragma unroll
      for (int i = 0; i < XI; ++i)
        lds_dma16(rs0, st + i * 8192, ((uint32_t)xoff0[i] + kofs) | (((xinv[i] >> tp) & 1u) << 31));
    } else {
      const bool seg1 = (ks * 4) >= ct0;                      // wave-uniform (segment 0 is K-step aligned)
      const uint32_t kofs = e & 0x8FFFFFFFu;
      if (!seg1) {
#pragma unroll
        for (int i = 0; i < XI; ++i) lds_dma16(rs0, st + i * 8192, ((uint32_t)xoff0[i] + kofs) | xinv[i]);
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) lds_dma16(rs1, st + i * 8192, ((uint32_t)xoff1[i] + kofs) | xinv[i]);
      }
    }
    lds_dma16(rsw, st + BM * DMH_ROW, woff + (uint32_t)(ks * 64));
    d_slot = (d_slot == NST - 1) ? 0 : d_slot + 1;
    ++d_issued;
    if (++d_ks == nk2) {
      d_ks = 0;
      d_tile += G;
      if (d_tile < ntiles) setup_tile(d_tile);
    }
  };

  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const uint32_t fl = (uint32_t)(frow * DMH_ROW + ((fq ^ ((4 - (frow >> 2)) & 3)) * 16));
  auto compute = [&](int slot) {
    const unsigned char* xs = smem + slot * STAGE + fl;
    const unsigned char* ws = xs + BM * DMH_ROW;
    uint4 af[TC], bf[TPW];
#pragma unroll
    for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + ((wc * TC + i) * 16) * DMH_ROW);
#pragma unroll
    for (int j = 0; j < TPW; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + ((wp * TPW + j) * 16) * DMH_ROW);
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TPW; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
  };

  setup_tile(d_tile);
  issue_next();
  if (total_steps > 1) issue_next();

  int c_tile = first, c_ks = 0, c_slot = 0;
  for (int c = 0; c < total_steps; ++c) {
    if (c + 1 < total_steps) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");   // NI = 3: one younger stage in flight
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (d_issued < total_steps) issue_next();
    compute(c_slot);
    c_slot = (c_slot == NST - 1) ? 0 : c_slot + 1;
    if (++c_ks == nk2) {
      // ---- epilogue of tile c_tile (the next tile's first stages are already in flight)
      const int mb = c_tile / NB, nb = c_tile - mb * NB;
      const int m0 = mb * BM, n0 = nb * BN;
      // bias through the SCALAR unit (wave-uniform address -> s_load, lgkmcnt): a vector load
      // here would make the compiler wait vmcnt(0) and drain the next tile's DMAs in flight.
      // The bias array is padded to a multiple of 128 floats by the host (weights.py).
      const float* __restrict__ bias = a.bias;
      auto run_epilogue = [&](auto outf32_tag) {
        constexpr bool OUTF32 = decltype(outf32_tag)::value;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
          const int nt = __builtin_amdgcn_readfirstlane(n0 + (wc * TC + i) * 16);
          const int n = nt + fq * 4;
          v4i_t s0, s1, s2, s3;            // 16 consecutive biases of this 16-channel tile, in SGPRs
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
            for (int j = 0; j < TPW; ++j) rv[j] = epilogue_res_load<T>(a, rres, m0 + (wp * TPW + j) * 16 + frow, n);
          } else {
#pragma unroll
            for (int j = 0; j < TPW; ++j) rv[j] = (v4ie_t){0, 0, 0, 0};
          }
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int m = m0 + (wp * TPW + j) * 16 + frow;
            epilogue_fast<T, OUTF32>(a, rdst, m, n, acc[i][j], bv, rv[j]);
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      };
      if (a.vec_ok) {
        if (a.out_f32) run_epilogue(std::true_type{}); else run_epilogue(std::false_type{});
      } else {                                   // odd channel counts (e.g. nc = 13): scalar path
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
    }
  }
}

template <typename T, int KS, int WC, int TC>
inline hipError_t launch_dmh_cfg(const ConvArgs& a, hipStream_t s, int ncu) {
  constexpr int BN = WC * TC * 16;
  const size_t lds = dmh_lds_bytes() + (size_t)a.nk * 32;
  if (lds > 80 * 1024) return hipErrorInvalidValue;                   // two workgroups must fit a CU's 160 KiB
  const long mbk = ((long)a.M + DMA_BM - 1) / DMA_BM, nb = (a.cout + BN - 1) / BN;
  long grid = std::min<long>(mbk * nb, 2L * ncu);
#if MIYOLO_DMAP_BALANCED_GRID
  {
    const long rounds = (mbk * nb + 2L * ncu - 1) / (2L * ncu);
    grid = std::min<long>((mbk * nb + rounds - 1) / rounds, 2L * ncu);
  }
#endif
  grid = (grid + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_dmh_kernel<T, KS, WC, TC>), dim3((unsigned)grid), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename T, int KS>
inline hipError_t launch_dmh_ks(const ConvArgs& a, ConvCfg c, hipStream_t s, int ncu) {
  if (c.wc == 2 && c.tc == 4) return launch_dmh_cfg<T, KS, 2, 4>(a, s, ncu);
  if (c.wc == 2 && c.tc == 3) return launch_dmh_cfg<T, KS, 2, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 4) return launch_dmh_cfg<T, KS, 1, 4>(a, s, ncu);
  if (c.wc == 1 && c.tc == 3) return launch_dmh_cfg<T, KS, 1, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 2) return launch_dmh_cfg<T, KS, 1, 2>(a, s, ncu);
  return launch_dmh_cfg<T, KS, 1, 1>(a, s, ncu);
}

inline bool dmh_eligible(const ConvArgs& a) { return dmh_lds_bytes() + (size_t)a.nk * 32 <= 80 * 1024; }

// When the default engine (conv_impl 3) takes this kernel instead of conv_dmap.h: layers whose tile count lies
// between one and two tiles per CU.  The one-workgroup-per-CU kernel runs them as a full round plus a mostly empty
// one; with two resident workgroups per CU all tiles run at once.  Measured (profiles/r01_dmh_vs_dmap.md): 6-17 %
// faster on exactly these layers (20x20 288-channel bottlenecks, the 40x40 64-channel box branch), 12-40 % slower
// everywhere else, where half-size stages only add barriers.
inline bool dmh_preferred_shape(int cout, long M, int ncu) {
  const ConvCfg cd = pick_dma_cfg(cout, M);                    // the shape conv_dmh would run
  const int bn = cd.wc * cd.tc * 16;
  const long mbk = (M + DMA_BM - 1) / DMA_BM, nb = (cout + bn - 1) / bn, tiles = mbk * nb;
  if (tiles <= ncu || tiles > 2L * ncu) return false;
  // two workgroups share a CU: each tile takes ~1.7x as long (same-box timings), but all tiles run in one round
  const double dmh_cost = 1.7 * (bn + MIYOLO_DMAP_C0) * (cd.wc == 1 ? 1.05 : 1.0);
  return dmh_cost < dmap_cost(cout, M, ncu, pick_dmap_cfg(cout, M, ncu));
}
inline bool dmh_preferred(const ConvArgs& a, int ncu) { return dmh_preferred_shape(a.cout, a.M, ncu); }

template <typename T>
inline hipError_t launch_conv_dmh(const ConvArgs& a, hipStream_t s, int ncu, int force_wc = 0, int force_tc = 0) {
  ConvCfg c = pick_dma_cfg(a.cout, a.M);
  if (force_wc > 0 && force_tc > 0 && force_tc <= 4) c = {force_wc, force_tc};
  if (a.ksize == 3) return launch_dmh_ks<T, 3>(a, c, s, ncu);
  return launch_dmh_ks<T, 1>(a, c, s, ncu);
}

}  // namespace miyolo

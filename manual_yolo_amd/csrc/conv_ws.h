// Implicit-GEMM convolution, warp-specialised: 4 producer waves feed an LDS ring by LDS-DMA, 4 consumer
// waves (one per SIMD) run the MFMAs.  Persistent workgroups, same tile math / ring / K table as conv_dmap.h.
//
// Why (profiles/r01_conv_stamps.md, r01_halop_vs_dmap.md): when every wave both fetches and computes, a K step is
// [all waves stalled in DMA issue ~900 cycles] + [both SIMD partners queueing on the matrix pipe ~770] +
// [barrier skew ~500]: the vector-memory path, the LDS and the matrix pipe take turns instead of overlapping,
// and MFMA utilisation stalls at ~0.25.  LDS-DMA issue blocks the issuing WAVE, not the CU - so the stall is
// moved onto waves that have nothing else to do:
//   * waves 4-7 (producers): per step issue the next stage's DMAs (a quarter each), wait for their part of
//     the stage after next, barrier.  They hold ~30 VGPRs of state and never touch the matrix pipe.
//   * waves 0-3 (consumers, one per SIMD): per step read fragments, run the MFMAs of a 128 x (48|64) or
//     64 x (16..64) wave tile (2x the register reuse of the 64 x 48 tiles: 25-35 % fewer LDS fragment bytes per
//     FLOP), barrier; at the end of a tile, the epilogue - during which the producers keep the ring filling.
//   vmcnt is per wave: the consumers' epilogue loads/stores no longer share a counter with the DMAs.
// One s_barrier per K step for all eight waves; stage c+1 is complete at the barrier that ends step c because each
// producer waits for its own quarter before arriving.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dmap.h"

namespace miyolo {

template <int CN, int TC>
constexpr int ws_bnp() { return (CN * TC * 16 + 31) / 32 * 32; }
template <int CN, int TC>
constexpr size_t ws_lds_bytes() { return (size_t)3 * (DMA_BM + ws_bnp<CN, TC>()) * ROW_BYTES; }

template <typename T, int KS, int CN, int TC>
__global__ __launch_bounds__(512) void conv_ws_kernel(const ConvArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int CM = 4 / CN;                    // consumer waves along pixels
  constexpr int TPW = DMA_BM / CM / 16;         // 16-pixel tiles per consumer wave: 8 (CN=2) or 4 (CN=1)
  constexpr int BM = DMA_BM;
  constexpr int BN = CN * TC * 16;
  constexpr int BNP = ws_bnp<CN, TC>();
  constexpr int ROWS = BM + BNP;
  constexpr int XIP = BM / 32;                  // activation DMAs per producer wave per stage (8)
  constexpr int WIP = BNP / 32;                 // weight DMAs per producer wave per stage (1..4)
  constexpr int NIP = XIP + WIP;
  constexpr int STAGE = ROWS * ROW_BYTES;
  constexpr int NST = 3;
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
  const int total_steps = my_tiles * a.nk;
  constexpr uint32_t kOob = 0x80000000u;
  const int ct0 = a.src[0].ch_cnt / CE;
  const int ct1 = (a.nsrc > 1) ? a.src[1].ch_cnt / CE : 0;

  // ---- K table (see conv_dmap.h), built by all eight waves
  uint32_t* const ktab = reinterpret_cast<uint32_t*>(smem + NST * STAGE);
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
  __syncthreads();

  if (wave >= 4) {
    // =========================================================================== producers
    const int p = wave - 4;
    const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
    const v4i_t rs1 = make_srd(a.src[1].ptr, a.src[1].bytes);
    const v4i_t rsw = make_srd(a.w, a.wbytes);
    const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int rsub = lane >> 3;
    const int cg = (lane & 7) ^ (((lane >> 4) + 4 * (p & 1)) & 7);
    const int HWo = a.Hout * a.Wout;
    int32_t xoff0[XIP];
    int32_t xoff1[KS == 1 ? XIP : 1];
    uint32_t xinv[XIP];
    uint32_t woff[WIP];
    int d_tile = first, d_ks = 0, d_slot = 0, d_issued = 0;

    auto setup_tile = [&](int tile) {
      const int mb = tile / NB, nb = tile - mb * NB;
      const int m0 = mb * BM, n0 = nb * BN;
#pragma unroll
      for (int i = 0; i < XIP; ++i) {
        const int m = m0 + 8 * (p + 4 * i) + rsub;
        const bool vm = m < a.M;
        const uint32_t mm = vm ? (uint32_t)m : 0u;
        const int b = (int)magic_div(mm, a.mg_hw_mul, a.mg_hw_shift);
        const uint32_t rem = mm - (uint32_t)b * (uint32_t)HWo;
        const int ho = (int)magic_div(rem, a.mg_w_mul, a.mg_w_shift);
        const int wo = (int)rem - ho * a.Wout;
        if constexpr (KS == 3) {
          const int hi0 = ho * a.stride - 1, wi0 = wo * a.stride - 1;
          xoff0[i] = (((b * a.src[0].h + hi0) * a.src[0].w + wi0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
          const uint32_t hm = (hi0 >= 0 ? 1u : 0u) | 2u | ((hi0 + 2 < a.Hin) ? 4u : 0u);
          const uint32_t wm = (wi0 >= 0 ? 1u : 0u) | 2u | ((wi0 + 2 < a.Win) ? 4u : 0u);
          const uint32_t msk = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? (wm << 3) : 0u) | ((hm & 4u) ? (wm << 6) : 0u);
          xinv[i] = (vm ? (~msk & 0x1FFu) : 0x1FFu) | 0x200u;
        } else {
          const int h0 = a.src[0].up ? (ho >> 1) : ho, w0 = a.src[0].up ? (wo >> 1) : wo;
          xoff0[i] = (((b * a.src[0].h + h0) * a.src[0].w + w0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
          const int h1 = a.src[1].up ? (ho >> 1) : ho, w1 = a.src[1].up ? (wo >> 1) : wo;
          xoff1[i] = (((b * a.src[1].h + h1) * a.src[1].w + w1) * a.src[1].ld + a.src[1].ch_off) * (int)sizeof(T);
          xinv[i] = vm ? 0u : kOob;
        }
      }
#pragma unroll
      for (int i = 0; i < WIP; ++i) {
        const int row = 8 * (p + 4 * i) + rsub, n = n0 + row;
        woff[i] = (row < BN && n < a.cout) ? (uint32_t)(n * a.kpad * (int)sizeof(T) + cg * 16) : kOob;
      }
    };
    const uint32_t ab_and = ABL(128) ? 0x800FFFFFu : 0xFFFFFFFFu;   // timing experiment: activations wrapped into 1 MiB (L2 resident)
    auto issue_next = [&]() {
      const uint32_t st = lds_base + (uint32_t)(d_slot * STAGE + p * 1024);
      const int ks = d_ks;
      if (!ABL(1)) {                 // timing experiment (results wrong): no tile DMA
      const uint32_t e = ktab[ks * 8 + cg];
      if constexpr (KS == 3) {
        const uint32_t tp = e >> 28, kofs = e & 0x0FFFFFFFu;
#pragma unroll
        for (int i = 0; i < XIP; ++i) {
          const uint32_t off = (((uint32_t)xoff0[i] + kofs) | (((xinv[i] >> tp) & 1u) << 31)) & ab_and;
          lds_dma16(rs0, st + i * 4096, off);
        }
      } else {
        const bool seg1 = (ks * 8) >= ct0;
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
      for (int i = 0; i < WIP; ++i) lds_dma16(rsw, st + BM * ROW_BYTES + i * 4096, woff[i] + (uint32_t)(ks * 128));
      }
      d_slot = (d_slot == NST - 1) ? 0 : d_slot + 1;
      ++d_issued;
      if (++d_ks == a.nk) {
        d_ks = 0;
        d_tile += G;
        if (d_tile < ntiles) setup_tile(d_tile);
      }
    };

    setup_tile(d_tile);
    issue_next();
    if (total_steps > 1) {
      issue_next();
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NIP) : "memory");     // stage 0 landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
#if MIYOLO_ABLATE
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, pa_issue = 0, pa_vm = 0, pa_bar = 0, t_begin = 0;
    STAMP(t_begin);
#endif
    for (int c = 0; c < total_steps; ++c) {
      // during step c: issue stage c+2, then make sure this wave's quarter of stage c+1 has landed
      STAMP(t0);
      if (d_issued < total_steps) {
        issue_next();
        STAMP(t1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIP) : "memory");
      } else {
        STAMP(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      STAMP(t2);
      asm volatile("s_barrier" ::: "memory");
      STAMP(t3);
#if MIYOLO_ABLATE
      pa_issue += t1 - t0; pa_vm += t2 - t1; pa_bar += t3 - t2;
#endif
    }
#if MIYOLO_ABLATE
    if (a.dbg && lane == 0) {          // per producer wave: total, vmcnt wait, issue, barrier wait
      unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
      d[0] = t3 - t_begin; d[1] = pa_vm; d[2] = pa_issue; d[3] = pa_bar; d[4] = 0; d[5] = (unsigned long long)total_steps;
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

  asm volatile("s_barrier" ::: "memory");                      // prologue barrier: stage 0 is in LDS
  int c_tile = first, c_ks = 0, c_slot = 0;
#if MIYOLO_ABLATE
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ca_comp = 0, ca_epi = 0, ca_bar = 0, t_begin = 0;
  STAMP(t_begin);
#endif
  for (int c = 0; c < total_steps; ++c) {
    STAMP(t0);
    {
      const unsigned char* xs = smem + c_slot * STAGE;
      const unsigned char* ws = xs + BM * ROW_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 af[TC], bf[TPW];
        if (!ABL(4)) {
#pragma unroll
          for (int i = 0; i < TC; ++i)
            af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((cn * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
          for (int j = 0; j < TPW; ++j)
            bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((cm * TPW + j) * 16 + frow, kk * 4 + fq));
        } else {
#pragma unroll
          for (int i = 0; i < TC; ++i) af[i] = make_uint4(c + i, lane, kk, 1);
#pragma unroll
          for (int j = 0; j < TPW; ++j) bf[j] = make_uint4(c + j, lane, kk, 2);
        }
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
    }
    c_slot = (c_slot == NST - 1) ? 0 : c_slot + 1;
    STAMP(t1);
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
          const float* bp = bias + nt;
          asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                       "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(bp));
          float bv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bv[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s1[r] : fq == 2 ? s2[r] : s3[r]);
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            const int m = m0 + (cm * TPW + j) * 16 + frow;
            if (!ABL(8)) epilogue_fast<T, OUTF32>(a, rdst, rres, m, n, acc[i][j], bv);
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
    }
    STAMP(t2);
    asm volatile("s_barrier" ::: "memory");                    // end of step c: stage c+1 complete, slot c free
    STAMP(t3);
#if MIYOLO_ABLATE
    ca_comp += t1 - t0; ca_epi += t2 - t1; ca_bar += t3 - t2;
#endif
  }
#if MIYOLO_ABLATE
  if (a.dbg && lane == 0) {            // per consumer wave: total, barrier wait, -, compute, epilogue
    unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = t3 - t_begin; d[1] = ca_bar; d[2] = 0; d[3] = ca_comp; d[4] = ca_epi; d[5] = (unsigned long long)total_steps;
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
  const size_t lds = ws_lds_bytes<CN, TC>() + (size_t)a.nk * 32;
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

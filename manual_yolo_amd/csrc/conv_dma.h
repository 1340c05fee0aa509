// Implicit-GEMM convolution, second generation: LDS-DMA staging, 3-stage ring, 8 waves.
//
// Same math, operand roles, flattened-K walk and LDS row format as conv_igemm.h (read its
// header first).  What changes is how tiles reach the LDS and how far ahead they are fetched:
//
// * `buffer_load_dwordx4 ... lds` (LDS-DMA): global -> LDS without passing through VGPRs and
//   without ds_write instructions.  One wave-instruction moves 64 x 16 B = 8 LDS rows of 128 B;
//   the LDS side is linear (base + lane*16), so the bank swizzle is applied on the SOURCE side:
//   the lane that owns LDS slot s of row r fetches global chunk s ^ ((r>>1)&7).
//   An out-of-range byte offset makes the DMA write ZEROS (profiles/r01_probe_lds_dma.log) -
//   that is the conv zero padding and every M/N/K tail, still branch-free.
// * 3-stage ring: while step k is in the MFMAs, the DMAs of steps k+1 and k+2 are in flight
//   (v1 had one step in flight and was latency bound at 2 waves/SIMD).  One raw s_barrier per
//   step, with a counted `s_waitcnt vmcnt(NI)` in front of it: only this wave's DMAs of step
//   k must have landed, the NI of step k+1 stay in flight across the barrier.
// * 512 threads = 8 waves, one workgroup per CU (3 x 48 KiB of LDS): 256 pixels x up to 128
//   channels per workgroup; waves are 4(pixels) x 2(channels) with 64x(48|64) tiles, or
//   8(pixels) x 1 with 32x(16..64) tiles for narrow layers.
#pragma once
#include "common.h"
#include "conv_igemm.h"

namespace miyolo {

constexpr int DMA_STAGES = 3;
constexpr int DMA_BM = 256;

// One LDS-DMA wave-instruction: 64 lanes x 16 B, lane l lands at LDS byte lds_addr + 16*l.
// Written as inline asm on purpose: with the clang builtin the compiler knows an LDS-DMA is in
// flight and, unable to tell ring slots apart, puts `s_waitcnt vmcnt(0)` in front of the first
// ds_read of every K step - which drains the two-step prefetch this kernel exists for.  As asm
// the DMA is invisible to that pass; the kernel counts vmcnt itself (see the ring below).
// M0 carries the LDS address (nothing else in this kernel uses M0; one wait state is required
// between the SALU write of M0 and the DMA).
__device__ __forceinline__ void lds_dma16(const v4i_t rsrc, uint32_t lds_addr, uint32_t voff) {
  // readfirstlane: the address IS wave-uniform, but under SGPR pressure the compiler may keep such a value
  // in a VGPR and then hand the "s" operand a VGPR (assembler error); this pins it to an SGPR.
  const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  const v4i_t rs = {__builtin_amdgcn_readfirstlane(rsrc[0]), __builtin_amdgcn_readfirstlane(rsrc[1]),
                    __builtin_amdgcn_readfirstlane(rsrc[2]), __builtin_amdgcn_readfirstlane(rsrc[3])};   // folded away when already in SGPRs
  // s_nop 2 (+ the two instructions behind it = 5 wait states): an operand SGPR may have been written by a VALU op right
  // before this statement (v_readlane reloading a spilled SGPR, v_readfirstlane above when it is not folded away); a VMEM op
  // needs 5 wait states behind such a write and the compiler's hazard pass does not look inside asm.  A scan of the
  // generated code found 4 such places (spill reloads in conv_dmap<*,1,2,6>) among 3 420 DMA statements.
  asm volatile("s_nop 2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
}
// The 4-byte form (lane l lands at lds_addr + 4*l): used to touch cache lines (L2 warm-up), the data is never read.
__device__ __forceinline__ void lds_dma4(const v4i_t rsrc, uint32_t lds_addr, uint32_t voff) {
  const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  const v4i_t rs = {__builtin_amdgcn_readfirstlane(rsrc[0]), __builtin_amdgcn_readfirstlane(rsrc[1]),
                    __builtin_amdgcn_readfirstlane(rsrc[2]), __builtin_amdgcn_readfirstlane(rsrc[3])};
  asm volatile("s_nop 2\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ v4i_t make_srd(const void* p, uint32_t bytes) {
  const unsigned long long a = (unsigned long long)p;
  return (v4i_t){(int)(uint32_t)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

template <typename T, int KS, int WC, int TC>
__global__ __launch_bounds__(512) void conv_dma_kernel(const ConvArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int WP = 8 / WC;                 // waves along pixels
  constexpr int TPW = DMA_BM / (WP * 16);    // 16-pixel tiles per wave: 4 (WC=2) or 2 (WC=1)
  constexpr int BM = DMA_BM;
  constexpr int BN = WC * TC * 16;
  constexpr int BNP = (BN + 63) / 64 * 64;   // W rows padded so every wave issues the same DMA count
  constexpr int ROWS = BM + BNP;
  constexpr int NI = ROWS / 64;              // DMA instructions per wave per stage (5 or 6)
  constexpr int XI = BM / 64;                // of which for activation rows (4)
  constexpr int STAGE = ROWS * ROW_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  if ABL(64) return;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC, wc = wave % WC;

  const int NB = (a.cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mb = bid / NB, nb = bid - mb * NB;
  const int m0 = mb * BM, n0 = nb * BN;

  constexpr uint32_t kOob = 0x80000000u;
  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rs1 = make_srd(a.src[1].ptr, a.src[1].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

  // DMA instruction i of this wave covers stage rows 8*(wave + 8*i) .. +7; this lane owns LDS
  // slot (lane&7) of row 8*(wave+8i) + (lane>>3) and fetches global chunk cg of that row.
  const int rsub = lane >> 3;
  const int cg = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);

  int32_t xoff0[XI];
  int32_t xoff1[KS == 1 ? XI : 1];
  uint32_t xmask[XI];
  const int HWo = a.Hout * a.Wout;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int m = m0 + 8 * (wave + 8 * i) + rsub;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int b = mm / HWo, rem = mm - b * HWo;
    const int ho = rem / a.Wout, wo = rem - ho * a.Wout;
    if constexpr (KS == 3) {
      const int hi0 = ho * a.stride - 1, wi0 = wo * a.stride - 1;
      xoff0[i] = (((b * a.src[0].h + hi0) * a.src[0].w + wi0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
      uint32_t msk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = hi0 + t / 3, wi = wi0 + t % 3;
        if (vm && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win) msk |= 1u << t;
      }
      xmask[i] = msk;
    } else {
      const int h0 = a.src[0].up ? (ho >> 1) : ho, w0 = a.src[0].up ? (wo >> 1) : wo;
      xoff0[i] = (((b * a.src[0].h + h0) * a.src[0].w + w0) * a.src[0].ld + a.src[0].ch_off) * (int)sizeof(T);
      const int h1 = a.src[1].up ? (ho >> 1) : ho, w1 = a.src[1].up ? (wo >> 1) : wo;
      xoff1[i] = (((b * a.src[1].h + h1) * a.src[1].w + w1) * a.src[1].ld + a.src[1].ch_off) * (int)sizeof(T);
      xmask[i] = vm ? 1u : 0u;
    }
  }
  // weight rows of this lane: stage row BM + 8*(wave + 8*(i-XI)) + rsub, i = XI..NI-1
  uint32_t woff[NI - XI];            // bit 31 set = row outside the tile / cout: the DMA writes zeros
#pragma unroll
  for (int i = 0; i < NI - XI; ++i) {
    const int row = 8 * (wave + 8 * i) + rsub, n = n0 + row;
    woff[i] = (row < BN && n < a.cout) ? (uint32_t)(n * a.kpad * (int)sizeof(T) + cg * 16) : kOob;
  }
  const int ct0 = a.src[0].ch_cnt / CE;
  const int ct1 = (a.nsrc > 1) ? a.src[1].ch_cnt / CE : 0;
  int tap = 0, coff = cg;
  if constexpr (KS == 3) {
    tap = cg / ct0;
    coff = cg - tap * ct0;
  }

  // Issue all DMAs of K step `ks` into ring slot `slot`.  EXACTLY NI instructions per wave, all
  // lanes active: the counted vmcnt below depends on it, so validity is folded into the byte
  // offset with plain ALU ops (bit 31 -> out of range -> zeros), never with a select the
  // compiler could turn into a branch around the load.
  auto issue = [&](int ks, int slot) {
    const uint32_t st = lds_base + (uint32_t)(slot * STAGE + wave * 1024);
    if constexpr (KS == 3) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int32_t toff = ((ky * a.src[0].w + kx) * a.src[0].ld + coff * CE) * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const uint32_t bit = (xmask[i] >> tap) & 1u;          // 0 once tap >= 9 (9-bit mask)
        const uint32_t off = (uint32_t)(xoff0[i] + toff) | ((bit ^ 1u) << 31);
        lds_dma16(rs0, st + i * 8192, off);
      }
      coff += 8;
      while (coff >= ct0) { coff -= ct0; ++tap; }
    } else {
      const int q = ks * 8 + cg;
      const bool seg1 = (ks * 8) >= ct0;                       // wave-uniform
      const int cq = seg1 ? q - ct0 : q;
      const int lim = seg1 ? ct1 : ct0;                        // ct1 = 0 without a second segment
      const uint32_t kvb = ((uint32_t)(cq - lim)) >> 31;       // 1 iff cq < lim
      const int32_t toff = cq * CE * (int)sizeof(T);
      if (!seg1) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const uint32_t off = (uint32_t)(xoff0[i] + toff) | (((kvb & xmask[i]) ^ 1u) << 31);
          lds_dma16(rs0, st + i * 8192, off);
        }
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const uint32_t off = (uint32_t)(xoff1[i] + toff) | (((kvb & xmask[i]) ^ 1u) << 31);
          lds_dma16(rs1, st + i * 8192, off);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NI - XI; ++i) {
      const uint32_t off = woff[i] + (uint32_t)(ks * 128);
      lds_dma16(rsw, st + BM * ROW_BYTES + i * 8192, off);
    }
  };

  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int slot) {
    const unsigned char* xs = smem + slot * STAGE;
    const unsigned char* ws = xs + BM * ROW_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[TC], bf[TPW];
      if (!ABL(4)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
          af[i] = *reinterpret_cast<const uint4*>(ws + lds_off((wc * TC + i) * 16 + frow, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < TPW; ++j)
          bf[j] = *reinterpret_cast<const uint4*>(xs + lds_off((wp * TPW + j) * 16 + frow, kk * 4 + fq));
      } else {
#pragma unroll
        for (int i = 0; i < TC; ++i) af[i] = make_uint4(slot, kk, i, lane);
#pragma unroll
        for (int j = 0; j < TPW; ++j) bf[j] = make_uint4(lane, j, kk, slot);
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
          for (int j = 0; j < TPW; ++j) acc[i][j][0] += __uint_as_float(af[i].x ^ bf[j].y);   // keep the reads live
      }
    }
  };

  // ---- 3-stage ring
  const int nk_eff = ABL(32) ? 0 : a.nk;
  if (nk_eff > 0) issue(0, 0);
  if (nk_eff > 1) issue(1, 1);
  int slot = 0;
  for (int ks = 0; ks < nk_eff; ++ks) {
    if (ks + 1 < a.nk) {
      if constexpr (NI == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (ks + 2 < a.nk && !ABL(1)) issue(ks + 2, slot >= 1 ? slot - 1 : 2);   // (slot + 2) % 3
    compute(slot);
    slot = (slot == 2) ? 0 : slot + 1;
  }

  // ---- epilogue (identical to conv_igemm.h)
  const float* __restrict__ bias = a.bias;
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int n = n0 + (wc * TC + i) * 16 + fq * 4;
    if (n >= a.cout) continue;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (n + r < a.cout) ? bias[n + r] : 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int m = m0 + (wp * TPW + j) * 16 + frow;
      if (m >= a.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[i][j][r] + bv[r];
        if (a.act && !ABL(16)) x = a.exact ? silu_exact(x) : silu_fast(x);
        v[r] = x;
      }
      if (!ABL(8) || v[0] == 123.456f) epilogue_store<T>(a, m, n, v);
    }
  }
}

template <int WC, int TC>
constexpr size_t dma_lds_bytes() {
  return (size_t)DMA_STAGES * (DMA_BM + (WC * TC * 16 + 63) / 64 * 64) * ROW_BYTES;
}

inline ConvCfg pick_dma_cfg(int cout, long M) {
  static const ConvCfg cands[] = {{2, 4}, {2, 3}, {1, 4}, {1, 3}, {1, 2}, {1, 1}};
  ConvCfg best = {1, 1};
  double best_cost = 1e30;
  for (const ConvCfg& c : cands) {
    const int bn = c.wc * c.tc * 16;
    const long nb = (cout + bn - 1) / bn, mbk = (M + DMA_BM - 1) / DMA_BM;
    double cost = (double)(nb * bn) * (double)(mbk * DMA_BM);
    if (nb * mbk < 256) cost *= 1.0 + 0.25 * (256.0 / (double)(nb * mbk) - 1.0);
    cost *= 1.0 + 0.02 * (128.0 / bn);
    if (c.wc == 1) cost *= 1.10;      // 32-pixel wave tiles re-read the weight tile twice as often
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <typename T, int KS, int WC, int TC>
inline hipError_t launch_dma_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WC * TC * 16;
  const long mbk = ((long)a.M + DMA_BM - 1) / DMA_BM, nb = (a.cout + BN - 1) / BN;
  const size_t lds = dma_lds_bytes<WC, TC>();
  hipLaunchKernelGGL((conv_dma_kernel<T, KS, WC, TC>), dim3((unsigned)(mbk * nb)), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename T, int KS>
inline hipError_t launch_dma_ks(const ConvArgs& a, ConvCfg c, hipStream_t s) {
  if (c.wc == 2 && c.tc == 4) return launch_dma_cfg<T, KS, 2, 4>(a, s);
  if (c.wc == 2 && c.tc == 3) return launch_dma_cfg<T, KS, 2, 3>(a, s);
  if (c.wc == 1 && c.tc == 4) return launch_dma_cfg<T, KS, 1, 4>(a, s);
  if (c.wc == 1 && c.tc == 3) return launch_dma_cfg<T, KS, 1, 3>(a, s);
  if (c.wc == 1 && c.tc == 2) return launch_dma_cfg<T, KS, 1, 2>(a, s);
  return launch_dma_cfg<T, KS, 1, 1>(a, s);
}

template <typename T>
inline hipError_t launch_conv_dma(const ConvArgs& a, hipStream_t s, int force_wc = 0, int force_tc = 0) {
  ConvCfg c = pick_dma_cfg(a.cout, a.M);
  if (force_wc > 0 && force_tc > 0) c = {force_wc, force_tc};
  if (a.ksize == 3) return launch_dma_ks<T, 3>(a, c, s);
  return launch_dma_ks<T, 1>(a, c, s);
}

}  // namespace miyolo

// 3x3 stride-1 convolution with the input tile (plus halo) held in LDS across the nine taps.
//
// Why: per-layer profiles of the im2col-style kernels (profiles/r01_v2_perop_f16.json) show every
// 3x3 layer pulling ~4-7 TB/s of tile fills from beyond the XCD L2 - each tap re-fetches the
// activation tile, and with 32 workgroups per XCD filling ~1.4 MB per K step the 4 MiB L2 turns
// over before the next kernel row comes round.  The layers are fill-bandwidth bound, not MFMA
// bound.  Here a workgroup fetches its 256 output pixels' input ONCE per 32-channel chunk, as a
// run of 256 + 2W + 2 consecutive pixels (flattened NHWC rows: one row above, one below, one
// pixel left/right), and every tap reads a shifted window of that LDS image:
//     halo row of (output pixel m_local, tap ky,kx) = m_local + ky*W + kx.
// Zero padding, image borders inside a tile that straddles two frames, and the M tail are a
// per-lane 9-bit mask: an invalid (pixel, tap) reads a dedicated zero row instead.
//
// K order is (channel chunk, tap): one step = one tap of one chunk = one MFMA K step
// (32 f16 / 16 f32 channels, 64-byte LDS rows, chunk slot = q ^ ((row>>1)&3): conflict-free
// for ds_read_b128 at EVERY 16-row window alignment, which shifted windows need).
// Per step a wave issues one weight DMA (16 rows of the tap's [cout][chunk] slice, 4-slot
// ring, three steps ahead) and, during the first XI taps of a chunk, one piece of the NEXT
// chunk's halo (2 halo buffers).  One raw s_barrier per step with a counted vmcnt, as in
// conv_dma.h; the loop is software pipelined by hand (fragments of step s+1 are read from LDS
// before the MFMAs of step s issue), because with one workgroup per CU there are only two
// waves per SIMD to hide the LDS latency behind.  Fill traffic per workgroup and chunk drops from 9*(256+BN) rows to
// (256+2W+2) + 9*BN rows, and the 9*BN weight rows are L2 hits.
#pragma once
#include "common.h"
#include "conv_dma.h"

namespace miyolo {

constexpr int HALO_BM = 256;
constexpr int HROW = 64;              // bytes per LDS row (one K step of one pixel / channel)
constexpr int HALO_WSLOT = 128 * HROW;   // weight ring slot: 128 rows
constexpr int HALO_WRING = 4;            // slots

__device__ __forceinline__ uint32_t halo_off(int row, int q) {
  return (uint32_t)(row * HROW + ((q ^ ((row >> 1) & 3)) << 4));
}

template <typename T> struct MmaH;   // one 16-byte chunk pair per call
template <> struct MmaH<float> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) { Mma<float>::run(a, b, c); }
};
template <> struct MmaH<half_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) { Mma<half_t>::run(a, b, c); }
};

template <typename T, int WC, int TC>
__global__ __launch_bounds__(512) void conv_halo_kernel(const ConvArgs a, const int XI) {
  constexpr int CE = DT<T>::CE;
  constexpr int CC = 4 * CE;                 // channels per chunk (32 f16 / 16 f32)
  constexpr int WP = 8 / WC;
  constexpr int TPW = HALO_BM / (WP * 16);   // 4 (WC=2) or 2 (WC=1)
  constexpr int BN = WC * TC * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [X buf 0: XI*128 rows][X buf 1][W ring 4 x 128 rows]; halo rows >= R are zero-filled by
  // the DMA (forced out of range), and the last row of each halo buffer serves as the zero row
  const int xbytes = XI * 128 * HROW;
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t wring = 2 * xbytes;
  const uint32_t zrow = (uint32_t)((XI * 128 - 1) * HROW);      // inside the current halo buffer

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC, wc = wave % WC;

  const int NB = (a.cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mb = bid / NB, nb = bid - mb * NB;
  const int m0 = mb * HALO_BM, n0 = nb * BN;
  const int W = a.Win, H = a.Hin, HW = H * W;
  const int cin = a.src[0].ch_cnt;
  const int nchunk = (cin + CC - 1) / CC;
  const int nsteps = nchunk * 9;
  const int R = HALO_BM + 2 * W + 2;         // halo rows actually needed


  const v4i_t rsx = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);

  // ---- DMA lane geometry: one instruction = 16 rows x 64 B; lane -> row (lane>>2), slot lane&3,
  // which holds chunk-quarter cgl of that row (source-side swizzle, rows are 16-aligned here)
  const int drow = lane >> 2;
  const int cgl = (lane & 3) ^ ((lane >> 3) & 3);
  // halo row hr <-> flattened input pixel p = m0 - W - 1 + hr (may be < 0 or past the end: the
  // byte offset then falls outside the descriptor and the DMA writes zeros)
  const int ld = a.src[0].ld;
  const int32_t xrow0 = ((m0 - W - 1 + wave * 16 + drow) * ld + a.src[0].ch_off + cgl * CE) * (int)sizeof(T);
  const int32_t xpiece = 128 * ld * (int)sizeof(T);        // one piece = 128 halo rows further
  // weights: row n0 + wave*16 + drow of [cout][kpad]
  const int wrow = wave * 16 + drow;
  const bool wrow_ok = (wrow < BN) && (n0 + wrow < a.cout);
  const uint32_t wbase = wrow_ok ? (uint32_t)(((n0 + wrow) * a.kpad + cgl * CE) * (int)sizeof(T)) : 0x80000000u;

  auto issue_x = [&](int chunk, int piece) {       // halo rows [piece*128, +128) of `chunk`
    const uint32_t chok = (uint32_t)((chunk * CC + cgl * CE) - cin) >> 31;   // 1 iff channel < cin
    const uint32_t rowok = (uint32_t)((piece * 128 + wave * 16 + drow) - R) >> 31;   // 1 iff halo row < R
    const uint32_t off = (uint32_t)(xrow0 + piece * xpiece + chunk * CC * (int)sizeof(T)) | (((chok & rowok) ^ 1u) << 31);
    lds_dma16(rsx, lds_base + (uint32_t)((chunk & 1) * xbytes + (piece * 128 + wave * 16) * HROW), off);
  };
  auto issue_w = [&](int chunk, int tap, int slot) {
    const uint32_t chok = (uint32_t)((chunk * CC + cgl * CE) - cin) >> 31;
    const uint32_t off = (wbase + (uint32_t)((tap * cin + chunk * CC) * (int)sizeof(T))) | ((chok ^ 1u) << 31);
    lds_dma16(rsw, lds_base + wring + (uint32_t)(slot * HALO_WSLOT + wave * 16 * HROW), off);
  };

  // ---- fragment geometry
  const int frow = lane & 15, fq = lane >> 4;
  uint32_t tmask[TPW];       // bit t: (this lane's pixel of tile j, tap t) is inside the image
  int mloc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    mloc[j] = (wp * TPW + j) * 16 + frow;
    const int m = m0 + mloc[j];
    const bool vm = m < a.M;
    const int rem = (vm ? m : 0) % HW;
    const int h = rem / W, w = rem - h * W;
    uint32_t msk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hi = h + t / 3 - 1, wi = w + t % 3 - 1;
      if (vm && hi >= 0 && hi < H && wi >= 0 && wi < W) msk |= 1u << t;
    }
    tmask[j] = msk;
  }
  uint32_t aoff[TC];          // weight fragment offsets inside a ring slot (rows are 16-aligned)
#pragma unroll
  for (int i = 0; i < TC; ++i) aoff[i] = halo_off((wc * TC + i) * 16 + frow, fq);

  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- software-pipelined main loop.  Iteration s:
  //   [A] wait until W(s+1) (and, at a chunk boundary, the next chunk's halo) has landed; barrier
  //   [B] issue DMAs: halo piece tap(s) of the next chunk, then W(s+3)
  //   [C] ds_read the fragments of step s+1 into the other register set
  //   [D] MFMAs of step s from the current register set (the reads of [C] complete underneath)
  // A wave reaches barrier(s+1) only after its MFMAs of step s-1, i.e. after it consumed the
  // fragments of step s-1: the DMA of W(s+3) (4-slot ring -> slot of W(s-1)) and the halo
  // pieces (buffer last read by the previous chunk) cannot overtake a reader.
  auto load_frags = [&](int st, uint4 (&af)[TC], uint4 (&bf)[TPW]) {
    const int ch = st / 9, tp = st - ch * 9;
    const unsigned char* xs = smem + (ch & 1) * xbytes;
    const unsigned char* ws = smem + wring + (st & 3) * HALO_WSLOT;
    const int shift = (tp / 3) * W + (tp % 3);
#pragma unroll
    for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + aoff[i]);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const uint32_t ok = (tmask[j] >> tp) & 1u;
      const uint32_t o = halo_off(mloc[j] + shift, fq);
      const unsigned char* p = xs + (ok ? o : zrow);
      bf[j] = *reinterpret_cast<const uint4*>(p);
    }
  };
  // Measured (profiles/r01_halo_pipelining.md): the hand-pipelined form ([A]..[D] above) is
  // SLOWER than the plain one (wait -> prefetch DMAs -> read fragments -> MFMAs) on every layer:
  // it costs 50 more VGPRs (one workgroup per CU instead of two on the <= 62-pixel-wide maps)
  // and does not raise MFMA utilisation where occupancy is equal, so LDS latency is not what
  // bounds this kernel.  The plain form is the one built (HALO_PIPELINED = 0).
#ifndef HALO_PIPELINED
#define HALO_PIPELINED 0
#endif
#if HALO_PIPELINED
  auto body = [&](int st, uint4 (&afc)[TC], uint4 (&bfc)[TPW], uint4 (&afn)[TC], uint4 (&bfn)[TPW]) {
    const int ch = st / 9, tp = st - ch * 9;
    const bool xp_prev = (st > 0) && ((tp == 0 ? 8 : tp - 1) < XI) && ((tp == 0 ? ch - 1 : ch) + 1 < nchunk);
    if (st + 1 < nsteps) {
      const int allow = ((st + 2 < nsteps) ? 1 : 0) + ((xp_prev && tp != 8) ? 1 : 0);
      if (allow == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
      else if (allow == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (tp < XI && ch + 1 < nchunk) issue_x(ch + 1, tp);
    if (st + 3 < nsteps) {
      const int s3 = st + 3, c3 = s3 / 9;
      issue_w(c3, s3 - c3 * 9, s3 & 3);
    }
    if (st + 1 < nsteps) load_frags(st + 1, afn, bfn);
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TPW; ++j) MmaH<T>::run(afc[i], bfc[j], acc[i][j]);
  };
  for (int p = 0; p < XI; ++p) issue_x(0, p);
  issue_w(0, 0, 0);
  issue_w(0, 1, 1);
  issue_w(0, 2, 2);
  asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
  uint4 af0[TC], bf0[TPW], af1[TC], bf1[TPW];
  load_frags(0, af0, bf0);
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    body(s, af0, bf0, af1, bf1);
    body(s + 1, af1, bf1, af0, bf0);
  }
  if (s < nsteps) body(s, af0, bf0, af1, bf1);
#else
  // plain form.  Iteration s: wait W(s) [and the chunk's halo at tap 0]; barrier; issue the halo
  // piece tap(s) of the next chunk, then W(s+2) (slot of W(s-2): its readers passed barrier(s-1)
  // after their MFMAs of step s-2); read fragments; MFMAs.
  for (int p = 0; p < XI; ++p) issue_x(0, p);
  issue_w(0, 0, 0);
  issue_w(0, 1, 1);
  {
    int chunk = 0, tap = 0;
    bool prev_x = false;
    uint4 af[TC], bf[TPW];
    for (int s = 0; s < nsteps; ++s) {
      const int allow = ((s + 1 < nsteps) ? 1 : 0) + (prev_x ? 1 : 0);
      if (allow == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
      else if (allow == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      prev_x = (tap < XI) && (chunk + 1 < nchunk);
      if (prev_x) issue_x(chunk + 1, tap);
      if (s + 2 < nsteps) {
        const int s2 = s + 2, c2 = s2 / 9;
        issue_w(c2, s2 - c2 * 9, s2 & 3);
      }
      load_frags(s, af, bf);
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) MmaH<T>::run(af[i], bf[j], acc[i][j]);
      if (++tap == 9) { tap = 0; ++chunk; }
    }
  }
#endif

  // ---- epilogue: bias, SiLU, residual, 4 consecutive channels per lane
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
      const int m = m0 + mloc[j];
      if (m >= a.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[i][j][r] + bv[r];
        if (a.act) x = a.exact ? silu_exact(x) : silu_fast(x);
        v[r] = x;
      }
      epilogue_store<T>(a, m, n, v);
    }
  }
}

inline size_t halo_lds_bytes(int XI) { return (size_t)2 * XI * 128 * HROW + HALO_WRING * HALO_WSLOT; }
inline int halo_xi(int W) { return (HALO_BM + 2 * W + 2 + 1 + 127) / 128; }   // +1: the zero row
inline bool halo_eligible(const ConvArgs& a) {
  return a.ksize == 3 && a.stride == 1 && a.nsrc == 1 && !a.src[0].up && halo_xi(a.Win) <= 8;
}

inline ConvCfg pick_halo_cfg(int cout, long M) {
  static const ConvCfg cands[] = {{2, 4}, {2, 3}, {1, 4}, {1, 3}, {1, 2}, {1, 1}};
  ConvCfg best = {1, 1};
  double best_cost = 1e30;
  for (const ConvCfg& c : cands) {
    const int bn = c.wc * c.tc * 16;
    const long nb = (cout + bn - 1) / bn, mbk = (M + HALO_BM - 1) / HALO_BM;
    double cost = (double)(nb * bn) * (double)(mbk * HALO_BM);
    if (nb * mbk < 256) cost *= 1.0 + 0.25 * (256.0 / (double)(nb * mbk) - 1.0);
    cost *= 1.0 + 0.05 * (128.0 / bn);          // every channel tile re-fetches the halo
    if (c.wc == 1) cost *= 1.10;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <typename T, int WC, int TC>
inline hipError_t launch_halo_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int BN = WC * TC * 16;
  const int XI = halo_xi(a.Win);
  const size_t lds = halo_lds_bytes(XI);
  const long mbk = ((long)a.M + HALO_BM - 1) / HALO_BM, nb = (a.cout + BN - 1) / BN;
  hipLaunchKernelGGL((conv_halo_kernel<T, WC, TC>), dim3((unsigned)(mbk * nb)), dim3(512), lds, s, a, XI);
  return hipGetLastError();
}

template <typename T>
inline hipError_t launch_conv_halo(const ConvArgs& a, hipStream_t s, int force_wc = 0, int force_tc = 0) {
  ConvCfg c = pick_halo_cfg(a.cout, a.M);
  if (force_wc > 0 && force_tc > 0) c = {force_wc, force_tc};
  if (c.wc == 2 && c.tc == 4) return launch_halo_cfg<T, 2, 4>(a, s);
  if (c.wc == 2 && c.tc == 3) return launch_halo_cfg<T, 2, 3>(a, s);
  if (c.wc == 1 && c.tc == 4) return launch_halo_cfg<T, 1, 4>(a, s);
  if (c.wc == 1 && c.tc == 3) return launch_halo_cfg<T, 1, 3>(a, s);
  if (c.wc == 1 && c.tc == 2) return launch_halo_cfg<T, 1, 2>(a, s);
  return launch_halo_cfg<T, 1, 1>(a, s);
}

}  // namespace miyolo

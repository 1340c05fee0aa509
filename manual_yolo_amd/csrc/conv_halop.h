// 3x3 stride-1 convolution: persistent workgroups + input halo tile held in LDS across the nine taps.
//
// Why (profiles/r01_conv_stamps.md): in conv_dmap.h a K step costs about MFMA (770 cycles per SIMD) +
// LDS-fill issue (800-1200, 48 KiB through the CU's 64 B/clk vector-memory path) + synchronisation, executed
// one after the other; two thirds of the filled bytes are the activation tile, re-fetched for each of the nine
// taps.  Here a workgroup fetches the 64-channel slab of its 256 output pixels ONCE per chunk, as the run of
// 256 + 2W + 2 consecutive NHWC pixels that covers all nine shifted windows
//     halo row of (output pixel m_local, tap ky,kx) = m_local + ky*W + kx,
// and a step (one tap of one chunk) fills only the tap's weight rows: ~22 KiB per step instead of 48.
// Image borders, tiles straddling two frames and the M tail are a per-lane 9-bit mask that redirects the
// fragment read to a zero row.  K order is (chunk, tap); a trailing 32-channel chunk runs one MFMA K step.
//
// LDS: 2 halo buffers (chunk c+1 - possibly the next tile's first chunk - streams in during the first taps of
// chunk c) + a 3-slot weight ring (two steps ahead).  Rows are 128 B, chunk slot = c ^ (row & 7): conflict-free
// for ds_read_b128 at EVERY window alignment (tests/test_conv_emulation.py), which shifted windows need.
// Same LDS-DMA / counted-vmcnt / persistent-tile machinery as conv_dmap.h; the DMA side needs no division at
// all (halo rows are consecutive pixels), the fragment masks use the magic divisions once per tile.
// Eligible: 3x3, stride 1, one unscaled source, map width <= 95 (halo buffer 448 rows), cout tile <= 128.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dmap.h"

namespace miyolo {

constexpr int HP_BM = 256;
constexpr int HP_WROWS = 128;                      // weight ring slot rows (2 DMAs per wave per step)
constexpr int HP_WSLOT = HP_WROWS * ROW_BYTES;     // 16 KiB
constexpr int HP_WRING = 3;

__device__ __forceinline__ uint32_t hp_off(int row, int chunk) { return (uint32_t)(row * ROW_BYTES + ((chunk ^ (row & 7)) << 4)); }

inline int hp_xrows(int W) { return (HP_BM + 2 * W + 2 + 1 + 63) / 64 * 64; }      // +1: zero row
inline size_t hp_lds_bytes(int W) { return (size_t)2 * hp_xrows(W) * ROW_BYTES + HP_WRING * HP_WSLOT; }

template <typename T, int WC, int TC>
__global__ __launch_bounds__(512) void conv_halop_kernel(const ConvArgs a, const int XRB) {
  constexpr int CE = DT<T>::CE;
  constexpr int CC = 8 * CE;                       // channels per chunk: 64 f16 / 32 f32 (one 128-B row)
  constexpr int WP = 8 / WC;
  constexpr int TPW = HP_BM / (WP * 16);
  constexpr int BN = WC * TC * 16;
  static_assert(BN <= HP_WROWS, "channel tile exceeds the weight ring slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WC, wc = wave % WC;
  const int W = a.Win, H = a.Hin, HW = H * W;
  const int cin = a.src[0].ch_cnt;
  const int nchunk = (cin + CC - 1) / CC;
  const int R = HP_BM + 2 * W + 2;                 // halo rows that are real pixels
  const int xbytes = XRB * ROW_BYTES;
  const int npieces = XRB / 64;                    // halo DMAs per wave per chunk (<= 7)

  const int NB = (a.cout + BN - 1) / BN;
  const int MB = (a.M + HP_BM - 1) / HP_BM;
  const int ntiles = MB * NB;
  const int G = gridDim.x;
  const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < ntiles) ? (ntiles - first + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int steps_per_tile = nchunk * 9;
  const int total_steps = my_tiles * steps_per_tile;

  const v4i_t rsx = make_srd(a.src[0].ptr, a.src[0].bytes);
  const v4i_t rsw = make_srd(a.w, a.wbytes);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t wring = 2 * xbytes;
  constexpr uint32_t kOob = 0x80000000u;

  // ---- DMA lane geometry: one instruction = 8 rows x 128 B; lane -> row (lane>>3), slot lane&7 = chunk cg
  const int rsub = lane >> 3;
  const int cg = (lane & 7) ^ rsub;
  const int ld = a.src[0].ld;
  const int32_t x_lane = ((wave * 8 + rsub - W - 1) * ld + a.src[0].ch_off + cg * CE) * (int)sizeof(T);
  const int32_t x_piece = 64 * ld * (int)sizeof(T);          // 8 waves x 8 rows further
  const int32_t x_tile = HP_BM * ld * (int)sizeof(T);        // per pixel tile (m0 = mb * 256)

  auto issue_x = [&](int tile, int chunk, int piece, int buf) {
    const int mb = tile / NB;
    const int hr = piece * 64 + wave * 8 + rsub;               // halo row this lane fills
    const uint32_t ok = ((uint32_t)(hr - R) >> 31) & ((uint32_t)((chunk * CC + cg * CE) - cin) >> 31);
    const uint32_t off = (uint32_t)(x_lane + mb * x_tile + piece * x_piece + chunk * ROW_BYTES) | ((ok ^ 1u) << 31);
    lds_dma16(rsx, lds_base + (uint32_t)(buf * xbytes + (piece * 64 + wave * 8) * ROW_BYTES), off);
  };
  auto issue_w = [&](int tile, int chunk, int tap, int slot) {
    const int nb = tile - (tile / NB) * NB;
    const uint32_t chok = (uint32_t)((chunk * CC + cg * CE) - cin) >> 31;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + 8 * i) * 8 + rsub, n = nb * BN + row;
      const uint32_t ok = chok & ((uint32_t)(row - BN) >> 31) & ((uint32_t)(n - a.cout) >> 31);
      const uint32_t off = (uint32_t)((n * a.kpad + tap * cin + chunk * CC + cg * CE) * (int)sizeof(T)) | ((ok ^ 1u) << 31);
      lds_dma16(rsw, lds_base + wring + (uint32_t)(slot * HP_WSLOT + (wave + 8 * i) * 1024), off);
    }
  };

  // ---- fragment geometry
  const int frow = lane & 15, fq = lane >> 4;
  uint32_t tmask[TPW];
  int mloc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) mloc[j] = (wp * TPW + j) * 16 + frow;
  auto setup_masks = [&](int tile) {
    const int m0 = (tile / NB) * HP_BM;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int m = m0 + mloc[j];
      const bool vm = m < a.M;
      const uint32_t mm = vm ? (uint32_t)m : 0u;
      const uint32_t b = magic_div(mm, a.mg_hw_mul, a.mg_hw_shift);
      const uint32_t rem = mm - b * (uint32_t)HW;
      const int h = (int)magic_div(rem, a.mg_w_mul, a.mg_w_shift);
      const int w = (int)rem - h * W;
      const uint32_t hm = (h > 0 ? 1u : 0u) | 2u | ((h + 1 < H) ? 4u : 0u);
      const uint32_t wm = (w > 0 ? 1u : 0u) | 2u | ((w + 1 < W) ? 4u : 0u);
      const uint32_t msk = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? (wm << 3) : 0u) | ((hm & 4u) ? (wm << 6) : 0u);
      tmask[j] = vm ? msk : 0u;
    }
  };
  uint32_t aoff[TC];
#pragma unroll
  for (int i = 0; i < TC; ++i) aoff[i] = (uint32_t)(((wc * TC + i) * 16 + frow) * ROW_BYTES);
  const int arow7 = frow & 7;                                   // (row & 7) of every weight fragment row
  const uint32_t zoff = (uint32_t)((XRB - 1) * ROW_BYTES);      // zero row (halo rows >= R are DMA'd as zeros)

  f32x4 acc[TC][TPW];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- stream cursors (wave-uniform): compute position c, weight DMA position w = c + 2 steps
  int c_tile = first, c_chunk = 0, c_tap = 0, c_buf = 0, c_wslot = 0;
  int w_tile = first, w_chunk = 0, w_tap = 0, w_slot = 0, w_issued = 0;
  auto w_advance = [&]() {
    w_slot = (w_slot == HP_WRING - 1) ? 0 : w_slot + 1;
    ++w_issued;
    if (++w_tap == 9) { w_tap = 0; if (++w_chunk == nchunk) { w_chunk = 0; w_tile += G; } }
  };

  // prologue: whole halo of the first chunk, weights of steps 0 and 1
  for (int p = 0; p < npieces; ++p) issue_x(c_tile, 0, p, 0);
  issue_w(w_tile, w_chunk, w_tap, w_slot); w_advance();
  issue_w(w_tile, w_chunk, w_tap, w_slot); w_advance();
  setup_masks(c_tile);

  bool prev_x = false;
  for (int g = 0; g < total_steps; ++g) {
    // W(g) [and, at tap 0, this chunk's halo] must have landed; W(g+1) (2 DMAs) and the halo piece issued in
    // step g-1 (1 DMA, older than W(g+1)) may stay in flight
    const int allow = ((g + 1 < total_steps) ? 2 : 0) + ((prev_x && c_tap != 0) ? 1 : 0);
    if (allow == 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
    else if (allow == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    else if (allow == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // halo piece c_tap of the NEXT chunk (next tile's first chunk after the last chunk), then W(g+2)
    {
      int n_tile = c_tile, n_chunk = c_chunk + 1;
      if (n_chunk == nchunk) { n_chunk = 0; n_tile += G; }
      prev_x = (c_tap < npieces) && (n_tile < ntiles);
      if (prev_x) issue_x(n_tile, n_chunk, c_tap, c_buf ^ 1);
    }
    if (w_issued < total_steps) { issue_w(w_tile, w_chunk, w_tap, w_slot); w_advance(); }

    // ---- MFMAs of step g = (c_tile, c_chunk, c_tap)
    {
      const unsigned char* xs = smem + c_buf * xbytes;
      const unsigned char* ws = smem + wring + c_wslot * HP_WSLOT;
      const int shift = (c_tap / 3) * W + (c_tap % 3);
      const int kkn = (cin - c_chunk * CC > 4 * CE) ? 2 : 1;
      uint32_t brow[TPW];       // byte offset of the halo row, or of the zero row
      int b7[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int hr = mloc[j] + shift;
        const bool ok = (tmask[j] >> c_tap) & 1u;
        brow[j] = ok ? (uint32_t)(hr * ROW_BYTES) : zoff;
        b7[j] = ok ? (hr & 7) : 0;
      }
      for (int kk = 0; kk < kkn; ++kk) {
        uint4 af[TC], bf[TPW];
        const int ch = kk * 4 + fq;
#pragma unroll
        for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + aoff[i] + ((ch ^ arow7) << 4));
#pragma unroll
        for (int j = 0; j < TPW; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + brow[j] + ((ch ^ b7[j]) << 4));
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
      }
    }
    c_wslot = (c_wslot == HP_WRING - 1) ? 0 : c_wslot + 1;

    // ---- advance the compute cursor; epilogue at the end of a tile
    if (++c_tap == 9) {
      c_tap = 0;
      c_buf ^= 1;
      if (++c_chunk == nchunk) {
        c_chunk = 0;
        const int mb = c_tile / NB, nb = c_tile - mb * NB;
        const int m0 = mb * HP_BM, n0 = nb * BN;
        const float* __restrict__ bias = a.bias;
        auto run_epilogue = [&](auto outf32_tag) {
          constexpr bool OUTF32 = decltype(outf32_tag)::value;
#pragma unroll
          for (int i = 0; i < TC; ++i) {
            const int nt = __builtin_amdgcn_readfirstlane(n0 + (wc * TC + i) * 16);
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
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
              const v4ie_t rv = a.res ? epilogue_res_load<T>(a, rres, m0 + mloc[j], n) : (v4ie_t){0, 0, 0, 0};
              epilogue_fast<T, OUTF32>(a, rdst, m0 + mloc[j], n, acc[i][j], bv, rv);
              acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        };
        if (a.out_f32) run_epilogue(std::true_type{}); else run_epilogue(std::false_type{});
        c_tile += G;
        if (c_tile < ntiles) setup_masks(c_tile);
      }
    }
  }
}

inline bool halop_eligible(const ConvArgs& a) {
  return a.ksize == 3 && a.stride == 1 && a.nsrc == 1 && !a.src[0].up && a.vec_ok && a.Win <= 95 &&
         hp_lds_bytes(a.Win) <= 160 * 1024;
}

inline ConvCfg pick_halop_cfg(int cout, long M) {
  static const ConvCfg cands[] = {{2, 4}, {2, 3}, {1, 4}, {1, 3}, {1, 2}, {1, 1}};
  ConvCfg best = {1, 1};
  double best_cost = 1e30;
  for (const ConvCfg& c : cands) {
    const int bn = c.wc * c.tc * 16;
    const long nb = (cout + bn - 1) / bn, mbk = (M + HP_BM - 1) / HP_BM;
    double cost = (double)(nb * bn) * (double)(mbk * HP_BM);
    if (nb * mbk < 256) cost *= 1.0 + 0.25 * (256.0 / (double)(nb * mbk) - 1.0);
    cost *= 1.0 + 0.05 * (128.0 / bn);
    if (c.wc == 1) cost *= 1.10;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <typename T, int WC, int TC>
inline hipError_t launch_halop_cfg(const ConvArgs& a, hipStream_t s, int ncu) {
  constexpr int BN = WC * TC * 16;
  const int XRB = hp_xrows(a.Win);
  const size_t lds = hp_lds_bytes(a.Win);
  const long mbk = ((long)a.M + HP_BM - 1) / HP_BM, nb = (a.cout + BN - 1) / BN;
  long grid = std::min<long>(mbk * nb, ncu);
  grid = (grid + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_halop_kernel<T, WC, TC>), dim3((unsigned)grid), dim3(512), lds, s, a, XRB);
  return hipGetLastError();
}

template <typename T>
inline hipError_t launch_conv_halop(const ConvArgs& a, hipStream_t s, int ncu, int force_wc = 0, int force_tc = 0) {
  ConvCfg c = pick_halop_cfg(a.cout, a.M);
  if (force_wc > 0 && force_tc > 0 && force_wc * force_tc * 16 <= HP_WROWS) c = {force_wc, force_tc};
  if (c.wc == 2 && c.tc == 4) return launch_halop_cfg<T, 2, 4>(a, s, ncu);
  if (c.wc == 2 && c.tc == 3) return launch_halop_cfg<T, 2, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 4) return launch_halop_cfg<T, 1, 4>(a, s, ncu);
  if (c.wc == 1 && c.tc == 3) return launch_halop_cfg<T, 1, 3>(a, s, ncu);
  if (c.wc == 1 && c.tc == 2) return launch_halop_cfg<T, 1, 2>(a, s, ncu);
  return launch_halop_cfg<T, 1, 1>(a, s, ncu);
}

}  // namespace miyolo

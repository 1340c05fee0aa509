// 3x3 stride-1 convolution for NARROW layers (cin, cout <= 64) on 2-D pixel tiles, weights resident in LDS.
//
// Why: profiles/r01_per_layer_f16.md - the 48->48 3x3 layers at 160x160 are HBM-bound by the layer-wise roofline
// (216 FLOP per compulsory byte) but ran at 0.18 of it: the ring kernels re-fetch the activation tile once per tap and
// the weight tile once per pixel tile, 1.06 KB of L2 -> LDS traffic per output pixel against 192 B of HBM traffic, and
// the fill (its DMA instructions), not the matrix pipe, is what bounds them.  Here a persistent workgroup
//   * loads the layer's WHOLE weight matrix into LDS once (48 x 432 halves = 42 KiB),
//   * per 16 x 16 pixel tile fetches the 18 x 18 halo tile ONCE (all channels, zero-filled outside the image by the
//     LDS-DMA's out-of-range rule), double-buffered against the MFMAs of the previous tile,
//   * runs the flattened K = (tap, channel) loop entirely out of LDS with NO barrier inside it: a 16-pixel MFMA column
//     tile is 16 consecutive x of one tile row, so a tap is a constant address offset - no masks, no per-tap setup.
// L2 -> LDS traffic drops to ~140 B per output pixel; one s_barrier per tile.
// LDS rows are padded to an odd number of 16-byte slots (pixel rows and weight rows), which spreads the 16 rows of a
// ds_read_b128 fragment over 16 distinct bank slots.  Accumulation order = conv_dmap.h's (flattened K ascending), so the
// exact-fp32 mode is bit-identical to the other kernels.  Epilogue shared with them (common.h).
#pragma once
#include <type_traits>

#include "common.h"
#include "conv_dmap.h"

namespace miyolo {

struct T2dGeom {
  int32_t tiles_x, tiles_y, ntiles;      // 16 x 16 tiles per image row / column, total over the batch
  uint32_t mg_img_mul, mg_img_shift;     // magic division by tiles_x * tiles_y
  uint32_t mg_tx_mul, mg_tx_shift;       // magic division by tiles_x
  uint32_t mg_spp_mul, mg_spp_shift;     // magic division by slots per halo pixel
  int32_t cpt;                           // 16-byte chunks per tap (cin / CE)
  int32_t nch, ng;                       // chunks of the flattened K axis, MFMA k-groups (4 chunks each)
  int32_t wrow, xrow, spp;               // LDS bytes per weight row / halo pixel, 16-byte slots per halo pixel
  int32_t nslots, ndw;                   // 16-byte slots of a halo tile, DMAs per wave per tile
  int32_t xbuf_bytes;                    // one halo buffer = 8 * ndw KiB
};

constexpr int kT2dHalo = 18;             // 16 + 2
constexpr int kT2dMaxNdw = 6;

template <typename T, int TC>
__global__ __launch_bounds__(512) void conv_t2d_kernel(const ConvArgs a, const T2dGeom g) {
  constexpr int CE = DT<T>::CE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;

  // LDS: weights [TC*16][wrow] | koff table [ng*4] | halo buffer 0 | halo buffer 1
  unsigned char* const wl = smem;
  int32_t* const koff = reinterpret_cast<int32_t*>(smem + TC * 16 * g.wrow);
  const int x_base = TC * 16 * g.wrow + ((g.ng * 16 + 1023) & ~1023);
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

  const int G = gridDim.x;
  const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int my_tiles = (first < g.ntiles) ? (g.ntiles - first + G - 1) / G : 0;
  if (my_tiles == 0) return;

  // ---- weights -> LDS (once per workgroup); rows beyond cout are zero
  {
    const int cpr = g.ng * 4;                              // 16-byte chunks per weight row kept
    const int total = TC * 16 * cpr;
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w);
    const size_t row_bytes = (size_t)a.kpad * sizeof(T);
    for (int e = tid; e < total; e += 512) {
      const int n = e / cpr, c = e - n * cpr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < a.cout && (size_t)(c + 1) * 16 <= row_bytes) v = *reinterpret_cast<const uint4*>(wg + (size_t)n * row_bytes + c * 16);
      *reinterpret_cast<uint4*>(wl + n * g.wrow + c * 16) = v;
    }
    // per-chunk B-operand offset of the flattened K axis: chunk q = (tap, channel chunk) -> (dy*18 + dx) pixels + channel bytes
    for (int q = tid; q < g.ng * 4; q += 512) {
      int v = 0;
      if (q < g.nch) {
        const int tap = q / g.cpt, co = q - tap * g.cpt;
        v = ((tap / 3 - 1) * kT2dHalo + (tap % 3 - 1)) * g.xrow + co * 16;
      }
      koff[q] = v;
    }
  }

  // ---- DMA-side per-lane state, tile independent: slot s of the halo tile -> (halo pixel, chunk)
  const v4i_t rs0 = make_srd(a.src[0].ptr, a.src[0].bytes);
  const int ldB = a.src[0].ld * (int)sizeof(T);
  const int cinB16 = (a.cin * (int)sizeof(T)) / 16;        // real chunks per pixel (the rest of a row is padding)
  int32_t rel[kT2dMaxNdw];                                 // byte offset relative to the tile's first pixel
  uint32_t edge[kT2dMaxNdw];                               // bit0 top halo row, bit1 bottom, bit2 left, bit3 right, bit31 never valid
#pragma unroll
  for (int d = 0; d < kT2dMaxNdw; ++d) {
    const uint32_t s = (uint32_t)((wave * g.ndw + d) * 64 + lane);
    const uint32_t p = magic_div(s, g.mg_spp_mul, g.mg_spp_shift);
    const int c = (int)(s - p * (uint32_t)g.spp);
    const int hy = (int)(p / kT2dHalo), hx = (int)(p % kT2dHalo);
    const bool ok = d < g.ndw && (int)s < g.nslots && c < cinB16;
    rel[d] = ((hy - 1) * a.Win + (hx - 1)) * ldB + c * 16;
    edge[d] = ok ? ((hy == 0 ? 1u : 0u) | (hy == kT2dHalo - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == kT2dHalo - 1 ? 8u : 0u)) : 0x80000000u;
  }
  auto tile_coords = [&](int tile, int* b, int* ty, int* tx) {
    const uint32_t bb = magic_div((uint32_t)tile, g.mg_img_mul, g.mg_img_shift);
    const uint32_t r = (uint32_t)tile - bb * (uint32_t)(g.tiles_x * g.tiles_y);
    const uint32_t yy = magic_div(r, g.mg_tx_mul, g.mg_tx_shift);
    *b = (int)bb; *ty = (int)yy; *tx = (int)(r - yy * (uint32_t)g.tiles_x);
  };
  auto issue_tile = [&](int tile, int buf) {
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
    const uint32_t tflags = (ty == 0 ? 1u : 0u) | (ty == g.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == g.tiles_x - 1 ? 8u : 0u) | 0x80000000u;
    const int32_t origin = ((b * a.Hin + ty * 16) * a.Win + tx * 16) * ldB + a.src[0].ch_off * (int)sizeof(T);
    const uint32_t st = lds_base + (uint32_t)(x_base + buf * g.xbuf_bytes + wave * g.ndw * 1024);
#pragma unroll
    for (int d = 0; d < kT2dMaxNdw; ++d) {
      if (d < g.ndw) {                                                       // wave-uniform
        const uint32_t inv = (edge[d] & tflags) ? 0x80000000u : 0u;          // outside the image / padding slot -> zeros
        lds_dma16(rs0, st + d * 1024, (uint32_t)(origin + rel[d]) | inv);
      }
    }
  };

  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.dst), 0, a.res ? a.res_bytes : 0u, 0x00020000);

  f32x4 acc[TC][2];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment bases: this wave owns tile rows 2*wave, 2*wave + 1 (32 pixels) x all channels
  uint32_t abase[TC], bbase[2];
#pragma unroll
  for (int i = 0; i < TC; ++i) abase[i] = (uint32_t)((i * 16 + frow) * g.wrow + fq * 16);
#pragma unroll
  for (int j = 0; j < 2; ++j) bbase[j] = (uint32_t)(((2 * wave + j + 1) * kT2dHalo + frow + 1) * g.xrow);

  issue_tile(first, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                          // weights + table + tile 0 in LDS

  int tile = first;
  for (int t = 0; t < my_tiles; ++t, tile += G) {
    const int buf = t & 1;
    if (t > 0) asm volatile("s_barrier" ::: "memory");      // tile t landed (every wave waited its DMAs before its epilogue) and
                                                            // buffer buf^1 is no longer read (everyone finished the MFMAs of t-1)
    if (t + 1 < my_tiles) issue_tile(tile + G, buf ^ 1);
    int b, ty, tx;
    tile_coords(tile, &b, &ty, &tx);
    // residual operand of this tile: loaded NOW, consumed after the K loop (its latency hides under the MFMAs; the
    // vmcnt(0) below covers these loads together with the DMAs)
    v4ie_t rv[TC][2];
    if (a.res) {
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          rv[i][j] = epilogue_res_load<T>(a, rres, (b * a.Hout + ty * 16 + 2 * wave + j) * a.Wout + tx * 16 + frow, i * 16 + fq * 4);
    } else {
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) rv[i][j] = (v4ie_t){0, 0, 0, 0};
    }
    const unsigned char* xs = smem + x_base + buf * g.xbuf_bytes;
#pragma unroll 2
    for (int kg = 0; kg < g.ng; ++kg) {
      const int ko = koff[kg * 4 + fq];
      uint4 af[TC], bf[2];
#pragma unroll
      for (int i = 0; i < TC; ++i) af[i] = *reinterpret_cast<const uint4*>(wl + abase[i] + kg * 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const uint4*>(xs + (int)bbase[j] + ko);
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's part of tile t+1 has landed (it had the whole K loop)

    // ---- epilogue of tile t
    const float* __restrict__ bias = a.bias;
    auto run_epilogue = [&](auto outf32_tag) {
      constexpr bool OUTF32 = decltype(outf32_tag)::value;
#pragma unroll
      for (int i = 0; i < TC; ++i) {
        const int nt = i * 16;
        const int n = nt + fq * 4;
        v4i_t s0, s1, s2, s3;
        const float* bp = sgpr_ptr(bias + nt);
        asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\t"
                     "s_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(bp));
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bv[r] = __int_as_float(fq == 0 ? s0[r] : fq == 1 ? s1[r] : fq == 2 ? s2[r] : s3[r]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int m = (b * a.Hout + ty * 16 + 2 * wave + j) * a.Wout + tx * 16 + frow;
          epilogue_fast<T, OUTF32>(a, rdst, m, n, acc[i][j], bv, rv[i][j]);
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    if (a.out_f32) run_epilogue(std::true_type{}); else run_epilogue(std::false_type{});
  }
}

// host side ------------------------------------------------------------------------------------------------------
inline bool t2d_shape(int cin, int cout, int B, int Ho, int Wo, size_t elem, int ce, T2dGeom* g, size_t* lds) {
  if (cin > 64 || cout > 64 || cin % ce || (Ho % 16) || (Wo % 16)) return false;
  const int tc = (cout + 15) / 16;
  g->tiles_x = Wo / 16; g->tiles_y = Ho / 16; g->ntiles = B * g->tiles_x * g->tiles_y;
  g->cpt = cin / ce; g->nch = 9 * g->cpt; g->ng = (g->nch + 3) / 4;
  g->wrow = g->ng * 64 + 16;
  const int cinb = cin * (int)elem;
  g->xrow = cinb + (((cinb / 16) % 2 == 0) ? 16 : 0);
  g->spp = g->xrow / 16;
  g->nslots = kT2dHalo * kT2dHalo * g->spp;
  const int ndma = (g->nslots + 63) / 64;
  g->ndw = (ndma + 7) / 8;
  if (g->ndw > kT2dMaxNdw) return false;
  g->xbuf_bytes = 8 * g->ndw * 1024;
  host_magic((uint32_t)(g->tiles_x * g->tiles_y), &g->mg_img_mul, &g->mg_img_shift);
  host_magic((uint32_t)g->tiles_x, &g->mg_tx_mul, &g->mg_tx_shift);
  host_magic((uint32_t)g->spp, &g->mg_spp_mul, &g->mg_spp_shift);
  *lds = (size_t)tc * 16 * g->wrow + (((size_t)g->ng * 16 + 1023) & ~(size_t)1023) + 2 * (size_t)g->xbuf_bytes;
  return *lds <= 160 * 1024;
}

inline bool t2d_geometry(const ConvArgs& a, size_t elem, int ce, T2dGeom* g, size_t* lds) {
  if (a.ksize != 3 || a.stride != 1 || a.nsrc != 1 || a.src[0].up || !a.vec_ok) return false;
  if (a.Hin != a.Hout || a.Win != a.Wout) return false;
  if ((a.src[0].ch_off * elem) % 16 || (a.src[0].ld * elem) % 16) return false;
  return t2d_shape(a.cin, a.cout, a.B, a.Hout, a.Wout, elem, ce, g, lds);
}

template <typename T>
inline bool t2d_eligible(const ConvArgs& a) {
  T2dGeom g; size_t lds;
  return t2d_geometry(a, sizeof(T), DT<T>::CE, &g, &lds);
}

template <typename T>
inline hipError_t launch_conv_t2d(const ConvArgs& a, hipStream_t s, int ncu) {
  T2dGeom g; size_t lds;
  if (!t2d_geometry(a, sizeof(T), DT<T>::CE, &g, &lds)) return hipErrorInvalidValue;
  long grid = std::min<long>(g.ntiles, ncu);
  grid = (grid + 7) / 8 * 8;
  switch ((a.cout + 15) / 16) {
    case 1: hipLaunchKernelGGL((conv_t2d_kernel<T, 1>), dim3((unsigned)grid), dim3(512), lds, s, a, g); break;
    case 2: hipLaunchKernelGGL((conv_t2d_kernel<T, 2>), dim3((unsigned)grid), dim3(512), lds, s, a, g); break;
    case 3: hipLaunchKernelGGL((conv_t2d_kernel<T, 3>), dim3((unsigned)grid), dim3(512), lds, s, a, g); break;
    default: hipLaunchKernelGGL((conv_t2d_kernel<T, 4>), dim3((unsigned)grid), dim3(512), lds, s, a, g); break;
  }
  return hipGetLastError();
}

}  // namespace miyolo

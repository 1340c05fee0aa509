// 1x1 convolution (pointwise: C2f cv1 / cv2, SPPF cv1 / cv2, the neck's convs over a concat, [3P] ultralytics Conv with
// k = 1 - SURVEY.md 8a a5/a6/a8), f16: STREAMING kernel - activations never touch the LDS.  EXPERIMENT OF ROUND 3, OFF BY
// DEFAULT (engine option "pw" = 1 selects it): correct and bit-identical to the ring kernel, but 6-8 % SLOWER on the sum
// of yolov8m's eighteen 1x1 layers (1 839-1 885 us vs 1 729 us per step, profiles/r03_pw_experiment.md).
//
// The idea: the ring kernel (conv_dmap.h) runs these layers at 3.3-4.9 TB/s of compulsory traffic where a
// plain elementwise kernel gets 6.2 TB/s on the same box (tools/probe_hbm_rates.py).  Its activation tile goes through a
// ring of LDS slots, and a 256 px x 192 ch tile leaves room for two: one K step (32 KiB of activations per CU) is in flight
// while the other is computed on.  Here
//   * a wave loads its OWN pixels' MFMA B fragments straight from global memory into registers - for a 1x1 conv the
//     fragment layout (pixel = lane & 15, 16 bytes of K at (lane >> 4) * 16) is the NHWC layout - two groups of two K steps,
//     16 KiB per wave in flight across tile boundaries;
//   * only the WEIGHTS go through the LDS (a 3-slot ring of BN x 128 B), fetched by a PRODUCER wave with LDS-DMA.
//     A wave's vector-memory operations retire in order, so a consumer that also waited for (short-latency) weight loads
//     would drag its (long-latency) pixel loads in with them.  The producer wave has its own counter; consumers see weights
//     only through the per-step barrier; the bias sits in the LDS for the same reason;
//   * a wave owns 32 pixels x ALL BN channels of the workgroup; channel tiles (N = 384, 576) are neighbours on one XCD;
//   * pixels are dealt to workgroups as balanced contiguous RANGES (not whole tiles).
// What the measurements say (same file): every variant - one 8-wave or two 4-wave workgroups per CU, fragments re-filled one
// at a time (64-byte pieces) or eight at a time (256-byte pieces), six weight fragments read ahead of their MFMAs or two,
// stores one by one or a pixel row at a time, workgroups started in step or staggered - lands within 3 % of 1 850 us.
// Ablations (timing builds, wrong results): without the pixel loads 1 160 us, without MFMAs 1 611, without weight DMAs 1 724,
// with none of the three (barriers, epilogue, stores) 575 - the parts ADD UP instead of overlapping, although 16 KiB per
// wave (112 KiB per CU, 28 MB on the chip) are in flight: by Little's law the memory system answers this request stream
// with ~5 us of latency at 3.8 TB/s, where it gives an elementwise kernel 6.2 TB/s.  One contribution is the access pattern
// the MFMA layout forces on a direct load - the 16 lanes of a quarter wave are 16 different PIXELS, 16 bytes from each of 16
// rows per quarter, where the ring kernel's LDS-DMA reads 8 rows x 128 contiguous bytes per instruction: with every
// instruction reading 1 KiB contiguous bytes instead (same bytes, wrong results) the sum drops 6 % (1 853 -> 1 747 us).  The
// rest is not explained; the ring kernel stays.
// Same MFMA, same flattened K order (source 0's channels, then source 1's; 64 per step, 32 per MFMA), same epilogue
// arithmetic (acc + bias, v_exp / v_rcp SiLU, f16 rounding) as conv_dmap.h: results are BIT-IDENTICAL to it
// (tests/test_gpu_conv.py::test_pw_*).  All consumer-side memory operations are compiler-visible loads (exact s_waitcnt
// by the compiler); the K loop is unrolled four steps deep so that every register buffer has a compile-time name, and a
// tile's K steps are padded to a multiple of four with void steps (out-of-range loads: zeros, no traffic; no MFMAs).
#pragma once
#include "common.h"
#include "conv_dma.h"
#include "conv_dmap.h"
#include "conv_igemm.h"

namespace miyolo {

// consumer waves per workgroup (the next wave is the weight producer): 7 = one 8-wave workgroup per CU, 3 = two 4-wave ones
#ifndef MIYOLO_PW_CW
#define MIYOLO_PW_CW 7
#endif
constexpr int kPwConsumers = MIYOLO_PW_CW;
constexpr int kPwTile = kPwConsumers * 32;        // pixels per workgroup tile
constexpr int kPwThreads = (kPwConsumers + 1) * 64;
constexpr int kPwWgPerCu = kPwConsumers == 7 ? 1 : 2;
constexpr int kPwSlots = 3;                       // weight ring (the producer runs two steps ahead)

struct PwGeom {
  int32_t nnt;            // channel tiles (cout / BN)
  int32_t gpx;            // pixel ranges per XCD (= workgroups per XCD / nnt)
  int32_t R;              // pixels per range (multiple of 16)
  int32_t S, Sp;          // K steps of 64 channels; padded to a multiple of 4
  int32_t ct0, ct1;       // channels of source 0 / 1 (ct1 = 0: one source)
  int32_t any_up;         // a source is read through the nearest x2 upsample
};

template <int NT>
__global__ __launch_bounds__(kPwThreads, kPwWgPerCu) void conv_pw_kernel(const ConvArgs a, const PwGeom g) {
  constexpr int BN = NT * 16, SLOT = BN * ROW_BYTES, NPAIR = NT / 2, ND = BN / 8;    // ND: weight DMAs (8 rows each) per step
  static_assert(NT % 2 == 0, "channel tiles are stored in pairs (8 channels per lane)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kOob = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;

  // ---- this workgroup's pixel range and channel tile: workgroups b, b + 8, ... share an XCD; nnt neighbours there share a range
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const int gi = wi / g.nnt, nt = wi - gi * g.nnt;
  if (gi >= g.gpx) return;
  const int m_begin = (xcd * g.gpx + gi) * g.R;
  const int m_end = min(a.M, m_begin + g.R);
  if (m_begin >= m_end) return;
  const int T = (m_end - m_begin + kPwTile - 1) / kPwTile;
  const int total = T * g.Sp;                                  // barriers every wave of this workgroup takes
  const int n0 = nt * BN;
  const uint32_t lds_base = (uint32_t)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

  if (wave == kPwConsumers) {
    // ================= producer: W(step) -> slot step % 3, two steps ahead of the consumers =================
    const v4i_t rsw = make_srd(a.w, a.wbytes);
    uint32_t woff[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {                             // LDS row r holds channel n0 + pi(r), the pair deal of conv_h2.h
      const int r = 8 * d + (lane >> 3);
      const int ti = r >> 4, rho = r & 15;
      const int ch = 32 * (ti >> 1) + 8 * (rho >> 2) + 4 * (ti & 1) + (rho & 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);               // source-side swizzle: position lane & 7 of the row holds chunk c
      woff[d] = (n0 + ch < a.cout) ? (uint32_t)((n0 + ch) * a.kpad * 2 + c * 16) : kOob;
    }
    auto issue_w = [&](int ks, int slot, bool live) {
      const uint32_t st = lds_base + (uint32_t)(slot * SLOT);
      const uint32_t inv = (live && ks < g.S) ? 0u : kOob;
      if (ABL(1)) return;                                      // timing experiment: no weight DMAs (results wrong)
#pragma unroll
      for (int d = 0; d < ND; ++d) lds_dma16(rsw, st + d * 1024, (woff[d] + (uint32_t)(ks * 128)) | inv);
    };
    int ks = 0, slot = 0;                                      // of the next group to issue
    auto advance = [&]() { ks = (ks + 1 == g.Sp) ? 0 : ks + 1; slot = (slot + 1 == kPwSlots) ? 0 : slot + 1; };
    issue_w(ks, slot, true); advance();
    issue_w(ks, slot, 1 < total); advance();
    asm volatile("s_barrier" ::: "memory");                    // the consumers' bias table (below)
    for (int gstep = 0; gstep < total; ++gstep) {
      // W(gstep) has landed when at most the ND DMAs of W(gstep + 1) are outstanding (always issued, void ones as
      // out-of-range DMAs, so that the count is uniform)
      if constexpr (ND == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");                  // consumers: done with slot (gstep - 1) % 3 = (gstep + 2) % 3
      issue_w(ks, slot, gstep + 2 < total); advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no DMA may land in the LDS of a workgroup that has left
    return;
  }

  // ================= consumers =================
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[0].ptr), 0, a.src[0].bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[1].ptr), 0, a.src[1].bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst_bytes, 0x00020000);
  const int HWo = a.Hout * a.Wout;
  // The bias goes to the LDS once: a bias LOAD in the epilogue would sit behind NB steps of pixel loads in the wave's in-order
  // vector-memory queue and wait for all of them (the compiler put s_waitcnt vmcnt(0) there: one HBM latency per tile).
  float* const bias_l = reinterpret_cast<float*>(smem + kPwSlots * SLOT);
  if (tid < BN) bias_l[tid] = (n0 + tid < a.cout) ? a.bias[n0 + tid] : 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // per-lane byte offsets of the PREFETCH stream's tile: pixel (wave * 32 + mt * 16 + frow), this lane's 16-byte K column
  uint32_t poff[2][2];
  auto set_tile = [&](int t) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m_begin + t * kPwTile + wave * 32 + mt * 16 + frow;
      const bool vm = t < T && m < m_end;
      const uint32_t mm = vm ? (uint32_t)m : 0u;
      if (g.any_up) {
        const int b = (int)magic_div(mm, a.mg_hw_mul, a.mg_hw_shift);
        const uint32_t rem = mm - (uint32_t)b * (uint32_t)HWo;
        const int ho = (int)magic_div(rem, a.mg_w_mul, a.mg_w_shift);
        const int wo = (int)rem - ho * a.Wout;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int hh = a.src[s].up ? (ho >> 1) : ho, ww = a.src[s].up ? (wo >> 1) : wo;
          poff[mt][s] = vm ? (uint32_t)((((b * a.src[s].h + hh) * a.src[s].w + ww) * a.src[s].ld + a.src[s].ch_off) * 2 + fq * 16) : kOob;
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) poff[mt][s] = vm ? (uint32_t)((m * a.src[s].ld + a.src[s].ch_off) * 2 + fq * 16) : kOob;
      }
    }
  };

  // B fragments: two GROUPS of two K steps each, [group][step][pixel tile][kk].  A group's eight loads go out back to back,
  // pixel tile by pixel tile in K order - 256 contiguous bytes of each of 16 pixel rows within a few instructions: HBM gives
  // 2.9 / 4.1 / 5.0 TB/s to 64- / 128- / 256-byte pieces of a 1152-byte row fetched microseconds apart
  // (tools/probe_hbm_strided.py), and the first form of this kernel, which re-filled each 64-byte fragment as soon as its
  // MFMAs were out, was slower than the ring kernel's 128-byte pieces on every HBM-bound layer.
  v4ie_t pb[2][2][2][2];
  auto load_frag = [&](auto g_tag, auto st_tag, auto mt_tag, auto kk_tag, int ks) __attribute__((always_inline)) {
    constexpr int gi_ = decltype(g_tag)::value, st = decltype(st_tag)::value, mt = decltype(mt_tag)::value, kk = decltype(kk_tag)::value;
    const int k = ks * 64 + kk * 32;                           // first channel of this MFMA's K slice (wave-uniform)
    const bool s1 = k >= g.ct0;
    const int kl = s1 ? k - g.ct0 : k;
    const bool kv = ks < g.S && kl < (s1 ? g.ct1 : g.ct0);
    uint32_t off = kv ? poff[mt][s1 ? 1 : 0] + (uint32_t)(kl * 2) : kOob;
    // timing experiment (results wrong): the same bytes of a 16-pixel tile with full rows (ld == cin), but every instruction
    // reads 1 KiB CONTIGUOUS bytes instead of 64 bytes from each of 16 rows
    if (ABL(8) && kv) off = poff[mt][0] - (uint32_t)(fq * 16 + frow * a.src[0].ld * 2) + (uint32_t)((ks * 2 + kk) * 1024 + lane * 16);
    const __amdgpu_buffer_rsrc_t rs = s1 ? rs1 : rs0;
    pb[gi_][st][mt][kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
  };
  auto load_group = [&](auto g_tag, int ks0) __attribute__((always_inline)) {
    if (ABL(4)) return;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    load_frag(g_tag, I0{}, I0{}, I0{}, ks0); load_frag(g_tag, I0{}, I0{}, I1{}, ks0);
    load_frag(g_tag, I1{}, I0{}, I0{}, ks0 + 1); load_frag(g_tag, I1{}, I0{}, I1{}, ks0 + 1);
    load_frag(g_tag, I0{}, I1{}, I0{}, ks0); load_frag(g_tag, I0{}, I1{}, I1{}, ks0);
    load_frag(g_tag, I1{}, I1{}, I0{}, ks0 + 1); load_frag(g_tag, I1{}, I1{}, I1{}, ks0 + 1);
  };

  f32x4 acc[NT][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NT; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  };
  const uint32_t aaddr0 = lds_off(frow, fq), aaddr1 = lds_off(frow, 4 + fq);      // (row & 15, chunk) part: rows 16 i + frow share (row >> 1) & 7

  // one K step.  Six weight fragments are read up front, then their twelve MFMAs run (the sched_barriers keep it so: left
  // alone the scheduler reads TWO fragments, waits, issues their four MFMAs, reads the next two into the same registers ... -
  // one exposed LDS latency per 64 cycles of matrix work, and the K = 1152 layers ran at 2.7 x their MFMA time).
  auto step = [&](auto g_tag, auto st_tag, int slot, bool real) __attribute__((always_inline)) {
    constexpr int gi_ = decltype(g_tag)::value, st = decltype(st_tag)::value;
    if (!real || ABL(2)) return;                               // ABL: timing experiments (results wrong)
    const unsigned char* ws = smem + slot * SLOT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int hf = 0; hf < NT / 6; ++hf) {
        uint4 af[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) af[i] = *reinterpret_cast<const uint4*>(ws + (kk ? aaddr1 : aaddr0) + (hf * 6 + i) * 16 * ROW_BYTES);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          Mma<half_t>::run(af[i], *reinterpret_cast<const uint4*>(&pb[gi_][st][0][kk]), acc[hf * 6 + i][0]);
          Mma<half_t>::run(af[i], *reinterpret_cast<const uint4*>(&pb[gi_][st][1][kk]), acc[hf * 6 + i][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // Epilogue: per pixel tile, all NPAIR 16-byte stores of a lane go out back to back (a pixel's whole output row within a few
  // instructions) behind the activations of all its channels.
  auto epilogue = [&](int t) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m_begin + t * kPwTile + wave * 32 + mt * 16 + frow;
      v4ie_t ov[NPAIR];
#pragma unroll
      for (int ip = 0; ip < NPAIR; ++ip) {
        const v4ie_t b0 = *reinterpret_cast<const v4ie_t*>(bias_l + 32 * ip + 8 * fq);
        const v4ie_t b1 = *reinterpret_cast<const v4ie_t*>(bias_l + 32 * ip + 8 * fq + 4);
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x0 = acc[2 * ip][mt][r] + __int_as_float(b0[r]);
          float x1 = acc[2 * ip + 1][mt][r] + __int_as_float(b1[r]);
          if (a.act) { x0 = silu_fast(x0); x1 = silu_fast(x1); }
          v[r] = x0; v[4 + r] = x1;
        }
        const f16x8 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
        ov[ip] = *reinterpret_cast<const v4ie_t*>(&hv);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ip = 0; ip < NPAIR; ++ip) {
        const int n = n0 + 32 * ip + 8 * fq;
        const uint32_t so = (m < m_end && n < a.cout) ? (uint32_t)((m * a.dst_ld + a.dst_choff + n) * 2) : kOob;
        __builtin_amdgcn_raw_buffer_store_b128(ov[ip], rdst, so, 0, MIYOLO_ST_AUX);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: the first four steps of tile 0 in flight
  using G0 = std::integral_constant<int, 0>; using G1 = std::integral_constant<int, 1>;
  int pt = 0, pks = 0;                     // prefetch stream: tile, first K step of the super-step (4 steps) it is in
  set_tile(0);
  load_group(G0{}, 0);
  load_group(G1{}, 2);
  auto next_super = [&]() { pks += 4; if (pks == g.Sp) { pks = 0; ++pt; set_tile(pt); } };
  next_super();
  zero_acc();
  int ct = 0, cks = 0, slot = 0;           // compute stream: tile, first K step of the current super-step, weight slot
  const int nsuper = total / 4;
  auto next_slot = [&]() { slot = (slot + 1 == kPwSlots) ? 0 : slot + 1; };
#pragma unroll 1
  for (int u = 0; u < nsuper; ++u) {
    asm volatile("s_barrier" ::: "memory");
    step(G0{}, G0{}, slot, cks + 0 < g.S); next_slot();
    asm volatile("s_barrier" ::: "memory");
    step(G0{}, G1{}, slot, cks + 1 < g.S); next_slot();
    load_group(G0{}, pks);                 // steps 0, 1 of the NEXT super-step: two steps of flight
    asm volatile("s_barrier" ::: "memory");
    step(G1{}, G0{}, slot, cks + 2 < g.S); next_slot();
    asm volatile("s_barrier" ::: "memory");
    step(G1{}, G1{}, slot, cks + 3 < g.S); next_slot();
    load_group(G1{}, pks + 2);
    next_super();
    cks += 4;
    if (cks == g.Sp) { epilogue(ct); zero_acc(); cks = 0; ++ct; }
  }
}

// host side ------------------------------------------------------------------------------------------------------
inline bool pw_eligible(const ConvArgs& a) {
  if (a.ksize != 1 || a.stride != 1 || a.res || a.out_f32) return false;
  if (a.cout % 96) return false;
  if (a.nsrc == 2 && (a.src[0].ch_cnt % 64 || a.src[1].ch_cnt % 32)) return false;
  if (a.nsrc == 1 && a.src[0].ch_cnt % 32) return false;
  for (int s = 0; s < a.nsrc; ++s)
    if (a.src[s].ld % 8 || a.src[s].ch_off % 8) return false;              // 16-byte aligned fragment loads
  if (a.dst_ld % 8 || a.dst_choff % 8) return false;                       // 16-byte stores
  return true;
}
inline int pw_nt(int cout) { return cout % 192 == 0 ? 12 : 6; }

inline hipError_t launch_conv_pw(const ConvArgs& a, hipStream_t s, int ncu) {
  if (!pw_eligible(a)) return hipErrorInvalidValue;
  const int NT = pw_nt(a.cout), BN = NT * 16;
  PwGeom g;
  g.nnt = a.cout / BN;
  const int per_xcd = std::max(1, ncu * kPwWgPerCu / 8);
  g.gpx = per_xcd / g.nnt;
  if (g.gpx < 1) return hipErrorInvalidValue;
  const int groups = g.gpx * 8;
  g.R = (int)(((a.M + groups - 1) / groups + 15) / 16 * 16);
  g.S = (a.cin + 63) / 64; g.Sp = (g.S + 3) / 4 * 4;
  g.ct0 = a.src[0].ch_cnt; g.ct1 = a.nsrc == 2 ? a.src[1].ch_cnt : 0;
  g.any_up = (a.src[0].up || (a.nsrc == 2 && a.src[1].up)) ? 1 : 0;
  const unsigned grid = (unsigned)(per_xcd * 8);
  const size_t lds = (size_t)kPwSlots * BN * ROW_BYTES + (size_t)BN * 4;
  if (NT == 12) hipLaunchKernelGGL((conv_pw_kernel<12>), dim3(grid), dim3(kPwThreads), lds, s, a, g);
  else hipLaunchKernelGGL((conv_pw_kernel<6>), dim3(grid), dim3(kPwThreads), lds, s, a, g);
  return hipGetLastError();
}

inline hipError_t set_pw_attrs() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pw_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pw_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

}  // namespace miyolo

// The whole yolov8n-cls forward as ONE launch: one workgroup per image, every activation of the image resident in LDS.
//
// Why (BASELINE config 2, SURVEY.md 8d: "in practice launch-bound; use one persistent kernel"): the layered path runs the
// classifier (reference detect.py:121 `rank_model(crop)`) as 27 dependent launches of 1-3 us of work each, and the GPU's
// front end spends ~8.5 us per dependent dispatch: 0.23 ms per batch of 256 whatever the kernels do (hipGraph replay and
// multi-stream forks measured slower).  A 64 x 64 crop's activations are tiny - 32 KiB after the stem, 177 KiB
// summed over all 20 buffers, 72 KiB alive at once - so a workgroup keeps its image's buffers in LDS (the host packs them
// by liveness), walks the layer table and only streams the weights (2.9 MB f16, L2 resident) through registers:
//   * 8 waves; a conv's work items are (16-channel tile, group of up to four 16-pixel tiles) pairs dealt to the waves;
//     an item's K loop loads ONE weight fragment per 32-deep step straight from global/L2 (16 B per lane, eight
//     steps in flight) and uses it for up to four MFMAs, the activation fragments are 16-byte LDS reads addressed per lane
//     (3x3 taps, stride 2 and channel slices are address arithmetic; buffers read by a 3x3 conv carry a zero halo, and
//     pixels are padded by 16 bytes so that a fragment read is free of bank conflicts);
//   * same MFMA (16x16x32 f16), same K order (tap, channel), same epilogue arithmetic (acc + bias -> SiLU -> + residual ->
//     f16) as the layered f16 kernels: logits are BIT-IDENTICAL to the layered path (tests/test_gpu_cls.py);
//   * the Classify tail (avg-pool, Linear, softmax) runs in the same launch with the arithmetic order of cls_head_kernel.
// f16 only (config 2's dtype); the exact-fp32 parity mode and models that do not fit keep the layered path.
#pragma once
#ifndef MEGA_BURST_KIND
#define MEGA_BURST_KIND 0
#endif
#include "common.h"
#include "conv_igemm.h"
#include "conv_dma.h"

namespace miyolo {

struct MegaOp {
  int32_t kind, ksize, stride, act;
  int32_t cin, cout, kpad, lg_cpt;          // lg_cpt: log2(cin / 8) for 3x3 convs
  int32_t hout, wout, lg_wout;
  // LDS views: org = byte offset of pixel (0,0), first channel of the view; wp = pixels per stored row (width + 2 when the
  // buffer carries a zero halo); ps = bytes per stored pixel (channels * 2 + 16: an odd number of 16-byte chunks, so the 16
  // pixels of a fragment read land in 16 different bank groups instead of one)
  int32_t src_org, src_wp, src_ps;
  int32_t dst_org, dst_wp, dst_ps;
  int32_t res_org, res_wp, res_ps;          // res_org < 0: no residual
  int32_t zero_off, zero_rows;              // first writer of a haloed buffer clears its halo ring: buffer offset, stored rows
  int32_t wbytes;
  const void* w;                            // convs: fragment order [cout/16][kpad/32][lane 64][8] f16 (mega_repack_kernel), one
                                            // contiguous KiB per wave and K step; stem: [cout][32], K' order of kernels_misc.h
  const float* bias;
};

constexpr int kMegaMaxOps = 30;
constexpr int kMegaPf = 4;                  // weight fragments in flight per wave (ring slots; 4 KiB is what one M0 reaches)
constexpr int kMegaWaves = 8;               // 512 threads per image

struct MegaArgs {
  const uint8_t* in;                        // [B][H][W][3]
  float* logits; float* probs;              // [B][nc] (either may be null)
  const float* lin_w; const float* lin_b;   // Classify.linear
  int32_t B, H, W, nc, nops;
  int32_t feat_off, feat_c, feat_hw, feat_ps;   // the Classify conv's output in LDS (no halo)
  int32_t pool_off;                         // fp32 scratch for pooled features + logits
  int32_t ring_off;                         // kMegaWaves weight rings of kMegaPf KiB
  int32_t bias_off;                         // kMegaWaves x 2 bias slots of 256 B
  int32_t desc_off;                         // the layer table, copied to LDS once (reads of it never touch the scalar cache)
  const MegaOp* ops_dev;                    // the same table in device memory
  unsigned long long* stamps;               // option dbg_op: cycle counter after every layer, first 8 images x 32 slots
};

// A wave's weight stream: a private LDS ring of kMegaPf slots (1 KiB = 64 lanes x 16 B each) always holds the next kMegaPf K
// steps the wave will consume - of the current item, then of its next item, then of its first item of the NEXT layer - so
// layer boundaries, epilogues and barriers overlap the L2 latency.  The fills are LDS-DMA loads written as inline asm with
// hand-counted waits: every visit of a slot waits vmcnt(kMegaPf - 1) (the slot's own fill is the oldest in flight),
// optionally computes, and issues EXACTLY ONE new fill into the slot, so the count never changes.  Nothing else in a conv
// layer uses VMEM: the bias comes through the scalar cache.
//   Measured on the way here (batch 256): compiler-visible register loads made the waitcnt pass put vmcnt(0) in front of
//   every MFMA (600-2000 cycles per K step, 0.21-0.29 ms per batch); asm loads into registers are not usable, the register
//   allocator copies a ring register (loop phi) while its load is still in flight.
struct MegaStream {
  v4i_t rs, rsb;                            // weights, bias
  uint32_t arow;                            // byte offset of this lane's 16 bytes of step 0 (a step is 1 KiB further)
  uint32_t boff;                            // byte offset of this lane's bias word (lanes 0..15; the others out of range)
  int nsteps;                               // 0: no such item
};
constexpr uint32_t kMegaOob = 0x80000000u;  // offset past any weight tensor: the load returns zeros

// Fill of ring slot U.  M0 (the LDS base of an LDS-DMA) is the SAME value for every ring fill of a wave; the slot is chosen by
// the instruction's 12-bit offset, which the hardware adds to the LDS address AND to the memory address - the descriptors
// therefore start kMegaBack bytes early and the lane offset carries kMegaBack - 1024 U (4 slots of 1 KiB = what one M0
// reaches).
// THE HAZARD of this stream, established by replaying batches against the layered path (tools/stress_cls_mega.py; the
// experiment switches below are compile-time, e.g. `build.sh -DMEGA_BURST=4 -DMEGA_NO_SPACING`): the vmcnt retirement of an
// LDS-DMA does not mean that its bytes have LANDED in LDS; when the DMA write path is congested the landing lags, and a
// fill can land after a YOUNGER fill to the same slot (write-after-write inversion: the slot ends up holding the older
// entry).  Measurements, wrong images per 2.5 x 10^5:
//   * MEGA_BURST=4 (16 extra refill-only fills before every item boundary, duplicates of what the slots already hold, so
//     only their timing can matter), unspaced: 13-31;   the same + 512 idle cycles before the next item: 0;
//     the same + vmcnt(0) after every burst fill: 0;   the same + 32 cycles between burst fills: 0;
//     the same + stricter waits on the READ side: 26 (it is not the reads);   out-of-range (zero) burst fills: 4;
//   * a first 8-slot ring with M0 rewritten per fill and six refill-only visits in a row: 20-40 % of all images wrong, which
//     looked like a late-sampled M0 and led to the fixed-M0 form (kept: one value is certainly safe); a variant with the
//     bias as a ring entry (up to four refill-only visits in a row, then the next item at once): 18-63 in 5 x 10^5.
// WHAT THIS KERNEL DOES ABOUT IT (round 3: ordering rules, not cycle margins - VERDICT r2 item 4).  The experiments say
// which orderings hold: a READ behind the wave's own covering vmcnt always saw its fill (no wrong result ever came from the
// consuming visits, and stricter read-side waits changed nothing), whereas the retirement of a slot's previous fill did NOT
// order the next fill of that slot behind it when no read lay between them (the bursts: every fill waited vmcnt(3), i.e. for
// its slot's previous fill), and a complete drain, vmcnt(0), did.  So:
//   R1  a slot is READ only by the wave that filled it, behind that wave's own counted `s_waitcnt vmcnt(N)` covering the
//       fill (the ordering the hardware gives a wave over its own LDS-DMAs, MI355X_MICROARCH.md "Two waves per SIMD" item
//       7); the wait covers one fill MORE than the slot's own, so the slot's fill is never the youngest retired one;
//   R2  AT MOST ONE fill per slot is ever in flight: a slot is refilled either right behind a read of it that R1 ordered
//       behind its previous fill (consuming visits: fill -> wait -> read -> MFMA -> refill), or - where there is no read,
//       the refill-only visits at an item's end and the four fills that start a stream - behind `s_waitcnt vmcnt(0)`, i.e.
//       with NOTHING in flight at all;
//   R3  M0 (the LDS base of an LDS-DMA) is written in the same asm statement as the DMA that uses it, one wait state
//       ahead (the ISA's M0-write -> LDS-DMA rule), and is read by the hardware at issue; the bias fill's excursion to its
//       own M0 window begins and ends inside one statement, so no ring fill can issue under the bias' M0.  (The `s_nop 15`
//       pairs around it are left from round 2 and are NOT what correctness rests on.)
// Cost of R2's vmcnt(0) against round 2's 32-cycle spacing: see DESIGN.md 4.5 (same-box A/B).  tools/stress_cls_mega.py
// replays batches against the layered path, also with the detect engine busy on another stream (--with-detect), which is
// the situation of chain.py.  The conv kernels read a stage only behind vmcnt AND a workgroup barrier, one phase later, and
// refill a slot only behind a barrier that follows its last read (one fill per slot in flight as well).
constexpr uint32_t kMegaBack = (kMegaPf - 1) * 1024;
template <int U>
__device__ __forceinline__ void mega_load(uint32_t ring, const v4i_t rs, uint32_t off) {
  // s_nop 4: the descriptor or the ring address may have been written by a VALU op (v_readlane reloading a spilled SGPR)
  // in the instruction before this statement; a VMEM op needs 5 wait states behind that, and the compiler's hazard pass
  // does not look inside asm
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%3 lds"
               :: "s"(ring), "v"(off + (kMegaBack - U * 1024)), "s"(rs), "n"(U * 1024) : "memory", "m0");
}
__device__ __forceinline__ v4i_t mega_srd(const void* p, uint32_t bytes) {
  return make_srd(static_cast<const char*>(p) - kMegaBack, bytes + kMegaBack);
}
// an item's 16 biases: one dword LDS-DMA (lane l < 16 lands at slot + 4 l), an extra load in the stream - extra loads only
// make the counted waits stricter, never wrong (loads complete in order).  Its slot lies outside the ring's M0 window: the
// statement waits out the fills' sampling window on both sides of its own M0 (once per item, next to an epilogue).
__device__ __forceinline__ void mega_load_bias(uint32_t slot_addr, uint32_t ring, const v4i_t rs, uint32_t off) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_nop 15\n\ts_nop 15\n\ts_mov_b32 m0, %3"
               :: "s"(slot_addr), "v"(off), "s"(rs), "s"(ring) : "memory", "m0");
}
__device__ __forceinline__ void mega_wait_slot() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMegaPf - 1) : "memory"); }
__device__ __forceinline__ void mega_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t mega_arow(int n_t, int nsteps, int lane) { return (uint32_t)((n_t * nsteps * 64 + lane) * 16); }
__device__ __forceinline__ uint32_t mega_boff(int n_t, int lane) { return lane < 16 ? (uint32_t)((n_t * 16 + lane) * 4) : kMegaOob; }

__device__ __forceinline__ MegaStream mega_first_item(const MegaOp& nx, bool has, int wave, int lane) {
  const int mt = (nx.hout * nx.wout + 15) / 16, nt = nx.cout / 16, ng = (mt + 3) / 4;
  MegaStream st;
  st.rs = mega_srd(nx.w, (uint32_t)nx.wbytes);
  st.rsb = make_srd(nx.bias, (uint32_t)(nx.cout * 4));
  const int n_t = wave / ng;
  const bool live = has && nx.kind == 1 && wave < nt * ng;
  st.nsteps = live ? nx.kpad / 32 : 0;
  st.arow = mega_arow(n_t, nx.kpad / 32, lane);
  st.boff = live ? mega_boff(n_t, lane) : kMegaOob;
  return st;
}

// (re)start the stream: bias of the wave's first item of the next layer, then its first kMegaPf steps
__device__ __forceinline__ void mega_prefetch(const MegaStream& st, uint32_t ring, uint32_t bslot) {
  mega_drain();                                                       // R2: a stream starts with nothing in flight
  mega_load_bias(bslot, ring, st.rsb, st.boff);
  mega_load<0>(ring, st.rs, (0 < st.nsteps) ? st.arow : kMegaOob);
  mega_load<1>(ring, st.rs, (1 < st.nsteps) ? st.arow + 1024u : kMegaOob);
  mega_load<2>(ring, st.rs, (2 < st.nsteps) ? st.arow + 2048u : kMegaOob);
  mega_load<3>(ring, st.rs, (3 < st.nsteps) ? st.arow + 3072u : kMegaOob);
  static_assert(kMegaPf == 4, "one M0 reaches 4 KiB of LDS");
}

__device__ __forceinline__ void mega_stem(const MegaArgs& a, const MegaOp& op, unsigned char* smem, int img, int wave, int lane) {
  // conv3x3 s2 on the uint8 image as one 32-deep MFMA step per 16 output pixels (K' order: kernels_misc.h stem_kernel).
  // The image bytes come from HBM: a wave loads the bytes of up to kStemTb of its tiles before it computes the first.
  constexpr int kStemTb = 8;
  const int frow = lane & 15, q = lane >> 4;
  const uint8_t* im = a.in + (size_t)img * a.H * a.W * 3;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(im), 0, (uint32_t)(a.H * a.W * 3), 0x00020000);
  const int W3 = a.W * 3;
  const int startq = (q < 3) ? q * W3 : 8, stepq = (q < 3) ? 1 : W3;
  const uint32_t m_always = (q < 3) ? 0u : 0xF8u, m_top = (q < 3) ? (q == 0 ? 0xFFu : 0u) : 1u, m_left = (q < 3) ? 7u : 0u;
  const int M = op.hout * op.wout, mt = (M + 15) / 16, ntc = op.cout / 16;
  const half_t* wp = reinterpret_cast<const half_t*>(op.w);
  const uint4 wf0 = *reinterpret_cast<const uint4*>(wp + frow * 32 + q * 8);
  const float4 b0 = *reinterpret_cast<const float4*>(op.bias + q * 4);
  for (int t0 = wave; t0 < mt; t0 += kMegaWaves * kStemTb) {
    uint32_t ub[kStemTb][8];
#pragma unroll
    for (int i = 0; i < kStemTb; ++i) {
      const int m = (t0 + i * kMegaWaves) * 16 + frow;
      const bool vm = m < M;
      const int mm = vm ? m : 0;
      const int ho = mm >> op.lg_wout, wo = mm & (op.wout - 1);
      const int rb = ((2 * ho - 1) * a.W + 2 * wo - 1) * 3;
      const uint32_t inval = (vm ? m_always : 0xFFu) | (ho == 0 ? m_top : 0u) | (wo == 0 ? m_left : 0u);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t off = (uint32_t)(rb + startq + j * stepq) | (((inval >> j) & 1u) << 31);
        ub[i][j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, off, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < kStemTb; ++i) {
      const int m = (t0 + i * kMegaWaves) * 16 + frow;
      const bool vm = m < M;
      const int ho = m >> op.lg_wout, wo = m & (op.wout - 1);
      f16x8 xb;
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = (half_t)((float)ub[i][j] * (1.0f / 255.0f));
      for (int tc = 0; tc < ntc; ++tc) {
        uint4 wf = wf0;
        float4 bb = b0;
        if (tc > 0) {
          wf = *reinterpret_cast<const uint4*>(wp + (tc * 16 + frow) * 32 + q * 8);
          bb = *reinterpret_cast<const float4*>(op.bias + tc * 16 + q * 4);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&wf), xb, acc, 0, 0, 0);
        if (vm) {
          const int n = tc * 16 + q * 4;
          const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float y = acc[r] + bv[r];
            if (op.act) y = silu_fast(y);
            v[r] = y;
          }
          const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<f16x4*>(smem + op.dst_org + (ho * op.dst_wp + wo) * op.dst_ps + n * 2) = hv;
        }
      }
    }
  }
}

// Per-item constants of the K loop
struct MegaItem {
  v4i_t rsw, rs_n;                          // weight tensor of this item / of the item the ring moves on to
  uint32_t arow, arow_n;
  int nsteps, nsteps_n;
  int ksize, lg_cpt, cptm, cpt1, src_wp, src_ps;
};

// byte offset of this lane's 8-channel chunk of K step ks from the tile's tap-(0,0) pixel.  No bounds tests: out-of-image
// taps read the buffer's zero halo; chunks in the zero tail of K (weights are 0 there), and the one step past the end the
// read-ahead touches, are clamped onto a real chunk so they multiply / read finite data
__device__ __forceinline__ int mega_toff(const MegaItem& it, int ks, int kq) {
  const int q = ks * 4 + kq;
  if (it.ksize == 3) {
    const int tap = min(q >> it.lg_cpt, 8);
    const int dy = (tap * 11) >> 5;                                    // tap / 3 for tap < 12
    const int dx = tap - 3 * dy;
    return (dy * it.src_wp + dx) * it.src_ps + (q & it.cptm) * 16;
  }
  return min(q, it.cpt1) * 16;
}

// One turn of the ring: slots 0..NU-1 hold K steps k0..k0+NU-1 of the item and are consumed; every slot is then refilled
// (the ring's rotation stays uniform, which is what makes the hand-counted vmcnt exact).  Branch-free: NPT pixel tiles per
// weight fragment, the activation fragments of the next step are read before the MFMAs of this one.
template <int NPT, int NU, int U>
__device__ __forceinline__ void mega_visit(const MegaItem& it, const unsigned char* smem, const int (&base)[4], int kq, int k0,
                                           uint32_t ring, int lane, f32x4 (&acc)[4], uint4 (&bf)[4]) {
  const int ks = k0 + U;
#ifdef MEGA_WAIT0
  if (U < NU) mega_drain(); else mega_wait_slot();      // experiment: a consuming visit waits for EVERY fill in flight
#elif defined(MEGA_R2_SPACING)
  if (U < NU) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMegaPf - 2) : "memory"); else mega_wait_slot();   // round 2's form (A/B only)
#else
  // R1: a consuming visit waits for its slot's fill AND the next slot's (one fill more than it needs);
  // R2: a refill-only visit issues its fill with nothing in flight
  if (U < NU) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMegaPf - 2) : "memory"); else mega_drain();
#endif
  if (U < NU) {
    const uint4 af = *reinterpret_cast<const uint4*>(smem + ring + U * 1024 + lane * 16);
    const int tn = mega_toff(it, ks + 1, kq);
    uint4 bn[4];
#pragma unroll
    for (int j = 0; j < NPT; ++j) bn[j] = *reinterpret_cast<const uint4*>(smem + base[j] + tn);
#pragma unroll
    for (int j = 0; j < NPT; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&af), *reinterpret_cast<const f16x8*>(&bf[j]), acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NPT; ++j) bf[j] = bn[j];
  }
  // slot U is free: step ks + kMegaPf of this item, or - past its end - step U of the wave's next item
  const int t = ks + kMegaPf;
  const bool cur = t < it.nsteps;
  const uint32_t off = cur ? it.arow + (uint32_t)(t * 1024) : ((U < it.nsteps_n) ? it.arow_n + (uint32_t)(U * 1024) : kMegaOob);
  mega_load<U>(ring, cur ? it.rsw : it.rs_n, off);
#if defined(MEGA_R2_SPACING)
  if (U >= NU) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // round 2: refill-only fills 32 cycles apart (A/B only)
#endif
}

// One turn of the ring: slots 0..NU-1 hold K steps k0..k0+NU-1 of the item and are consumed; every slot is then refilled
// (the ring's rotation stays uniform, which is what makes the hand-counted vmcnt exact).  Branch-free: NPT pixel tiles per
// weight fragment, the activation fragments of the next step are read before the MFMAs of this one.
template <int NPT, int NU>
__device__ __forceinline__ void mega_block(const MegaItem& it, const unsigned char* smem, const int (&base)[4], int kq, int k0,
                                           uint32_t ring, int lane, f32x4 (&acc)[4], uint4 (&bf)[4]) {
  mega_visit<NPT, NU, 0>(it, smem, base, kq, k0, ring, lane, acc, bf);
  mega_visit<NPT, NU, 1>(it, smem, base, kq, k0, ring, lane, acc, bf);
  mega_visit<NPT, NU, 2>(it, smem, base, kq, k0, ring, lane, acc, bf);
  mega_visit<NPT, NU, 3>(it, smem, base, kq, k0, ring, lane, acc, bf);
}

template <int NPT>
__device__ __forceinline__ void mega_kloop(const MegaItem& it, const unsigned char* smem, const int (&base)[4], int kq,
                                           uint32_t ring, int lane, f32x4 (&acc)[4]) {
  uint4 bf[4];
  const int t0 = mega_toff(it, 0, kq);
#pragma unroll
  for (int j = 0; j < NPT; ++j) bf[j] = *reinterpret_cast<const uint4*>(smem + base[j] + t0);
  int k0 = 0;
  for (; k0 + kMegaPf <= it.nsteps; k0 += kMegaPf) mega_block<NPT, kMegaPf>(it, smem, base, kq, k0, ring, lane, acc, bf);
  if (it.nsteps - k0 == 2) { mega_block<NPT, 2>(it, smem, base, kq, k0, ring, lane, acc, bf); k0 += kMegaPf; }   // K is padded to 64: the step count is even
#ifdef MEGA_BURST
  // hazard experiment (tools/stress_cls_mega.py): extra ring turns of refill-only visits, each re-fetching the next item's
  // first entries - harmless by construction, so any wrong result comes from the burst of fills itself
  for (int r = 0; r < MEGA_BURST; ++r) {
#if MEGA_BURST_KIND == 1      // fills that touch no memory (out of range): complete at once
    MegaItem ib = it; ib.nsteps_n = 0;
    mega_block<NPT, 0>(ib, smem, base, kq, (it.nsteps + 3) & ~3, ring, lane, acc, bf);
    if (r == MEGA_BURST - 1) mega_block<NPT, 0>(it, smem, base, kq, (it.nsteps + 3) & ~3, ring, lane, acc, bf);   // put the real entries back
#else
    mega_block<NPT, 0>(it, smem, base, kq, (it.nsteps + 3) & ~3, ring, lane, acc, bf);
#endif
  }
#ifdef MEGA_BURST_PAUSE
  for (int r = 0; r < MEGA_BURST_PAUSE; ++r) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // 64 cycles each
#endif
#endif
}

__device__ __forceinline__ void mega_conv(const MegaOp& op, const MegaStream& nx, unsigned char* smem, int wave, int lane,
                                          uint32_t ring, uint32_t bring, int& bsel, unsigned long long* fine) {
  const int frow = lane & 15, kq = lane >> 4;
  const int M = op.hout * op.wout, mt = (M + 15) / 16, nt = op.cout / 16, ng = (mt + 3) / 4;
  const int nitems = nt * ng;
  const int pad = op.ksize == 3 ? 1 : 0;
  if (wave >= nitems) {                                                // idle in this layer: fetch for the next one
    bsel ^= 1;
    mega_prefetch(nx, ring, bring + bsel * 256);
    return;
  }
  const v4i_t rsb = make_srd(op.bias, (uint32_t)(op.cout * 4));
  MegaItem it;
  it.rsw = mega_srd(op.w, (uint32_t)op.wbytes);
  it.nsteps = op.kpad / 32;
  it.ksize = op.ksize; it.lg_cpt = op.lg_cpt; it.cptm = (1 << op.lg_cpt) - 1; it.cpt1 = op.cin / 8 - 1;
  it.src_wp = op.src_wp; it.src_ps = op.src_ps;
  for (int item = wave; item < nitems; item += kMegaWaves) {
    const int n_t = item / ng, g = item - n_t * ng;
    const int n = n_t * 16 + kq * 4;                                   // lane's 4 output channels (C/D map of the 16x16 MFMA)
    // what the stream fetches once this item's steps are all in flight; the next item's bias goes out now (slot bsel holds
    // this item's, fetched an item ago)
    const bool more = item + kMegaWaves < nitems;
    const int n_tn = (item + kMegaWaves) / ng;
    it.arow = mega_arow(n_t, it.nsteps, lane);
    it.arow_n = more ? mega_arow(n_tn, it.nsteps, lane) : nx.arow;
    it.nsteps_n = more ? it.nsteps : nx.nsteps;
    it.rs_n = more ? it.rsw : nx.rs;
    mega_load_bias(bring + (bsel ^ 1) * 256, ring, more ? rsb : nx.rsb, more ? mega_boff(n_tn, lane) : nx.boff);
    int base[4], pdst[4];
    bool pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = (4 * g + j) * 16 + frow;
      pv[j] = p < M;
      const int pp = pv[j] ? p : 0;                                    // columns past the image compute on pixel 0, never stored
      const int oy = pp >> op.lg_wout, ox = pp & (op.wout - 1);
      base[j] = op.src_org + ((oy * op.stride - pad) * op.src_wp + (ox * op.stride - pad)) * op.src_ps;   // tap (0,0)
      pdst[j] = oy * op.dst_wp + ox;
    }
    const int npt = min(4, mt - 4 * g);                                // pixel tiles of this group (wave-uniform)
    if (fine && item == wave && lane == 0) fine[1] = __builtin_readcyclecounter();
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (npt == 1) mega_kloop<1>(it, smem, base, kq, ring, lane, acc);
    else mega_kloop<4>(it, smem, base, kq, ring, lane, acc);                   // 2 or 3 tiles: the others recompute pixel 0, never stored
    // epilogue: lane holds channels n..n+3 of pixel p
    if (fine && item == wave && lane == 0) fine[2] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kMegaPf) : "memory");      // the bias is older than the ring's fills in flight
    const float4 b4 = *reinterpret_cast<const float4*>(smem + bring + bsel * 256 + kq * 16);
    bsel ^= 1;
    const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < npt && pv[j]) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[j][r] + bv[r];
          if (op.act) x = silu_fast(x);
          v[r] = x;
        }
        if (op.res_org >= 0) {
          const int p = (4 * g + j) * 16 + frow;
          const int rp = (p >> op.lg_wout) * op.res_wp + (p & (op.wout - 1));
          const f16x4 h = *reinterpret_cast<const f16x4*>(smem + op.res_org + rp * op.res_ps + n * 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)h[r];
        }
        const f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<f16x4*>(smem + op.dst_org + pdst[j] * op.dst_ps + n * 2) = hv;
      }
    }
  }
}

// [cout][kpad] f16 -> fragment order: chunk (n_t, ks, lane) = the 16 bytes lane (kq = lane / 16, row = lane % 16) feeds the MFMA
__global__ void mega_repack_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int cout, int kpad) {
  const int nsteps = kpad / 32, total = (cout / 16) * nsteps * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, ks = (i >> 6) % nsteps, n_t = (i >> 6) / nsteps;
    const uint4 v = *reinterpret_cast<const uint4*>(w + (long)(n_t * 16 + (lane & 15)) * kpad + ks * 32 + (lane >> 4) * 8);
    *reinterpret_cast<uint4*>(out + (long)i * 8) = v;
  }
}

// workgroup barrier for LDS traffic only: no vmcnt wait, the weight ring stays in flight across it
__device__ __forceinline__ void mega_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(kMegaWaves * 64) void cls_mega_kernel(const MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.stamps && img < 8 && tid == 0) a.stamps[img * 32 + 31] = __builtin_readcyclecounter();
  const uint32_t ring = (uint32_t)(a.ring_off + wave * (kMegaPf * 1024));   // this wave's weight ring
  const uint32_t bring = (uint32_t)(a.bias_off + wave * 512);              // and its two bias slots
  int bsel = 0;
  // layer table -> LDS (one coalesced copy); a wave reads a descriptor with broadcast LDS reads and pins the values to
  // SGPRs.  From the kernel arguments each descriptor was ~30 scalar loads whose latency the SGPR spill code exposed at
  // every layer (~1 k cycles x 26 layers).
  {
    constexpr int kDw = (int)(sizeof(MegaOp) / 4);
    const int32_t* src = reinterpret_cast<const int32_t*>(a.ops_dev);
    int32_t* dst = reinterpret_cast<int32_t*>(smem + a.desc_off);
    for (int i = tid; i < a.nops * kDw; i += kMegaWaves * 64) dst[i] = src[i];
    __syncthreads();
  }
  auto desc = [&](int oi) {
    constexpr int kDw = (int)(sizeof(MegaOp) / 4);
    const int32_t* p = reinterpret_cast<const int32_t*>(smem + a.desc_off) + oi * kDw;
    MegaOp o;
    int32_t* q = reinterpret_cast<int32_t*>(&o);
#pragma unroll
    for (int i = 0; i < kDw; ++i) q[i] = __builtin_amdgcn_readfirstlane(p[i]);
    return o;
  };
  MegaOp op = desc(0);
  MegaOp nxo = desc(a.nops > 1 ? 1 : 0);
  for (int oi = 0; oi < a.nops; ++oi) {
    const bool has = oi + 1 < a.nops;
    const MegaStream nx = mega_first_item(nxo, has, wave, lane);
    if (op.zero_rows > 0) {
      // a buffer a 3x3 conv will read starts its life: clear its halo ring (disjoint from what this layer writes)
      const int hp = op.zero_rows, wp = op.dst_wp, cps = op.dst_ps >> 4;
      const int nbp = 2 * wp + 2 * (hp - 2);
      for (int i = tid; i < nbp * cps; i += kMegaWaves * 64) {
        const int b = i / cps, c = i - b * cps;
        int row, col;
        if (b < wp) { row = 0; col = b; }
        else if (b < 2 * wp) { row = hp - 1; col = b - wp; }
        else { const int r = b - 2 * wp; row = 1 + (r >> 1); col = (r & 1) ? wp - 1 : 0; }
        *reinterpret_cast<uint4*>(smem + op.zero_off + (row * wp + col) * op.dst_ps + c * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if (op.kind == 0) { mega_stem(a, op, smem, img, wave, lane); mega_prefetch(nx, ring, bring); }
    else {
      unsigned long long* fine = (a.stamps && img == 0 && wave == 0) ? a.stamps + 256 + oi * 4 : nullptr;
      if (fine && lane == 0) fine[0] = __builtin_readcyclecounter();
      mega_conv(op, nx, smem, wave, lane, ring, bring, bsel, fine);
      if (fine && lane == 0) fine[3] = __builtin_readcyclecounter();
    }
    mega_barrier();
    if (a.stamps && img < 8 && tid == 0) a.stamps[img * 32 + oi] = __builtin_readcyclecounter();
    op = nxo;
    nxo = desc(oi + 2 < a.nops ? oi + 2 : oi);
  }
  mega_drain();
  // ---- Classify tail (arithmetic order of cls_head_kernel): avg-pool -> Linear -> softmax
  float* pooled = reinterpret_cast<float*>(smem + a.pool_off);
  float* lg = pooled + a.feat_c;
  const half_t* f = reinterpret_cast<const half_t*>(smem + a.feat_off);
  constexpr int kLin = 20;                                               // Linear weights of the wave's first class, in registers
  float wv[kLin];
  const bool wreg = a.feat_c <= kLin * 64;
  if (wreg) {
#pragma unroll
    for (int i = 0; i < kLin; ++i) {
      const int ch = lane + 64 * i;
      wv[i] = (wave < a.nc && ch < a.feat_c) ? a.lin_w[(long)wave * a.feat_c + ch] : 0.f;
    }
  }
  for (int ch = tid; ch < a.feat_c; ch += kMegaWaves * 64) {
    float s = 0.f;
    for (int p = 0; p < a.feat_hw; ++p) s += (float)f[p * (a.feat_ps / 2) + ch];
    pooled[ch] = s / (float)a.feat_hw;
  }
  __syncthreads();
  for (int k = wave; k < a.nc; k += kMegaWaves) {
    const float* wr = a.lin_w + (long)k * a.feat_c;
    float s = 0.f;
    if (wreg && k == wave) {
#pragma unroll
      for (int i = 0; i < kLin; ++i) { const int ch = lane + 64 * i; if (ch < a.feat_c) s = fmaf(pooled[ch], wv[i], s); }
    } else {
      for (int ch = lane; ch < a.feat_c; ch += 64) s = fmaf(pooled[ch], wr[ch], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) lg[k] = s + a.lin_b[k];
  }
  __syncthreads();
  if (tid == 0) {
    float mx = lg[0];
    for (int k = 1; k < a.nc; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f;
    for (int k = 0; k < a.nc; ++k) s += expf(lg[k] - mx);
    for (int k = 0; k < a.nc; ++k) {
      if (a.logits) a.logits[(long)img * a.nc + k] = lg[k];
      if (a.probs) a.probs[(long)img * a.nc + k] = expf(lg[k] - mx) / s;
    }
    if (a.stamps && img < 8) a.stamps[img * 32 + 30] = __builtin_readcyclecounter();
  }
}

}  // namespace miyolo

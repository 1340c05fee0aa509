// Non-GEMM kernels of the YOLOv8 path: stem conv on uint8 frames, SPPF max-pool,
// Detect decode, Classify tail.  All HBM/latency bound: coalesced NHWC access, no MFMA.
#pragma once
#include "common.h"

namespace miyolo {

// ------------------------------------------------------------------------------------
// Stem: conv3x3 s2 p1 over the uint8 frame with the `/255` of the reference's preprocess
// ([3P] DetectionPredictor.preprocess: im.float(); im /= 255) folded in, + bias + SiLU.
//
// The layer is bound by WRITING its output (2*cout bytes per pixel against 6.75 input bytes);
// the 27-deep contraction is padded to one 32-deep MFMA step so the arithmetic disappears
// behind the stores.  K' order (what `lane>>4 = q` holds, 8 values each):
//     q = 0,1,2 : kernel row ky = q, the first 8 of its 9 consecutive input bytes (kx*3 + c)
//     q = 3     : byte 8 (kx=2, c=2) of rows ky = 0,1,2, then five zeros
// i.e. a lane reads 8 adjacent bytes of one image row - NHWC uint8 makes a kernel row's three
// pixels contiguous.  Weights arrive as [cout][32] in the same K' order (weights.py).
// One wave = 16 output pixels x all output channels per iteration; out-of-image taps are
// out-of-range buffer loads (zero).
struct StemArgs {
  const uint8_t* in;   // [B,H,W,3]
  const void* w;       // [cout][32] T, K' order
  const float* bias;   // [cout]
  void* out;           // [B,H/2,W/2,cout] T
  int32_t B, H, W, Ho, Wo, cout, act, exact;
  float out_inv_scale;  // OUT8: stored byte = fp8(value * out_inv_scale)
  uint32_t in_bytes;
  uint32_t mg_hw_mul, mg_hw_shift, mg_w_mul, mg_w_shift;   // host_magic(Ho*Wo), host_magic(Wo): no 64-bit divides per tile
};

// OUT8: compute as T (= half), store fp8 e4m3 bytes (the fp8 engine's first activation buffer).
template <typename T, int TCS, bool OUT8 = false>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
  constexpr int CE = DT<T>::CE;
  const int lane = threadIdx.x & 63, frow = lane & 15, q = lane >> 4;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const long total = (long)a.B * a.Ho * a.Wo;
  const long ntiles = (total + 15) / 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.in), 0, a.in_bytes, 0x00020000);
  // weight fragments: A[ch = tc*16 + frow][k' = 8q .. 8q+7]
  uint4 wf[TCS][sizeof(T) == 4 ? 2 : 1];
  const T* wp = reinterpret_cast<const T*>(a.w);
#pragma unroll
  for (int tc = 0; tc < TCS; ++tc) {
    const T* p = wp + (tc * 16 + frow) * 32 + q * 8;
    wf[tc][0] = *reinterpret_cast<const uint4*>(p);
    if constexpr (sizeof(T) == 4) wf[tc][1] = *reinterpret_cast<const uint4*>(p + 4);
  }
  float bv[TCS][4];
#pragma unroll
  for (int tc = 0; tc < TCS; ++tc)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[tc][r] = a.bias[tc * 16 + q * 4 + r];

  const int W3 = a.W * 3;
  // Gather of the 3x3x3 window (round 3).  Lane (pixel frow, q < 3) needs the 9 consecutive bytes of window row q - 8 for
  // its own K' slot, the 9th for the q = 3 lane of its pixel.  Rounds 1-2 fetched them as 8 single-byte loads per lane
  // (8 texture-path instructions of 64 scattered bytes per 16 outputs: the kernel sat at 0.32 of its HBM roofline with 67 %
  // of its wave cycles parked on memory).  Now ONE 12-byte load per lane from the dword below the window start, the window
  // cut out with v_alignbyte; the q = 3 lanes receive their three bytes from the q = 0..2 lanes by ds_bpermute.  The left
  // image column (wo = 0: the window starts 3 bytes before the row) loads from the row start and shifts, so no offset is
  // negative except a row above the image, which is out of range as a whole (zeros).  The frame batch is a multiple of 4
  // bytes (even H and W), and the last window ends on its last byte: no load reaches past the buffer.
  for (long tile = wave_global; tile < ntiles; tile += nwaves) {
    const long m = tile * 16 + frow;
    const bool vm = m < total;
    const uint32_t mm = vm ? (uint32_t)m : 0u;                 // total < 2^31 (engine: buffers below 2 GiB)
    const int b = (int)magic_div(mm, a.mg_hw_mul, a.mg_hw_shift);
    const uint32_t rem = mm - (uint32_t)b * (uint32_t)(a.Ho * a.Wo);
    const int ho = (int)magic_div(rem, a.mg_w_mul, a.mg_w_shift);
    const int wo = (int)rem - ho * a.Wo;
    const int wi0 = 2 * wo - 1, hi0 = 2 * ho - 1;
    const bool left = wo == 0;
    const int o = ((b * a.H + hi0 + (q < 3 ? q : 0)) * a.W + wi0) * 3 + (left ? 3 : 0);
    const bool rowok = vm && q < 3 && !(ho == 0 && q == 0);
    const uint32_t sh = (uint32_t)o & 3u;
    const auto d3 = __builtin_amdgcn_raw_buffer_load_b96(rs, rowok ? ((uint32_t)o & ~3u) : 0x80000000u, 0, 0);
    uint32_t lo = __builtin_amdgcn_alignbyte((uint32_t)d3[1], (uint32_t)d3[0], sh);      // window bytes 0..3
    uint32_t hi = __builtin_amdgcn_alignbyte((uint32_t)d3[2], (uint32_t)d3[1], sh);      // 4..7
    uint32_t b8 = ((uint32_t)d3[2] >> (8u * sh)) & 0xFFu;                                 // 8
    if (left) {                       // loaded from the row start: the true window is three zero bytes, then bytes 0..5 of it
      b8 = (hi >> 8) & 0xFFu;
      hi = (hi << 24) | (lo >> 8);
      lo = lo << 24;
    }
    // the q = 3 lanes: byte 8 of window rows 0, 1, 2 of their pixel = b8 of lanes frow, frow + 16, frow + 32
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(frow * 4, (int)b8);
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute((frow + 16) * 4, (int)b8);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute((frow + 32) * 4, (int)b8);
    if (q == 3) { lo = r0 | (r1 << 8) | (r2 << 16); hi = 0u; }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t u = ((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xFFu;
      x[j] = a.exact ? (float)u / 255.0f : (float)u * (1.0f / 255.0f);   // f16: rounded to half right after
    }
    f32x4 acc[TCS];
#pragma unroll
    for (int tc = 0; tc < TCS; ++tc) acc[tc] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int tc = 0; tc < TCS; ++tc) {
        const float* w0 = reinterpret_cast<const float*>(&wf[tc][0]);
        const float* w1 = reinterpret_cast<const float*>(&wf[tc][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], x[j], acc[tc], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], x[4 + j], acc[tc], 0, 0, 0);
      }
    } else {
      f16x8 xb;
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = (half_t)x[j];
#pragma unroll
      for (int tc = 0; tc < TCS; ++tc)
        acc[tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&wf[tc][0]), xb, acc[tc], 0, 0, 0);
    }
    if (vm) {
      T* op = reinterpret_cast<T*>(a.out) + m * a.cout + q * 4;
      unsigned char* op8 = reinterpret_cast<unsigned char*>(a.out) + m * a.cout + q * 4;
#pragma unroll
      for (int tc = 0; tc < TCS; ++tc) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = acc[tc][r] + bv[tc][r];
          if (a.act) y = a.exact ? silu_exact(y) : silu_fast(y);
          v[r] = y;
        }
        if constexpr (OUT8) {
          const float qs = a.out_inv_scale;
          *reinterpret_cast<uint32_t*>(op8 + tc * 16) = pack_fp8x4(v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs);
        } else if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(op + tc * 16) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          f16x4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<f16x4*>(op + tc * 16) = hv;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// MaxPool2d(k=5, s=1, p=2) on a channel slice ([3P] SPPF.m). One thread = one pixel x one
// 16-byte channel chunk; max is exact, so any evaluation order matches the reference.
struct PoolArgs {
  const void* src; void* dst;
  int32_t src_ld, src_choff, dst_ld, dst_choff, ch, B, H, W;
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool5_kernel(const PoolArgs a) {
  constexpr int CE = DT<T>::CE;
  typedef T vec_t __attribute__((ext_vector_type(CE)));
  const int nch = a.ch / CE;
  const long total = (long)a.B * a.H * a.W * nch;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cc = (int)(idx % nch);
  long p = idx / nch;
  const int w = (int)(p % a.W);
  p /= a.W;
  const int h = (int)(p % a.H), b = (int)(p / a.H);
  const T* sp = reinterpret_cast<const T*>(a.src);
  vec_t best = *reinterpret_cast<const vec_t*>(sp + (((long)b * a.H + h) * a.W + w) * a.src_ld + a.src_choff + cc * CE);
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const int hh = h + dy;
    if (hh < 0 || hh >= a.H) continue;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int ww = w + dx;
      if (ww < 0 || ww >= a.W) continue;
      const vec_t v = *reinterpret_cast<const vec_t*>(sp + (((long)b * a.H + hh) * a.W + ww) * a.src_ld + a.src_choff + cc * CE);
      best = __builtin_elementwise_max(best, v);
    }
  }
  T* dp = reinterpret_cast<T*>(a.dst);
  *reinterpret_cast<vec_t*>(dp + (((long)b * a.H + h) * a.W + w) * a.dst_ld + a.dst_choff + cc * CE) = best;
}

// SPPF's three chained pools as ONE launch: pool5(x), pool5(pool5(x)), pool5(pool5(pool5(x))) for one image and one
// 16-byte channel chunk per workgroup, the map ping-ponged through LDS (row max, then column max - a 5x5 window max is
// separable, and max is exact, so the three outputs are bit-identical to three maxpool5_kernel launches); the input is
// read once instead of three times through 25-tap gathers.
struct Sppf3Args {
  const void* src; void* dst[3];
  int32_t src_ld, src_choff, dst_ld[3], dst_choff[3], ch, B, H, W;
};

template <typename T>
__global__ __launch_bounds__(256) void sppf3_kernel(const Sppf3Args a) {
  constexpr int CE = DT<T>::CE;
  typedef T vec_t __attribute__((ext_vector_type(CE)));
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
  const int hw = a.H * a.W;
  vec_t* A = reinterpret_cast<vec_t*>(sp_lds);
  vec_t* Bf = A + hw;
  const int cc = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const T* sp = reinterpret_cast<const T*>(a.src);
  for (int p = tid; p < hw; p += 256) A[p] = *reinterpret_cast<const vec_t*>(sp + ((long)b * hw + p) * a.src_ld + a.src_choff + cc * CE);
  __syncthreads();
  for (int st = 0; st < 3; ++st) {
    for (int p = tid; p < hw; p += 256) {
      const int y = p / a.W, x = p - y * a.W;
      vec_t m = A[p];
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) { const int xx = x + dx; if (dx != 0 && xx >= 0 && xx < a.W) m = __builtin_elementwise_max(m, A[y * a.W + xx]); }
      Bf[p] = m;
    }
    __syncthreads();
    T* dp = reinterpret_cast<T*>(a.dst[st]);
    for (int p = tid; p < hw; p += 256) {
      const int y = p / a.W;
      vec_t m = Bf[p];
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) { const int yy = y + dy; if (dy != 0 && yy >= 0 && yy < a.H) m = __builtin_elementwise_max(m, Bf[p + dy * a.W]); }
      A[p] = m;
      *reinterpret_cast<vec_t*>(dp + ((long)b * hw + p) * a.dst_ld[st] + a.dst_choff[st] + cc * CE) = m;
    }
    __syncthreads();
  }
}

// fp8: 16 channels per thread; e4m3 bytes are compared through their float values (the byte patterns are sign-magnitude,
// not two's complement) and the winning BYTE is kept, so the result is exact; the slice keeps its producer's scale.
template <>
__global__ __launch_bounds__(256) void maxpool5_kernel<fp8_t>(const PoolArgs a) {
  const int nch = a.ch / 16;
  const long total = (long)a.B * a.H * a.W * nch;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cc = (int)(idx % nch);
  long p = idx / nch;
  const int w = (int)(p % a.W);
  p /= a.W;
  const int h = (int)(p % a.H), b = (int)(p / a.H);
  const unsigned char* sp = reinterpret_cast<const unsigned char*>(a.src);
  float best[16];
  unsigned char bb[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { best[k] = -1e30f; bb[k] = 0; }
  for (int dy = -2; dy <= 2; ++dy) {
    const int hh = h + dy;
    if (hh < 0 || hh >= a.H) continue;
    for (int dx = -2; dx <= 2; ++dx) {
      const int ww = w + dx;
      if (ww < 0 || ww >= a.W) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(sp + (((long)b * a.H + hh) * a.W + ww) * a.src_ld + a.src_choff + cc * 16);
      const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float f[4];
        unpack_fp8x4(wv[q], f);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (f[r] > best[q * 4 + r]) { best[q * 4 + r] = f[r]; bb[q * 4 + r] = (unsigned char)(wv[q] >> (8 * r)); }
      }
    }
  }
  uint32_t o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = (uint32_t)bb[q * 4] | ((uint32_t)bb[q * 4 + 1] << 8) | ((uint32_t)bb[q * 4 + 2] << 16) | ((uint32_t)bb[q * 4 + 3] << 24);
  unsigned char* dp = reinterpret_cast<unsigned char*>(a.dst);
  *reinterpret_cast<uint4*>(dp + (((long)b * a.H + h) * a.W + w) * a.dst_ld + a.dst_choff + cc * 16) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------
// Detect decode ([3P] Detect._inference): per anchor, DFL = softmax over the 16 bins of each
// box side followed by the expectation sum_k k*p_k, dist2bbox(xywh=True) around the cell
// centre (+0.5), x stride, and sigmoid on the class logits.
//   raw  : fp32 [B, hw_l, 64+nc] per level (box logits | class logits), written by the head's
//          final 1x1 convs
//   y    : fp32 [B, 4+nc, A]  (A = anchors of all levels, level-major like the reference's cat)
// One block = 64 anchors: rows are pulled through LDS (stride padded to an odd number of
// words) so both the row-wise global reads and the channel-major global writes coalesce.
struct DecodeArgs {
  const float* raw[3];
  int32_t lh[3], lw[3], lstride[3], aoff[3];   // per level grid, stride, first anchor index
  int32_t nlevel, nc, A, B;
  float* y;
  // optional fused score filter of the NMS (nms.h, nms_prefilter_kernel - same compares, same key): keys != null
  unsigned long long* keys;       // [B, P]
  int32_t* count;                 // [B], zeroed before this launch
  int32_t* cls_idx;               // [B, A]
  int32_t P, use_mask;
  float conf;
  uint32_t cls_mask[8];
};

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
  extern __shared__ float sm[];   // [64][no+1]
  const int no = 64 + a.nc, ldr = no + 1;
  const int b = blockIdx.y;
  int a0 = blockIdx.x * 64;       // first anchor of this block (blocks never straddle levels:
  int lvl = 0;                    //  grid.x is built per level from ceil(hw/64) blocks)
  int blk = blockIdx.x;
  for (lvl = 0; lvl < a.nlevel; ++lvl) {
    const int nb = (a.lh[lvl] * a.lw[lvl] + 63) / 64;
    if (blk < nb) break;
    blk -= nb;
  }
  const int hw = a.lh[lvl] * a.lw[lvl];
  a0 = blk * 64;
  const int na = min(64, hw - a0);
  const float* rp = a.raw[lvl] + ((long)b * hw + a0) * no;
  for (int i = threadIdx.x; i < na * no; i += 256) {
    const int r = i / no, c = i - r * no;
    sm[r * ldr + c] = rp[i];
  }
  __syncthreads();
  // phase 1: thread = (anchor, side); softmax expectation over 16 bins
  {
    const int an = threadIdx.x >> 2, side = threadIdx.x & 3;
    float d = 0.f;
    if (an < na) {
      const float* l = sm + an * ldr + side * 16;
      float mx = l[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) mx = fmaxf(mx, l[k]);
      float e[16], s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) { e[k] = expf(l[k] - mx); s += e[k]; }
#pragma unroll
      for (int k = 0; k < 16; ++k) d += (e[k] / s) * (float)k;
    }
    __syncthreads();
    if (an < na) sm[an * ldr + side] = d;   // overwrite the first 4 words of the row
  }
  __syncthreads();
  float* yb = a.y + (long)b * (4 + a.nc) * a.A + a.aoff[lvl] + a0;
  const int an = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (an < na) {
    if (grp == 0) {
      const int idx = a0 + an;
      const float ax = (float)(idx % a.lw[lvl]) + 0.5f, ay = (float)(idx / a.lw[lvl]) + 0.5f;
      const float* d = sm + an * ldr;
      const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
      const float st = (float)a.lstride[lvl];
      yb[0L * a.A + an] = ((x1 + x2) / 2.0f) * st;
      yb[1L * a.A + an] = ((y1 + y2) / 2.0f) * st;
      yb[2L * a.A + an] = (x2 - x1) * st;
      yb[3L * a.A + an] = (y2 - y1) * st;
    }
    for (int c = grp; c < a.nc; c += 4) {
      const float v = sm[an * ldr + 64 + c];
      const float sg = 1.0f / (1.0f + expf(-v));
      // With the fused score filter (miyolo_detect) nobody reads the class planes of y afterwards - the NMS takes score and class
      // from the keys / cls_idx this kernel writes and only the four box planes from y: 137 MB of stores per 64 frames saved.
      // miyolo_head_raw (keys == null) gets the full y.
      if (a.keys) sm[an * ldr + 64 + c] = sg;           // each word is read and rewritten by the same thread
      else yb[(long)(4 + c) * a.A + an] = sg;
    }
  }
  if (!a.keys) return;
  // fused NMS score filter: the class scores of the 64 anchors are still in LDS - the separate pass would read all of y
  // back (137 MB at batch 64).  First arg-max in class order, `classes=` mask, strict > conf: nms_prefilter_kernel's rules
  __syncthreads();
  if (grp == 0 && an < na) {
    const float* sc = sm + an * ldr + 64;
    float best = sc[0];
    int j = 0;
    for (int c = 1; c < a.nc; ++c) {
      const float v = sc[c];
      if (v > best) { best = v; j = c; }
    }
    if (a.use_mask && !((a.cls_mask[j >> 5] >> (j & 31)) & 1u)) return;
    if (best > a.conf) {
      const int ag = a.aoff[lvl] + a0 + an;
      const int slot = atomicAdd(a.count + b, 1);
      // slot < P always holds when the counter was zeroed for this pass (an anchor appends at most once); the test keeps a
      // counter that was NOT reset (see zero_i32_kernel) from writing past the image's key slots
      if (slot < a.P) a.keys[(long)b * a.P + slot] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ag);
      a.cls_idx[(long)b * a.A + ag] = j;
    }
  }
}

// ------------------------------------------------------------------------------------
// Classify tail ([3P] Classify.forward after its Conv): AdaptiveAvgPool2d(1) -> flatten ->
// Linear(c -> nc) -> softmax.  One block per image.
struct ClsHeadArgs {
  const void* feat;    // [B, hw, c] T
  const float* w;      // [nc][c]
  const float* bias;   // [nc]
  float* logits;       // [B, nc] or null
  float* probs;        // [B, nc] or null
  int32_t B, hw, c, nc;
};

template <typename T>
__global__ __launch_bounds__(256) void cls_head_kernel(const ClsHeadArgs a) {
  extern __shared__ float sm[];          // pooled[c] + logits[nc]
  float* pooled = sm;
  float* lg = sm + a.c;
  const int b = blockIdx.x;
  const T* f = reinterpret_cast<const T*>(a.feat) + (long)b * a.hw * a.c;
  for (int ch = threadIdx.x; ch < a.c; ch += 256) {
    float s = 0.f;
    for (int p = 0; p < a.hw; ++p) {
      // fp8 features: the stored e4m3 values; their activation scale is folded into the Linear's columns by the host
      if constexpr (is_fp8<T>::value) s += __builtin_amdgcn_cvt_f32_fp8((int)f[(long)p * a.c + ch].v, 0);
      else s += (float)f[(long)p * a.c + ch];
    }
    pooled[ch] = s / (float)a.hw;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < a.nc; k += 4) {
    const float* wr = a.w + (long)k * a.c;
    float s = 0.f;
    for (int ch = lane; ch < a.c; ch += 64) s = fmaf(pooled[ch], wr[ch], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) lg[k] = s + a.bias[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mx = lg[0];
    for (int k = 1; k < a.nc; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f;
    for (int k = 0; k < a.nc; ++k) s += expf(lg[k] - mx);
    for (int k = 0; k < a.nc; ++k) {
      if (a.logits) a.logits[(long)b * a.nc + k] = lg[k];
      if (a.probs) a.probs[(long)b * a.nc + k] = expf(lg[k] - mx) / s;
    }
  }
}

// The NMS candidate counters are zeroed by a KERNEL, not by hipMemsetAsync: under hipGraph capture (option "graph") every
// node of a captured call is then a kernel node.  Round 2's first-call capture "replayed into a memory access fault", and
// it deferred captures to a key's second call without knowing why that helped.  Round 3: the one non-kernel node of that
// capture was the 4 x B byte hipMemsetAsync of these counters, and launching a graph that holds this memset node is what
// faults on this runtime (ROCm 7.2): with the reset as a kernel, first-call capture and replays of an older graph after a
// newer capture are clean (tests/test_gpu_detect.py::test_detect_hip_graph_replay, census 63 kernel nodes = 63 launches);
// with the memset put back a graph launch faulted again within four calls although candidate slots were range-checked by
// then (profiles/r03_graph_memset_node_fault.log), so it is the node itself, not a counter overflow.  The memset form is
// gone from the library (why round 2's second-sighting order survived with it was not looked into further).
__global__ void zero_i32_kernel(int32_t* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

// ------------------------------------------------------------------------------------
// Debug / parity taps: fp32 <-> activation dtype copies of a whole buffer.
template <typename T>
__global__ void copy_to_f32_kernel(const T* src, float* dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
template <typename T>
__global__ void copy_from_f32_kernel(const float* src, T* dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}

// fp8 buffers: the taps move the stored e4m3 values (UNSCALED: the host multiplies by the buffer's activation scale)
template <>
__global__ void copy_to_f32_kernel<fp8_t>(const fp8_t* src, float* dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = __builtin_amdgcn_cvt_f32_fp8((int)src[i].v, 0);
}
template <>
__global__ void copy_from_f32_kernel<fp8_t>(const float* src, fp8_t* dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i].v = (unsigned char)(pack_fp8x4(src[i], 0.f, 0.f, 0.f) & 0xFFu);
}

}  // namespace miyolo

// miyolo engine: executes the layer program of include/miyolo.h with the gfx950 kernels.
// C ABI only at the boundary; no torch types, no allocation or synchronisation on the hot
// path (the caller passes the workspace and the stream).
#include "../../include/miyolo.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "conv_igemm.h"
#include "conv_dma.h"
#include "conv_dmap.h"
#include "conv_t2d.h"
#include "conv_h2.h"
#include "conv_h3.h"
#include "conv_pw.h"
#include "conv_h4.h"
// Earlier kernel generations / experiments (persistent halo, warp-specialised, two-workgroup, first halo prototype):
// compiled only with `build.sh experiments` (-DMIYOLO_EXPERIMENTS=1); the shipped library carries conv_igemm.h
// (conv_impl 0) and conv_dma.h (1) as the simple bit-exact cross-checks of the default kernels.
#ifndef MIYOLO_EXPERIMENTS
#define MIYOLO_EXPERIMENTS 0
#endif
#if MIYOLO_EXPERIMENTS
#include "conv_halo.h"
#include "conv_halop.h"
#include "conv_ws.h"
#include "conv_dmh.h"
#endif
#include "kernels_misc.h"
#include "cls_mega.h"
#include "conv_bneck.h"
#include "conv_stem2.h"
#include "nms.h"
#include "preprocess.h"
#include "nmm.h"

using namespace miyolo;

namespace {

thread_local std::string g_create_error;     // last error of a call without a handle (create, letterbox, crop_resize), per thread

struct Plan {
  int B = 0, H = 0, W = 0;            // chunk batch and frame size the offsets are valid for
  std::vector<size_t> buf_off;        // byte offset of every activation buffer
  size_t y_off = 0, keys_off = 0, count_off = 0, cls_off = 0, total = 0;
  int A = 0, P = 1;
};

}  // namespace

struct miyolo_engine {
  miyolo_desc desc;
  std::vector<miyolo_buf> bufs;
  std::vector<miyolo_op> ops;
  std::vector<const void*> weights;
  int device = 0;
  int max_chunk = 0;        // 0 = automatic
  int force_wc = 0, force_tc = 0;
  int conv_impl = 3;        // 0: register-staged (conv_igemm.h); 1: LDS-DMA ring (conv_dma.h); 2: 1 + halo kernel for
                            // 3x3 s1 (conv_halo.h); 3: persistent LDS-DMA ring (conv_dmap.h)
  int ncu = 256;
  int graph = 0;            // 1: capture the launches of a classify/detect call into a hipGraph and replay it while the
                            // call's shape, thresholds, stream and pointers stay the same (launch-bound classifier path)
  struct GraphRec { unsigned char key[160]; size_t klen; hipGraph_t g; hipGraphExec_t ex; };
  std::vector<GraphRec> graphs;
  int dmh_auto = 0;         // conv_impl 3: two-workgroup kernel for launches with 1-2 tiles per CU (conv_dmh.h); off since the
                            // balanced grids: +0.2 % without it (same-box A/B), it won only by removing a half-empty round
  int h2 = 1;               // conv_impl 3: 3x3 stride-1 layers on the halo-slab kernel (conv_h2.h) where its tiles cover the map well
  int h2_warm = 0;          // ... 1: persistent form (next tile's slab issued inside the epilogue): measured 1.4 % SLOWER on the step, off
  int h2_min_util = 70;     // ... = pixel utilisation of its 256-pixel tiles, in percent
  int pw = 0;               // 1: conv_impl 3, f16: eligible 1x1 layers on the streaming kernel (conv_pw.h) - round-3 experiment, 6-8 % slower than the ring kernel: off
  int pair8 = 1;            // conv_dmap.h: 16-byte f16 stores over channel-tile pairs (0: 8-byte stores; same results)
  int h4 = 0;               // experiment (conv_h4.h): 3x3 stride-1 f16 layers whose map its 480/512-pixel tiles cover >= h4_min_util % on the one-workgroup-per-CU kernel
  int h4_min_util = 85;
  int h3 = 2;               // conv_impl 3, f16: 3x3 stride-1 layers on the three-workgroups-per-CU form of the halo-slab kernel (conv_h3.h):
                            // 0 never, 1 wherever eligible, 2 (default) where its tile count fills the chip better (h3_preferred)
  int h3_min_util = 75;     // ... and its 128-pixel tiles cover at least this share of the map
  int h3_max_w = 0;         // ... and the map is at most this wide (0: any width)
  int t2d = 1;              // conv_impl 3: narrow 3x3 layers on 16x16 tiles with resident weights (conv_t2d.h)
  int ablate = 0;           // timing experiments (conv_dma.h), never set in production
  unsigned long long* dbg = nullptr;   // MIYOLO_ABLATE: 256*8*8 u64 stamp buffer (last conv launch wins)
  int dbg_op = -1;          // op index whose stamps are wanted
  int cls_mega = 1;         // classify, f16: the whole forward as one launch with LDS-resident activations (cls_mega.h)
  MegaArgs mega;            // layer table of that launch for the cached frame size
  int mega_h = 0, mega_w = 0, mega_ok = 0;
  size_t mega_lds = 0;
  void* mega_wbuf = nullptr;   // conv weights in MFMA fragment order (mega_repack_kernel)
  void* mega_ops_dev = nullptr;   // the layer table in device memory
  std::vector<MegaOp> mega_ops;
  int cls_streams = 1;      // classify: sub-batches on this many internal streams, joined by events.  Measured at batch 256: 1 stream
                            // 1.12 M img/s, 2: 0.85 M, 4: 0.40 M, 8: 0.29 M (captured in a hipGraph: 0.88 / 0.84 / 0.57 / 0.42 M) - the
                            // front end takes ~8 us per dispatch whether or not the chains are independent; only fewer launches help
  long launches = 0;        // kernel launches issued by this handle (run_ops, run_nms, counter reset): the census of a capture
  int graph_nodes = 0, graph_kernel_nodes = 0, graph_launches = 0, graph_rejected = 0;   // of the most recent capture
  std::vector<hipStream_t> lanes;
  int batch_split = 0;      // detect: K > 1 runs a single-chunk batch as K part batches on K streams (measured +0.8..1.4 %, off)
  std::vector<hipStream_t> split_streams;
  std::vector<hipEvent_t> split_ev;   // [0] fork, [k] join of part k
  int stem_fuse = 1;        // f16: the stem and the first stride-2 conv as one launch (conv_stem2.h; round 3: two wave groups a phase apart, 558 -> 342 us)
  int bneck_fuse = 1;       // a narrow Bottleneck's two 3x3 convs as one launch (conv_bneck.h), f16
  int sppf_fuse = 1;        // three chained MAXPOOL5 ops (SPPF) as one launch (sppf3_kernel)
  int fuse_pre = 0;         // set by miyolo_detect around run_ops: the decode op also runs the NMS score filter
  int fuse_pre_opt = 1;     // option "fuse_prefilter"
  float fuse_conf = 0.f;
  int nms_async = 0;        // detect: the NMS of a call on an internal stream, not joined (miyolo_wait_outputs); see miyolo.h
  hipStream_t nms_stream = nullptr;
  hipEvent_t ev_dec = nullptr, ev_nms = nullptr;
  bool nms_pending = false;
  int head_lanes = 1;       // detect: the Detect head's independent conv chains on side streams (build_lanes)
  std::vector<int> op_lane;                 // lane of each op (0 = the caller's stream)
  std::vector<std::vector<int>> op_waits;   // ops on OTHER lanes op i must wait for (latest per lane)
  std::vector<char> op_signal;              // op i has a consumer on another lane: record its event
  std::vector<hipEvent_t> op_ev;
  int n_lanes = 1;
  std::vector<hipEvent_t> lane_ev;   // [0] fork, [1 + i] join of lane i
  uint32_t cls_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // `classes=` filter of the NMS prefilter (miyolo_set_classes)
  int use_cls_mask = 0;
  std::vector<char> buf_stale;   // buffer never written under the current options (its producer runs inside a fused launch and
                                 // keeps it in LDS): miyolo_read_buffer refuses it instead of returning uninitialised memory
  int profile = 0;          // 1: bracket every op launch with hipEvents (bench/roofline only)
  struct ProfRec { int op, cfg; hipEvent_t e0, e1; };
  std::vector<ProfRec> prof;
  std::string err;
  Plan plan;
};

namespace {

int fail(miyolo_engine* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(h, MIYOLO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

inline size_t elem_size(const miyolo_engine* h, const miyolo_buf& b) {
  if (b.dtype == -2) return 1;                       // uint8 input
  if (b.dtype == MIYOLO_F32) return 4;
  return h->desc.dtype == MIYOLO_F16 ? 2 : h->desc.dtype == MIYOLO_F8 ? 1 : 4;        // -1: activation dtype
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T, int KS, int WC, int TC>
hipError_t set_conv_attr() {
  constexpr int BM = (4 / WC) * TP * 16, BN = WC * TC * 16;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, KS, WC, TC>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * ROW_BYTES);
}
template <typename T, int KS>
hipError_t set_conv_attrs_ks() {
  hipError_t e;
  if ((e = set_conv_attr<T, KS, 2, 4>()) != hipSuccess) return e;
  if ((e = set_conv_attr<T, KS, 2, 3>()) != hipSuccess) return e;
  if ((e = set_conv_attr<T, KS, 1, 4>()) != hipSuccess) return e;
  if ((e = set_conv_attr<T, KS, 1, 3>()) != hipSuccess) return e;
  if ((e = set_conv_attr<T, KS, 1, 2>()) != hipSuccess) return e;
  return set_conv_attr<T, KS, 1, 1>();
}

template <typename T, int KS, int WC, int TC>
hipError_t set_dma_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<T, KS, WC, TC>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)dma_lds_bytes<WC, TC>());
}
template <typename T, int KS>
hipError_t set_dma_attrs_ks() {
  hipError_t e;
  if ((e = set_dma_attr<T, KS, 2, 4>()) != hipSuccess) return e;
  if ((e = set_dma_attr<T, KS, 2, 3>()) != hipSuccess) return e;
  if ((e = set_dma_attr<T, KS, 1, 4>()) != hipSuccess) return e;
  if ((e = set_dma_attr<T, KS, 1, 3>()) != hipSuccess) return e;
  if ((e = set_dma_attr<T, KS, 1, 2>()) != hipSuccess) return e;
  return set_dma_attr<T, KS, 1, 1>();
}

template <typename T, int KS>
hipError_t set_dmap_attrs_ks() {
  hipError_t e;
#define MIYOLO_DMAP_ATTR(WC, TC)                                                                        \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dmap_kernel<T, KS, WC, TC>),         \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  MIYOLO_DMAP_ATTR(2, 6) MIYOLO_DMAP_ATTR(2, 4) MIYOLO_DMAP_ATTR(2, 3) MIYOLO_DMAP_ATTR(1, 4) MIYOLO_DMAP_ATTR(1, 3)
  MIYOLO_DMAP_ATTR(1, 2) MIYOLO_DMAP_ATTR(1, 1)
#undef MIYOLO_DMAP_ATTR
  return hipSuccess;
}

template <typename T>
hipError_t set_t2d_attrs() {
  hipError_t e;
#define MIYOLO_T2D_ATTR(TC)                                                                             \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t2d_kernel<T, TC>),                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  MIYOLO_T2D_ATTR(1) MIYOLO_T2D_ATTR(2) MIYOLO_T2D_ATTR(3) MIYOLO_T2D_ATTR(4)
#undef MIYOLO_T2D_ATTR
  return hipSuccess;
}

#if MIYOLO_EXPERIMENTS
template <typename T, int KS>
hipError_t set_dmh_attrs_ks() {
  hipError_t e;
#define MIYOLO_DMH_ATTR(WC, TC)                                                                         \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dmh_kernel<T, KS, WC, TC>),           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)) != hipSuccess) return e;
  MIYOLO_DMH_ATTR(2, 4) MIYOLO_DMH_ATTR(2, 3) MIYOLO_DMH_ATTR(1, 4) MIYOLO_DMH_ATTR(1, 3) MIYOLO_DMH_ATTR(1, 2) MIYOLO_DMH_ATTR(1, 1)
#undef MIYOLO_DMH_ATTR
  return hipSuccess;
}

template <typename T, int KS>
hipError_t set_ws_attrs_ks() {
  hipError_t e;
#define MIYOLO_WS_ATTR(CN, TC)                                                                          \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<T, KS, CN, TC>),            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  MIYOLO_WS_ATTR(2, 4) MIYOLO_WS_ATTR(2, 3) MIYOLO_WS_ATTR(2, 2) MIYOLO_WS_ATTR(1, 4) MIYOLO_WS_ATTR(1, 3)
  MIYOLO_WS_ATTR(1, 2) MIYOLO_WS_ATTR(1, 1)
#undef MIYOLO_WS_ATTR
  return hipSuccess;
}

template <typename T>
hipError_t set_halop_attrs() {
  hipError_t e;
#define MIYOLO_HALOP_ATTR(WC, TC)                                                                      \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halop_kernel<T, WC, TC>),           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  MIYOLO_HALOP_ATTR(2, 4) MIYOLO_HALOP_ATTR(2, 3) MIYOLO_HALOP_ATTR(1, 4) MIYOLO_HALOP_ATTR(1, 3)
  MIYOLO_HALOP_ATTR(1, 2) MIYOLO_HALOP_ATTR(1, 1)
#undef MIYOLO_HALOP_ATTR
  return hipSuccess;
}

template <typename T>
hipError_t set_halo_attrs() {
  hipError_t e;
  const int lds = (int)halo_lds_bytes(8);
#define MIYOLO_HALO_ATTR(WC, TC)                                                                       \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<T, WC, TC>),            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  MIYOLO_HALO_ATTR(2, 4) MIYOLO_HALO_ATTR(2, 3) MIYOLO_HALO_ATTR(1, 4) MIYOLO_HALO_ATTR(1, 3)
  MIYOLO_HALO_ATTR(1, 2) MIYOLO_HALO_ATTR(1, 1)
#undef MIYOLO_HALO_ATTR
  return hipSuccess;
}
#endif  // MIYOLO_EXPERIMENTS

int nms_lds_bytes(int max_det) { return nms_keys_offset(max_det) + kNmsLdsKeys * 8; }

// Largest number of images per pass such that every buffer stays below 2 GiB (the kernels use
// 32-bit byte offsets, and raw-buffer loads use offset 0x80000000 as the "reads zero" marker).
int auto_chunk(const miyolo_engine* h, int B, int H, int W) {
  size_t per_img = 1;
  for (const miyolo_buf& b : h->bufs) {
    const size_t e = (size_t)(H / b.down) * (W / b.down) * b.channels * elem_size(h, b);
    if (e > per_img) per_img = e;
  }
  size_t lim = (((size_t)1 << 31) - 1) / per_img;
  if (lim < 1) lim = 1;
  int c = (int)std::min<size_t>(lim, (size_t)B);
  if (h->max_chunk > 0 && c > h->max_chunk) c = h->max_chunk;
  return c;
}

// sub-batch size of the multi-stream classify path (0: single stream)
int cls_lane_batch(const miyolo_engine* h, int B) {
  const int ns = h->cls_streams;
  if (ns < 2 || h->profile || B < 2 * ns) return 0;
  return (B + ns - 1) / ns;
}

int total_anchors(const miyolo_engine* h, int H, int W) {
  int A = 0;
  for (const miyolo_op& op : h->ops)
    if (op.kind == MIYOLO_OP_DECODE)
      for (int l = 0; l < 3; ++l) A += (H / op.level_stride[l]) * (W / op.level_stride[l]);
  return A;
}

void make_plan(const miyolo_engine* h, int Bc, int H, int W, Plan* p) {
  p->B = Bc; p->H = H; p->W = W;
  p->buf_off.assign(h->bufs.size(), 0);
  size_t off = 0;
  for (size_t i = 1; i < h->bufs.size(); ++i) {   // buffer 0 is the caller's input
    const miyolo_buf& b = h->bufs[i];
    p->buf_off[i] = off;
    off += align_up((size_t)Bc * (H / b.down) * (W / b.down) * b.channels * elem_size(h, b), 256);
  }
  p->A = total_anchors(h, H, W);
  p->P = 1;
  while (p->P < p->A) p->P <<= 1;
  if (h->desc.task == 0) {
    p->y_off = off;     off += align_up((size_t)Bc * (4 + h->desc.nc) * p->A * 4, 256);
    p->keys_off = off;  off += align_up((size_t)Bc * p->P * 8, 256);
    p->count_off = off; off += align_up((size_t)Bc * 4, 256);
    p->cls_off = off;   off += align_up((size_t)Bc * p->A * 4, 256);
  }
  p->total = off;
}

int check_shape(miyolo_engine* h, int B, int H, int W) {
  if (B < 1 || H < 1 || W < 1) return fail(h, MIYOLO_ERR_SHAPE, "bad batch/frame size B=%d H=%d W=%d", B, H, W);
  if (H % h->desc.max_stride || W % h->desc.max_stride)
    return fail(h, MIYOLO_ERR_SHAPE, "H=%d W=%d must be multiples of the model stride %d", H, W, h->desc.max_stride);
  return 0;
}

struct DevGuard {
  int prev = -1; bool sw = false;
  explicit DevGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) { sw = hipSetDevice(dev) == hipSuccess; } }
  ~DevGuard() { if (sw) (void)hipSetDevice(prev); }
};

void* buf_ptr(const miyolo_engine* h, const Plan& p, int buf, const void* in, void* ws) {
  if (buf == 0) return const_cast<void*>(in);
  return static_cast<unsigned char*>(ws) + p.buf_off[buf];
}

template <typename T>
int run_op(miyolo_engine* h, const miyolo_op& op, const Plan& p, const void* in, void* ws,
           float* cls_logits, float* cls_probs, hipStream_t s) {
  const int Bc = p.B, H = p.H, W = p.W;
  switch (op.kind) {
    case MIYOLO_OP_STEM: {
      const miyolo_buf& ob = h->bufs[op.dst.buf];
      StemArgs a;
      a.in = static_cast<const uint8_t*>(in);
      a.w = h->weights[op.weight];
      a.bias = static_cast<const float*>(h->weights[op.bias]);
      a.out = buf_ptr(h, p, op.dst.buf, in, ws);
      a.B = Bc; a.H = H; a.W = W; a.Ho = H / ob.down; a.Wo = W / ob.down;
      a.cout = op.cout; a.act = op.act; a.exact = (h->desc.dtype == MIYOLO_F32);
      if ((size_t)Bc * H * W * 3 >= ((size_t)1 << 31)) return fail(h, MIYOLO_ERR_SHAPE, "input batch exceeds 2 GiB per pass");
      a.in_bytes = (uint32_t)((size_t)Bc * H * W * 3);
      host_magic((uint32_t)(a.Ho * a.Wo), &a.mg_hw_mul, &a.mg_hw_shift);
      host_magic((uint32_t)a.Wo, &a.mg_w_mul, &a.mg_w_shift);
      if (op.cout % 16 || op.cout > 80 || op.dst.ch_off != 0 || ob.channels != op.cout || ob.down != 2)
        return fail(h, MIYOLO_ERR_UNSUPPORTED, "stem: cout=%d must be a multiple of 16 (<= 80) and own its buffer", op.cout);
      const long ntiles = ((long)Bc * a.Ho * a.Wo + 15) / 16;
      const unsigned grid = (unsigned)std::min<long>((ntiles + 3) / 4, 256 * 8);
      a.out_inv_scale = op.out_inv_scale;
      if constexpr (is_fp8<T>::value) {         // fp8 engine: the stem computes in f16 (f16 weights) and stores e4m3
        switch (op.cout / 16) {
          case 1: hipLaunchKernelGGL((stem_kernel<half_t, 1, true>), dim3(grid), dim3(256), 0, s, a); break;
          case 2: hipLaunchKernelGGL((stem_kernel<half_t, 2, true>), dim3(grid), dim3(256), 0, s, a); break;
          case 3: hipLaunchKernelGGL((stem_kernel<half_t, 3, true>), dim3(grid), dim3(256), 0, s, a); break;
          case 4: hipLaunchKernelGGL((stem_kernel<half_t, 4, true>), dim3(grid), dim3(256), 0, s, a); break;
          default: hipLaunchKernelGGL((stem_kernel<half_t, 5, true>), dim3(grid), dim3(256), 0, s, a); break;
        }
      } else {
      switch (op.cout / 16) {
        case 1: hipLaunchKernelGGL((stem_kernel<T, 1>), dim3(grid), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((stem_kernel<T, 2>), dim3(grid), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((stem_kernel<T, 3>), dim3(grid), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((stem_kernel<T, 4>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((stem_kernel<T, 5>), dim3(grid), dim3(256), 0, s, a); break;
      }
      }
      break;
    }
    case MIYOLO_OP_CONV: {
      constexpr int CE = DT<T>::CE;
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      const miyolo_buf& ob = h->bufs[op.dst.buf];
      a.nsrc = op.n_src;
      int down_in = 0, ktot = 0;
      for (int i = 0; i < op.n_src; ++i) {
        const miyolo_view& v = op.src[i];
        const miyolo_buf& sb = h->bufs[v.buf];
        if (v.buf == 0 || sb.dtype != -1) return fail(h, MIYOLO_ERR_UNSUPPORTED, "conv source must be an activation buffer");
        a.src[i].ptr = buf_ptr(h, p, v.buf, in, ws);
        a.src[i].ld = sb.channels; a.src[i].ch_off = v.ch_off; a.src[i].ch_cnt = v.ch_cnt; a.src[i].up = v.upsample;
        a.src[i].h = H / sb.down; a.src[i].w = W / sb.down;
        a.src[i].bytes = (uint32_t)((size_t)Bc * a.src[i].h * a.src[i].w * sb.channels * sizeof(T));
        const int d = v.upsample ? sb.down / 2 : sb.down;
        if (i == 0) down_in = d; else if (d != down_in) return fail(h, MIYOLO_ERR_UNSUPPORTED, "concat of different resolutions");
        if (v.ch_cnt % CE || v.ch_off % CE || sb.channels % CE)
          return fail(h, MIYOLO_ERR_UNSUPPORTED, "conv view channels (%d @%d of %d) must be multiples of %d", v.ch_cnt, v.ch_off, sb.channels, CE);
        ktot += v.ch_cnt;
      }
      if (op.n_src == 1) a.src[1] = a.src[0];
      if (op.n_src == 2 && op.src[0].ch_cnt % (8 * CE))
        return fail(h, MIYOLO_ERR_UNSUPPORTED, "first concat segment (%d ch) must be a multiple of %d channels", op.src[0].ch_cnt, 8 * CE);
      if (ktot != op.cin) return fail(h, MIYOLO_ERR_ARG, "conv cin mismatch");
      if (op.ksize == 3 && (op.n_src != 1 || op.src[0].upsample)) return fail(h, MIYOLO_ERR_UNSUPPORTED, "3x3 conv over concat/upsample");
      if (op.ksize == 1 && op.stride != 1) return fail(h, MIYOLO_ERR_UNSUPPORTED, "strided 1x1 conv");
      if (op.ksize != 1 && op.ksize != 3) return fail(h, MIYOLO_ERR_UNSUPPORTED, "conv kernel size %d", op.ksize);
      a.w = h->weights[op.weight];
      a.bias = static_cast<const float*>(h->weights[op.bias]);
      a.dst = buf_ptr(h, p, op.dst.buf, in, ws);
      a.dst_ld = ob.channels; a.dst_choff = op.dst.ch_off; a.out_f32 = (ob.dtype == MIYOLO_F32);
      a.dst_bytes = (uint32_t)((size_t)Bc * (H / ob.down) * (W / ob.down) * ob.channels * elem_size(h, ob));
      if (op.res.buf >= 0) {
        a.res = buf_ptr(h, p, op.res.buf, in, ws);
        a.res_ld = h->bufs[op.res.buf].channels; a.res_choff = op.res.ch_off;
        a.res_bytes = (uint32_t)((size_t)Bc * (H / h->bufs[op.res.buf].down) * (W / h->bufs[op.res.buf].down) * a.res_ld * sizeof(T));
      }
      a.B = Bc; a.Hin = H / down_in; a.Win = W / down_in; a.Hout = H / ob.down; a.Wout = W / ob.down;
      if (down_in * op.stride != ob.down) return fail(h, MIYOLO_ERR_ARG, "conv resolution mismatch");
      a.cin = op.cin; a.cout = op.cout; a.ksize = op.ksize; a.stride = op.stride; a.act = op.act;
      a.M = Bc * a.Hout * a.Wout;
      const int K = op.cin * op.ksize * op.ksize, BK = 8 * CE;
      a.kpad = (K + BK - 1) / BK * BK; a.nk = a.kpad / BK;
      a.wbytes = (uint32_t)((size_t)op.cout * a.kpad * sizeof(T));
      a.vec_ok = (op.cout % 4 == 0) && (ob.channels % 4 == 0) && (op.dst.ch_off % 4 == 0) &&
                 (op.res.buf < 0 || (h->bufs[op.res.buf].channels % 4 == 0 && op.res.ch_off % 4 == 0));
      a.exact = (h->desc.dtype == MIYOLO_F32);
      if constexpr (is_fp8<T>::value) {
        if (op.qscale < 0 || op.qscale >= (int)h->weights.size() || op.bias_init < 0 || op.bias_init >= (int)h->weights.size() || !a.vec_ok)
          return fail(h, MIYOLO_ERR_ARG, "fp8 conv needs qscale / bias_init weight entries and 4-channel aligned views");
        a.qscale = static_cast<const float*>(h->weights[op.qscale]);
        a.bias_init = static_cast<const float*>(h->weights[op.bias_init]);
        a.out_inv_scale = op.out_inv_scale; a.res_scale = op.res_scale;
      }
      a.ablate = h->ablate;
      a.dbg = (h->dbg && (&op - h->ops.data()) == h->dbg_op) ? h->dbg : nullptr;
      a.res_vec = a.res && (a.res_ld % 4 == 0) && (a.res_choff % 4 == 0) && (op.cout % 4 == 0);
      host_magic((uint32_t)(a.Hout * a.Wout), &a.mg_hw_mul, &a.mg_hw_shift);
      host_magic((uint32_t)a.Wout, &a.mg_w_mul, &a.mg_w_shift);
      if constexpr (is_fp8<T>::value) {         // fp8: the halo-slab kernel where eligible, else the persistent ring kernel
        if (h->conv_impl != 3 && h->conv_impl != 8) return fail(h, MIYOLO_ERR_UNSUPPORTED, "fp8 runs on conv_impl 3 / 8 only");
        if ((h->conv_impl == 8 || (h->h2 && h->force_wc == 0)) && h2_eligible<T>(a, h->conv_impl == 8 ? 0.0 : 0.01 * h->h2_min_util))
          HIP_TRY(h, launch_conv_h2<T>(a, s, h->ncu, h->h2_warm));
        else HIP_TRY(h, launch_conv_dmap<T>(a, s, h->ncu, h->force_wc, h->force_tc, h->pair8));
      } else {
      bool done_h3 = false;
      if constexpr (sizeof(T) == 2) {
        if (h->conv_impl == 9 && h3_eligible(a, 0.0)) { HIP_TRY(h, launch_conv_h3(a, s)); done_h3 = true; }
        else if ((h->conv_impl == 10 && h4_eligible(a, 0.0)) || (h->conv_impl == 3 && h->h4 && h->force_wc == 0 && h4_eligible(a, 0.01 * h->h4_min_util))) {
          HIP_TRY(h, launch_conv_h4(a, s)); done_h3 = true;
        }
      }
      if (done_h3) {}
      else if (h->conv_impl == 8 && h2_eligible<T>(a, 0.0)) HIP_TRY(h, launch_conv_h2<T>(a, s, h->ncu, h->h2_warm));
      // narrow layers: the 2-D-tile kernel where there is a residual (it reads it under its K loop), the halo-slab kernel
      // where there is none (f16, detect models only: the classifier's layered path stays on the kernels whose K order the
      // one-launch classifier reproduces bit for bit; per-layer A/B at 48 channels, 160 x 160: 162 vs 182 us without,
      // 220 / 183 vs 199 / 167 us with)
      else if ((h->conv_impl == 7 || (h->conv_impl == 3 && h->t2d && h->force_wc == 0 &&
                                     !(sizeof(T) == 2 && !a.res && h->desc.task == 0 && h->h2 && h2_eligible<T>(a, 0.01 * h->h2_min_util)))) && t2d_eligible<T>(a))
        HIP_TRY(h, launch_conv_t2d<T>(a, s, h->ncu));
      else if (h->conv_impl == 3 && h->h2 && h->force_wc == 0 && h2_eligible<T>(a, 0.01 * h->h2_min_util)) {
        bool h3 = false;
        if constexpr (sizeof(T) == 2)
          h3 = h->h3 && (h->h3_max_w == 0 || a.Wout <= h->h3_max_w) && h3_eligible(a, 0.01 * h->h3_min_util) && (h->h3 == 1 || h3_preferred(a, h->ncu));
        if constexpr (sizeof(T) == 2) { if (h3) HIP_TRY(h, launch_conv_h3(a, s)); }
        if (!h3) HIP_TRY(h, launch_conv_h2<T>(a, s, h->ncu, h->h2_warm));
      }
#if MIYOLO_EXPERIMENTS
      else if ((h->conv_impl == 6 || (h->conv_impl == 3 && h->dmh_auto && dmh_preferred(a, h->ncu) && h->force_wc == 0)) && dmh_eligible(a))
        HIP_TRY(h, launch_conv_dmh<T>(a, s, h->ncu, h->force_wc, h->force_tc));
      else if (h->conv_impl == 5) HIP_TRY(h, launch_conv_ws<T>(a, s, h->ncu, h->force_wc, h->force_tc));
      else if (h->conv_impl == 4 && halop_eligible(a)) HIP_TRY(h, launch_conv_halop<T>(a, s, h->ncu, h->force_wc, h->force_tc));
      else if (h->conv_impl == 2 && halo_eligible(a)) HIP_TRY(h, launch_conv_halo<T>(a, s, h->force_wc, h->force_tc));
#endif
      else if (sizeof(T) == 2 && h->conv_impl == 3 && h->pw && h->force_wc == 0 && pw_eligible(a)) {
        HIP_TRY(h, launch_conv_pw(a, s, h->ncu));
      }
      else if (h->conv_impl >= 3) HIP_TRY(h, launch_conv_dmap<T>(a, s, h->ncu, h->force_wc, h->force_tc, h->pair8));
      else if (h->conv_impl >= 1) HIP_TRY(h, launch_conv_dma<T>(a, s, h->force_wc, h->force_tc));
      else HIP_TRY(h, launch_conv<T>(a, s, h->force_wc, h->force_tc));
      }
      break;
    }
    case MIYOLO_OP_MAXPOOL5: {
      constexpr int CE = DT<T>::CE;
      const miyolo_buf& sb = h->bufs[op.src[0].buf];
      const miyolo_buf& ob = h->bufs[op.dst.buf];
      PoolArgs a;
      a.src = buf_ptr(h, p, op.src[0].buf, in, ws); a.dst = buf_ptr(h, p, op.dst.buf, in, ws);
      a.src_ld = sb.channels; a.src_choff = op.src[0].ch_off; a.dst_ld = ob.channels; a.dst_choff = op.dst.ch_off;
      a.ch = op.src[0].ch_cnt; a.B = Bc; a.H = H / sb.down; a.W = W / sb.down;
      if (a.ch % CE || a.src_choff % CE || a.dst_choff % CE || a.src_ld % CE || a.dst_ld % CE)
        return fail(h, MIYOLO_ERR_UNSUPPORTED, "maxpool channel slice must be a multiple of %d", CE);
      const long total = (long)Bc * a.H * a.W * (a.ch / CE);
      hipLaunchKernelGGL(maxpool5_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
      break;
    }
    case MIYOLO_OP_DECODE: {
      DecodeArgs a;
      memset(&a, 0, sizeof(a));
      a.nlevel = 3; a.nc = h->desc.nc; a.A = p.A; a.B = Bc;
      int aoff = 0, nblk = 0;
      for (int l = 0; l < 3; ++l) {
        const miyolo_buf& rb = h->bufs[op.src[l].buf];
        if (rb.dtype != MIYOLO_F32 || rb.channels != 64 + a.nc || h->desc.reg_max != 16)
          return fail(h, MIYOLO_ERR_UNSUPPORTED, "decode expects fp32 raw maps of 64+nc channels (reg_max 16)");
        a.raw[l] = static_cast<const float*>(buf_ptr(h, p, op.src[l].buf, in, ws));
        a.lh[l] = H / rb.down; a.lw[l] = W / rb.down; a.lstride[l] = op.level_stride[l]; a.aoff[l] = aoff;
        aoff += a.lh[l] * a.lw[l];
        nblk += (a.lh[l] * a.lw[l] + 63) / 64;
      }
      a.y = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + p.y_off);
      if (h->fuse_pre) {
        unsigned char* w8 = static_cast<unsigned char*>(ws);
        a.keys = reinterpret_cast<unsigned long long*>(w8 + p.keys_off);
        a.count = reinterpret_cast<int32_t*>(w8 + p.count_off);
        a.cls_idx = reinterpret_cast<int32_t*>(w8 + p.cls_off);
        a.P = p.P; a.conf = h->fuse_conf; a.use_mask = h->use_cls_mask;
        for (int i = 0; i < 8; ++i) a.cls_mask[i] = h->cls_mask[i];
      }
      const size_t lds = (size_t)64 * (64 + a.nc + 1) * 4;
      hipLaunchKernelGGL(decode_kernel, dim3(nblk, Bc), dim3(256), lds, s, a);
      break;
    }
    case MIYOLO_OP_CLS_HEAD: {
      {
      const miyolo_buf& sb = h->bufs[op.src[0].buf];
      ClsHeadArgs a;
      a.feat = buf_ptr(h, p, op.src[0].buf, in, ws);
      a.w = static_cast<const float*>(h->weights[op.weight]);
      a.bias = static_cast<const float*>(h->weights[op.bias]);
      a.logits = cls_logits; a.probs = cls_probs;
      a.B = Bc; a.hw = (H / sb.down) * (W / sb.down); a.c = op.cin; a.nc = op.cout;
      if (op.src[0].ch_off != 0 || sb.channels != op.cin) return fail(h, MIYOLO_ERR_UNSUPPORTED, "cls head must read a whole buffer");
      hipLaunchKernelGGL(cls_head_kernel<T>, dim3(Bc), dim3(256), (size_t)(a.c + a.nc) * 4, s, a);
      }
      break;
    }
    default:
      return fail(h, MIYOLO_ERR_UNSUPPORTED, "unknown op kind %d", op.kind);
  }
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// pw_eligible (conv_pw.h) from the op description alone (conv_cfg_id has no ConvArgs)
bool pw_op_eligible(const miyolo_engine* h, const miyolo_op& op) {
  const miyolo_buf& ob = h->bufs[op.dst.buf];
  if (op.ksize != 1 || op.stride != 1 || op.res.buf >= 0 || ob.dtype == MIYOLO_F32 || op.cout % 96) return false;
  if (op.n_src == 2 && (op.src[0].ch_cnt % 64 || op.src[1].ch_cnt % 32)) return false;
  if (op.n_src == 1 && op.src[0].ch_cnt % 32) return false;
  for (int i = 0; i < op.n_src; ++i)
    if (h->bufs[op.src[i].buf].channels % 8 || op.src[i].ch_off % 8) return false;
  return ob.channels % 8 == 0 && op.dst.ch_off % 8 == 0;
}

int conv_cfg_id(const miyolo_engine* h, const miyolo_op& op, const Plan& p) {
  if (op.kind != MIYOLO_OP_CONV) return 0;
  const miyolo_buf& ob = h->bufs[op.dst.buf];
  const long M = (long)p.B * (p.H / ob.down) * (p.W / ob.down);
  T2dGeom tg; size_t tlds;
  const bool t2d = (h->conv_impl == 7 || (h->conv_impl == 3 && h->t2d && h->force_wc == 0)) && op.ksize == 3 && op.stride == 1 &&
                   op.n_src == 1 && !op.src[0].upsample &&
                   h->desc.dtype != MIYOLO_F8 &&
                   t2d_shape(op.cin, op.cout, p.B, p.H / ob.down, p.W / ob.down, h->desc.dtype == MIYOLO_F16 ? 2 : 4, h->desc.dtype == MIYOLO_F16 ? 8 : 4, &tg, &tlds);
  const bool s1 = op.ksize == 3 && op.stride == 1 && op.n_src == 1 && !op.src[0].upsample && ob.dtype != MIYOLO_F32 && op.cout % 8 == 0;
  const bool h2_first = h->desc.dtype == MIYOLO_F16 && op.res.buf < 0 && h->desc.task == 0;   // as run_op: narrow layers without a residual
  if (s1 && (h->conv_impl == 10 || (h->conv_impl == 3 && h->h4 && h->force_wc == 0)) && h->desc.dtype == MIYOLO_F16) {
    H2Geom g4; size_t l4; int geo4;
    const int tc = h2_pick_tc(op.cout, 2);
    if ((tc == 6 || tc == 4) && h4_shape(op.cin, op.cout, p.B, p.H / ob.down, p.W / ob.down, tc, &g4, &l4, &geo4) &&
        (h->conv_impl == 10 || h4_util(g4, p.H / ob.down, p.W / ob.down) >= 0.01 * h->h4_min_util))
      return 8000 + 300 + 60 + tc;                          // conv_h4_kernel<TC>
  }
  if (s1 && h->conv_impl == 9 && h->desc.dtype == MIYOLO_F16) {
    H2Geom g3; size_t l3; int geo3;
    const int tc = h2_pick_tc(op.cout, 2);
    if (h3_shape(op.cin, op.cout, p.B, p.H / ob.down, p.W / ob.down, tc, &g3, &l3, &geo3)) return 8000 + 300 + 50 + tc;
  }
  if (s1 && (h->conv_impl == 8 || (h->conv_impl == 3 && h->h2 && h->force_wc == 0 && (!t2d || h2_first)))) {
    H2Geom hg; size_t hl; int hgeo;
    const int es = h->desc.dtype == MIYOLO_F16 ? 2 : h->desc.dtype == MIYOLO_F8 ? 1 : 4, tc = h2_pick_tc(op.cout, es);
    if (h2_shape(op.cin, op.cout, p.B, p.H / ob.down, p.W / ob.down, es, 16 / es, tc, &hg, &hl, &hgeo) &&
        (h->conv_impl == 8 || h2_util(hg, p.H / ob.down, p.W / ob.down) >= 0.01 * h->h2_min_util)) {
      H2Geom g3; size_t l3; int geo3;
      auto eff = [](int tiles, int slots) { return (double)tiles / ((double)slots * ((tiles + slots - 1) / slots)); };
      if (es == 2 && h->conv_impl == 3 && h->h3 && (h->h3_max_w == 0 || p.W / ob.down <= h->h3_max_w) && h3_shape(op.cin, op.cout, p.B, p.H / ob.down, p.W / ob.down, tc, &g3, &l3, &geo3) &&
          h3_util(g3, p.H / ob.down, p.W / ob.down) >= 0.01 * h->h3_min_util &&
          (h->h3 == 1 || eff(g3.ntiles, 3 * h->ncu) >= eff(hg.ntiles, 2 * h->ncu) + 0.15))
        return 8000 + 300 + 50 + tc;                        // conv_h3_kernel<TC>
      return 8000 + 300 + 40 + tc;                          // conv_h2_kernel<T,TC>
    }
  }
  if (t2d) return 7000 + 300 + 10 + (op.cout + 15) / 16;   // conv_t2d_kernel<T,TC>
  if (op.ksize == 1 && h->conv_impl == 3 && h->pw && h->force_wc == 0 && h->desc.dtype == MIYOLO_F16 && pw_op_eligible(h, op))
    return 10000 + 100 + pw_nt(op.cout);                    // conv_pw_kernel<NT,NB>: 10112 / 10106
  int impl = h->conv_impl >= 3 ? 3 : (h->conv_impl >= 1 ? 1 : 0);
  ConvCfg c = impl == 3 ? pick_dmap_cfg(op.cout, M, h->ncu, op.ksize) : impl == 1 ? pick_dma_cfg(op.cout, M) : pick_conv_cfg(op.cout, M);
#if MIYOLO_EXPERIMENTS
  {
    const bool halo = h->conv_impl == 2 && op.ksize == 3 && op.stride == 1 && op.n_src == 1 && !op.src[0].upsample &&
                      halo_xi(p.W / ob.down) <= 8;
    const bool halop = h->conv_impl == 4 && op.ksize == 3 && op.stride == 1 && op.n_src == 1 && !op.src[0].upsample &&
                       (p.W / ob.down) <= 95 && op.cout % 4 == 0;
    const int bk = 8 * (h->desc.dtype == MIYOLO_F16 ? 8 : 4);
    const size_t dmh_lds = dmh_lds_bytes() + (size_t)((op.cin * op.ksize * op.ksize + bk - 1) / bk) * 32;
    const bool dmh_auto = h->conv_impl == 3 && h->dmh_auto && h->force_wc == 0 && dmh_preferred_shape(op.cout, M, h->ncu) && dmh_lds <= 80 * 1024;
    impl = (h->conv_impl == 6 || dmh_auto) ? 6 : h->conv_impl == 5 ? 5 : halop ? 4 : h->conv_impl >= 3 ? 3 : halo ? 2 : (h->conv_impl >= 1 ? 1 : 0);
    c = impl == 6 ? pick_dma_cfg(op.cout, M) : impl == 5 ? pick_ws_cfg(op.cout, M) : impl == 4 ? pick_halop_cfg(op.cout, M) : impl == 3 ? pick_dmap_cfg(op.cout, M, h->ncu, op.ksize) : impl == 2 ? pick_halo_cfg(op.cout, M)
        : impl == 1 ? pick_dma_cfg(op.cout, M) : pick_conv_cfg(op.cout, M);
  }
#endif
  if (h->force_wc > 0 && h->force_tc > 0) c = {h->force_wc, h->force_tc};
  return impl * 1000 + op.ksize * 100 + c.wc * 10 + c.tc;   // e.g. 3323 = conv_dmap_kernel<T,3,2,3>
}

// Which ops may run beside each other.  Dependencies are read off the views (RAW, WAR, WAW on overlapping channel ranges of
// one buffer).  Ops whose results only the decode op consumes - the Detect head: per level one (fused) first conv and two
// conv chains - leave the caller's stream: each chain gets a side stream, forked by an event behind its producer and
// joined in front of the decode.  The P3 chains then run beside the P4 / P5 neck, and the small P5 kernels (20 x 20 maps,
// too few tiles for 256 CUs) beside each other.  Same kernels, same arguments: results cannot change.
void build_lanes(miyolo_engine* h) {
  const int n = (int)h->ops.size();
  h->op_lane.assign(n, 0); h->op_waits.assign(n, {}); h->op_signal.assign(n, 0); h->n_lanes = 1;
  int D = -1;
  for (int i = 0; i < n; ++i) if (h->ops[i].kind == MIYOLO_OP_DECODE) D = i;
  if (h->desc.task != 0 || D < 0) return;
  auto overlap = [](const miyolo_view& a, const miyolo_view& b) {
    return a.buf > 0 && a.buf == b.buf && a.ch_off < b.ch_off + b.ch_cnt && b.ch_off < a.ch_off + a.ch_cnt;
  };
  auto reads = [&](const miyolo_op& op, std::vector<miyolo_view>* v) {
    v->clear();
    const int ns = op.kind == MIYOLO_OP_CONV ? op.n_src : op.kind == MIYOLO_OP_DECODE ? 3 : 1;
    for (int k = 0; k < ns; ++k) v->push_back(op.src[k]);
    if (op.kind == MIYOLO_OP_CONV && op.res.buf >= 0) v->push_back(op.res);
  };
  std::vector<std::vector<int>> raw(n), any(n);          // predecessors: true data flow / every ordering constraint
  std::vector<miyolo_view> ri, rj;
  for (int j = 0; j < n; ++j) {
    reads(h->ops[j], &rj);
    for (int i = 0; i < j; ++i) {
      reads(h->ops[i], &ri);
      const bool wi = h->ops[i].kind != MIYOLO_OP_DECODE, wj = h->ops[j].kind != MIYOLO_OP_DECODE;   // decode writes y, not a buffer
      bool r = false, o = false;
      for (const auto& v : rj) if (wi && overlap(v, h->ops[i].dst)) r = true;
      for (const auto& v : ri) if (wj && overlap(v, h->ops[j].dst)) o = true;
      if (wi && wj && overlap(h->ops[i].dst, h->ops[j].dst)) o = true;
      if (r) raw[j].push_back(i);
      if (r || o) any[j].push_back(i);
    }
  }
  std::vector<char> head(n, 0);
  for (int i = D - 1; i >= 0; --i) {
    bool used = false, only_head = true;
    for (int j = i + 1; j <= D; ++j)
      for (int pi : raw[j]) if (pi == i) { used = true; if (j != D && !head[j]) only_head = false; }
    head[i] = used && only_head;
  }
  std::vector<char> handed(n, 0);
  for (int i = 0; i < D; ++i) {
    if (!head[i]) continue;
    int lane = 0;
    for (int pi : raw[i]) if (head[pi] && !handed[pi]) { lane = h->op_lane[pi]; handed[pi] = 1; break; }
    if (lane == 0) lane = h->n_lanes++;
    h->op_lane[i] = lane;
  }
  for (int j = 0; j < n; ++j) {
    std::vector<int> latest(h->n_lanes, -1);
    for (int pi : any[j]) if (h->op_lane[pi] != h->op_lane[j]) latest[h->op_lane[pi]] = std::max(latest[h->op_lane[pi]], pi);
    for (int l = 0; l < h->n_lanes; ++l) if (latest[l] >= 0) { h->op_waits[j].push_back(latest[l]); h->op_signal[latest[l]] = 1; }
  }
}

// ops i, i+1, i+2 = SPPF's pool chain (each pools the previous one's output, same width, same map)?  Then one launch.
template <typename T>
bool try_sppf3(miyolo_engine* h, int i, int last, const Plan& p, const void* in, void* ws, hipStream_t s) {
  constexpr int CE = DT<T>::CE;
  if (!h->sppf_fuse || i + 2 >= last) return false;
  const miyolo_op* o = &h->ops[i];
  for (int k = 0; k < 3; ++k) if (o[k].kind != MIYOLO_OP_MAXPOOL5) return false;
  auto same = [](const miyolo_view& a, const miyolo_view& b) { return a.buf == b.buf && a.ch_off == b.ch_off && a.ch_cnt == b.ch_cnt; };
  if (!same(o[1].src[0], o[0].dst) || !same(o[2].src[0], o[1].dst) || o[0].src[0].ch_cnt != o[0].dst.ch_cnt) return false;
  const miyolo_buf& sb = h->bufs[o[0].src[0].buf];
  Sppf3Args a;
  a.src = buf_ptr(h, p, o[0].src[0].buf, in, ws);
  a.src_ld = sb.channels; a.src_choff = o[0].src[0].ch_off; a.ch = o[0].src[0].ch_cnt;
  a.B = p.B; a.H = p.H / sb.down; a.W = p.W / sb.down;
  if (a.ch % CE || a.src_choff % CE || a.src_ld % CE) return false;
  for (int k = 0; k < 3; ++k) {
    const miyolo_buf& ob = h->bufs[o[k].dst.buf];
    if (ob.down != sb.down || o[k].dst.ch_off % CE || ob.channels % CE) return false;
    a.dst[k] = buf_ptr(h, p, o[k].dst.buf, in, ws); a.dst_ld[k] = ob.channels; a.dst_choff[k] = o[k].dst.ch_off;
  }
  const size_t lds = (size_t)2 * a.H * a.W * 16;
  if (lds > 64 * 1024) return false;
  hipLaunchKernelGGL(sppf3_kernel<T>, dim3((unsigned)(a.ch / CE), (unsigned)a.B), dim3(256), lds, s, a);
  return true;
}

// ops i (the stem) and i+1 (conv3x3 stride 2 on the stem's output, which nobody else reads)?  Then one launch (conv_stem2.h).
// If op i+2 is a 1x1 conv C1 -> C1 on layer 1's whole output (a C2f's cv1) and nobody else reads that output, it joins the
// launch (layer 1's activations are its MFMA operand as they stand): returns the number of ops absorbed (0: no fusion).
int try_stem2(miyolo_engine* h, int i, int last, const Plan& p, const void* in, void* ws, hipStream_t s, bool allow3) {
  if (!h->stem_fuse || h->desc.dtype != MIYOLO_F16 || i + 1 >= last) return 0;
  const miyolo_op& o0 = h->ops[i];
  const miyolo_op& o1 = h->ops[i + 1];
  if (o0.kind != MIYOLO_OP_STEM || o1.kind != MIYOLO_OP_CONV || o1.ksize != 3 || o1.stride != 2 || o1.n_src != 1 || o1.src[0].upsample || o1.res.buf >= 0) return 0;
  const miyolo_buf& sb = h->bufs[o0.dst.buf];
  const miyolo_buf& yb1 = h->bufs[o1.dst.buf];
  if (sb.dtype != -1 || yb1.dtype != -1 || sb.down != 2 || yb1.down != 4 || sb.channels != o0.cout || o0.dst.ch_off != 0) return 0;
  if (o1.src[0].buf != o0.dst.buf || o1.src[0].ch_off != 0 || o1.src[0].ch_cnt != o0.cout || o1.cin != o0.cout) return 0;
  auto read_only_by = [&](int buf, int reader) {              // is `buf` read by op `reader` only?
    for (int k = 0; k < (int)h->ops.size(); ++k) {
      if (k == reader) continue;
      const miyolo_op& o = h->ops[k];
      const int ns = o.kind == MIYOLO_OP_CONV ? o.n_src : o.kind == MIYOLO_OP_DECODE ? 3 : 1;
      if (o.kind != MIYOLO_OP_STEM) for (int q = 0; q < ns; ++q) if (o.src[q].buf == buf) return false;
      if (o.kind == MIYOLO_OP_CONV && o.res.buf == buf) return false;
    }
    return true;
  };
  if (!read_only_by(o0.dst.buf, i + 1)) return 0;
  bool has3 = false;
  if (allow3 && h->stem_fuse >= 1 && h->stem_fuse != 2 && i + 2 < last) {        // option stem_fuse = 2: two ops only
    const miyolo_op& o2 = h->ops[i + 2];
    const miyolo_buf& y2 = h->bufs[o2.dst.buf];
    has3 = o2.kind == MIYOLO_OP_CONV && o2.ksize == 1 && o2.stride == 1 && o2.n_src == 1 && !o2.src[0].upsample && o2.res.buf < 0 &&
           o2.src[0].buf == o1.dst.buf && o2.src[0].ch_off == 0 && o2.src[0].ch_cnt == o1.cout && o2.cin == o1.cout && o2.cout == o1.cout &&
           o1.dst.ch_off == 0 && yb1.channels == o1.cout && o1.cout % 32 == 0 &&
           y2.dtype == -1 && y2.down == 4 && y2.channels % 8 == 0 && o2.dst.ch_off % 8 == 0 && read_only_by(o1.dst.buf, i + 2);
  }
  const miyolo_op& od = has3 ? h->ops[i + 2] : o1;
  const miyolo_buf& yb = h->bufs[od.dst.buf];
  size_t lds;
  if (!stem2_shape_ok(o0.cout, o1.cout, p.H, p.W, has3, &lds)) return 0;
  if (yb.channels % 8 || od.dst.ch_off % 8) return 0;           // 16-byte stores
  const size_t inb = (size_t)p.B * p.H * p.W * 3, yb_bytes = (size_t)p.B * (p.H / 4) * (p.W / 4) * yb.channels * 2;
  if (inb >= ((size_t)1 << 31) || yb_bytes >= ((size_t)1 << 31)) return 0;
  Stem2Args a;
  memset(&a, 0, sizeof(a));
  a.in = static_cast<const uint8_t*>(in); a.in_bytes = (uint32_t)inb;
  a.w0 = h->weights[o0.weight]; a.b0 = static_cast<const float*>(h->weights[o0.bias]);
  a.w1 = h->weights[o1.weight]; a.b1 = static_cast<const float*>(h->weights[o1.bias]);
  if (has3) {
    const miyolo_op& o2 = h->ops[i + 2];
    a.w2 = h->weights[o2.weight]; a.b2 = static_cast<const float*>(h->weights[o2.bias]);
    a.kpad2 = (o2.cin + 63) / 64 * 64; a.act2 = o2.act;
  }
  a.dst = buf_ptr(h, p, od.dst.buf, in, ws); a.dst_bytes = (uint32_t)yb_bytes; a.dst_ld = yb.channels; a.dst_choff = od.dst.ch_off;
  a.kpad = (9 * o0.cout + 63) / 64 * 64;
  a.B = p.B; a.H = p.H; a.W = p.W; a.act0 = o0.act; a.act1 = o1.act;
  if (launch_conv_stem2(a, o0.cout, o1.cout, has3, s, h->ncu) != hipSuccess) return 0;
  return has3 ? 2 : 1;
}

// ops i, i+1 = a narrow Bottleneck (conv3x3 -> conv3x3 + residual of the first one's input, C -> C -> C channels, the
// intermediate buffer read by nobody else)?  Then one launch (conv_bneck.h).  f16 only.
bool try_bneck(miyolo_engine* h, int i, int last, const Plan& p, const void* in, void* ws, hipStream_t s) {
  if (!h->bneck_fuse || h->desc.dtype != MIYOLO_F16 || i + 1 >= last) return false;
  const miyolo_op& o1 = h->ops[i];
  const miyolo_op& o2 = h->ops[i + 1];
  auto same = [](const miyolo_view& a, const miyolo_view& b) { return a.buf == b.buf && a.ch_off == b.ch_off && a.ch_cnt == b.ch_cnt; };
  auto c3 = [](const miyolo_op& o) { return o.kind == MIYOLO_OP_CONV && o.ksize == 3 && o.stride == 1 && o.n_src == 1 && !o.src[0].upsample; };
  if (!c3(o1) || !c3(o2) || o1.res.buf >= 0 || o2.res.buf < 0) return false;
  const int C = o1.cin;
  if (o1.cout != C || o2.cin != C || o2.cout != C) return false;
  if (!same(o2.src[0], o1.dst) || !same(o2.res, o1.src[0]) || o1.src[0].buf <= 0) return false;
  const miyolo_buf& xb = h->bufs[o1.src[0].buf];
  const miyolo_buf& tb = h->bufs[o1.dst.buf];
  const miyolo_buf& yb = h->bufs[o2.dst.buf];
  if (xb.dtype != -1 || tb.dtype != -1 || yb.dtype != -1 || xb.down != tb.down || xb.down != yb.down) return false;
  if (tb.channels != C || o1.dst.ch_off != 0) return false;                 // the intermediate owns its buffer ...
  for (int k = 0; k < (int)h->ops.size(); ++k) {                            // ... and only the second conv reads it
    if (k == i + 1) continue;
    const miyolo_op& o = h->ops[k];
    const int ns = o.kind == MIYOLO_OP_CONV ? o.n_src : o.kind == MIYOLO_OP_DECODE ? 3 : 1;
    for (int q = 0; q < ns; ++q) if (o.src[q].buf == o1.dst.buf) return false;
    if (o.kind == MIYOLO_OP_CONV && o.res.buf == o1.dst.buf) return false;
    if (k != i && o.kind != MIYOLO_OP_DECODE && o.kind != MIYOLO_OP_CLS_HEAD && o.dst.buf == o1.dst.buf) return false;
  }
  const int H = p.H / xb.down, W = p.W / xb.down;
  size_t lds;
  if (!bneck_shape_ok(C, H, W, &lds)) return false;
  if (xb.channels % 8 || o1.src[0].ch_off % 8 || yb.channels % 4 || o2.dst.ch_off % 4) return false;
  const size_t xbytes = (size_t)p.B * H * W * xb.channels * 2, ybytes = (size_t)p.B * H * W * yb.channels * 2;
  if (xbytes >= ((size_t)1 << 31) || ybytes >= ((size_t)1 << 31)) return false;
  BneckArgs a;
  memset(&a, 0, sizeof(a));
  a.x = buf_ptr(h, p, o1.src[0].buf, in, ws); a.x_bytes = (uint32_t)xbytes; a.x_ld = xb.channels; a.x_choff = o1.src[0].ch_off;
  a.dst = buf_ptr(h, p, o2.dst.buf, in, ws); a.dst_bytes = (uint32_t)ybytes; a.dst_ld = yb.channels; a.dst_choff = o2.dst.ch_off;
  a.w1 = h->weights[o1.weight]; a.w2 = h->weights[o2.weight];
  a.b1 = static_cast<const float*>(h->weights[o1.bias]); a.b2 = static_cast<const float*>(h->weights[o2.bias]);
  a.kpad = (9 * C + 63) / 64 * 64;
  a.B = p.B; a.H = H; a.W = W; a.act1 = o1.act; a.act2 = o2.act;
  return launch_conv_bneck(a, C, s, h->ncu) == hipSuccess;
}

int run_ops(miyolo_engine* h, int first, int last, const Plan& p, const void* in, void* ws,
            float* cls_logits, float* cls_probs, hipStream_t s) {
  // under graph capture a call keeps everything on the caller's stream: round 2 saw the event fork / join across streams
  // replay an almost empty graph (with_graph's census would now reject such a capture; not pursued - graphs do not speed
  // the detect step up, its launches already run back to back)
  const bool lanes = h->head_lanes && !h->graph && !h->profile && h->n_lanes > 1 && first == 0 && last == (int)h->ops.size();
  if (lanes) {
    while ((int)h->lanes.size() < h->n_lanes - 1) {
      hipStream_t st;
      HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      h->lanes.push_back(st);
    }
    while (h->op_ev.size() < h->ops.size()) {
      hipEvent_t ev;
      HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      h->op_ev.push_back(ev);
    }
  }
  for (int i = first; i < last; ++i) {
    miyolo_engine::ProfRec rec{i, 0, nullptr, nullptr};
    hipStream_t si = s;
    if (lanes) {
      if (h->op_lane[i] > 0) si = h->lanes[h->op_lane[i] - 1];
      for (int pi : h->op_waits[i]) HIP_TRY(h, hipStreamWaitEvent(si, h->op_ev[pi], 0));
    }
    if (h->nms_pending && h->ops[i].kind == MIYOLO_OP_DECODE) HIP_TRY(h, hipStreamWaitEvent(si, h->ev_nms, 0));   // it still reads y
    if (h->profile) {
      rec.cfg = conv_cfg_id(h, h->ops[i], p);
      HIP_TRY(h, hipEventCreate(&rec.e0));
      HIP_TRY(h, hipEventCreate(&rec.e1));
      HIP_TRY(h, hipEventRecord(rec.e0, s));
    }
    int fusedn = 0;
    bool stem_fused = false;
    if (h->ops[i].kind == MIYOLO_OP_STEM && i + 1 < last && (!lanes || (h->op_lane[i] == h->op_lane[i + 1] && h->op_waits[i + 1].empty())) &&
        (fusedn = try_stem2(h, i, last, p, in, ws, si,
                            i + 2 < last && (!lanes || (h->op_lane[i] == h->op_lane[i + 2] && h->op_waits[i + 2].empty())))) > 0) {
      rec.cfg = 9000 + (fusedn == 2 ? 500 : 400) + h->ops[i + 1].cout / 16;        // conv_stem2_kernel (95xx: + the 1x1 conv behind layer 1)
      stem_fused = true;
    } else if (h->ops[i].kind == MIYOLO_OP_CONV && h->ops[i].ksize == 3 && i + 1 < last &&
        (!lanes || (h->op_lane[i] == h->op_lane[i + 1] && h->op_waits[i + 1].empty())) && try_bneck(h, i, last, p, in, ws, si)) {
      fusedn = 1;
      rec.cfg = 9000 + 300 + h->ops[i].cout / 16;           // conv_bneck_kernel<TC>
    }
    else if (h->ops[i].kind == MIYOLO_OP_MAXPOOL5 && h->desc.dtype != MIYOLO_F8 && i + 2 < last &&
        (!lanes || (h->op_lane[i] == h->op_lane[i + 1] && h->op_lane[i] == h->op_lane[i + 2] && h->op_waits[i + 1].empty() && h->op_waits[i + 2].empty())))
      fusedn = ((h->desc.dtype == MIYOLO_F16) ? try_sppf3<half_t>(h, i, last, p, in, ws, si) : try_sppf3<float>(h, i, last, p, in, ws, si)) ? 2 : 0;
    const int rc = fusedn ? 0
                   : (h->desc.dtype == MIYOLO_F8) ? run_op<fp8_t>(h, h->ops[i], p, in, ws, cls_logits, cls_probs, si)
                   : (h->desc.dtype == MIYOLO_F16)
                       ? run_op<half_t>(h, h->ops[i], p, in, ws, cls_logits, cls_probs, si)
                       : run_op<float>(h, h->ops[i], p, in, ws, cls_logits, cls_probs, si);
    if (rc) return rc;
    ++h->launches;                                         // every op (fused or not) is exactly one kernel launch
    if (h->ops[i].kind == MIYOLO_OP_STEM || h->ops[i].kind == MIYOLO_OP_CONV) {
      // the stem+conv and Bottleneck fusions (one absorbed op) keep the FIRST op's output in LDS: that buffer is never written
      const bool kept_in_lds = fusedn == 1 || stem_fused;
      h->buf_stale[h->ops[i].dst.buf] = kept_in_lds;
      if (kept_in_lds) {
        for (int k = 1; k < fusedn; ++k) h->buf_stale[h->ops[i + k].dst.buf] = 1;      // stem + layer 1 + 1x1: layer 1's map too
        h->buf_stale[h->ops[i + fusedn].dst.buf] = 0;
      }
    }
    for (int k = 0; k < fusedn; ++k, ++i)                  // the two absorbed pools: same lane, their events mean the same launch
      if (lanes && h->op_signal[i]) HIP_TRY(h, hipEventRecord(h->op_ev[i], si));
    if (lanes && h->op_signal[i]) HIP_TRY(h, hipEventRecord(h->op_ev[i], si));
    if (h->profile) {
      HIP_TRY(h, hipEventRecord(rec.e1, s));
      h->prof.push_back(rec);
    }
  }
  return 0;
}

int run_nms(miyolo_engine* h, const Plan& p, const float* y, int Bc, int A, float conf, float iou, int agnostic,
            int max_det, const float* scale, float* out_dets, int32_t* out_counts, int32_t* out_anchor,
            void* ws, hipStream_t s, bool prefiltered = false) {
  NmsArgs a;
  a.y = y; a.B = Bc; a.A = A; a.nc = h->desc.nc; a.max_det = max_det; a.agnostic = agnostic; a.P = p.P;
  a.conf = conf; a.iou = iou; a.scale = scale;
  unsigned char* w = static_cast<unsigned char*>(ws);
  a.keys = reinterpret_cast<unsigned long long*>(w + p.keys_off);
  a.count = reinterpret_cast<int32_t*>(w + p.count_off);
  a.cls_idx = reinterpret_cast<int32_t*>(w + p.cls_off);
  a.out_dets = out_dets; a.out_counts = out_counts; a.out_anchor = out_anchor;
  for (int i = 0; i < 8; ++i) a.cls_mask[i] = h->cls_mask[i];
  a.use_mask = h->use_cls_mask;
  if (!prefiltered) {       // else: the decode op filled keys / count / cls_idx (decode_kernel's fused filter)
    hipLaunchKernelGGL(zero_i32_kernel, dim3((Bc + 255) / 256), dim3(256), 0, s, a.count, Bc);      // a kernel, never a memset node
    ++h->launches;
    ++h->launches;
    hipLaunchKernelGGL(nms_prefilter_kernel, dim3((A + 255) / 256, Bc), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(nms_sort_greedy_kernel, dim3(Bc), dim3(kNmsThreads), (size_t)nms_lds_bytes(max_det), s, a);
  ++h->launches;
  HIP_TRY(h, hipGetLastError());
  return 0;
}

// an asynchronous NMS of an earlier detect call may still use y and the NMS scratch: order this stream behind it
int join_pending_nms(miyolo_engine* h, hipStream_t s) {
  if (h->nms_pending) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_nms, 0));
  return 0;
}

int prepare(miyolo_engine* h, int B, int H, int W, size_t ws_bytes, void* ws) {
  if (int rc = check_shape(h, B, H, W)) return rc;
  if (!ws) return fail(h, MIYOLO_ERR_ARG, "workspace is null");
  const int Bc = auto_chunk(h, B, H, W);
  if (h->plan.B != Bc || h->plan.H != H || h->plan.W != W) make_plan(h, Bc, H, W, &h->plan);
  if (ws_bytes < h->plan.total)
    return fail(h, MIYOLO_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, h->plan.total);
  return 0;
}

// hipGraph replay of one entry point's launch sequence (option "graph").  Everything the launches bake in is part of the
// key: entry, shape, thresholds, stream, every pointer; any option change drops the graphs.  A changed key re-captures
// (at most 8 graphs are kept).  Capture needs a non-default stream; on the null stream, or while profiling, the call
// runs directly.
template <class F>
int with_graph(miyolo_engine* h, hipStream_t s, const void* key, size_t klen, F&& body) {
  // batch_split / cls_streams fork over internal streams inside body(): not captured, the call runs directly
  if (!h->graph || h->profile || s == nullptr || klen > sizeof(miyolo_engine::GraphRec::key) || h->batch_split > 1 || h->cls_streams > 1) return body();
  for (auto& r : h->graphs)
    if (r.klen == klen && !memcmp(r.key, key, klen)) { HIP_TRY(h, hipGraphLaunch(r.ex, s)); return 0; }
  // A key is captured on its first call (round 2 deferred the capture to the second sighting after a first-call capture
  // had replayed into a memory access fault; the cause was the captured hipMemsetAsync node, see zero_i32_kernel).  Nothing inside body()
  // creates streams, events or memory: those are made in miyolo_create / before with_graph.  CENSUS: the captured graph is
  // used only if it holds exactly one kernel node per launch body() issued and nothing else - a capture that lost launches
  // (round 2 saw one with the head lanes' cross-stream fork: "140 k frames/s") is thrown away and the call runs directly.
  const long l0 = h->launches;
  HIP_TRY(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess || !g) return fail(h, MIYOLO_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  {
    size_t nn = 0;
    HIP_TRY(h, hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    if (nn) HIP_TRY(h, hipGraphGetNodes(g, nodes.data(), &nn));
    int nk = 0;
    for (size_t i = 0; i < nn; ++i) {
      hipGraphNodeType t;
      HIP_TRY(h, hipGraphNodeGetType(nodes[i], &t));
      if (t == hipGraphNodeTypeKernel) ++nk;
    }
    h->graph_nodes = (int)nn; h->graph_kernel_nodes = nk; h->graph_launches = (int)(h->launches - l0);
    const bool ok = nk == h->graph_launches && (int)nn == nk;
    if (!ok) {
      ++h->graph_rejected;
      (void)hipGraphDestroy(g);
      return body();                                       // not captured: run the call directly, keep no graph for this key
    }
  }
  hipGraphExec_t ex = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e2 != hipSuccess) { (void)hipGraphDestroy(g); return fail(h, MIYOLO_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2)); }
  if (h->graphs.size() >= 8) {
    (void)hipGraphExecDestroy(h->graphs.front().ex); (void)hipGraphDestroy(h->graphs.front().g);
    h->graphs.erase(h->graphs.begin());
  }
  miyolo_engine::GraphRec r;
  memset(&r, 0, sizeof(r));
  memcpy(r.key, key, klen); r.klen = klen; r.g = g; r.ex = ex;
  h->graphs.push_back(r);
  HIP_TRY(h, hipGraphLaunch(ex, s));
  return 0;
}
void drop_graphs(miyolo_engine* h) {
  for (auto& r : h->graphs) { (void)hipGraphExecDestroy(r.ex); (void)hipGraphDestroy(r.g); }
  h->graphs.clear();
}

// Layer table + LDS packing of the one-launch classifier (cls_mega.h) for H x W crops; false: not applicable.
bool build_mega(miyolo_engine* h, int H, int W) {
  h->mega_h = H; h->mega_w = W; h->mega_ok = 0;
  if (h->desc.task != 1 || h->desc.dtype != MIYOLO_F16 || (int)h->ops.size() > kMegaMaxOps + 1) return false;
  const int nops = (int)h->ops.size();
  if (nops < 2 || h->ops[0].kind != MIYOLO_OP_STEM || h->ops[nops - 1].kind != MIYOLO_OP_CLS_HEAD) return false;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return ((1 << l) == v) ? l : -1; };
  // liveness of every activation buffer over the op sequence
  const int nb = (int)h->bufs.size();
  std::vector<int> first(nb, -1), last(nb, -1);
  auto touch = [&](int b, int i) { if (b > 0) { if (first[b] < 0) first[b] = i; last[b] = i; } };
  for (int i = 0; i < nops; ++i) {
    const miyolo_op& op = h->ops[i];
    if (op.kind == MIYOLO_OP_CONV || op.kind == MIYOLO_OP_STEM || op.kind == MIYOLO_OP_CLS_HEAD)
      for (int k = 0; k < (op.kind == MIYOLO_OP_CONV ? op.n_src : 1); ++k) touch(op.src[k].buf, i);
    if (op.kind != MIYOLO_OP_CLS_HEAD) touch(op.dst.buf, i);
    if (op.kind == MIYOLO_OP_CONV && op.res.buf >= 0) touch(op.res.buf, i);
  }
  // stored layout of a buffer: pixels padded to an odd number of 16-byte chunks; a zero halo when a 3x3 conv reads it
  std::vector<int> halo(nb, 0), ps(nb, 0), wp(nb, 0);
  for (int i = 0; i < nops; ++i)
    if (h->ops[i].kind == MIYOLO_OP_CONV && h->ops[i].ksize == 3 && h->ops[i].src[0].buf > 0) halo[h->ops[i].src[0].buf] = 1;
  std::vector<size_t> off(nb, 0), size(nb, 0);
  for (int b = 1; b < nb; ++b) {
    const int hb = H / h->bufs[b].down, wb = W / h->bufs[b].down, c = h->bufs[b].channels;
    if (c % 8) return false;
    ps[b] = c * 2 + ((c / 8) % 2 == 0 ? 16 : 0);
    wp[b] = wb + 2 * halo[b];
    size[b] = (size_t)(hb + 2 * halo[b]) * wp[b] * ps[b];
  }
  auto org = [&](int b, int choff) { return (int)off[b] + halo[b] * (wp[b] + 1) * ps[b] + choff * 2; };
  size_t peak = 0;
  for (int i = 0; i < nops; ++i)
    for (int b = 1; b < nb; ++b) {
      if (first[b] != i) continue;
      size_t cand = 0;                                // lowest offset that does not overlap a buffer alive at op i
      for (bool moved = true; moved;) {
        moved = false;
        for (int o = 1; o < nb; ++o)
          if (o != b && first[o] >= 0 && first[o] <= i && last[o] >= i && (first[o] < i || o < b) && cand < off[o] + size[o] && off[o] < cand + size[b]) {
            cand = off[o] + size[o]; moved = true;
          }
      }
      off[b] = cand;
      peak = std::max(peak, cand + size[b]);
    }
  MegaArgs& m = h->mega;
  memset(&m, 0, sizeof(m));
  h->mega_ops.assign((size_t)(nops - 1), MegaOp{});
  m.H = H; m.W = W; m.nc = h->desc.nc; m.nops = nops - 1;
  for (int i = 0; i < nops - 1; ++i) {
    const miyolo_op& op = h->ops[i];
    MegaOp& o = h->mega_ops[i];
    const int db = op.dst.buf;
    if (db <= 0) return false;
    const miyolo_buf& ob = h->bufs[db];
    if (ob.dtype != -1 || op.cout % 16 || op.dst.ch_off % 4) return false;
    o.kind = op.kind == MIYOLO_OP_STEM ? 0 : 1; o.ksize = op.ksize; o.stride = op.stride; o.act = op.act;
    o.cin = op.cin; o.cout = op.cout;
    o.hout = H / ob.down; o.wout = W / ob.down; o.lg_wout = ilog2(o.wout);
    if (o.lg_wout < 0) return false;
    o.dst_org = org(db, op.dst.ch_off); o.dst_wp = wp[db]; o.dst_ps = ps[db];
    if (halo[db] && first[db] == i) { o.zero_off = (int)off[db]; o.zero_rows = H / ob.down + 2; }
    o.res_org = -1;
    o.w = h->weights[op.weight]; o.bias = static_cast<const float*>(h->weights[op.bias]);
    if (op.kind == MIYOLO_OP_STEM) {
      if (ob.down != 2 || op.dst.ch_off != 0) return false;
      o.kpad = 32; o.wbytes = op.cout * 32 * 2;
      continue;
    }
    if (op.kind != MIYOLO_OP_CONV || op.n_src != 1 || op.src[0].upsample || op.src[0].buf <= 0) return false;
    const int sbi = op.src[0].buf;
    const miyolo_buf& sb = h->bufs[sbi];
    if (op.cin % 8 || op.src[0].ch_off % 8) return false;
    if (o.hout * op.stride != H / sb.down || o.wout * op.stride != W / sb.down) return false;
    o.src_org = org(sbi, op.src[0].ch_off); o.src_wp = wp[sbi]; o.src_ps = ps[sbi];
    const int K = op.cin * op.ksize * op.ksize;
    o.kpad = (K + 63) / 64 * 64; o.wbytes = op.cout * o.kpad * 2;
    o.lg_cpt = 0;
    if (op.ksize == 3) { o.lg_cpt = ilog2(op.cin / 8); if (o.lg_cpt < 0) return false; }
    else if (op.ksize != 1 || op.stride != 1) return false;
    if (op.res.buf >= 0) {
      const int rbi = op.res.buf;
      if (rbi <= 0 || op.res.ch_off % 4) return false;
      o.res_org = org(rbi, op.res.ch_off); o.res_wp = wp[rbi]; o.res_ps = ps[rbi];
    }
  }
  const miyolo_op& hd = h->ops[nops - 1];
  const miyolo_buf& fb = h->bufs[hd.src[0].buf];
  if (hd.src[0].ch_off != 0 || fb.channels != hd.cin) return false;
  if (halo[hd.src[0].buf]) return false;
  m.feat_off = (int)off[hd.src[0].buf]; m.feat_c = hd.cin; m.feat_hw = (H / fb.down) * (W / fb.down); m.feat_ps = ps[hd.src[0].buf];
  m.pool_off = (int)align_up(peak, 16);
  peak = m.pool_off + (size_t)(hd.cin + hd.cout) * 4;
  m.ring_off = (int)align_up(peak, 1024);
  peak = m.ring_off + (size_t)kMegaWaves * kMegaPf * 1024;
  m.bias_off = (int)peak;
  peak += (size_t)kMegaWaves * 512;
  m.desc_off = (int)align_up(peak, 16);
  peak = m.desc_off + sizeof(MegaOp) * (size_t)(nops - 1);
  m.lin_w = static_cast<const float*>(h->weights[hd.weight]); m.lin_b = static_cast<const float*>(h->weights[hd.bias]);
  if (peak > 160 * 1024) return false;
  // weights of the convs in fragment order (one buffer; repacked on the engine's device, default stream, synchronous)
  size_t wtot = 0;
  for (int i = 1; i < nops - 1; ++i) wtot += align_up((size_t)h->mega_ops[i].wbytes, 256);
  if (h->mega_wbuf) { (void)hipFree(h->mega_wbuf); h->mega_wbuf = nullptr; }
  if (hipMalloc(&h->mega_wbuf, wtot) != hipSuccess) return false;
  size_t wo = 0;
  for (int i = 1; i < nops - 1; ++i) {
    MegaOp& o = h->mega_ops[i];
    half_t* dst = reinterpret_cast<half_t*>(static_cast<char*>(h->mega_wbuf) + wo);
    const int total = (o.cout / 16) * (o.kpad / 32) * 64;
    hipLaunchKernelGGL(mega_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0,
                       static_cast<const half_t*>(o.w), dst, o.cout, o.kpad);
    o.w = dst;
    wo += align_up((size_t)o.wbytes, 256);
  }
  if (h->mega_ops_dev) { (void)hipFree(h->mega_ops_dev); h->mega_ops_dev = nullptr; }
  if (hipMalloc(&h->mega_ops_dev, sizeof(MegaOp) * (size_t)(nops - 1)) != hipSuccess) return false;
  if (hipMemcpy(h->mega_ops_dev, h->mega_ops.data(), sizeof(MegaOp) * (size_t)(nops - 1), hipMemcpyHostToDevice) != hipSuccess) return false;
  m.ops_dev = static_cast<const MegaOp*>(h->mega_ops_dev);
  if (hipDeviceSynchronize() != hipSuccess) return false;
  h->mega_lds = peak;
  h->mega_ok = 1;
  return true;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int miyolo_abi_version(void) { return MIYOLO_ABI_VERSION; }

int miyolo_k_align(int dtype) { return dtype == MIYOLO_F8 ? 128 : dtype == MIYOLO_F16 ? 64 : 32; }

int miyolo_create(const miyolo_desc* desc, const miyolo_buf* bufs, const miyolo_op* ops,
                  const void* const* weights, int device, miyolo_handle* out) {
  if (!desc || !bufs || !ops || !weights || !out) return fail(nullptr, MIYOLO_ERR_ARG, "null argument");
  if (desc->abi_version != MIYOLO_ABI_VERSION) return fail(nullptr, MIYOLO_ERR_ARG, "ABI version %d != %d", desc->abi_version, MIYOLO_ABI_VERSION);
  if (desc->dtype != MIYOLO_F32 && desc->dtype != MIYOLO_F16 && desc->dtype != MIYOLO_F8) return fail(nullptr, MIYOLO_ERR_ARG, "bad dtype %d", desc->dtype);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, MIYOLO_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(nullptr, MIYOLO_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
  if (!strstr(prop.gcnArchName, "gfx950"))
    return fail(nullptr, MIYOLO_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  DevGuard guard(device);
  miyolo_engine* h = new miyolo_engine();
  h->desc = *desc;
  h->device = device;
  h->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  h->bufs.assign(bufs, bufs + desc->n_bufs);
  h->ops.assign(ops, ops + desc->n_ops);
  h->weights.assign(weights, weights + desc->n_weights);
  h->buf_stale.assign(h->bufs.size(), 0);
  for (const miyolo_buf& b : h->bufs)
    if (b.down < 1 || b.channels < 1) { delete h; return fail(nullptr, MIYOLO_ERR_ARG, "buffer with channels %d / down %d", b.channels, b.down); }
  for (const miyolo_op& op : h->ops) {
    auto bad_view = [&](const miyolo_view& v, bool optional) {
      if (v.buf < 0) return !optional;
      return v.buf >= desc->n_bufs || v.ch_off < 0 || v.ch_cnt < 1 || v.ch_off + v.ch_cnt > h->bufs[v.buf].channels;
    };
    bool bad = false;
    if (op.kind == MIYOLO_OP_CONV) {
      bad = op.n_src < 1 || op.n_src > 2 || bad_view(op.dst, false) || bad_view(op.res, true);
      for (int i = 0; i < op.n_src && !bad; ++i) bad = bad_view(op.src[i], false);
    } else if (op.kind == MIYOLO_OP_STEM) {
      bad = bad_view(op.dst, false);
    } else if (op.kind == MIYOLO_OP_MAXPOOL5) {
      bad = bad_view(op.src[0], false) || bad_view(op.dst, false);
    } else if (op.kind == MIYOLO_OP_DECODE) {
      for (int i = 0; i < 3 && !bad; ++i) bad = bad_view(op.src[i], false) || op.level_stride[i] < 1;
    } else if (op.kind == MIYOLO_OP_CLS_HEAD) {
      bad = bad_view(op.src[0], false);
    }
    if (bad) { delete h; return fail(nullptr, MIYOLO_ERR_ARG, "op %d: view outside its buffer, or n_src out of range", (int)(&op - h->ops.data())); }
    const bool needs_w = op.kind == MIYOLO_OP_STEM || op.kind == MIYOLO_OP_CONV || op.kind == MIYOLO_OP_CLS_HEAD;
    if (needs_w && (op.weight < 0 || op.weight >= desc->n_weights || op.bias < 0 || op.bias >= desc->n_weights ||
                    !h->weights[op.weight] || !h->weights[op.bias])) {
      delete h;
      return fail(nullptr, MIYOLO_ERR_ARG, "op references a missing weight");
    }
  }
  build_lanes(h);
  hipError_t e = hipSuccess;
  {   // the head lanes' streams and events exist from here on: nothing is created on a call (run_ops only looks them up)
    DevGuard g2(device);
    for (int i = 1; i < h->n_lanes && e == hipSuccess; ++i) {
      hipStream_t st;
      if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) == hipSuccess) h->lanes.push_back(st);
    }
    for (size_t i = 0; i < h->ops.size() && e == hipSuccess && h->n_lanes > 1; ++i) {
      hipEvent_t ev;
      if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) == hipSuccess) h->op_ev.push_back(ev);
    }
  }
  if (e == hipSuccess) e = set_conv_attrs_ks<float, 1>();
  if (e == hipSuccess) e = set_conv_attrs_ks<float, 3>();
  if (e == hipSuccess) e = set_conv_attrs_ks<half_t, 1>();
  if (e == hipSuccess) e = set_conv_attrs_ks<half_t, 3>();
  if (e == hipSuccess) e = set_dma_attrs_ks<float, 1>();
  if (e == hipSuccess) e = set_dma_attrs_ks<float, 3>();
  if (e == hipSuccess) e = set_dma_attrs_ks<half_t, 1>();
  if (e == hipSuccess) e = set_dma_attrs_ks<half_t, 3>();
  if (e == hipSuccess) e = set_dmap_attrs_ks<float, 1>();
  if (e == hipSuccess) e = set_dmap_attrs_ks<float, 3>();
  if (e == hipSuccess) e = set_dmap_attrs_ks<half_t, 1>();
  if (e == hipSuccess) e = set_dmap_attrs_ks<half_t, 3>();
  if (e == hipSuccess) e = set_h2_attrs<float>();
  if (e == hipSuccess) e = set_h2_attrs<half_t>();
  if (e == hipSuccess) e = set_h2_attrs<fp8_t>();
  if (e == hipSuccess) e = set_h3_attrs();
  if (e == hipSuccess) e = set_bneck_attrs();
  if (e == hipSuccess) e = set_pw_attrs();
  if (e == hipSuccess) e = set_h4_attrs();
  if (e == hipSuccess) e = set_stem2_attrs();
  if (e == hipSuccess) e = set_dmap_attrs_ks<fp8_t, 1>();
  if (e == hipSuccess) e = set_dmap_attrs_ks<fp8_t, 3>();
  if (e == hipSuccess) e = set_t2d_attrs<float>();
  if (e == hipSuccess) e = set_t2d_attrs<half_t>();
#if MIYOLO_EXPERIMENTS
  if (e == hipSuccess) e = set_dmh_attrs_ks<float, 1>();
  if (e == hipSuccess) e = set_dmh_attrs_ks<float, 3>();
  if (e == hipSuccess) e = set_dmh_attrs_ks<half_t, 1>();
  if (e == hipSuccess) e = set_dmh_attrs_ks<half_t, 3>();
  if (e == hipSuccess) e = set_ws_attrs_ks<float, 1>();
  if (e == hipSuccess) e = set_ws_attrs_ks<float, 3>();
  if (e == hipSuccess) e = set_ws_attrs_ks<half_t, 1>();
  if (e == hipSuccess) e = set_ws_attrs_ks<half_t, 3>();
  if (e == hipSuccess) e = set_halop_attrs<float>();
  if (e == hipSuccess) e = set_halop_attrs<half_t>();
  if (e == hipSuccess) e = set_halo_attrs<float>();
  if (e == hipSuccess) e = set_halo_attrs<half_t>();
#endif
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(cls_mega_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_greedy_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, nms_lds_bytes(1024));
  if (e != hipSuccess) {
    delete h;
    return fail(nullptr, MIYOLO_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  *out = h;
  return 0;
}

void miyolo_destroy(miyolo_handle h) {
  if (!h) return;
  drop_graphs(h);
  for (hipStream_t st : h->lanes) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : h->lane_ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : h->op_ev) (void)hipEventDestroy(ev);
  if (h->nms_stream) { (void)hipStreamSynchronize(h->nms_stream); (void)hipStreamDestroy(h->nms_stream); (void)hipEventDestroy(h->ev_dec); (void)hipEventDestroy(h->ev_nms); }
  for (hipStream_t st : h->split_streams) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : h->split_ev) (void)hipEventDestroy(ev);
  if (h->dbg) (void)hipFree(h->dbg);
  if (h->mega_wbuf) (void)hipFree(h->mega_wbuf);
  if (h->mega_ops_dev) (void)hipFree(h->mega_ops_dev);
  delete h;
}

const char* miyolo_last_error(miyolo_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int miyolo_classify_launches(miyolo_handle h, int H, int W, size_t* lds_bytes) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 1) return fail(h, MIYOLO_ERR_ARG, "not a classify engine");
  if (lds_bytes) *lds_bytes = 0;
  if (h->cls_mega && !h->profile && h->desc.dtype == MIYOLO_F16) {
    if (h->mega_h != H || h->mega_w != W) build_mega(h, H, W);
    if (h->mega_ok) { if (lds_bytes) *lds_bytes = h->mega_lds; return 1; }
  }
  return (int)h->ops.size();
}

int miyolo_wait_outputs(miyolo_handle h, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  DevGuard guard(h->device);
  return join_pending_nms(h, static_cast<hipStream_t>(stream));
}

int miyolo_set_option(miyolo_handle h, const char* key, int value) {
  if (!h || !key) return MIYOLO_ERR_ARG;
  drop_graphs(h);                                    // every option can change which kernels a call launches
  if (!strcmp(key, "graph")) { h->graph = value; return 0; }
  if (!strcmp(key, "max_chunk")) { h->max_chunk = value; h->plan = Plan(); return 0; }
  if (!strcmp(key, "force_wc")) { h->force_wc = value; return 0; }
  if (!strcmp(key, "force_tc")) { h->force_tc = value; return 0; }
  if (!strcmp(key, "profile")) { h->profile = value; return 0; }
  if (!strcmp(key, "conv_impl")) {
    if (!MIYOLO_EXPERIMENTS && (value == 2 || value == 4 || value == 5 || value == 6))
      return fail(h, MIYOLO_ERR_UNSUPPORTED, "conv_impl %d is an experimental kernel: rebuild with csrc/build.sh experiments", value);
    h->conv_impl = value; return 0;
  }
  if (!strcmp(key, "t2d")) { h->t2d = value; return 0; }
  if (!strcmp(key, "h2")) { h->h2 = value; return 0; }
  if (!strcmp(key, "pw")) { h->pw = value; return 0; }
  if (!strcmp(key, "h4")) { h->h4 = value; return 0; }
  if (!strcmp(key, "h4_min_util")) { h->h4_min_util = value; return 0; }
  if (!strcmp(key, "pair8")) { h->pair8 = value; return 0; }
  if (!strcmp(key, "h3")) { h->h3 = value; return 0; }
  if (!strcmp(key, "h3_min_util")) { h->h3_min_util = value; return 0; }
  if (!strcmp(key, "h3_max_w")) { h->h3_max_w = value; return 0; }
  if (!strcmp(key, "cls_mega")) { h->cls_mega = value; return 0; }
  if (!strcmp(key, "head_lanes")) { h->head_lanes = value; return 0; }
  if (!strcmp(key, "nms_async")) { h->nms_async = value; return 0; }
  if (!strcmp(key, "fuse_prefilter")) { h->fuse_pre_opt = value; return 0; }
  if (!strcmp(key, "sppf_fuse")) { h->sppf_fuse = value; return 0; }
  if (!strcmp(key, "bneck_fuse")) { h->bneck_fuse = value; return 0; }
  if (!strcmp(key, "stem_fuse")) { h->stem_fuse = value; return 0; }
  if (!strcmp(key, "batch_split")) { h->batch_split = value; return 0; }
  if (!strcmp(key, "cls_streams")) { if (value < 1 || value > 16) return fail(h, MIYOLO_ERR_ARG, "cls_streams out of range"); h->cls_streams = value; return 0; }
  if (!strcmp(key, "h2_warm")) { h->h2_warm = value; return 0; }
  if (!strcmp(key, "h2_min_util")) { h->h2_min_util = value; return 0; }
  if (!strcmp(key, "dmh_auto")) { h->dmh_auto = value; return 0; }
  if (!strcmp(key, "ncu")) { if (value < 8 || value > 1024) return fail(h, MIYOLO_ERR_ARG, "ncu out of range"); h->ncu = value; return 0; }   // persistent-grid width (A/B)
  if (!strcmp(key, "ablate")) { h->ablate = value; return 0; }
  if (!strcmp(key, "dbg_op")) {
    h->dbg_op = value;
    if (!h->dbg) { if (hipMalloc(reinterpret_cast<void**>(&h->dbg), 2 * 256 * 8 * 8 * 8) != hipSuccess) return fail(h, MIYOLO_ERR_HIP, "hipMalloc dbg"); }
    (void)hipMemset(h->dbg, 0, 2 * 256 * 8 * 8 * 8);
    return 0;
  }
  return fail(h, MIYOLO_ERR_ARG, "unknown option %s", key);
}

int miyolo_set_classes(miyolo_handle h, const int32_t* classes, int n) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 0) return fail(h, MIYOLO_ERR_ARG, "not a detection model");
  drop_graphs(h);
  if (n <= 0 || !classes) { h->use_cls_mask = 0; return 0; }
  if (h->desc.nc > 256) return fail(h, MIYOLO_ERR_UNSUPPORTED, "classes= filter supports at most 256 classes (model has %d)", h->desc.nc);
  uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    if (classes[i] < 0 || classes[i] >= h->desc.nc) continue;      // a class the model does not have never matches, as upstream
    m[classes[i] >> 5] |= 1u << (classes[i] & 31);
  }
  for (int i = 0; i < 8; ++i) h->cls_mask[i] = m[i];
  h->use_cls_mask = 1;
  return 0;
}

size_t miyolo_workspace_bytes(miyolo_handle h, int B, int H, int W) {
  if (!h || check_shape(h, B, H, W)) return 0;
  Plan p;
  make_plan(h, auto_chunk(h, B, H, W), H, W, &p);
  size_t need = p.total;
  if (h->desc.task == 1) {                              // multi-stream classify: one workspace slice per lane
    const int Bs = cls_lane_batch(h, B);
    if (Bs > 0) {
      Plan ps;
      make_plan(h, Bs, H, W, &ps);
      need = std::max(need, (size_t)((B + Bs - 1) / Bs) * ps.total);
    }
  }
  return need;
}

int miyolo_chunk(miyolo_handle h, int B, int H, int W) {
  if (!h || check_shape(h, B, H, W)) return 0;
  return auto_chunk(h, B, H, W);
}

int miyolo_detect(miyolo_handle h, const uint8_t* in, int B, int H, int W, float conf, float iou,
                  int agnostic, int max_det, const float* scale, float* out_dets, int32_t* out_counts,
                  int32_t* out_anchor, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 0) return fail(h, MIYOLO_ERR_ARG, "not a detection model");
  if (!in || !out_dets || !out_counts) return fail(h, MIYOLO_ERR_ARG, "null argument");
  if (max_det < 1 || max_det > 1024) return fail(h, MIYOLO_ERR_ARG, "max_det %d out of range [1,1024]", max_det);
  if (int rc = prepare(h, B, H, W, workspace_bytes, workspace)) return rc;
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Plan& p = h->plan;
  const float* y = reinterpret_cast<const float*>(static_cast<unsigned char*>(workspace) + p.y_off);
  struct { int entry, B, H, W, agnostic, max_det; float conf, iou; const void* in; const void* scale; void* d; void* c; void* a; void* ws; hipStream_t s; }
      key = {1, B, H, W, agnostic, max_det, conf, iou, in, scale, out_dets, out_counts, out_anchor, workspace, s};
  const bool async_nms = h->nms_async && !h->graph && !h->profile && h->batch_split <= 1;
  if (async_nms && !h->nms_stream) {
    HIP_TRY(h, hipStreamCreateWithFlags(&h->nms_stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_dec, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_nms, hipEventDisableTiming));
  }
  if (!async_nms) { if (int rc = join_pending_nms(h, s)) return rc; }
  return with_graph(h, s, &key, sizeof(key), [&]() -> int {
    const int K = h->batch_split;
    if (K > 1 && B == p.B && B % K == 0 && !h->profile) {
      Plan p2;
      make_plan(h, B / K, H, W, &p2);
      const size_t part = align_up(p2.total, 256);
      if ((size_t)K * part <= workspace_bytes) {
        while ((int)h->split_streams.size() < K - 1) {
          hipStream_t st;
          HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
          h->split_streams.push_back(st);
        }
        while ((int)h->split_ev.size() < K) {
          hipEvent_t ev;
          HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
          h->split_ev.push_back(ev);
        }
        HIP_TRY(h, hipEventRecord(h->split_ev[0], s));
        for (int k = 0; k < K; ++k) {
          hipStream_t sk = k ? h->split_streams[k - 1] : s;
          if (k) HIP_TRY(h, hipStreamWaitEvent(sk, h->split_ev[0], 0));
          unsigned char* wk = static_cast<unsigned char*>(workspace) + k * part;
          const int b0 = k * (B / K);
          if (int rc = run_ops(h, 0, (int)h->ops.size(), p2, in + (size_t)b0 * H * W * 3, wk, nullptr, nullptr, sk)) return rc;
          if (int rc = run_nms(h, p2, reinterpret_cast<const float*>(wk + p2.y_off), p2.B, p2.A, conf, iou, agnostic, max_det,
                               scale ? scale + (size_t)b0 * 5 : nullptr, out_dets + (size_t)b0 * max_det * 6, out_counts + b0,
                               out_anchor ? out_anchor + (size_t)b0 * max_det : nullptr, wk, sk)) return rc;
          if (k) {
            HIP_TRY(h, hipEventRecord(h->split_ev[k], sk));
            HIP_TRY(h, hipStreamWaitEvent(s, h->split_ev[k], 0));
          }
        }
        return 0;
      }
    }
    for (int b0 = 0; b0 < B; b0 += p.B) {
      Plan pc = p;
      pc.B = std::min(p.B, B - b0);
      // the decode op runs the NMS score filter while the class scores are still in LDS (decode_kernel); its candidate
      // counters are zeroed first.  Not with the asynchronous NMS, whose previous instance may still read them.
      const bool fused = !async_nms && h->fuse_pre_opt;
      if (fused) {
        int32_t* cnt = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(workspace) + p.count_off);
        hipLaunchKernelGGL(zero_i32_kernel, dim3((pc.B + 255) / 256), dim3(256), 0, s, cnt, pc.B);
        ++h->launches;
        h->fuse_pre = 1; h->fuse_conf = conf;
      }
      const int rc_ops = run_ops(h, 0, (int)h->ops.size(), pc, in + (size_t)b0 * H * W * 3, workspace, nullptr, nullptr, s);
      h->fuse_pre = 0;
      if (rc_ops) return rc_ops;
      hipStream_t sn = s;
      if (async_nms) {
        HIP_TRY(h, hipEventRecord(h->ev_dec, s));
        HIP_TRY(h, hipStreamWaitEvent(h->nms_stream, h->ev_dec, 0));
        sn = h->nms_stream;
      }
      if (int rc = run_nms(h, pc, y, pc.B, p.A, conf, iou, agnostic, max_det, scale ? scale + (size_t)b0 * 5 : nullptr,
                           out_dets + (size_t)b0 * max_det * 6, out_counts + b0,
                           out_anchor ? out_anchor + (size_t)b0 * max_det : nullptr, workspace, sn, fused)) return rc;
      if (async_nms) {
        HIP_TRY(h, hipEventRecord(h->ev_nms, sn));
        h->nms_pending = true;
      }
    }
    return 0;
  });
}

int miyolo_head_raw(miyolo_handle h, const uint8_t* in, int B, int H, int W, float* y_out, void* workspace,
                    size_t workspace_bytes, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 0) return fail(h, MIYOLO_ERR_ARG, "not a detection model");
  if (!in || !y_out) return fail(h, MIYOLO_ERR_ARG, "null argument");
  if (int rc = prepare(h, B, H, W, workspace_bytes, workspace)) return rc;
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = join_pending_nms(h, s)) return rc;
  const Plan& p = h->plan;
  const size_t per = (size_t)(4 + h->desc.nc) * p.A;
  for (int b0 = 0; b0 < B; b0 += p.B) {
    Plan pc = p;
    pc.B = std::min(p.B, B - b0);
    if (int rc = run_ops(h, 0, (int)h->ops.size(), pc, in + (size_t)b0 * H * W * 3, workspace, nullptr, nullptr, s)) return rc;
    HIP_TRY(h, hipMemcpyAsync(y_out + (size_t)b0 * per, static_cast<unsigned char*>(workspace) + p.y_off,
                              (size_t)pc.B * per * 4, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

int miyolo_nms(miyolo_handle h, const float* y, int B, int A, int H, int W, float conf, float iou, int agnostic,
               int max_det, const float* scale, float* out_dets, int32_t* out_counts, int32_t* out_anchor,
               void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 0) return fail(h, MIYOLO_ERR_ARG, "not a detection model");
  if (!y || !out_dets || !out_counts) return fail(h, MIYOLO_ERR_ARG, "null argument");
  if (max_det < 1 || max_det > 1024) return fail(h, MIYOLO_ERR_ARG, "max_det %d out of range [1,1024]", max_det);
  if (int rc = prepare(h, B, H, W, workspace_bytes, workspace)) return rc;
  // A < the frame's anchor count is allowed (candidate lists of the sliced-inference merge step): the key / class-index
  // scratch is sized for plan.A
  if (A < 1 || A > h->plan.A) return fail(h, MIYOLO_ERR_SHAPE, "A=%d exceeds the %d anchors of a %dx%d frame", A, h->plan.A, H, W);
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = join_pending_nms(h, s)) return rc;
  const Plan& p = h->plan;
  const size_t per = (size_t)(4 + h->desc.nc) * p.A;
  for (int b0 = 0; b0 < B; b0 += p.B) {
    Plan pc = p;
    pc.B = std::min(p.B, B - b0);
    if (int rc = run_nms(h, pc, y + (size_t)b0 * per, pc.B, A, conf, iou, agnostic, max_det,
                         scale ? scale + (size_t)b0 * 5 : nullptr, out_dets + (size_t)b0 * max_det * 6, out_counts + b0,
                         out_anchor ? out_anchor + (size_t)b0 * max_det : nullptr, workspace, s)) return rc;
  }
  return 0;
}

int miyolo_classify(miyolo_handle h, const uint8_t* in, int B, int H, int W, float* logits, float* probs,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 1) return fail(h, MIYOLO_ERR_ARG, "not a classification model");
  if (!in) return fail(h, MIYOLO_ERR_ARG, "null argument");
  if (int rc = prepare(h, B, H, W, workspace_bytes, workspace)) return rc;
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Plan& p = h->plan;
  const int nc = h->desc.nc;
  if (h->cls_mega && !h->profile && h->desc.dtype == MIYOLO_F16) {
    if (h->mega_h != H || h->mega_w != W) build_mega(h, H, W);
    if (h->mega_ok) {
      MegaArgs m = h->mega;
      m.in = in; m.logits = logits; m.probs = probs; m.B = B; m.stamps = h->dbg;
      hipLaunchKernelGGL(cls_mega_kernel, dim3((unsigned)B), dim3(kMegaWaves * 64), h->mega_lds, s, m);
      HIP_TRY(h, hipGetLastError());
      std::fill(h->buf_stale.begin(), h->buf_stale.end(), 1);      // every activation stayed in LDS
      return 0;
    }
  }
  const int Bs = cls_lane_batch(h, B);
  if (Bs > 0 && Bs <= auto_chunk(h, Bs, H, W)) {
    // fork / join on events (no host synchronisation): lane i runs images [i*Bs, (i+1)*Bs) in its own slice of the workspace
    Plan ps;
    make_plan(h, Bs, H, W, &ps);
    const int ns = (B + Bs - 1) / Bs;
    if ((size_t)ns * ps.total <= workspace_bytes) {
      while ((int)h->lanes.size() < ns) {
        hipStream_t st; hipEvent_t ev;
        HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        h->lanes.push_back(st);
        while (h->lane_ev.size() < h->lanes.size() + 1) { HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); h->lane_ev.push_back(ev); }
      }
      struct { int entry, B, H, W, ns; const void* in; void* lg; void* pr; void* ws; hipStream_t s; } keyl = {3, B, H, W, ns, in, logits, probs, workspace, s};
      return with_graph(h, s, &keyl, sizeof(keyl), [&]() -> int {
        HIP_TRY(h, hipEventRecord(h->lane_ev[0], s));
        for (int i = 0; i < ns; ++i) {
          const int b0 = i * Bs;
          Plan pc = ps;
          pc.B = std::min(Bs, B - b0);
          hipStream_t si = h->lanes[i];
          HIP_TRY(h, hipStreamWaitEvent(si, h->lane_ev[0], 0));
          if (int rc = run_ops(h, 0, (int)h->ops.size(), pc, in + (size_t)b0 * H * W * 3, static_cast<unsigned char*>(workspace) + (size_t)i * ps.total,
                               logits ? logits + (size_t)b0 * nc : nullptr, probs ? probs + (size_t)b0 * nc : nullptr, si)) return rc;
          HIP_TRY(h, hipEventRecord(h->lane_ev[1 + i], si));
          HIP_TRY(h, hipStreamWaitEvent(s, h->lane_ev[1 + i], 0));
        }
        return 0;
      });
    }
  }
  struct { int entry, B, H, W; const void* in; void* lg; void* pr; void* ws; hipStream_t s; } key = {2, B, H, W, in, logits, probs, workspace, s};
  return with_graph(h, s, &key, sizeof(key), [&]() -> int {
    for (int b0 = 0; b0 < B; b0 += p.B) {
      Plan pc = p;
      pc.B = std::min(p.B, B - b0);
      if (int rc = run_ops(h, 0, (int)h->ops.size(), pc, in + (size_t)b0 * H * W * 3, workspace,
                           logits ? logits + (size_t)b0 * nc : nullptr, probs ? probs + (size_t)b0 * nc : nullptr, s)) return rc;
    }
    return 0;
  });
}

int miyolo_run_ops(miyolo_handle h, int first, int last, const uint8_t* in, int B, int H, int W, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  if (first < 0 || last > (int)h->ops.size() || first > last) return fail(h, MIYOLO_ERR_ARG, "bad op range");
  if (int rc = prepare(h, B, H, W, workspace_bytes, workspace)) return rc;
  if (h->plan.B < B) return fail(h, MIYOLO_ERR_SHAPE, "debug entry points need B <= chunk (%d)", h->plan.B);
  DevGuard guard(h->device);
  return run_ops(h, first, last, h->plan, in, workspace, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

static int debug_buf(miyolo_handle h, int buf, int B, int H, int W, long* n) {
  if (!h) return MIYOLO_ERR_ARG;
  if (buf < 1 || buf >= (int)h->bufs.size()) return fail(h, MIYOLO_ERR_ARG, "bad buffer index %d", buf);
  if (int rc = check_shape(h, B, H, W)) return rc;
  const int Bc = auto_chunk(h, B, H, W);
  if (Bc < B) return fail(h, MIYOLO_ERR_SHAPE, "debug entry points need B <= chunk (%d)", Bc);
  if (h->plan.B != Bc || h->plan.H != H || h->plan.W != W) make_plan(h, Bc, H, W, &h->plan);
  const miyolo_buf& b = h->bufs[buf];
  *n = (long)B * (H / b.down) * (W / b.down) * b.channels;
  return 0;
}

int miyolo_read_buffer(miyolo_handle h, int buf, int B, int H, int W, float* out, void* workspace, void* stream) {
  long n = 0;
  if (int rc = debug_buf(h, buf, B, H, W, &n)) return rc;
  if (!out || !workspace) return fail(h, MIYOLO_ERR_ARG, "null argument");
  if (h->buf_stale[buf])
    return fail(h, MIYOLO_ERR_UNSUPPORTED, "buffer %d was not written by the last run: its producer ran inside a fused launch that keeps it in "
                "LDS (options bneck_fuse / stem_fuse); switch the fusion off to tap it", buf);
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const void* src = static_cast<unsigned char*>(workspace) + h->plan.buf_off[buf];
  const unsigned g = (unsigned)((n + 255) / 256);
  if (elem_size(h, h->bufs[buf]) == 4) {
    HIP_TRY(h, hipMemcpyAsync(out, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  } else if (elem_size(h, h->bufs[buf]) == 1) {
    hipLaunchKernelGGL(copy_to_f32_kernel<fp8_t>, dim3(g), dim3(256), 0, s, static_cast<const fp8_t*>(src), out, n);
    HIP_TRY(h, hipGetLastError());
  } else {
    hipLaunchKernelGGL(copy_to_f32_kernel<half_t>, dim3(g), dim3(256), 0, s, static_cast<const half_t*>(src), out, n);
    HIP_TRY(h, hipGetLastError());
  }
  return 0;
}

int miyolo_write_buffer(miyolo_handle h, int buf, int B, int H, int W, const float* in, void* workspace, void* stream) {
  long n = 0;
  if (int rc = debug_buf(h, buf, B, H, W, &n)) return rc;
  if (!in || !workspace) return fail(h, MIYOLO_ERR_ARG, "null argument");
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  void* dst = static_cast<unsigned char*>(workspace) + h->plan.buf_off[buf];
  const unsigned g = (unsigned)((n + 255) / 256);
  if (elem_size(h, h->bufs[buf]) == 4) {
    HIP_TRY(h, hipMemcpyAsync(dst, in, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  } else if (elem_size(h, h->bufs[buf]) == 1) {
    hipLaunchKernelGGL(copy_from_f32_kernel<fp8_t>, dim3(g), dim3(256), 0, s, in, static_cast<fp8_t*>(dst), n);
    HIP_TRY(h, hipGetLastError());
  } else {
    hipLaunchKernelGGL(copy_from_f32_kernel<half_t>, dim3(g), dim3(256), 0, s, in, static_cast<half_t*>(dst), n);
    HIP_TRY(h, hipGetLastError());
  }
  return 0;
}

int miyolo_graph_info(miyolo_handle h, int32_t* out6) {
  if (!h || !out6) return MIYOLO_ERR_ARG;
  out6[0] = (int32_t)h->graphs.size(); out6[1] = h->graph_nodes; out6[2] = h->graph_kernel_nodes; out6[3] = h->graph_launches;
  out6[4] = h->graph_rejected; out6[5] = (int32_t)h->launches;
  return 0;
}

int miyolo_debug_candidate_counts(miyolo_handle h, const void* workspace, int B, int32_t* out_host) {
  if (!h || !workspace || !out_host || B < 1 || B > h->plan.B) return fail(h, MIYOLO_ERR_ARG, "debug_candidate_counts: B outside the last call's chunk");
  if (h->desc.task != 0) return fail(h, MIYOLO_ERR_ARG, "not a detection model");
  DevGuard guard(h->device);
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(out_host, static_cast<const unsigned char*>(workspace) + h->plan.count_off, (size_t)B * 4, hipMemcpyDeviceToHost));
  return 0;
}

int miyolo_profile_read(miyolo_handle h, int max_records, int32_t* op_index, int32_t* cfg, float* ms) {
  if (!h) return MIYOLO_ERR_ARG;
  DevGuard guard(h->device);
  int n = 0;
  for (auto& r : h->prof) {
    if (n < max_records && op_index && cfg && ms) {
      HIP_TRY(h, hipEventSynchronize(r.e1));
      float t = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&t, r.e0, r.e1));
      op_index[n] = r.op; cfg[n] = r.cfg; ms[n] = t;
    }
    ++n;
  }
  if (op_index) {       // a read with buffers consumes the records
    for (auto& r : h->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    h->prof.clear();
  }
  return n;
}

int miyolo_debug_stamps(miyolo_handle h, unsigned long long* out /* host, 256*8*8 */) {
  if (!h || !h->dbg || !out) return MIYOLO_ERR_ARG;
  DevGuard guard(h->device);
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(out, h->dbg, 2 * 256 * 8 * 8 * 8, hipMemcpyDeviceToHost));
  return 0;
}

int miyolo_op_work(miyolo_handle h, int op_index, int B, int H, int W, double* flops, double* bytes) {
  if (!h || op_index < 0 || op_index >= (int)h->ops.size()) return MIYOLO_ERR_ARG;
  if (int rc = check_shape(h, B, H, W)) return rc;
  const miyolo_op& op = h->ops[op_index];
  double fl = 0, by = 0;
  const double es = h->desc.dtype == MIYOLO_F16 ? 2 : h->desc.dtype == MIYOLO_F8 ? 1 : 4;
  if (op.kind == MIYOLO_OP_CONV || op.kind == MIYOLO_OP_STEM) {
    const miyolo_buf& ob = h->bufs[op.dst.buf];
    const double mo = (double)B * (H / ob.down) * (W / ob.down);
    fl = 2.0 * mo * op.cout * op.cin * op.ksize * op.ksize;
    if (op.kind == MIYOLO_OP_STEM) {
      by += (double)B * H * W * 3;
    } else {
      for (int i = 0; i < op.n_src; ++i) {
        const miyolo_buf& sb = h->bufs[op.src[i].buf];
        by += (double)B * (H / sb.down) * (W / sb.down) * op.src[i].ch_cnt * es;
      }
    }
    by += mo * op.cout * (ob.dtype == MIYOLO_F32 ? 4 : es);
    by += (double)op.cout * op.cin * op.ksize * op.ksize * (op.kind == MIYOLO_OP_STEM ? 4 : es);
    if (op.res.buf >= 0) by += mo * op.cout * es;
  } else if (op.kind == MIYOLO_OP_MAXPOOL5) {
    const miyolo_buf& sb = h->bufs[op.src[0].buf];
    by = 2.0 * B * (H / sb.down) * (W / sb.down) * op.src[0].ch_cnt * es;
  } else if (op.kind == MIYOLO_OP_DECODE) {
    by = (double)B * total_anchors(h, H, W) * ((64 + h->desc.nc) + (4 + h->desc.nc)) * 4;
  }
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return 0;
}

int miyolo_work(miyolo_handle h, int B, int H, int W, double* flops, double* bytes) {
  if (!h) return MIYOLO_ERR_ARG;
  if (int rc = check_shape(h, B, H, W)) return rc;
  double fl = 0, by = 0;
  for (int i = 0; i < (int)h->ops.size(); ++i) {
    double f = 0, b = 0;
    if (int rc = miyolo_op_work(h, i, B, H, W, &f, &b)) return rc;
    fl += f; by += b;
  }
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return 0;
}

int miyolo_letterbox(const void* src, int B, int src_h, int src_w, void* dst, int dst_h, int dst_w, int top, int left,
                     int new_h, int new_w, int pad_value, void* stream) {
  if (!src || !dst || B < 1 || src_h < 1 || src_w < 1 || new_h < 1 || new_w < 1 || top < 0 || left < 0 ||
      top + new_h > dst_h || left + new_w > dst_w || pad_value < 0 || pad_value > 255)
    return fail(nullptr, MIYOLO_ERR_ARG, "letterbox: bad geometry %dx%d -> %dx%d at (%d,%d) in %dx%d", src_h, src_w, new_h, new_w,
                top, left, dst_h, dst_w);
  LetterboxArgs a;
  a.src = static_cast<const uint8_t*>(src); a.dst = static_cast<uint8_t*>(dst);
  a.B = B; a.sh = src_h; a.sw = src_w; a.dh = dst_h; a.dw = dst_w; a.top = top; a.left = left; a.nh = new_h; a.nw = new_w;
  a.pad = pad_value;
  a.scale_x = 1.0 / ((double)new_w / (double)src_w);      // OpenCV: scale = 1. / inv_scale, inv_scale = dsize / ssize
  a.scale_y = 1.0 / ((double)new_h / (double)src_h);
  hipLaunchKernelGGL(letterbox_kernel, dim3((dst_w + 63) / 64, (dst_h + 3) / 4, B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, MIYOLO_ERR_HIP, "letterbox launch: %s", hipGetErrorString(e));
  return 0;
}

int miyolo_slice_batch(const void* frame, int H, int W, const int32_t* boxes, int n, void* out, int sh, int sw, int pad_value, void* stream) {
  if (!frame || !boxes || !out || H < 1 || W < 1 || n < 1 || sh < 1 || sw < 1 || pad_value < 0 || pad_value > 255)
    return fail(nullptr, MIYOLO_ERR_ARG, "slice_batch: bad arguments");
  SliceArgs a;
  a.frame = static_cast<const uint8_t*>(frame); a.boxes = boxes; a.out = static_cast<uint8_t*>(out);
  a.H = H; a.W = W; a.n = n; a.sh = sh; a.sw = sw; a.pad = pad_value;
  hipLaunchKernelGGL(slice_batch_kernel, dim3((sw + 63) / 64, (sh + 3) / 4, n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, MIYOLO_ERR_HIP, "slice_batch launch: %s", hipGetErrorString(e));
  return 0;
}

int miyolo_merge_slices(miyolo_handle h, const float* dets, const int32_t* counts, const int32_t* boxes, int n_slices, int slice_max_det,
                        int H, int W, float iou, int agnostic, int max_det, float* y_scratch, float* out_dets, int32_t* out_counts,
                        int32_t* out_index, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return MIYOLO_ERR_ARG;
  if (h->desc.task != 0) return fail(h, MIYOLO_ERR_ARG, "not a detection model");
  if (!dets || !counts || !boxes || !y_scratch || !out_dets || !out_counts || n_slices < 1 || slice_max_det < 1)
    return fail(h, MIYOLO_ERR_ARG, "merge_slices: null argument");
  const int cap = n_slices * slice_max_det;
  DevGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  MergeArgs m;
  m.dets = dets; m.counts = counts; m.boxes = boxes; m.y = y_scratch; m.ns = n_slices; m.max_det = slice_max_det; m.nc = h->desc.nc; m.cap = cap;
  hipLaunchKernelGGL(merge_pack_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, m);
  HIP_TRY(h, hipGetLastError());
  // every candidate already passed its slice's confidence threshold: conf = 0 keeps all real ones (score > 0)
  return miyolo_nms(h, y_scratch, 1, cap, H, W, 0.0f, iou, agnostic, max_det, nullptr, out_dets, out_counts, out_index, workspace, workspace_bytes, stream);
}

int miyolo_merge_slices_nmm(const float* dets, const int32_t* counts, const int32_t* boxes, int n_slices, int slice_max_det, int frame_h,
                            int frame_w, int metric, float threshold, int agnostic, int max_out, float* out_dets, int32_t* out_counts,
                            int32_t* out_index, void* stream) {
  if (!dets || !counts || !boxes || !out_dets || !out_counts || n_slices < 1 || n_slices > 250 || slice_max_det < 1 || frame_h < 1 ||
      frame_w < 1 || max_out < 1 || (metric != 0 && metric != 1))
    return fail(nullptr, MIYOLO_ERR_ARG, "merge_slices_nmm: bad arguments (1..250 slices, metric 0 = IOS / 1 = IOU)");
  NmmArgs a;
  a.dets = dets; a.counts = counts; a.boxes = boxes; a.ns = n_slices; a.max_det = slice_max_det; a.H = frame_h; a.W = frame_w;
  a.metric = metric; a.agnostic = agnostic; a.max_out = max_out; a.thr = threshold;
  a.out_dets = out_dets; a.out_count = out_counts; a.out_index = out_index;
  {   // per device, idempotent, cheap next to the launch: no cached flag (no global state)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(greedy_nmm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNmmLds);
    if (e != hipSuccess) return fail(nullptr, MIYOLO_ERR_HIP, "merge_slices_nmm attribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(greedy_nmm_kernel, dim3(1), dim3(kNmmThreads), kNmmLds, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, MIYOLO_ERR_HIP, "merge_slices_nmm launch: %s", hipGetErrorString(e));
  return 0;
}

int miyolo_crop_resize(const void* frame, int H, int W, const int32_t* boxes, int n, int size, int max_short, void* out, void* stream) {
  if (!frame || !boxes || !out || H < 1 || W < 1 || n < 0 || size < 8 || size > 128 || max_short < 1)
    return fail(nullptr, MIYOLO_ERR_ARG, "crop_resize: bad arguments");
  if (max_short > kCrMaxShort)
    return fail(nullptr, MIYOLO_ERR_UNSUPPORTED, "crop_resize: crops with a short side above %d px are not supported (got %d)", kCrMaxShort, max_short);
  if (n == 0) return 0;
  CropResizeArgs a;
  a.frame = static_cast<const uint8_t*>(frame); a.boxes = boxes; a.out = static_cast<uint8_t*>(out);
  a.H = H; a.W = W; a.n = n; a.S = size;
  a.tmp_rows = max_short + 2 * ((max_short + size - 1) / size) + 8;      // rows the vertical pass of S outputs can touch
  const size_t lds = crop_resize_lds_bytes(size, a.tmp_rows);
  if (lds > 160 * 1024) return fail(nullptr, MIYOLO_ERR_UNSUPPORTED, "crop_resize: %zu bytes of LDS needed", lds);
  {   // the attribute is per device; setting it is idempotent and cheap next to the launch, so no cached flag (no global state)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(crop_resize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(nullptr, MIYOLO_ERR_HIP, "crop_resize attribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(crop_resize_kernel, dim3(n), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(nullptr, MIYOLO_ERR_HIP, "crop_resize launch: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

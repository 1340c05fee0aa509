/*
 * miyolo.h - C ABI of the MI355X-native YOLOv8 detect + classify inference path.
 *
 * The reference (kanaksharma67/manual-yolo) has no FFI of its own: its hot path is the
 * Python call  results = model(frame)  into ultralytics' YOLO object
 *     detect.py:541   model(frame)[0]                    (yolov8m detector)
 *     detect.py:121   rank_model(crop)[0]                (yolov8n-cls classifier)
 *     pipe.py:179     model.predict(source=frame, imgsz=1280, conf=0.35, verbose=False)
 *     yolo.py:361     model(frame, conf=...)
 * which runs LetterBox -> DetectionModel/ClassificationModel forward -> non_max_suppression
 * -> Results on torch CPU kernels.  This library is what a replacement YOLO object binds
 * instead of those torch kernels (manual_yolo_amd/engine.py is that binding, via ctypes;
 * INTEGRATION.md shows the stub).  Each entry point names the reference-side step it
 * replaces.
 *
 * Conventions
 *   - plain C types only; all tensors are caller-owned DEVICE pointers (HBM), never freed
 *     or reallocated by the library; weights must stay alive as long as the handle.
 *   - every call is asynchronous on the hipStream_t passed as `void* stream` (0 = null
 *     stream); the library never synchronises and never allocates on the hot path: the
 *     caller provides the workspace (size from miyolo_workspace_bytes).
 *   - return value: 0 = OK, negative = miyolo_status; miyolo_last_error() gives the text.
 *     Nothing throws across the ABI.
 *   - a handle is bound to one device and is NOT re-entrant (one call in flight per
 *     handle); no global state (the text of errors raised without a handle is thread-local), so one
 *     process per GPU or one handle per thread both work.
 *   - activations are NHWC; the input image batch is uint8 NHWC with 3 channels in the
 *     channel order the packed stem weights expect (the Python host packs the stem for
 *     BGR frames, as the reference passes them).
 */
#ifndef MIYOLO_H
#define MIYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIYOLO_ABI_VERSION 1

typedef enum {
  MIYOLO_OK = 0,
  MIYOLO_ERR_ARG = -1,         /* null pointer, bad size, bad enum */
  MIYOLO_ERR_SHAPE = -2,       /* H/W not a multiple of the model stride, B < 1, ... */
  MIYOLO_ERR_UNSUPPORTED = -3, /* op table uses something the kernels do not implement */
  MIYOLO_ERR_WORKSPACE = -4,   /* workspace too small */
  MIYOLO_ERR_HIP = -5,         /* a HIP call failed; text has the HIP error */
  MIYOLO_ERR_NO_DEVICE = -6    /* no gfx950 device visible */
} miyolo_status;

typedef enum { MIYOLO_F32 = 0, MIYOLO_F16 = 1, MIYOLO_F8 = 2 /* e4m3 (OCP) activations + weights, fp32 accumulate: config 5 */ } miyolo_dtype;

typedef enum {
  MIYOLO_OP_STEM = 0,     /* conv3x3 s2 p1 on the uint8 3-channel input, /255 folded in, +bias, SiLU */
  MIYOLO_OP_CONV = 1,     /* conv kxk (k=1|3, s=1|2, p=k/2) +bias, optional SiLU, optional residual add */
  MIYOLO_OP_MAXPOOL5 = 2, /* max_pool2d(k=5, s=1, p=2) on a channel slice (SPPF) */
  MIYOLO_OP_DECODE = 3,   /* Detect: DFL softmax-expectation, dist2bbox, x stride, sigmoid(cls) */
  MIYOLO_OP_CLS_HEAD = 4  /* Classify: global avg-pool + Linear + softmax */
} miyolo_op_kind;

/* A view of `ch_cnt` channels starting at `ch_off` inside activation buffer `buf`.
 * upsample = 1 reads pixel (h>>1, w>>1) of a buffer at half the consumer's resolution
 * (nn.Upsample(scale_factor=2, mode='nearest') folded into the consumer). buf = -1: absent. */
typedef struct {
  int32_t buf, ch_off, ch_cnt, upsample;
} miyolo_view;

/* Activation buffer: per image (H/down) x (W/down) x channels, NHWC.
 * dtype: -1 = the handle's activation dtype, MIYOLO_F32 = fp32 (head outputs). */
typedef struct {
  int32_t channels, down, dtype, reserved;
} miyolo_buf;

/* One step of the layer program.  Replaces one fused Conv2d+BatchNorm2d+SiLU module
 * ([3P] ultralytics.nn.modules.conv.Conv after model.fuse()), MaxPool2d, Detect._inference
 * or Classify tail; chunk/cat/Upsample modules are expressed through views and cost nothing. */
typedef struct {
  int32_t kind;          /* miyolo_op_kind */
  int32_t ksize, stride, act;     /* CONV/STEM: kernel 1|3, stride 1|2, act 0 none / 1 SiLU */
  int32_t cin, cout;     /* CONV/STEM: logical channel counts (cin = sum of src ch_cnt) */
  int32_t n_src;         /* CONV: 1 or 2 (channel concat of two views, k=1 only) */
  miyolo_view src[3];    /* CONV: src[0..n_src); DECODE: the three raw head maps (box|cls, f32) */
  miyolo_view dst;       /* CONV/STEM/MAXPOOL5: where the result goes */
  miyolo_view res;       /* CONV: residual added AFTER the activation (Bottleneck.add) or buf=-1 */
  int32_t weight, bias;  /* indices into the weight pointer table (CONV/STEM/CLS_HEAD) */
  int32_t level_stride[3]; /* DECODE: stride of each level (8,16,32) */
  /* MIYOLO_F8 only (ignored otherwise).  Quantisation is static, chosen by the host (manual_yolo_amd/quant.py):
   * activation buffers hold e4m3 values q = x / s with one scale s per producing op; conv weights are e4m3 with one
   * scale per output channel, the input scales folded in per input channel before quantising.  A conv computes
   *   x[n] = acc[n] * qscale[n] + bias[n];  y = act(x) (+ res * res_scale);  stored = fp8(y * out_inv_scale)
   * (fp32 raw head maps store y).  qscale / bias_init index the weight table: float [cout] padded like the bias;
   * bias_init[n] = bias[n] / qscale[n] (accumulators of the halo-slab kernel start there). */
  int32_t qscale, bias_init;
  float out_inv_scale, res_scale;
  int32_t reserved[1];
} miyolo_op;

typedef struct {
  int32_t abi_version;   /* MIYOLO_ABI_VERSION */
  int32_t task;          /* 0 = detect, 1 = classify */
  int32_t dtype;         /* miyolo_dtype of activations and packed conv weights */
  int32_t nc;            /* classes */
  int32_t reg_max;       /* Detect.reg_max (16) */
  int32_t max_stride;    /* H and W must be multiples of this (32) */
  int32_t n_bufs, n_ops, n_weights;
  int32_t reserved[7];
} miyolo_desc;

typedef struct miyolo_engine* miyolo_handle;

/* ABI/version probes (no GPU needed). */
int miyolo_abi_version(void);
/* Conv weight rows are [cout][kpad]: K = (ky,kx,cin) flattened (concat segments in order),
 * zero padded at the END to a multiple of this many elements (one staging step of the conv
 * kernel, 128 B): 32 for F32, 64 for F16.  Every source view must hold a multiple of
 * 16 B worth of channels (4 for F32, 8 for F16, 16 for F8; F8 rows are padded to 128 elements). */
int miyolo_k_align(int dtype);

/* Replaces: YOLO(path) model construction + AutoBackend(fuse=True) (detect.py:20-21).
 * `weights[i]` are device pointers; conv weights are BN-folded, laid out [cout][kpad]
 * (see miyolo_k_align) in `desc->dtype`; biases (and the F8 qscale / bias_init arrays) fp32 [cout] zero padded to a
 * multiple of 128 floats (the kernels fetch them 16 at a time through the scalar cache, which is not range-checked; the
 * host side of this repository adds 256 floats of slack on top); the stem weight is [cout][32] in
 * `desc->dtype`, K' = 8q+j with q<3: (ky=q, byte j = kx*3+c, j<8), q=3: j<3 -> (ky=j,kx=2,c=2),
 * rest zero; the 1/255 input scale is NOT folded (the kernel divides the uint8 pixel by 255 as
 * the reference's preprocess does); CLS_HEAD weight fp32 [nc][c]. */
int miyolo_create(const miyolo_desc* desc, const miyolo_buf* bufs, const miyolo_op* ops,
                  const void* const* weights, int device, miyolo_handle* out);
void miyolo_destroy(miyolo_handle h);
const char* miyolo_last_error(miyolo_handle h); /* h may be NULL: last create() error */

/* Replaces: the `classes=` keyword of `model(frame, classes=[...])` ([3P] non_max_suppression(classes=...)): only
 * candidates whose best class is in `classes` enter the sort / NMS / max_det steps (filter BEFORE NMS, as upstream).
 * State of the handle, used by miyolo_detect and miyolo_nms until changed; n = 0 or classes = NULL removes the filter.
 * Models with more than 256 classes: MIYOLO_ERR_UNSUPPORTED. */
int miyolo_set_classes(miyolo_handle h, const int32_t* classes, int n);

/* Bytes of scratch the caller must pass for a batch of B images of H x W pixels. */
size_t miyolo_workspace_bytes(miyolo_handle h, int B, int H, int W);

/* Replaces: DetectionModel forward + Detect._inference + non_max_suppression + scale_boxes
 * (everything inside `model(frame)` after LetterBox, detect.py:541).
 *   in          uint8 [B,H,W,3] letterboxed frames
 *   conf, iou   thresholds (reference defaults 0.25 / 0.7); agnostic: class-agnostic NMS
 *   max_det     rows per image in the outputs (reference default 300)
 *   scale       optional device float [B,5] = gain, pad_x, pad_y, orig_w, orig_h to undo the
 *               letterbox as scale_boxes/clip_boxes do; NULL leaves boxes in network pixels
 *   out_dets    float [B,max_det,6] = x1,y1,x2,y2,conf,cls in NMS keep order (zero padded)
 *   out_counts  int32 [B] detections kept per image
 *   out_anchor  optional int32 [B,max_det] anchor index of every kept box (parity checks) */
int miyolo_detect(miyolo_handle h, const uint8_t* in, int B, int H, int W, float conf, float iou,
                  int agnostic, int max_det, const float* scale, float* out_dets,
                  int32_t* out_counts, int32_t* out_anchor, void* workspace, size_t workspace_bytes,
                  void* stream);

/* Option "nms_async" = 1 (default 0): miyolo_detect enqueues its score filter + NMS on an internal stream behind the
 * decode and does NOT join the caller's stream, so the next call's backbone runs beside it (the NMS is one workgroup per
 * image: 64 of 256 CUs for ~0.4 ms at batch 64).  The outputs of the call are then complete only after
 * miyolo_wait_outputs(h, stream), which orders `stream` behind the most recent NMS (an event wait, no host
 * synchronisation); a later call's decode, and every other entry point that uses the workspace, waits by itself.
 * Ignored while "graph", "profile" or "batch_split" are on.  Measured on configuration 1: no gain (8 109 vs 8 262 frames/s) -
 * the NMS workgroups hold 128 KiB of LDS and cannot share a CU with the conv kernels', so they only delay them; kept for
 * callers whose next work is not LDS-bound.  With the option off (default) outputs are complete in stream
 * order as before and miyolo_wait_outputs is a no-op. */
int miyolo_wait_outputs(miyolo_handle h, void* stream);

/* Replaces: the forward up to Detect's return value `y` ([3P] Detect._inference):
 * y float [B, 4+nc, A] = (cx,cy,w,h in network pixels, sigmoid class scores). The parity
 * gate on head values (1e-4 vs the CPU path) reads this. */
int miyolo_head_raw(miyolo_handle h, const uint8_t* in, int B, int H, int W, float* y,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: non_max_suppression + scale_boxes alone, on a caller-supplied y [B,4+nc,A]
 * (used to test the post-process bit-for-bit against the oracle on identical inputs). */
int miyolo_nms(miyolo_handle h, const float* y, int B, int A, int H, int W, float conf, float iou,
               int agnostic, int max_det, const float* scale, float* out_dets, int32_t* out_counts,
               int32_t* out_anchor, void* workspace, size_t workspace_bytes, void* stream);

/* Replaces: ClassificationModel forward (detect.py:121, `rank_model(crop)` after the
 * classify transforms).  in uint8 [B,H,W,3]; logits/probs float [B,nc] (either may be NULL). */
int miyolo_classify(miyolo_handle h, const uint8_t* in, int B, int H, int W, float* logits,
                    float* probs, void* workspace, size_t workspace_bytes, void* stream);

/* Images processed per pass for a B x H x W call (the batch is split so that 32-bit element
 * offsets suffice and, optionally, so that activations stay cache resident). */
int miyolo_chunk(miyolo_handle h, int B, int H, int W);

/* Kernel launches one miyolo_classify() call of H x W crops enqueues under the current options: 1 when the one-launch
 * classifier (csrc/cls_mega.h) serves that size, else the layer count.  lds_bytes (may be NULL) receives the LDS one
 * workgroup (= one image) of the one-launch kernel holds, 0 for the layered path.  < 0: error. */
int miyolo_classify_launches(miyolo_handle h, int H, int W, size_t* lds_bytes);

/* Options: "max_chunk" (images per pass, 0 = automatic), "force_wc"/"force_tc" (pin the conv
 * tile shape: waves along channels 1|2, 16-channel tiles per wave 1..6; tests and tuning),
 * "profile" (see miyolo_profile_read), "conv_impl" (0: register-staged double-buffered conv
 * kernel; 1: LDS-DMA 3-stage ring kernel; 2: as 1, plus the LDS halo-tile kernel for 3x3
 * stride-1 convolutions; 3 (default): persistent LDS-DMA ring kernel, with the 2-D-tile kernel for
 * narrow 3x3 layers; 4: persistent halo kernel; 5: warp-specialised kernel; 6: two workgroups
 * per CU; 7: 2-D-tile kernel where eligible, else 3), "t2d" (1 default: conv_impl 3 uses the
 * 2-D-tile kernel), "dmh_auto" (0 default: conv_impl 3 hands launches with 1-2 tiles per CU to
 * kernel 6), "ncu" (width of the persistent grids, default = the device's CU count), "graph"
 * (1: detect/classify calls are captured into a hipGraph and replayed while shape, thresholds,
 * stream and pointers stay the same - captured on the first call with that key; a capture is kept only if it holds exactly
 * one kernel node per launch and nothing else (miyolo_graph_info), otherwise the call runs directly; the head chains stay on
 * the caller's stream; not with "batch_split" / "cls_streams" > 1; needs a non-default stream; default 0), "h2" (1 default: 3x3 stride-1 layers run on the halo-slab kernel conv_h2.h where its tiles cover at least
 * "h2_min_util" percent (70) of the map; "h2_warm" = 1 selects its persistent form), "h3" (2 default: in F16 a 3x3 stride-1 layer
 * runs on conv_h3.h - the halo-slab kernel cut for three workgroups per CU, 128-pixel tiles, bit-identical results - where its
 * tile count fills the chip better than conv_h2's, i.e. on the 20 x 20 level; 1: wherever eligible; 0: never), "pair8" (1 default: the
 * ring kernel stores f16 outputs 16 bytes at a time over channel-tile pairs where the views are 8-channel aligned; 0: 8 bytes;
 * same results), "h4" (0 default; 1: 3x3 stride-1 F16 layers on 80- / 160-wide maps run on conv_h4.h - one workgroup per CU,
 * 512-register waves; bit-identical, measured 38-45 % slower: an experiment), "pw" (0 default;
 * 1: in F16 an eligible 1x1 layer runs on the streaming kernel conv_pw.h - pixel fragments loaded straight into registers,
 * weights through the LDS by a producer wave; bit-identical results, measured 6-8 % slower than the ring kernel), "cls_mega" (1 default: an f16 classifier whose activations fit LDS runs as ONE launch, cls_mega.h; 0: one launch per
 * layer; bit-identical results), "head_lanes" (1 default: detect runs the Detect head's independent conv chains - per level
 * the first conv and the box / class branches behind it - on internal side streams, forked and joined by events around
 * the caller's stream; 0: everything in order on the caller's stream; same kernels, same results), "fuse_prefilter" (1
 * default: miyolo_detect's decode kernel also runs the NMS score filter; 0: separate pass; same results), "sppf_fuse" (1
 * default: three chained MAXPOOL5 ops run as one launch; same results), "bneck_fuse" (1 default: in F16 a narrow Bottleneck -
 * conv3x3, conv3x3 + residual of the first one's input, 16 / 32 / 48 channels, the intermediate read by nobody else - runs
 * as one launch with the intermediate tile in LDS; same results as the two launches on the 2-D-tile kernel), "stem_fuse" (1
 * default: in F16 the stem, the stride-2 conv behind it and - where it is a C1 -> C1 1x1 conv read by nobody else - the conv behind
 * that run as one launch, the maps between them never written; 2: the first two only; 0: one launch each - same results), "batch_split" (0 default;
 * K > 1: a batch that fits one pass runs as K part batches on K streams when the workspace holds K part plans - measured
 * +0.8..1.4 % on configuration 1, left off), "cls_streams" (1 default:
 * > 1 makes miyolo_classify fork the batch over that many internal streams, joined by events - measured slower), and the
 * timing-experiment switches "ablate" / "dbg_op" of the non-shipped builds.  Setting any option
 * drops the captured graphs. */
int miyolo_set_option(miyolo_handle h, const char* key, int value);

/* Debug/parity taps (B must not exceed miyolo_chunk): copy activation buffer `buf` out as /
 * in from fp32 NHWC [B, H/down, W/down, channels], and run ops [first,last) of the program.
 * Per-layer parity tests against the oracle are built from these. */
int miyolo_read_buffer(miyolo_handle h, int buf, int B, int H, int W, float* out, void* workspace,
                       void* stream);
int miyolo_write_buffer(miyolo_handle h, int buf, int B, int H, int W, const float* in,
                        void* workspace, void* stream);
int miyolo_run_ops(miyolo_handle h, int first, int last, const uint8_t* in, int B, int H, int W,
                   void* workspace, size_t workspace_bytes, void* stream);

/* hipGraph census (option "graph"): out6 = { graphs held, nodes / kernel nodes of the most recent capture, kernel launches the
 * captured call issued, captures rejected because those numbers disagreed (the call then ran directly), kernel launches
 * issued by the handle in total }.  A captured call is replayed only if its graph holds exactly one kernel node per launch
 * and no other node. */
int miyolo_graph_info(miyolo_handle h, int32_t* out6);

/* Debug: the NMS candidate counters (anchors above conf per image) of the last detect call's chunk, copied to the host
 * (synchronises the device).  Lets a test see that a replayed graph resets them (tests/test_gpu_detect.py). */
int miyolo_debug_candidate_counts(miyolo_handle h, const void* workspace, int B, int32_t* out_host);

/* Bench support: with option "profile" = 1 every op launch is bracketed by hipEvents on the
 * call's stream; miyolo_profile_read synchronises on them and returns, per recorded launch,
 * the op index, the conv kernel variant (conv_impl*1000 + ksize*100 + waves_along_channels*10 +
 * tiles, 0 for non-conv ops) and the duration in ms.  Returns the number of records; passing NULL arrays
 * only counts, passing arrays consumes the records.  Not for the hot path. */
int miyolo_profile_read(miyolo_handle h, int max_records, int32_t* op_index, int32_t* cfg, float* ms);
/* Timing-experiment builds only (-DMIYOLO_ABLATE=1, option "dbg_op"): per-wave cycle stamps of the
 * persistent conv kernels for one op; `out` is a HOST buffer of 2*256*8*8 uint64 (second half: event trace of
 * workgroup 0, conv_ws.h). Synchronises. */
int miyolo_debug_stamps(miyolo_handle h, unsigned long long* out);
/* Algorithmic flops (2*MAC) and compulsory bytes of ONE op for a B x H x W batch. */
int miyolo_op_work(miyolo_handle h, int op_index, int B, int H, int W, double* flops, double* bytes);

/* Bench support: algorithmic work of one forward for a B x H x W batch, from the op table:
 * flops = 2*MAC of the conv ops; bytes = compulsory layer-wise traffic (each op reads its
 * inputs once, writes its output once, weights once per batch). */
int miyolo_work(miyolo_handle h, int B, int H, int W, double* flops, double* bytes);

/* Detect pre-processing on the device: what the reference's `model(frame)` does first
 * (detect.py:541 -> [3P] LetterBox: cv2.resize INTER_LINEAR to new_w x new_h, constant pad to dst_w x dst_h).
 * src: device uint8 [B][src_h][src_w][3], dst: device uint8 [B][dst_h][dst_w][3]; the resized image lands at
 * (top, left), everything else is pad_value (114).  The host computes the geometry exactly as LetterBox does
 * (manual_yolo_amd/preprocess.py letterbox_geometry).  Bit-exact against oracle/pre_ref.py letterbox.  Stateless
 * (no handle); errors are reported through miyolo_last_error(NULL).  Asynchronous on `stream`. */
int miyolo_letterbox(const void* src, int B, int src_h, int src_w, void* dst, int dst_h, int dst_w, int top, int left,
                     int new_h, int new_w, int pad_value, void* stream);

/* Classifier input on the device, fused with the crop: for each box, what the reference does
 * between a detection and `rank_model(crop)` (detect.py:100-113 safe_crop, detect.py:121 -> the checkpoint's pickled
 * transforms Resize(size, bilinear, antialias) + CenterCrop(size), i.e. Pillow's 8-bit resample).
 * frame: device uint8 [H][W][3]; boxes: device int32 [n][4] = x1,y1,x2,y2 already clamped to the frame (safe_crop's
 * arithmetic is the host's, manual_yolo_amd/chain.py); out: device uint8 [n][size][size][3], same channel order as the
 * frame.  max_short: largest min(width, height) over the boxes (<= 640).  Byte-exact against PIL
 * (tests/test_gpu_preprocess.py).  Stateless; errors through miyolo_last_error(NULL).  Asynchronous on `stream`. */
int miyolo_crop_resize(const void* frame, int H, int W, const int32_t* boxes, int n, int size, int max_short, void* out, void* stream);

/* Sliced ("SAHI-style") inference on the device (reference pipe.py:43-45,183-194 ->
 * [3P] sahi.get_sliced_prediction with 640 x 640 slices, 20 % overlap).  miyolo_slice_batch cuts one device frame
 * [H][W][3] into the batch out [n][sh][sw][3]: slice i = frame[y1:y2, x1:x2] (boxes int32 [n][4], inside the frame) at the
 * top-left of its canvas, the rest pad_value (slices smaller than the canvas only at frames smaller than a slice).
 * Stateless; errors through miyolo_last_error(NULL). */
int miyolo_slice_batch(const void* frame, int H, int W, const int32_t* boxes, int n, void* out, int sh, int sw, int pad_value,
                       void* stream);

/* The merge step of sliced inference: the per-slice detections (dets float [n_slices][slice_max_det][6] in slice
 * coordinates + counts, as miyolo_detect wrote them; boxes = the slice origins) are shifted into frame coordinates and
 * merged by the SAME class-aware NMS as the per-frame post-process (score order, cls * 7680 offset unless agnostic, strict
 * IoU > iou), keeping at most max_det.  H x W: any frame size the handle's workspace covers with at least
 * n_slices * slice_max_det anchors (the slice size does).  y_scratch: device float [(4 + nc) * n_slices * slice_max_det].
 * out_index (optional): for every kept box its candidate slot slice * slice_max_det + row. */
int miyolo_merge_slices(miyolo_handle h, const float* dets, const int32_t* counts, const int32_t* boxes, int n_slices, int slice_max_det,
                        int H, int W, float iou, int agnostic, int max_det, float* y_scratch, float* out_dets, int32_t* out_counts,
                        int32_t* out_index, void* workspace, size_t workspace_bytes, void* stream);

/* The merge step as the reference's call reaches it: pipe.py:186-188 calls [3P] sahi.get_sliced_prediction WITHOUT any
 * postprocess_* argument, i.e. with sahi's defaults - postprocess_type "GREEDYNMM", match metric "IOS", threshold 0.5,
 * class-aware.  Candidates = the per-slice detections (layout as for miyolo_merge_slices; a full-frame pass is passed as
 * one more "slice" at origin (0, 0) LAST, as sahi appends it), shifted into frame coordinates, clipped to frame_w x
 * frame_h, degenerate boxes dropped.  Per class (all together if `agnostic`): in descending score order a candidate
 * still in the pool becomes a keep and removes every later candidate whose inter / min(area) (metric 0 = IOS) or IoU
 * (metric 1) is not below `threshold` (fp32, as torch); then each keep absorbs its matched candidates in score order,
 * each only if it still matches the keep's grown box (metric > threshold, fp64, as numpy): box = hull, score and class =
 * the keep's.  Restated in oracle/post_ref.py greedy_nmm_merge; kept-set, boxes and order identical
 * (tests/test_gpu_sahi.py).  out_dets [max_out][6] rows x1,y1,x2,y2,score,cls by descending score (ties: sahi's output
 * order), zero padded; out_counts [2] = rows written, merged boxes in total (or -n when more than 4096 candidates came
 * in: nothing written); out_index (optional) [max_out] the keep's candidate slot slice * slice_max_det + row.
 * Stateless; errors through miyolo_last_error(NULL).  Asynchronous on `stream`. */
int miyolo_merge_slices_nmm(const float* dets, const int32_t* counts, const int32_t* boxes, int n_slices, int slice_max_det,
                            int frame_h, int frame_w, int metric, float threshold, int agnostic, int max_out, float* out_dets,
                            int32_t* out_counts, int32_t* out_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIYOLO_H */

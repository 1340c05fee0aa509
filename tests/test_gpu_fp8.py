"""-m gpu: the fp8 (e4m3) path of BASELINE config 5 - per-conv numerics against torch conv2d on the DEQUANTISED operands,
and yolov8m detections against the CPU oracle at the documented fp8 bar."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from manual_yolo_amd.engine import Engine, engine_from_weights
from manual_yolo_amd.quant import FP8_MAX, QuantSpec, dequant_fp8_bytes, fp8_round, quantize_conv_weight
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
from oracle.post_ref import non_max_suppression
from oracle.yolo_ref import RefYolo
from tests.gpu_util import conv_program

pytestmark = pytest.mark.gpu


def _run_conv_fp8(x_list, w, b, srcs, k, s, act, res, B, H, W, in_scales, out_scale, res_scale=None, impl=3, out_f32=False):
    """One fp8 conv through the C ABI.  x_list real-valued fp32 NHWC inputs; returns (real-valued output, reference)."""
    cout = w.shape[0]
    prog = conv_program(srcs, cout, k, s, act, res is not None, None, 0, out_f32)
    nb = len(prog.bufs)
    dst_buf = nb - 2 if res is not None else nb - 1
    bs = {}
    for i, (ld, off, cnt, up) in enumerate(srcs):
        bs[1 + i] = np.full(ld, in_scales[i], np.float32)
    if not out_f32:
        bs[dst_buf] = np.full(cout, out_scale, np.float32)
    if res is not None:
        bs[nb - 1] = np.full(cout, res_scale, np.float32)
    q = QuantSpec(bs, {} if out_f32 else {0: out_scale})
    sd = {"t.weight": torch.from_numpy(w), "t.bias": torch.from_numpy(b)}
    eng = Engine(prog, sd, 1e-3, "f8", 0, quant=q)
    eng.set_option("conv_impl", impl)
    for i, x in enumerate(x_list):
        eng.write_buffer(1 + i, torch.from_numpy(x), H, W)
    if res is not None:
        eng.write_buffer(nb - 1, torch.from_numpy(res), H, W)
    eng.run_ops(0, 1, None, B, H, W)
    y = eng.read_buffer(dst_buf, B, H, W).cpu().numpy()
    # reference on the values the kernel actually sees: inputs / weights rounded to e4m3 at their scales
    xs = []
    for i, x in enumerate(x_list):
        xq = fp8_round(torch.from_numpy(x) / in_scales[i]) * in_scales[i]
        if srcs[i][3]:
            xq = xq.repeat_interleave(2, 1).repeat_interleave(2, 2)
        xs.append(xq[..., srcs[i][1]:srcs[i][1] + srcs[i][2]])
    xin = torch.cat(xs, -1)
    s_in = np.concatenate([np.full(c[2], in_scales[i], np.float32) for i, c in enumerate(srcs)])
    qw, qs = quantize_conv_weight(torch.from_numpy(w), s_in)
    K = w.shape[1] * k * k
    wq = (dequant_fp8_bytes(qw)[:, :K] * qs.view(-1, 1)).view(cout, k, k, w.shape[1]).permute(0, 3, 1, 2) / torch.from_numpy(s_in).view(1, -1, 1, 1)
    ref = F.conv2d(xin.permute(0, 3, 1, 2), wq, torch.from_numpy(b), stride=s, padding=k // 2)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + fp8_round(torch.from_numpy(res) / res_scale) * res_scale
    return y, ref.numpy()


@pytest.mark.parametrize("impl", [3, 8])
@pytest.mark.parametrize("cin,cout,k,s,H,W,B,res", [
    (96, 96, 3, 1, 80, 80, 1, True),        # halo-slab kernel: one 128-channel chunk, 75 % full
    (192, 192, 3, 1, 40, 40, 2, False),     # 1.5 chunks, three 64-channel tiles
    (288, 288, 3, 1, 20, 20, 2, True),      # 2.25 chunks, 48-channel tiles
    (48, 48, 3, 1, 32, 32, 2, True),
    (384, 64, 3, 1, 20, 20, 1, False),
    (48, 96, 3, 2, 32, 32, 2, False),       # stride 2: ring kernel
    (192, 384, 3, 2, 16, 16, 1, False),
    (96, 96, 1, 1, 16, 16, 2, False),       # 1x1
    (576, 192, 1, 1, 8, 8, 2, False),
    (1152, 576, 1, 1, 8, 8, 1, False),
    (96, 96, 1, 1, 16, 16, 2, True),        # ring kernel with a residual: the paired (8-byte) fp8 epilogue's residual path
    (64, 128, 1, 1, 12, 12, 2, True),
])
def test_fp8_conv_vs_dequantised_reference(cin, cout, k, s, H, W, B, res, impl):
    rng = np.random.default_rng(cin + cout * 3 + k)
    x = rng.standard_normal((B, H, W, cin)).astype(np.float32) * 2.0
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.5
    r = rng.standard_normal((B, H // s, W // s, cout)).astype(np.float32) * 1.5 if res else None
    in_scale, out_scale, res_scale = float(np.abs(x).max()) / FP8_MAX, 8.0 / FP8_MAX, 8.0 / FP8_MAX
    y, ref = _run_conv_fp8([x], w, b, [(cin, 0, cin, 0)], k, s, True, r, B, H, W, [in_scale], out_scale, res_scale, impl)
    # the output itself is rounded to e4m3 at out_scale: half an ulp is 2^-4 relative (6.25 %), plus the smallest step;
    # values beyond 448 * out_scale saturate (e4m3fn has no infinity)
    ref = np.clip(ref, -FP8_MAX * out_scale, FP8_MAX * out_scale)
    err = np.abs(y - ref)
    bar = 0.0665 * np.abs(ref) + out_scale * 2.0 ** -9 * 1.01 + 2e-3
    assert (err <= bar).all(), f"{int((err > bar).sum())} of {err.size} outside the e4m3 rounding bar; worst {float((err / bar).max()):.2f}x"
    assert np.abs(y - ref).mean() < 0.03 * np.abs(ref).mean() + 1e-3


def test_fp8_concat_upsample_scales_folded_into_weights():
    """C2f.cv2 / FPN 1x1 over a concat of slices with DIFFERENT activation scales (one upsampled): the scales are folded
    into the weights per input channel, fp32 raw-map output (head final convs)."""
    rng = np.random.default_rng(3)
    B, H, W, c0, c1, cout = 2, 8, 12, 96, 128, 64      # 96 is not a multiple of 128: the engine swaps the views
    x0 = rng.standard_normal((B, H // 2, W // 2, c0)).astype(np.float32) * 3.0
    x1 = rng.standard_normal((B, H, W, c1)).astype(np.float32) * 0.3
    w = (rng.standard_normal((cout, c0 + c1, 1, 1)) / np.sqrt(c0 + c1)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y, ref = _run_conv_fp8([x0, x1], w, b, [(c0, 0, c0, 1), (c1, 0, c1, 0)], 1, 1, False, None, B, H, W,
                           [float(np.abs(x0).max()) / FP8_MAX, float(np.abs(x1).max()) / FP8_MAX], 1.0, out_f32=True)
    assert np.abs(y - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())          # fp32 output: only accumulation-order noise


@pytest.fixture(scope="module")
def m8():
    sd, meta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
    calib = torch.from_numpy(np.concatenate([synth_frames(4, 640, 640, seed=101), synth_frames(2, 640, 640, seed=102, kind="blocks")]))
    return sd, meta, engine_from_weights(sd, meta, "f8", 0, bgr_input=False, calib_frames=calib)


def _head_stats(got, want):
    """per level and branch: (corr, relative rms) of raw head maps `got` against `want` (lists of [B, 64+nc, h, w] arrays)."""
    out = {}
    for lvl, (g_, w_) in enumerate(zip(got, want)):
        for nm, sl in (("box", slice(0, 64)), ("cls", slice(64, None))):
            g, w = g_[:, sl].ravel().astype(np.float64), w_[:, sl].ravel().astype(np.float64)
            gc, wc = g - g.mean(), w - w.mean()
            out[lvl, nm] = (float((gc * wc).sum() / (np.linalg.norm(gc) * np.linalg.norm(wc))), float(np.linalg.norm(g - w) / np.linalg.norm(wc)))
    return out


def test_fp8_every_layer_matches_the_fakequant_oracle_on_the_same_input_bytes(m8):
    """The tight model-level bar for the fp8 arithmetic.  Every STEM / CONV op of yolov8m at 640 x 640,
    with the engine's real scales, bias corrections, swapped concat views and residual scales: the op's input bytes are
    read back from the engine, oracle/quant_ref.py computes that ONE layer from them on the CPU, and the stored e4m3 codes
    are compared.  Two correct executions differ only where a value sits within fp32 summation-order distance (~1e-6
    relative for an fp32 accumulator; the fp8 MFMA's K = 128 reduction keeps fewer bits, tests/fp8_util.py) of a rounding
    boundary: measured 0.2-2.7 codes in 10^3, by 1-3 steps where the value is a small difference of large terms.  Bars: at
    most 5 codes in 10^3 per op, and EVERY stored code inside the interval the accumulator tolerance allows; the stem (f16
    MFMA, fp32-exact accumulation) at most 1 in 10^4, one step; head raw maps (fp32) within 1e-4."""
    from tests.fp8_util import teacher_forced_layers
    sd, meta, eng = m8
    frames = torch.from_numpy(synth_frames(1, 640, 640, seed=1))
    rows = teacher_forced_layers(eng, sd, meta, frames)
    assert len(rows) == 83
    for name, n, nd, mx, nout in rows:
        if nd < 0:
            assert mx < 1e-4, (name, mx)
        elif name == "model.0":
            assert nd <= 1e-4 * n and mx <= 1, (name, nd, mx)
        else:
            assert nd <= 5e-3 * n and nout == 0, f"{name}: {nd} of {n} codes differ (max {mx} steps), {nout} outside the accumulator interval"


def test_fp8_end_to_end_agrees_with_the_oracle_as_well_as_the_oracle_agrees_with_itself(m8):
    """End to end a tight bar is impossible for ANY two executions of a deep e4m3 network that differ in fp32 summation
    order: one rounding that lands on the other code (a 6-12 % change of that element) shifts ~1000 downstream sums enough
    to flip ~5 % of their roundings, and the flips avalanche.  Measured on the CPU alone: the fake-quant oracle against
    itself with every accumulator perturbed by 1e-6 relative (oracle/quant_ref.py acc_noise) differs at the head by as
    much as fp8 differs from fp32.  So the end-to-end bar is that yardstick: the HIP engine may differ from the oracle by
    at most 1.3 x what the oracle differs from its own perturbed run (relative rms per head map), and must share at least
    0.8 x as many kept anchors with it."""
    from tests.fp8_util import oracle_from_engine
    sd, meta, eng = m8
    frames = synth_frames(2, 640, 640, seed=1)
    u8 = torch.from_numpy(frames).permute(0, 3, 1, 2).contiguous()
    ref = oracle_from_engine(eng, sd, meta)
    y0, raw0 = ref.forward_u8(u8)
    ref2 = oracle_from_engine(eng, sd, meta)
    ref2.acc_noise = 1e-6
    y1, raw1 = ref2.forward_u8(u8)
    x = torch.from_numpy(frames).cuda()
    yg = eng.head_raw(x).cpu().numpy()
    dec = [op for op in eng.prog.ops if op.kind == 3][0]
    rawg = [eng.read_buffer(v.buf, 2, 640, 640).cpu().numpy().transpose(0, 3, 1, 2) for v in dec.src]
    self_ = _head_stats([r.numpy() for r in raw1], [r.numpy() for r in raw0])
    hip = _head_stats(rawg, [r.numpy() for r in raw0])
    for k in self_:
        print(f"level {k[0]} {k[1]}: oracle vs perturbed oracle corr {self_[k][0]:.4f} rel {self_[k][1]:.3f} | HIP vs oracle corr {hip[k][0]:.4f} rel {hip[k][1]:.3f}")
        assert hip[k][1] <= 1.3 * self_[k][1] + 0.01, k
    _, i0 = non_max_suppression(y0.numpy(), 0.25, 0.7)
    _, i1 = non_max_suppression(y1.numpy(), 0.25, 0.7)
    _, ig = non_max_suppression(yg, 0.25, 0.7)
    tot = sum(len(i) for i in i0)
    c_self = sum(len(np.intersect1d(a, b)) for a, b in zip(i0, i1))
    c_hip = sum(len(np.intersect1d(a, b)) for a, b in zip(i0, ig))
    print(f"kept anchors shared with the oracle's {tot}: perturbed oracle {c_self}, HIP engine {c_hip}")
    assert c_hip >= 0.8 * c_self


def test_fp8_yolov8m_detections_vs_fp32_oracle(m8):
    """Accuracy of the fp8 mode as measured on seeded random-init yolov8m (640 x 640 noise frames NOT in the calibration
    set) against the fp32 CPU path - NOT evidence for a trained detector (none exists here; the trained classifier is in
    test_fp8_rank_classifier).  Head logits: correlation / relative rms per level; detections: share of the oracle's kept
    anchors, box and score differences on the shared ones.  Bars = measured values with margin."""
    sd, meta, eng = m8
    frames = synth_frames(4, 640, 640, seed=1)
    ref = RefYolo(sd, "detect", 64, "m", 1e-3, nc_quirk=False)
    y, raws = ref.forward(torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255)
    y = y.numpy()
    outs, idxs = non_max_suppression(y, 0.25, 0.7)
    x = torch.from_numpy(frames).cuda()
    dets, counts, anchor = eng.detect(x)
    dec = [op for op in eng.prog.ops if op.kind == 3][0]
    rawg = [eng.read_buffer(v.buf, 4, 640, 640).cpu().numpy().transpose(0, 3, 1, 2) for v in dec.src]
    st = _head_stats(rawg, [r.numpy() for r in raws])
    for k, (corr, rel) in st.items():
        print(f"level {k[0]} {k[1]}: corr {corr:.4f} rel rms {rel:.3f}")
    tot = com = 0
    be_all = []
    for b in range(4):
        n = int(counts[b])
        got = anchor[b, :n].cpu().numpy()
        cm, gi, oi = np.intersect1d(got, idxs[b], return_indices=True)
        tot += len(idxs[b]); com += len(cm)
        d = dets[b, :n].cpu().numpy()
        so = y[b, 4:, :].max(0)
        print(f"image {b}: kept {n} vs {len(idxs[b])}, common {len(cm)}, min oracle score at kept anchors {so[got].min() if n else 1:.3f}")
        if len(cm):
            be_all.append(np.abs(d[gi, :4] - outs[b][oi, :4]).max(1))
    be = np.concatenate(be_all)
    print(f"common-anchor fraction {com / tot:.3f}; box error on shared anchors: median {np.median(be):.2f} px, 95 % {np.quantile(be, 0.95):.2f} px, max {be.max():.2f} px")
    for k, (corr, rel) in st.items():
        # measured (round 3, = the CPU fake-quant study at 640 x 640 to three digits): box 0.989-0.995 / 0.10-0.15,
        # cls 0.908-0.954 / 0.31-0.44; shared kept anchors 63 %, box error median 1.3 px, 95 % 3.1 px, max 9 px
        assert corr >= (0.98 if k[1] == "box" else 0.88) and rel <= (0.19 if k[1] == "box" else 0.50), (k, corr, rel)
    assert com / tot >= 0.55
    assert np.median(be) < 2.5 and np.quantile(be, 0.95) < 6.0 and be.max() < 25.0


def test_fp8_fullsize_1280_batch16_properties(m8):
    """Config 5 at full size (1280x1280, batch 16, fp8): deterministic, batch independent, NMS invariants."""
    sd, meta, eng = m8
    frames = torch.from_numpy(synth_frames(16, 1280, 1280, seed=7)).cuda()
    d1, c1, a1 = eng.detect(frames, conf=0.35)                     # pipe.py:42,179: imgsz=1280, conf=0.35
    d2, c2, a2 = eng.detect(frames, conf=0.35)
    assert torch.equal(d1, d2) and torch.equal(c1, c2) and torch.equal(a1, a2)
    ds, cs, as_ = eng.detect(frames[3:6].contiguous(), conf=0.35)
    assert torch.equal(ds, d1[3:6]) and torch.equal(cs, c1[3:6]) and torch.equal(as_, a1[3:6])
    assert int(c1.sum()) > 0 and torch.isfinite(d1).all()
    for b in range(16):
        n = int(c1[b])
        s = d1[b, :n, 4]
        assert bool((s[:-1] >= s[1:]).all()) and bool((s > 0.35).all())
        assert bool((d1[b, n:] == 0).all())
        assert bool((d1[b, :n, 2] >= d1[b, :n, 0]).all()) and bool((d1[b, :n, 3] >= d1[b, :n, 1]).all())


def test_fp8_rank_classifier_on_the_reference_weights(rank_bundles, rank_valid):
    """The only accuracy evidence on TRAINED weights available here: the reference's rank classifier
    (`/root/reference/runs/rank_classifier/weights/best.pt`, fp32 top-1 63/67 = results.csv:21) in fp8 on the layered HIP
    path.  Calibration (ranges + channel means) on the FIRST 32 validation crops only; top-1 is reported on all 67 and on
    the 35 crops the calibration never saw.  Measured on the CPU fake-quant walk with the oracle's own calibration
    (tools/fp8_cpu_study.py classify --bias-corr): 63/67, without the bias correction 61/67.  Bars: >= 61/67 overall (the
    reference's own second checkpoint scores 61/67), arg-max equal to fp32's on >= 62 crops, and every layer code-for-code
    against the oracle on the engine's own input bytes."""
    from tests.fp8_util import oracle_from_engine, teacher_forced_layers
    sd, meta = rank_bundles["best"]
    pre = rank_valid["pre_u8"]
    labels = rank_valid["labels"]
    eng = engine_from_weights(sd, meta, "f8", 0, bgr_input=False, calib_frames=torch.from_numpy(pre[:32]))
    logits, probs = eng.classify(torch.from_numpy(pre).cuda())
    top1 = probs.argmax(1).cpu().numpy()
    ref32 = rank_valid["logits_best"].argmax(1)
    n_ok, n_held = int((top1 == labels).sum()), int((top1[32:] == labels[32:]).sum())
    dl = float(np.abs(logits.cpu().numpy() - rank_valid["logits_best"]).max())
    print(f"fp8 top-1 {n_ok}/67 (held-out crops {n_held}/35; fp32 {int((ref32 == labels).sum())}/67, held-out {int((ref32[32:] == labels[32:]).sum())}/35), "
          f"same arg-max as fp32 on {int((top1 == ref32).sum())}/67, max |dlogit| {dl:.2f}")
    oq = oracle_from_engine(eng, sd, meta)
    po, lo = oq.forward_u8(torch.from_numpy(pre).permute(0, 3, 1, 2).contiguous())
    to = lo.argmax(1).numpy()
    print(f"fake-quant oracle (same scales) top-1 {int((to == labels).sum())}/67, same arg-max as the HIP engine on {int((to == top1).sum())}/67, "
          f"max |dlogit| HIP vs oracle {float(np.abs(lo.numpy() - logits.cpu().numpy()).max()):.2f}")
    assert n_ok >= 61 and int((top1 == ref32).sum()) >= 62
    rows = teacher_forced_layers(eng, sd, meta, torch.from_numpy(pre[:8]))
    for name, n, nd, mx, nout in rows:
        # trained first layers are differences of large terms: up to 2 codes in 100 move (model.1), all inside the interval
        assert nd >= 0 and nd <= 3e-2 * n and nout == 0, f"{name}: {nd} of {n} codes differ (max {mx} steps), {nout} outside the accumulator interval"
    lg2, _ = eng.classify(torch.from_numpy(pre).cuda())              # deterministic
    assert torch.equal(lg2, logits)

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


def test_fp8_yolov8m_detections_vs_oracle(m8):
    """Documented fp8 bar on seeded yolov8m, 640x640 noise frames NOT in the calibration set.  e4m3 keeps 3
    mantissa bits: every stored tensor carries 3.6 % rms rounding noise, ~80 sequential layers of a RANDOM-INIT network
    (no trained-in margins; its "detections" are 3.5-sigma tail events of the class logits) end at ~20 % relative rms on
    the head logits.  So the bar is stated on the logits first - correlation >= 0.97, relative rms <= 0.30, least-squares
    slope within 0.9..1.1 of the oracle's (the gain correction of quant.py) - and on detections as measured: >= 30 % of the
    oracle's kept anchors kept, no kept anchor whose oracle score is below 0.05; boxes of matched anchors are DFL
    expectations over 16 noisy logits per side (up to 16 bins x stride 32): median < 12 px, 95 % < 40 px, every one < 160 px (boxes are 60-400 px wide);
    scores within 0.45 (measured: median 7 px, 95 % 19 px, max 22-100 px, 0.41)."""
    sd, meta, eng = m8
    frames = synth_frames(4, 640, 640, seed=1)
    ref = RefYolo(sd, "detect", 64, "m", 1e-3, nc_quirk=False)
    y, raws = ref.forward(torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255)
    y = y.numpy()
    outs, idxs = non_max_suppression(y, 0.25, 0.7)
    x = torch.from_numpy(frames).cuda()
    dets, counts, anchor = eng.detect(x)
    dec = [op for op in eng.prog.ops if op.kind == 3][0]
    for lvl, v in enumerate(dec.src):
        got = eng.read_buffer(v.buf, 4, 640, 640).cpu().numpy()
        want = raws[lvl].permute(0, 2, 3, 1).numpy()
        for nm, sl in (("box", slice(0, 64)), ("cls", slice(64, 128))):
            g, w = got[..., sl].ravel().astype(np.float64), want[..., sl].ravel().astype(np.float64)
            corr = np.corrcoef(g, w)[0, 1]
            rel = np.linalg.norm(g - w) / np.linalg.norm(w - w.mean())
            slope = ((g - g.mean()) * (w - w.mean())).sum() / ((w - w.mean()) ** 2).sum()
            print(f"level {lvl} {nm}: corr {corr:.4f} rel rms {rel:.3f} slope {slope:.3f}")
            assert corr >= 0.97 and rel <= 0.30 and 0.9 <= slope <= 1.1
    tot = com = 0
    for b in range(4):
        n = int(counts[b])
        got = anchor[b, :n].cpu().numpy()
        cm, gi, oi = np.intersect1d(got, idxs[b], return_indices=True)
        tot += len(idxs[b]); com += len(cm)
        d = dets[b, :n].cpu().numpy()
        so = y[b, 4:, :].max(0)
        print(f"image {b}: kept {n} vs {len(idxs[b])}, common {len(cm)}, min oracle score at kept anchors {so[got].min() if n else 1:.3f}")
        assert n == 0 or so[got].min() > 0.05
        if len(cm):
            be = np.abs(d[gi, :4] - outs[b][oi, :4]).max(1)
            assert np.median(be) < 12.0 and np.quantile(be, 0.95) < 40.0 and be.max() < 160.0, (np.median(be), np.quantile(be, 0.95), be.max())
            assert np.abs(d[gi, 4] - outs[b][oi, 4]).max() < 0.45
    print("common-anchor fraction", com / tot)
    assert com / tot >= 0.30


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

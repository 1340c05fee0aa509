"""Glue between the HIP fp8 engine's quantisation record (manual_yolo_amd.quant.QuantSpec: per buffer / per op index) and
the CPU fake-quant oracle (oracle/quant_ref.py: dicts keyed by the conv's state-dict prefix)."""
import numpy as np
import torch

from manual_yolo_amd.arch import OP_CONV, OP_STEM, build_program
from oracle.quant_ref import QT, RefYoloQuant


def plain_program(meta):
    """The layer program in reference order (the fp8 engine swaps the two views of some concats; the oracle does not)."""
    return build_program(meta["task"], meta["nc"], meta["scale"], meta.get("spec"), meta.get("nc_quirk", True), fuse_head=False)


def oracle_from_engine(eng, sd, meta):
    """RefYoloQuant executing exactly the quantised network the engine `eng` (dtype f8) was built with."""
    prog = plain_program(meta)
    q = eng.quant
    out_scale = {prog.ops[i].name: s for i, s in q.out_scale.items()}
    gains = {prog.ops[i].name: g for i, g in getattr(eng, "gains", {}).items()}
    in_mean = {}
    if q.buf_mean:
        for op in prog.ops:
            if op.kind == OP_CONV:
                in_mean[op.name] = torch.from_numpy(np.concatenate([q.buf_mean[v.buf][v.ch_off:v.ch_off + v.ch_cnt] for v in op.src]))
    return RefYoloQuant(sd, meta["task"], meta["nc"], meta["scale"], meta["bn_eps"], out_scale, gains,
                        nc_quirk=meta.get("nc_quirk", True), in_mean=in_mean)


def view_qt(eng, view, B, H, W, cache):
    """QT (stored e4m3 values NCHW + per-channel scales) of a channel-slice view of the engine's buffers."""
    if view.buf not in cache:
        cache[view.buf] = eng.read_buffer(view.buf, B, H, W, raw=True).cpu()
    q = cache[view.buf][..., view.ch_off:view.ch_off + view.ch_cnt].permute(0, 3, 1, 2).contiguous()
    if view.upsample:
        q = q.repeat_interleave(2, 2).repeat_interleave(2, 3)
    s = torch.from_numpy(eng.quant.buf_scale[view.buf][view.ch_off:view.ch_off + view.ch_cnt].copy())
    return QT(q, s)


E4M3_CODES = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()


def ulp_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|index(a) - index(b)| on the ordered list of finite e4m3 values (a, b hold exact e4m3 values)."""
    vals = torch.unique(E4M3_CODES[torch.isfinite(E4M3_CODES)])
    ia = torch.searchsorted(vals, a.flatten().contiguous())
    ib = torch.searchsorted(vals, b.flatten().contiguous())
    return (ia - ib).abs().view(a.shape)


ACC_EPS = 2.0 ** -13      # see teacher_forced_layers


def _code_bounds(ref, xin, res, op, fused_bn):
    """Lowest / highest e4m3 value the stored output may take when the accumulator (times the weight scale) deviates from
    the oracle's by at most ACC_EPS x sum |q_w q_x| x qs.  Same epilogue as RefYoloQuant._qconv; SiLU has its minimum
    (-0.27846 at x = -1.27846) inside the range it is not monotonic on, which is added where the interval straddles it."""
    import torch.nn.functional as F
    from oracle.quant_ref import e4m3
    qw, qs, bc = ref._qw[(op.name, tuple(xin.s.tolist()))]
    fw, fb = ref._fold(op.name, fused_bn)
    pad = op.ksize // 2
    acc = F.conv2d(xin.q, qw, None, stride=op.stride, padding=pad)
    mag = F.conv2d(xin.q.abs(), qw.abs(), None, stride=op.stride, padding=pad)
    g = float(ref.gains.get(op.name, 1.0))
    sc = (qs * g).view(1, -1, 1, 1)
    d = ACC_EPS * mag * sc
    lo, hi = acc * sc - d + (fb - bc).view(1, -1, 1, 1), acc * sc + d + (fb - bc).view(1, -1, 1, 1)
    if op.act:
        slo, shi = F.silu(lo), F.silu(hi)
        straddle = (lo < -1.27846) & (hi > -1.27846)
        vlo = torch.where(straddle, torch.full_like(lo, -0.278465), torch.minimum(slo, shi))
        vhi = torch.maximum(slo, shi)
    else:
        vlo, vhi = lo, hi
    if res is not None:
        vlo, vhi = vlo + res.real(), vhi + res.real()
    so = float(ref.out_scale[op.name])
    inv = float(torch.tensor(1.0 / so, dtype=torch.float32))
    return e4m3(vlo * inv * (1 - 1e-6) - 1e-30), e4m3(vhi * inv * (1 + 1e-6) + 1e-30)


def teacher_forced_layers(eng, sd, meta, frames_u8, report=print):
    """Every STEM / CONV op of the fp8 engine against the fake-quant oracle ON THE ENGINE'S OWN INPUT BYTES: the op's inputs
    (and residual) are read back from the engine's buffers, the oracle computes that one layer from them, and the stored
    outputs are compared code for code.  Returns [(name, n_elements, n_different, max_ulp or max_rel_err, n_outside)].

    What two correct executions may differ by is the accumulator: v_mfma_scale_f32_16x16x128_f8f6f4 does NOT reduce its 128
    products like an fp32 fma chain - tools/probe_fp8_mfma_acc.py (profiles/r03_probe_fp8_mfma_acc.log): next to one
    product of 2^8, products of 2^-4 ... 2^-9 lose up to 4e-4 of the sum (alignment to the largest term, truncated), and on
    random operands the result is up to 2^-16.6 (rms 2e-6) of sum|terms| away from an fp64 sum.  Where the pre-activation
    is a small difference of large terms that moves the stored value by a code or more.  So besides the count of differing
    codes the check is an interval: `n_outside` = stored codes outside [e4m3(f(acc - d)), e4m3(f(acc + d))] with
    d = 2^-13 x sum |q_w q_x| x qs (ACC_EPS) - must be 0."""
    prog = plain_program(meta)
    ref = oracle_from_engine(eng, sd, meta)
    x = frames_u8.to(eng.device)
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    eng.head_raw(x) if meta["task"] == "detect" else eng.classify(x)
    torch.cuda.synchronize()
    u8 = frames_u8.permute(0, 3, 1, 2).contiguous().cpu()
    out = []
    cache = {}
    for i, op in enumerate(prog.ops):
        if op.kind not in (OP_STEM, OP_CONV):
            continue
        fused_bn = prog.weights[op.weight].fused_bn
        out_f32 = prog.bufs[op.dst.buf][2] == 0
        if op.kind == OP_STEM:
            want = ref._stem(u8, op.name, op.ksize, op.stride)
        else:
            from oracle.quant_ref import qcat
            xin = qcat([view_qt(eng, v, B, H, W, cache) for v in op.src])
            res = view_qt(eng, op.res, B, H, W, cache) if op.res is not None else None
            want = ref._qconv(xin, op.name, op.ksize, op.stride, bool(op.act), res, fused_bn, out_f32)
            blo, bhi = (None, None) if out_f32 else _code_bounds(ref, xin, res, op, fused_bn)
        got = eng.read_buffer(op.dst.buf, B, H, W, raw=True).cpu()[..., op.dst.ch_off:op.dst.ch_off + op.dst.ch_cnt].permute(0, 3, 1, 2)
        if out_f32:
            err = float((got - want).abs().max() / max(1.0, float(want.abs().max())))
            out.append((op.name, got.numel(), -1, err, 0))
            report(f"op {i:3d} {op.name:24s} fp32 raw map: max rel err {err:.2e}")
        else:
            d = ulp_distance(got.contiguous(), want.q)
            nd, mx = int((d > 0).sum()), int(d.max())
            nout = 0 if op.kind == OP_STEM else int(((got < blo) | (got > bhi)).sum())
            out.append((op.name, got.numel(), nd, mx, nout))
            report(f"op {i:3d} {op.name:24s} k{op.ksize} {op.cin:4d}->{op.cout:4d}  codes different {nd:8d} of {got.numel():9d} ({nd / got.numel():.2e}), max {mx} ulp, "
                   f"outside the accumulator interval {nout}")
    return out

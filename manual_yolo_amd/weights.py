"""State dict -> device tensors in the layouts ``include/miyolo.h`` documents.

Host-side, one-time plumbing (torch on CPU, then one H2D copy per tensor); the reference does
the equivalent inside ``YOLO(path)`` + ``AutoBackend(fuse=True)`` (reference
``detect.py:20-21``): ``model.float().fuse()`` folds every BatchNorm into its conv
([3P] ``fuse_conv_and_bn``: ``W' = W * g/sqrt(var+eps)``, ``b' = beta - mean*g/sqrt(var+eps)``).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .arch import Program, WeightRecipe

K_ALIGN = {"f32": 32, "f16": 64, "f8": 128}      # must equal miyolo_k_align(); checked in engine.py
CHUNK_ELEMS = {"f32": 4, "f16": 8, "f8": 16}
TORCH_DTYPE = {"f32": torch.float32, "f16": torch.float16}


def fold_conv(sd: Dict[str, torch.Tensor], r: WeightRecipe, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(weight [cout,cin,k,k] fp32, bias [cout] fp32) of one conv with its BN folded in.  A prefix "a|b" names two convs
    over the same input run as one (arch.py, Detect's first convs): their folded weights concatenated along cout."""
    if "|" in r.prefix:
        parts = [fold_conv(sd, WeightRecipe(r.kind, p, r.fused_bn, r.seg_channels), eps) for p in r.prefix.split("|")]
        return torch.cat([w for w, _ in parts], 0), torch.cat([b for _, b in parts], 0)
    if r.fused_bn:
        w = sd[r.prefix + ".conv.weight"].float()
        g = sd[r.prefix + ".bn.weight"].float()
        beta = sd[r.prefix + ".bn.bias"].float()
        mean = sd[r.prefix + ".bn.running_mean"].float()
        var = sd[r.prefix + ".bn.running_var"].float()
        scale = g.div(torch.sqrt(eps + var))
        wf = w * scale.view(-1, 1, 1, 1)
        bf = beta - g.mul(mean).div(torch.sqrt(var + eps))
        return wf, bf
    return sd[r.prefix + ".weight"].float(), sd[r.prefix + ".bias"].float()


def pack_conv_weight(wf: torch.Tensor, dtype: str) -> torch.Tensor:
    """[cout,cin,kh,kw] -> [cout, kpad]: K = (ky, kx, cin) flattened, zero tail to K_ALIGN."""
    cout, cin, kh, kw = wf.shape
    flat = wf.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
    al = K_ALIGN[dtype]
    kpad = (flat.shape[1] + al - 1) // al * al
    out = torch.zeros((cout, kpad), dtype=torch.float32)
    out[:, :flat.shape[1]] = flat
    return out.to(TORCH_DTYPE[dtype]).contiguous()


def pack_stem_weight(wf: torch.Tensor, bgr_input: bool, dtype: str) -> torch.Tensor:
    """[cout,3,3,3] (cin = RGB as the model was trained) -> [cout][32] in the stem kernel's K'
    order (kernels_misc.h): k' = 8q + j; q<3: kernel row ky=q, byte j = kx*3 + c of its 9-byte
    run (j < 8); q=3: j<3 -> (ky=j, kx=2, c=2), rest zero.  With ``bgr_input`` the c axis is
    reversed so the kernel reads BGR frames directly (the reference flips BGR->RGB in its
    preprocess instead, [3P] ``im[..., ::-1]``)."""
    if bgr_input:
        wf = wf.flip(1)
    cout = wf.shape[0]
    run = wf.permute(0, 2, 3, 1).reshape(cout, 3, 9).float()      # [cout][ky][kx*3 + c]
    out = torch.zeros((cout, 32), dtype=torch.float32)
    for q in range(3):
        out[:, 8 * q:8 * q + 8] = run[:, q, :8]
    out[:, 24:27] = run[:, :, 8]
    return out.to(TORCH_DTYPE[dtype]).contiguous()


def _pad128(v: torch.Tensor) -> torch.Tensor:
    # padded: the conv kernels s_load 16 biases at a time (scalar loads are not range-checked).  A multiple of 128 covers
    # every channel tile that starts below cout; the 256 floats on top are slack for any tile shape (include/miyolo.h)
    out = torch.zeros(((v.numel() + 127) // 128 * 128 + 256,), dtype=torch.float32)
    out[:v.numel()] = v
    return out


def build_weight_tensors(prog: Program, sd: Dict[str, torch.Tensor], eps: float, dtype: str,
                         bgr_input: bool = True, quant=None):
    """One CPU tensor per ``prog.weights`` entry, ready for ``.to(device)``.  dtype "f8" (``quant`` = quant.QuantSpec):
    conv weights are e4m3 bytes with the input scales folded in, and the list gets two extra fp32 tensors per conv op
    (qscale, bias_init = bias / qscale); returns ``(tensors, extra)`` with ``extra[op_index] = (qscale_idx, bias_init_idx)``."""
    out: List[torch.Tensor] = []
    folded: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    ce = CHUNK_ELEMS[dtype]
    op_of_weight = {op.weight: (i, op) for i, op in enumerate(prog.ops) if op.weight >= 0}
    qscales: Dict[int, torch.Tensor] = {}
    bias_shift: Dict[int, torch.Tensor] = {}
    for wi, r in enumerate(prog.weights):
        if r.kind in ("conv", "stem", "bias"):
            if r.prefix not in folded:
                folded[r.prefix] = fold_conv(sd, r, eps)
            wf, bf = folded[r.prefix]
            if r.kind == "conv":
                for c in r.seg_channels:
                    if c % ce:
                        raise ValueError(f"{r.prefix}: input view of {c} channels is not a multiple of {ce} "
                                         f"({dtype} needs 16-byte channel chunks)")
                if dtype == "f8":
                    import numpy as np
                    from .quant import quantize_conv_weight
                    oi, op = op_of_weight[wi]
                    c0 = getattr(op, "swapped", 0)
                    if c0:                                    # engine swapped the two views: same order for the weight columns
                        wf = torch.cat([wf[:, c0:], wf[:, :c0]], 1)
                    s_in = np.concatenate([quant.buf_scale[v.buf][v.ch_off:v.ch_off + v.ch_cnt] for v in op.src])
                    q, qs = quantize_conv_weight(wf, s_in)
                    qscales[oi] = qs
                    out.append(q)
                    if quant.buf_mean:                        # bias correction (quant.py): E[rounding error of W] . E[x]
                        from .quant import weight_rounding_shift
                        m_in = np.concatenate([quant.buf_mean[v.buf][v.ch_off:v.ch_off + v.ch_cnt] for v in op.src])
                        bias_shift[oi] = weight_rounding_shift(wf, q, qs, s_in, m_in)
                else:
                    out.append(pack_conv_weight(wf, dtype))
            elif r.kind == "stem":
                out.append(pack_stem_weight(wf, bgr_input, "f16" if dtype == "f8" else dtype))
            else:
                out.append(_pad128(bf))
        elif r.kind == "linear":
            lw = sd[r.prefix + ".weight"].float()
            if dtype == "f8":                                 # the head pools the stored e4m3 values: their scale goes into the columns
                _, hop = op_of_weight[wi]
                v = hop.src[0]
                lw = lw * torch.from_numpy(quant.buf_scale[v.buf][v.ch_off:v.ch_off + v.ch_cnt]).view(1, -1)
            out.append(lw.contiguous())
        elif r.kind == "linear_bias":
            out.append(sd[r.prefix + ".bias"].float().contiguous())
        else:
            raise ValueError(r.kind)
    if dtype != "f8":
        return out
    extra = {}
    for oi, qs in qscales.items():
        op = prog.ops[oi]
        if oi in bias_shift:
            out[op.bias][:qs.numel()] -= bias_shift[oi]
        bias = out[op.bias][:qs.numel()]
        extra[oi] = (len(out), len(out) + 1)
        out.append(_pad128(qs))
        out.append(_pad128(bias / qs))
    return out, extra

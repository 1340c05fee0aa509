#!/usr/bin/env python3
"""How many bits does v_mfma_scale_f32_16x16x128_f8f6f4 keep inside its K = 128 reduction?  (gfx950, through the engine's
fp8 1x1 conv with an fp32 output - the MFMA's accumulator, times the weight scale, comes out unrounded.)

One input channel holds BIG = 2^8, the other 127 hold 2^-k; all weights are equal, so the exact result is 2^8 + 127 * 2^-k,
representable in fp32 for k <= 15 (an fp32 fma chain would return it exactly).  The table shows what comes out."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.engine import Engine  # noqa: E402
from manual_yolo_amd.quant import QuantSpec  # noqa: E402
from tests.gpu_util import conv_program  # noqa: E402


def run(x, w, impl=3):
    cin, cout = x.shape[-1], w.shape[0]
    prog = conv_program([(cin, 0, cin, 0)], cout, 1, 1, False, False, None, 0, True)
    q = QuantSpec({1: np.ones(cin, np.float32)}, {})
    eng = Engine(prog, {"t.weight": torch.from_numpy(w), "t.bias": torch.zeros(cout)}, 1e-3, "f8", 0, quant=q)
    eng.set_option("conv_impl", impl)
    B, H, W = x.shape[:3]
    eng.write_buffer(1, torch.from_numpy(x), H, W)
    eng.run_ops(0, 1, None, B, H, W)
    return eng.read_buffer(2, B, H, W).cpu().numpy()


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    print(f"K = {cin}: one channel 2^8, the rest 2^-k, weights 1")
    for k in range(0, 10):
        x = np.full((1, 4, 4, cin), 2.0 ** -k, np.float32)
        x[..., 0] = 256.0
        w = np.ones((16, cin, 1, 1), np.float32)
        y = run(x, w)[0, 0, 0, 0]
        exact = 256.0 + (cin - 1) * 2.0 ** -k
        print(f"k={k:2d}  exact {exact:.8f}  MFMA {y:.8f}  error {y - exact:+.3e}  ({(y - exact) / exact:+.2e} relative; small terms are 2^{-k - 8} of the big one)")
    # random data: error against an fp64 sum, relative to the sum of |terms|
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1, 8, 8, cin)) * 4).astype(np.float32)
    x = torch.from_numpy(x).clamp(-448, 448).to(torch.float8_e4m3fn).float().numpy()
    w = rng.standard_normal((16, cin, 1, 1)).astype(np.float32)
    y = run(x, w)
    from manual_yolo_amd.quant import dequant_fp8_bytes, quantize_conv_weight
    qw, qs = quantize_conv_weight(torch.from_numpy(w), np.ones(cin, np.float32))
    wq = (dequant_fp8_bytes(qw)[:, :cin].double() * qs.double().view(-1, 1)).numpy()
    ref = np.einsum("bhwc,nc->bhwn", x.astype(np.float64), wq)
    mag = np.einsum("bhwc,nc->bhwn", np.abs(x).astype(np.float64), np.abs(wq))
    r = np.abs(y - ref) / mag
    print(f"random operands: |MFMA - fp64| / sum|terms|: max {r.max():.2e} = 2^{np.log2(r.max()):.1f}, rms {np.sqrt((r ** 2).mean()):.2e}")


if __name__ == "__main__":
    main()

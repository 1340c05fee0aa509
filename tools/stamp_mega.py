"""Per-layer cycle breakdown of the one-launch classifier (csrc/cls_mega.h), from the stamps option dbg_op turns on.

usage (GPU box): python tools/stamp_mega.py [batch]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import ctypes as C

from manual_yolo_amd.ckpt import load_bundle  # noqa: E402
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    sd, meta = load_bundle("tests/golden/rank_best.safetensors")
    eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
    x = torch.randint(0, 255, (B, 64, 64, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        eng.classify(x)
    eng.set_option("dbg_op", 0)
    eng.classify(x)
    torch.cuda.synchronize()
    out = np.zeros(2 * 256 * 8 * 8, dtype=np.uint64)
    eng._check(eng.lib.miyolo_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p)), "stamps")
    st = out[: 8 * 32].reshape(8, 32).astype(np.int64)
    n, lds = eng.classify_launches(64, 64)
    print(f"launches {n}, LDS per image {lds} B, batch {B}")
    prog = eng.prog
    for img in (0, 3):
        t0 = st[img, 31]
        prev = t0
        print(f"image {img}: total {st[img, 30] - t0} cycles")
        for i in range(len(prog.ops) - 1):
            op = prog.ops[i]
            d = st[img, i] - prev
            prev = st[img, i]
            print(f"  op {i:2d} {op.name[-14:]:14s} k{op.ksize} s{op.stride} {op.cin:4d}->{op.cout:4d} @{64 // op.down_out:2d}  {d:8d}")
        print(f"  tail {st[img, 30] - prev}")
    fine = out[256: 256 + 4 * 32].reshape(32, 4).astype(np.int64)
    print("image 0, wave 0: cycles from the layer's start to [first item set up, its K loop done, all items done]; then barrier wait")
    for i in range(1, len(prog.ops) - 1):
        f = fine[i]
        print(f"  op {i:2d}  setup {f[1] - f[0]:6d}  kloop {f[2] - f[1]:6d}  rest {f[3] - f[2]:6d}  barrier {st[0, i] - f[3]:6d}  (prev barrier -> start {f[0] - st[0, i - 1]:6d})")


if __name__ == "__main__":
    main()

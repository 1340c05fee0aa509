"""CPU check of the conv kernel's index arithmetic through the lane-level model in emu_conv.py,
and of the LDS layout's bank behaviour (no GPU needed)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from manual_yolo_amd.weights import K_ALIGN, pack_conv_weight
from tests.emu_conv import emulate_conv, lds_slot


def _ref_conv(x_nhwc, w, b, stride, act, res=None):
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).double()
    y = F.conv2d(x, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=w.shape[2] // 2)
    if act:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1).reshape(-1, w.shape[0]).numpy()
    return y + res if res is not None else y


@pytest.mark.parametrize("dtype,cin,cout,k,s,H,W,B,wc,tc", [
    ("f16", 16, 20, 3, 1, 6, 6, 1, 1, 2),     # ct0=2: several taps inside one K step; cout tail
    ("f16", 24, 48, 3, 2, 8, 8, 2, 1, 3),     # stride 2, ct0=3 (taps straddle K steps), two images
    ("f16", 48, 96, 3, 1, 20, 20, 1, 2, 3),   # two M blocks x two waves along channels, M tail
    ("f32", 12, 16, 3, 1, 5, 7, 1, 1, 1),     # fp32 chunks of 4
    ("f32", 20, 40, 3, 2, 8, 8, 1, 2, 4),
    ("f16", 72, 64, 1, 1, 9, 9, 1, 1, 4),     # 1x1 with a K tail (72 -> 128)
])
def test_conv_single_source(dtype, cin, cout, k, s, H, W, B, wc, tc):
    rng = np.random.default_rng(0)
    ce = 8 if dtype == "f16" else 4
    ld, choff = cin + 2 * ce, ce                     # view inside a wider buffer
    x = rng.standard_normal((B, H, W, ld)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    wp = pack_conv_weight(torch.from_numpy(w), dtype).double().numpy()
    assert wp.shape[1] % K_ALIGN[dtype] == 0
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = rng.standard_normal((B * Ho * Wo, cout))
    out = emulate_conv([dict(arr=x.astype(np.float64), ch_off=choff, ch_cnt=cin, up=0)], wp, b.astype(np.float64),
                       ksize=k, stride=s, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, cout=cout, CE=ce, WC=wc, TC=tc,
                       act=True, res=res)
    wq = pack_conv_weight(torch.from_numpy(w), dtype)[:, :k * k * cin].float().reshape(cout, k, k, cin).permute(0, 3, 1, 2).numpy()
    ref = _ref_conv(x[..., choff:choff + cin], wq, b, s, True, res)
    assert np.abs(out - ref).max() < 1e-9


@pytest.mark.parametrize("dtype,c0,c1,up0", [("f16", 64, 24, 1), ("f32", 32, 12, 1), ("f16", 128, 64, 0)])
def test_conv_1x1_concat_upsample(dtype, c0, c1, up0):
    rng = np.random.default_rng(1)
    ce = 8 if dtype == "f16" else 4
    B, H, W, cout = 1, 8, 8, 32
    x0 = rng.standard_normal((B, H // 2, W // 2, c0) if up0 else (B, H, W, c0))
    x1 = rng.standard_normal((B, H, W, c1 + ce))
    w = (rng.standard_normal((cout, c0 + c1, 1, 1)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout)
    wp = pack_conv_weight(torch.from_numpy(w), dtype).double().numpy()
    out = emulate_conv([dict(arr=x0, ch_off=0, ch_cnt=c0, up=up0), dict(arr=x1, ch_off=ce, ch_cnt=c1, up=0)],
                       wp, b, ksize=1, stride=1, B=B, Hin=H, Win=W, Hout=H, Wout=W, cout=cout, CE=ce, WC=1, TC=2)
    x0u = np.repeat(np.repeat(x0, 2, 1), 2, 2) if up0 else x0
    xc = np.concatenate([x0u, x1[..., ce:]], -1)
    wq = wp[:, :c0 + c1].reshape(cout, c0 + c1, 1, 1)
    ref = _ref_conv(xc, wq, b, 1, False)
    assert np.abs(out - ref).max() < 1e-9


# ----------------------------------------------------------------------- LDS bank model
# MI355X_MICROARCH.md, section LDS: ds_read_b128 is serviced in four 16-lane groups
# {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; bank of
# byte address a is (a/4) mod 64; lanes of one group conflict when they touch the same bank
# row slot (16 B) with different addresses.  ds_write_b128: 8 groups of 8 contiguous lanes,
# bank (a/4) mod 32.
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _addr(row, chunk):
    r, c = lds_slot(row, chunk)
    return r * 128 + c * 16


@pytest.mark.parametrize("tile_base", [0, 16, 48, 112])
@pytest.mark.parametrize("kk", [0, 1])
def test_fragment_reads_are_conflict_free(tile_base, kk):
    for grp in READ_GROUPS:
        slots = [(_addr(tile_base + (l & 15), kk * 4 + (l >> 4)) // 16) % 16 for l in grp]
        assert len(set(slots)) == 16, slots


def test_staging_writes_are_conflict_free():
    for g in range(8):                      # one wave = 64 threads = rows r0..r0+7, 8 chunks each
        lanes = range(g * 8, g * 8 + 8)
        banks = [(_addr(l >> 3, l & 7) // 16) % 8 for l in lanes]   # 128 B = 8 slots of 16 B (32 banks)
        assert len(set(banks)) == 8


# ----------------------------------------------------------------------- conv_h2 / conv_h3 slab swizzle (round 3)
def _h2_swz(hx):
    return ((hx >> 1) & 3) << 1          # csrc/conv_h2.h h2_swz


def _fragment_conflicts(TW, NPX, swz, rowflip):
    """(worst multiplicity, mean extra LDS cycles per lane group) over every pixel fragment x tap column dx of a TW-wide tile of
    NPX pixels: lane (frow, fq) reads pixel p = 16 t + frow -> halo pixel (py, px + dx); slot = LDS row parity (= halo column
    parity: the slab pitch is even) x chunk position fq ^ swz(hx) ^ 4 (py & 1 if rowflip)."""
    worst, extra, n = 1, 0, 0
    for t in range((NPX + 15) // 16):
        for dx in (0, 1, 2):
            for grp in READ_GROUPS:
                slots = {}
                for l in grp:
                    frow, fq = l & 15, l >> 4
                    p = 16 * t + frow
                    if p >= NPX:
                        p = 0                      # lanes past the tile read pixel 0
                    py, px = divmod(p, TW)
                    hx = px + dx
                    pos = fq ^ swz(hx) ^ (4 * (py & 1) if rowflip else 0)
                    slots.setdefault(((hx & 1) * 8 + (pos & 7)) % 16, set()).add((py, hx, fq))
                m = max(len(v) for v in slots.values())
                worst = max(worst, m); extra += m - 1; n += 1
    return worst, extra / n


def test_h2_slab_swizzle_is_conflict_free_for_every_tap():
    """the halo-slab kernels' pixel-fragment reads start at halo column dx = 0, 1 or 2 (and wrap a halo
    row where the tile width is not a multiple of 16).  The ring kernels' swizzle (hx >> 1) & 7 pays 2- and 3-way conflicts
    there; h2_swz (+ the row-parity flip of the 20-wide geometry) pays none, on every geometry of conv_h2 and conv_h3."""
    old = lambda hx: (hx >> 1) & 7  # noqa: E731
    assert _fragment_conflicts(16, 256, old, False)[0] == 2 and _fragment_conflicts(20, 240, old, False)[0] == 3
    assert abs(_fragment_conflicts(16, 256, old, False)[1] - 2 / 3) < 1e-9        # x 4 reads of 10 per half: 21 % of the LDS cycles
    for TW, NPX, flip in ((16, 256, False), (40, 240, False), (20, 240, True),      # conv_h2: 16x16, 40x6, 20x12
                          (16, 128, False)):                                        # conv_h3: 16x8
        assert _fragment_conflicts(TW, NPX, _h2_swz, flip) == (1, 0.0), (TW, NPX)
    # conv_h3's 120-pixel tiles (40x3, 20x6): the last fragment is half empty and the idle lanes (which read pixel 0) meet
    # the live ones in one slot of one fragment in eight
    assert _fragment_conflicts(40, 120, _h2_swz, False) == (2, 0.125) and _fragment_conflicts(20, 120, _h2_swz, True) == (2, 0.125)


def _group_cycles(addr_of_lane):
    """Mean LDS cycles per ds_read_b128 lane group (1.0 = conflict-free): bank = 16-byte slot mod 16."""
    tot = 0
    for grp in READ_GROUPS:
        slots = {}
        for l in grp:
            a = addr_of_lane(l)
            slots.setdefault((a // 16) % 16, set()).add(a)
        tot += max(len(v) for v in slots.values())
    return tot / len(READ_GROUPS)


def test_bneck_lds_layout_bank_model():
    """conv_bneck.h (C = 48): weight-row pitch and halo-row pitch of the X tile.  Round 2's layout (odd weight pitch 57 chunks,
    dense 20-pixel halo rows) costs 2.0 / 1.74 cycles per lane group on the weight / conv-A pixel fragments - 43 % of all the
    kernel's read cycles, the counter said 45.6 % (profiles/r03_pmc_f16_b64.md); pitch 58 and 124 chunks cost 1.0 / 1.08."""
    CPT, NCH, NG = 6, 54, 14

    def weights(wrow):
        return sum(_group_cycles(lambda l, kg=kg: (l & 15) * wrow + (l >> 4) * 16 + kg * 64) for kg in range(NG)) / NG

    def conv_a_pixels(xpc):
        def ko(q):
            tap, co = divmod(q, CPT)
            return ((tap // 3) * xpc + (tap % 3) * CPT + co) * 16 if q < NCH else 0
        tot = 0
        for pt in range(21):
            for kg in range(NG):
                def addr(l, pt=pt, kg=kg):
                    p = pt * 16 + (l & 15)
                    p = p if p < 324 else 0
                    return ((p // 18) * xpc + (p % 18) * CPT) * 16 + ko(kg * 4 + (l >> 4))
                tot += _group_cycles(addr)
        return tot / (21 * NG)

    assert weights(912) == 2.0 and weights(928) == 1.0
    assert 1.7 < conv_a_pixels(120) < 1.8 and conv_a_pixels(124) < 1.1
    # the shares of the old layout: conv A reads 3 weight + 2.625 pixel fragments per k group, conv B 3 + 2 (its pixel reads: 1.04)
    old = 3 * 2.0 + 2.625 * conv_a_pixels(120) + 3 * 2.0 + 2 * 1.04
    assert 0.40 < 1 - (3 + 2.625 + 3 + 2) / old < 0.46

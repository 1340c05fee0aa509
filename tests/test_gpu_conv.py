"""-m gpu: the implicit-GEMM conv kernel (every tile shape, 3x3/1x1, stride 2, concat,
upsample, residual, channel tails) against torch conv2d on the CPU, through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import q, rel_err, run_conv

pytestmark = pytest.mark.gpu

TOL = {"f32": 2e-5, "f16": 2e-3}


def ref_conv(x, w, b, s, act, res=None):
    y = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), torch.from_numpy(b), stride=s,
                 padding=w.shape[2] // 2)
    if act:
        y = F.silu(y)
    y = y.permute(0, 2, 3, 1).numpy()
    return y + res if res is not None else y


IMPLS = [0, 1, 3, 7, 8]   # 8: halo-slab kernel for 3x3 stride-1 layers (conv_h2.h; other shapes fall through to the ring kernel); impls 2, 4, 5, 6 (conv_halo/halop/ws/dmh) are experiments, compiled only by `csrc/build.sh experiments` and then tested by MIYOLO_TEST_EXPERIMENTS=1; 7: narrow 3x3 layers on 16x16 tiles with LDS-resident weights (conv_t2d.h; other shapes fall through to the ring kernel); 6: half-size stages, two workgroups per CU (conv_dmh.h); 5: warp-specialised producer/consumer ring (conv_ws.h); 3: persistent LDS-DMA ring (conv_dmap.h); 4: 3 + persistent halo kernel (conv_halop.h); 0: register-staged (conv_igemm.h); 1: LDS-DMA ring (conv_dma.h); 2: 1 + halo kernel for 3x3 s1 (conv_halo.h)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,k,s,H,W,B", [
    (48, 96, 3, 2, 32, 32, 2),      # yolov8m L1-like (cin 48: taps straddle K steps)
    (48, 48, 3, 1, 20, 24, 1),      # L2 bottleneck
    (96, 96, 3, 1, 16, 16, 2),
    (192, 64, 3, 1, 12, 12, 1),     # head box branch
    (288, 288, 3, 1, 10, 10, 1),    # L8 bottleneck (288 = 4.5 K steps per tap)
    (384, 576, 3, 2, 8, 8, 1),      # L7
    (96, 96, 1, 1, 16, 16, 1),      # C2f cv1
    (576, 192, 1, 1, 8, 8, 2),
    (64, 64, 1, 1, 12, 12, 1),
    (16, 16, 3, 1, 16, 16, 3),      # classifier sizes
    (16, 32, 3, 2, 32, 32, 2),
    (256, 1280, 1, 1, 2, 2, 5),
])
def test_conv_layers(dtype, cin, cout, k, s, H, W, B, impl):
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    res = q(rng.standard_normal((B, H // s, W // s, cout)).astype(np.float32), dtype)
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, s, True, res, B, H, W, impl=impl)
    ref = ref_conv(x, w, b, s, True, res)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < TOL[dtype], rel_err(y, ref)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("wc,tc", [(2, 4), (2, 3), (1, 4), (1, 3), (1, 2), (1, 1)])
@pytest.mark.parametrize("k", [1, 3])
def test_every_tile_shape(dtype, wc, tc, k, impl):
    rng = np.random.default_rng(7)
    cin, cout, H, W, B = 96, 112, 20, 28, 2           # cout 112: channel tail for every tile width; M = 1120: M tail
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, 1, True, None, B, H, W, force=(wc, tc), impl=impl)
    assert rel_err(y, ref_conv(x, w, b, 1, True)) < TOL[dtype]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_channel_slices_and_untouched_channels(dtype, impl):
    """Reads a slice of a wider buffer, writes a slice of a wider buffer (C2f layout)."""
    rng = np.random.default_rng(3)
    B, H, W, ld, off, cin, cout = 2, 12, 12, 192, 48, 48, 48
    x = q(rng.standard_normal((B, H, W, ld)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / 20).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 3, 1, True, None, B, H, W, dst_ld=192, dst_off=96, impl=impl)
    ref = ref_conv(x[..., off:off + cin], w, b, 1, True)
    assert rel_err(y[..., 96:144], ref) < TOL[dtype]
    assert np.all(y[..., :96] == 7.0) and np.all(y[..., 144:] == 7.0)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("c0,c1,up", [(576, 384, 1), (192, 384, 0), (64, 32, 1)])
def test_concat_upsample_1x1(dtype, c0, c1, up, impl):
    """C2f.cv1 over cat(upsample(a), b) without materialising either (FPN layers 11-12, 14-15)."""
    rng = np.random.default_rng(11)
    B, H, W, cout = 2, 8, 12, 128
    x0 = q(rng.standard_normal((B, H // 2, W // 2, c0) if up else (B, H, W, c0)).astype(np.float32), dtype)
    x1 = q(rng.standard_normal((B, H, W, c1)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, c0 + c1, 1, 1)) / np.sqrt(c0 + c1)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x0, x1], w, b, [(c0, 0, c0, up), (c1, 0, c1, 0)], 1, 1, True, None, B, H, W, impl=impl)
    x0u = np.repeat(np.repeat(x0, 2, 1), 2, 2) if up else x0
    ref = ref_conv(np.concatenate([x0u, x1], -1), w, b, 1, True)
    assert rel_err(y, ref) < TOL[dtype]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cout", [13, 64, 80])
def test_head_final_conv_f32_out(dtype, cout, impl):
    """Detect's last 1x1 (bias, no activation) writes fp32 into a slice of the raw head map;
    cout=13 exercises the scalar-store path (unaligned slice)."""
    rng = np.random.default_rng(5)
    B, H, W, cin = 1, 10, 10, 64
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 1, 1)) / 8).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 1, 1, False, None, B, H, W, dst_ld=64 + cout, dst_off=64,
                 out_f32=True, impl=impl)
    assert rel_err(y[..., 64:], ref_conv(x, w, b, 1, False)) < TOL[dtype]
    assert np.all(y[..., :64] == 7.0)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,B,force", [
    (96, 96, 20, 20, 3, None),        # M = 1200: tiles straddle frames, M tail
    (48, 48, 12, 160, 1, None),       # widest yolov8m@640 map (W=160), cin 48 = 1.5 chunks
    (16, 32, 8, 320, 1, (1, 2)),      # W=320 (1280 input): XI = 8, the largest halo that fits
    (16, 16, 4, 382, 1, (1, 1)),      # widest eligible row (W=382)
    (192, 192, 18, 22, 2, (2, 4)),    # rect inputs such as 544 wide
    (32, 32, 6, 62, 2, (2, 3)),       # R + 1 = 383..384: zero row is the last LDS row of the halo buffer
    (64, 64, 40, 40, 1, (1, 4)),
    (288, 96, 6, 10, 5, (2, 3)),
    (32, 16, 4, 4, 70, (1, 1)),       # many tiny frames per tile
    (96, 96, 80, 80, 2, None),        # the 80x80 level: halo buffer at its 448-row / 160 KiB limit
    (288, 288, 20, 20, 4, None),      # 4.5 chunks: trailing 32-channel chunk
    (80, 64, 24, 94, 1, (1, 4)),      # widest map the persistent halo kernel takes, 1.25 chunks
])
@pytest.mark.parametrize("impl", [2, 4, 8])
def test_halo_kernel_shapes(dtype, cin, cout, H, W, B, force, impl):
    """conv_halo.h / conv_halop.h: shifted LDS windows, zero-row masking at frame borders, halo pieces
    (shapes the persistent variant does not take - width > 95 - fall back to the ring kernel)."""
    rng = np.random.default_rng(cin + cout + W)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    res = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype)
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, res, B, H, W, force=force, impl=impl)
    assert rel_err(y, ref_conv(x, w, b, 1, True, res)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,k,s,H,W,B", [(192, 192, 3, 1, 40, 40, 2), (384, 384, 1, 1, 24, 24, 1),
                                               (96, 192, 3, 2, 32, 32, 2), (64, 200, 3, 1, 12, 20, 3)])
def test_wide_tile_256x192(dtype, cin, cout, k, s, H, W, B):
    """conv_dmap.h <WC=2,TC=6>: 256 px x 192 ch tile on a 2-slot ring (forced; also with a channel tail)."""
    rng = np.random.default_rng(cin + cout)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    res = q(rng.standard_normal((B, H // s, W // s, cout)).astype(np.float32), dtype)
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, s, True, res, B, H, W, force=(2, 6), impl=3)
    assert rel_err(y, ref_conv(x, w, b, s, True, res)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("k,s,cin,cout,H,W,B,res", [(3, 1, 96, 64, 40, 40, 2, True), (1, 1, 192, 80, 24, 20, 3, False), (3, 2, 48, 64, 36, 28, 2, False)])
def test_ws_2x2_consumer_grid(dtype, k, s, cin, cout, H, W, B, res):
    """conv_ws.h's 2 x 2 consumer grid with 32-channel wave tiles (the {2,2} shape only that kernel has)."""
    rng = np.random.default_rng(11)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H // s, W // s, cout)).astype(np.float32), dtype) if res else None
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, s, True, r, B, H, W, force=(2, 2), impl=5)
    assert rel_err(y, ref_conv(x, w, b, s, True, r)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,k,H,W,B", [(96, 96, 3, 40, 40, 48), (192, 64, 1, 40, 40, 64)])
def test_one_to_two_tiles_per_cu(dtype, cin, cout, k, H, W, B):
    """Launches with more than one and at most two 256-pixel tiles per CU (300 / 400 tiles): the default engine's
    balanced persistent grid (150 / 200 workgroups, two tiles each) and, with the `dmh_auto` option, the two-workgroup
    kernel conv_dmh.h; residual, every tail."""
    rng = np.random.default_rng(5)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    res = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype)
    y3 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, 1, True, res, B, H, W, impl=3, opts={"h2": 0})
    y1 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, 1, True, res, B, H, W, impl=1)
    assert rel_err(y3, ref_conv(x, w, b, 1, True, res)) < TOL[dtype]
    if dtype == "f32":
        assert np.array_equal(y3, y1)   # same accumulation order in the ring kernels: bit-identical in the exact mode
    yd = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, 1, True, res, B, H, W, impl=3)      # default engine (halo-slab kernel for the 3x3)
    assert rel_err(yd, ref_conv(x, w, b, 1, True, res)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,B,res", [(48, 48, 32, 32, 3, True), (48, 48, 16, 16, 2, False), (16, 16, 32, 48, 5, True),
                                                (64, 48, 16, 48, 1, False), (32, 64, 48, 16, 2, True), (8, 24, 16, 16, 70, False),
                                                (48, 48, 160, 160, 2, True)])
def test_t2d_kernel_shapes(dtype, cin, cout, H, W, B, res):
    """conv_t2d.h (conv_impl 7, and the default engine for these shapes): 16x16 tiles, image borders on every side of a
    tile, one-tile images, more tiles than workgroups (B = 70), channel tails (cout 24), full 160x160 maps; in f32 the
    result must be bit-identical to the first ring kernel (same accumulation order)."""
    rng = np.random.default_rng(cin * 131 + cout + H)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype) if res else None
    y7 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=7)
    assert rel_err(y7, ref_conv(x, w, b, 1, True, r)) < TOL[dtype]
    y1 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=1)
    if dtype == "f32":
        assert np.array_equal(y7, y1)
    y3 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=3, opts={"h2": 0})   # ring-kernel engine: same kernel
    assert np.array_equal(y3, y7)
    yd = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=3)      # default engine (t2d where it fits, else the halo-slab kernel)
    assert rel_err(yd, ref_conv(x, w, b, 1, True, r)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_t2d_channel_slices(dtype):
    """input = channels 48..95 of a 96-channel buffer, output into channels 96..143 of a 192-channel buffer."""
    rng = np.random.default_rng(77)
    B, H, W, ld, off, cin, cout = 2, 32, 32, 96, 48, 48, 48
    x = q(rng.standard_normal((B, H, W, ld)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 3, 1, True, None, B, H, W, dst_ld=192, dst_off=96, impl=7)
    ref = ref_conv(x[..., off:off + cin], w, b, 1, True)
    assert rel_err(y[..., 96:96 + cout], ref) < TOL[dtype]
    assert np.all(y[..., :96] == 7.0) and np.all(y[..., 96 + cout:] == 7.0)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,B,res", [
    (96, 96, 80, 80, 2, True),        # 16x16 tiles, pitch 24 (odd dy flips the row halves), 1.5 chunks
    (192, 192, 40, 40, 2, False),     # 40x6 tiles (pitch 48), 240 of 256 pixels used, two channel tiles
    (288, 288, 20, 20, 3, True),      # 20x12 tiles, 4.5 chunks, three channel tiles
    (48, 48, 160, 160, 1, True),      # TC = 3: one pair + an unpaired channel tile
    (192, 64, 80, 80, 1, False),      # head box branch, TC = 4
    (384, 192, 40, 40, 1, False),
    (576, 64, 20, 20, 2, False),
    (64, 64, 20, 20, 2, False),       # one chunk exactly
    (96, 72, 68, 80, 1, True),        # letterboxed 544x640 frame: 68-wide map; channel tail (72 of 96)
    (96, 96, 18, 20, 2, False),       # partial tile rows
    (16, 16, 16, 16, 3, True),        # classifier sizes: a quarter chunk
    (32, 200, 12, 20, 2, False),      # cout 200 = 2 x 96 + 8
    (24, 40, 10, 6, 5, True),         # tiny maps, several tiles idle rows/columns
])
def test_h2_kernel_shapes(dtype, cin, cout, H, W, B, res):
    """conv_h2.h (conv_impl 8, and the default engine where its tiles cover the map): slab pitch classes, partial tiles
    on every side, channel chunk tails, paired / unpaired channel tiles, residual."""
    rng = np.random.default_rng(cin * 7 + cout + H)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype) if res else None
    y = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=8)
    assert rel_err(y, ref_conv(x, w, b, 1, True, r)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_h2_channel_slices(dtype):
    """input = channels 48..95 of a 96-channel buffer, output into channels 96..143 of a 192-channel buffer."""
    rng = np.random.default_rng(78)
    B, H, W, ld, off, cin, cout = 2, 40, 40, 96, 48, 48, 48
    x = q(rng.standard_normal((B, H, W, ld)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 3, 1, True, None, B, H, W, dst_ld=192, dst_off=96, impl=8)
    ref = ref_conv(x[..., off:off + cin], w, b, 1, True)
    assert rel_err(y[..., 96:96 + cout], ref) < TOL[dtype]
    assert np.all(y[..., :96] == 7.0) and np.all(y[..., 96 + cout:] == 7.0)


@pytest.mark.parametrize("cin,cout,H,W,B,res", [(96, 96, 80, 80, 22, True), (192, 192, 40, 40, 40, False), (64, 64, 80, 80, 24, True)])
def test_h2_persistent_variant(cin, cout, H, W, B, res):
    """conv_h2.h's persistent form (option h2_warm = 1, off by default): more tiles than the 512 workgroup slots, the next
    tile's slab issued inside the epilogue, with and without a residual operand; f16, vs torch and vs the default form."""
    dtype = "f16"
    rng = np.random.default_rng(cin + cout + B)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype) if res else None
    yp = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=8, opts={"h2_warm": 1})
    y0 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=8)
    assert rel_err(yp, ref_conv(x, w, b, 1, True, r)) < TOL[dtype]
    assert np.array_equal(yp, y0)          # same arithmetic in both forms


@pytest.mark.parametrize("cin,cout,H,W,B,res", [
    (96, 96, 80, 80, 2, True),        # 16x8 tiles (slab pitch 18), 1.5 chunks
    (192, 192, 40, 40, 2, False),     # 40x3 tiles (pitch 42), 120 of 128 pixels used, two channel tiles
    (288, 288, 20, 20, 3, True),      # 20x6 tiles (pitch 22), 4.5 chunks, three channel tiles
    (48, 48, 160, 160, 1, True),      # TC = 3: one pair + an unpaired channel tile
    (192, 64, 80, 80, 1, False),      # head box branch, TC = 4
    (384, 192, 40, 40, 1, False),
    (64, 64, 20, 20, 2, False),       # one chunk exactly
    (96, 72, 68, 80, 1, True),        # letterboxed 544x640 frame: 68-wide map; channel tail (72 of 96)
    (96, 96, 18, 20, 2, False),       # partial tile rows
    (16, 16, 16, 16, 3, True),        # a quarter chunk
    (32, 200, 12, 20, 2, False),      # cout 200 = 2 x 96 + 8
    (24, 40, 10, 6, 5, True),         # tiny maps, tiles with idle rows / columns
    (96, 96, 80, 80, 13, True),       # more tiles than the 768 workgroup slots of the chip
])
def test_h3_kernel_shapes_and_bit_identity_with_h2(cin, cout, H, W, B, res):
    """conv_h3.h (conv_impl 9: conv_h2's halo-slab kernel re-cut for three workgroups per CU - 128-pixel tiles, linearly
    stored slab): against torch, and BIT-IDENTICAL to conv_h2 (same K order, same epilogue arithmetic) on every shape class:
    the three tile geometries, partial tiles on every side, channel chunk tails, paired / unpaired channel tiles, residual."""
    dtype = "f16"
    rng = np.random.default_rng(cin * 7 + cout + H)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype) if res else None
    y3 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=9)
    y2 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=8)
    assert rel_err(y3, ref_conv(x, w, b, 1, True, r)) < TOL[dtype]
    assert np.array_equal(y3, y2)


def test_h3_channel_slices():
    """input = channels 48..95 of a 96-channel buffer, output into channels 96..143 of a 192-channel buffer."""
    dtype = "f16"
    rng = np.random.default_rng(79)
    B, H, W, ld, off, cin, cout = 2, 40, 40, 96, 48, 48, 48
    x = q(rng.standard_normal((B, H, W, ld)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 3, 1, True, None, B, H, W, dst_ld=192, dst_off=96, impl=9)
    ref = ref_conv(x[..., off:off + cin], w, b, 1, True)
    assert rel_err(y[..., 96:96 + cout], ref) < TOL[dtype]
    assert np.all(y[..., :96] == 7.0) and np.all(y[..., 96 + cout:] == 7.0)


# ---- conv_pw.h: the streaming 1x1 kernel (conv_impl 3, option pw = 1; the default, pw = 0, is the ring kernel conv_dmap.h)
@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,H,W,B", [
    (96, 96, 16, 16, 2),        # K = 1.5 steps (half step + void steps), one channel tile of 96
    (192, 96, 20, 12, 3),       # model.2.cv2's shape class; pixel count not a multiple of anything
    (192, 192, 16, 16, 2),      # BN = 192
    (576, 192, 24, 24, 2),      # 9 K steps + 3 void ones
    (384, 384, 8, 8, 5),        # two channel tiles sharing a pixel range
    (1152, 576, 6, 6, 3),       # three channel tiles, 18 steps
    (576, 288, 6, 6, 3),        # 288 = 3 x 96 (SPPF cv1)
    (64, 192, 10, 10, 1),       # one K step only; fewer pixels than one tile per workgroup
    (128, 96, 96, 96, 9),       # 82 944 pixels: several tiles per workgroup, partial last tile in every range
])
def test_pw_kernel_shapes_and_bit_identity_with_ring_kernel(cin, cout, H, W, B):
    """Against torch, and BIT-IDENTICAL to conv_dmap (same MFMA, same K order, same epilogue arithmetic)."""
    dtype = "f16"
    rng = np.random.default_rng(cin + 3 * cout + H)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    for act in (True, False):
        y1 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 1, 1, act, None, B, H, W, impl=3, opts={"pw": 1})
        y0 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 1, 1, act, None, B, H, W, impl=3, opts={"pw": 0})
        assert rel_err(y1, ref_conv(x, w, b, 1, act)) < TOL[dtype]
        assert np.array_equal(y1, y0)


@pytest.mark.gpu
@pytest.mark.parametrize("c0,c1,up", [(576, 384, 1), (192, 384, 0), (384, 192, 1), (64, 32, 0)])
def test_pw_concat_upsample(c0, c1, up):
    """cat(upsample(a), b) and cat(a, b) as address arithmetic (FPN layers 12, 15, 18, 21), bit-identical to the ring kernel."""
    dtype = "f16"
    rng = np.random.default_rng(13 + c0)
    B, H, W, cout = 3, 12, 20, 192
    x0 = q(rng.standard_normal((B, H // 2, W // 2, c0) if up else (B, H, W, c0)).astype(np.float32), dtype)
    x1 = q(rng.standard_normal((B, H, W, c1)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, c0 + c1, 1, 1)) / np.sqrt(c0 + c1)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    srcs = [(c0, 0, c0, up), (c1, 0, c1, 0)]
    y1 = run_conv(dtype, [x0, x1], w, b, srcs, 1, 1, True, None, B, H, W, impl=3, opts={"pw": 1})
    y0 = run_conv(dtype, [x0, x1], w, b, srcs, 1, 1, True, None, B, H, W, impl=3, opts={"pw": 0})
    x0u = np.repeat(np.repeat(x0, 2, 1), 2, 2) if up else x0
    assert rel_err(y1, ref_conv(np.concatenate([x0u, x1], -1), w, b, 1, True)) < TOL[dtype]
    assert np.array_equal(y1, y0)


@pytest.mark.gpu
def test_pw_channel_slices_and_untouched_channels():
    """input = channels 96..287 of a 384-channel buffer (a C2f's branches), output into channels 192..383 of a 576-channel one."""
    dtype = "f16"
    rng = np.random.default_rng(83)
    B, H, W, ld, off, cin, cout = 2, 20, 20, 384, 96, 192, 192
    x = q(rng.standard_normal((B, H, W, ld)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    y = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 1, 1, True, None, B, H, W, dst_ld=576, dst_off=192, impl=3, opts={"pw": 1})
    y0 = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 1, 1, True, None, B, H, W, dst_ld=576, dst_off=192, impl=3, opts={"pw": 0})
    assert rel_err(y[..., 192:384], ref_conv(x[..., off:off + cin], w, b, 1, True)) < TOL[dtype]
    assert np.all(y[..., :192] == 7.0) and np.all(y[..., 384:] == 7.0)
    assert np.array_equal(y, y0)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,k,s,H,W,B,res,force", [
    (96, 96, 1, 1, 16, 16, 2, False, None),       # BN = 96 as 2 waves x 3 tiles: one pair + one unpaired tile per wave
    (192, 192, 1, 1, 16, 16, 2, False, None),     # 256 x 192 tile: three pairs per wave
    (64, 128, 3, 2, 32, 32, 2, False, (2, 4)),    # stride-2 3x3, two pairs per wave
    (48, 96, 3, 2, 32, 32, 2, False, None),       # model.1's shape class
    (64, 64, 3, 1, 12, 12, 3, True, (1, 4)),      # 8 x 1 wave layout (32-pixel wave tiles), residual read 16 bytes at a time
    (32, 32, 1, 1, 10, 10, 2, True, (1, 2)),      # one pair, residual
    (64, 80, 1, 1, 10, 10, 2, False, (1, 3)),     # cout = 80: channel tail inside a pair (channels 80..95 do not exist)
    (64, 16, 1, 1, 10, 10, 2, False, (1, 1)),     # one channel tile: nothing to pair
])
def test_dmap_paired_stores_change_no_bit(cin, cout, k, s, H, W, B, res, force):
    """conv_dmap.h option pair8 (default 1): weight rows dealt so that a lane holds 8 consecutive channels over a channel-tile pair,
    one 16-byte store per pair - against torch, and bit-identical to the 8-byte-store form."""
    dtype = "f16"
    rng = np.random.default_rng(cin * 5 + cout + k)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H // s, W // s, cout)).astype(np.float32), dtype) if res else None
    kw = dict(force=force, impl=3, opts={"h2": 0, "t2d": 0, "pair8": 1})
    y1 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, s, True, r, B, H, W, **kw)
    kw["opts"] = {"h2": 0, "t2d": 0, "pair8": 0}
    y0 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], k, s, True, r, B, H, W, **kw)
    assert rel_err(y1, ref_conv(x, w, b, s, True, r)) < TOL[dtype]
    assert np.array_equal(y1, y0)


@pytest.mark.gpu
def test_dmap_paired_stores_channel_slices():
    """8-aligned slices use the paired stores and leave the neighbours alone; a 4-aligned destination falls back to 8-byte stores."""
    dtype = "f16"
    rng = np.random.default_rng(97)
    B, H, W, ld, off, cin, cout = 2, 12, 12, 192, 96, 96, 96
    x = q(rng.standard_normal((B, H, W, ld)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = ref_conv(x[..., off:off + cin], w, b, 1, True)
    for dst_off in (96, 100):
        y = run_conv(dtype, [x], w, b, [(ld, off, cin, 0)], 1, 1, True, None, B, H, W, dst_ld=288, dst_off=dst_off, impl=3)
        assert rel_err(y[..., dst_off:dst_off + cout], ref) < TOL[dtype]
        assert np.all(y[..., :dst_off] == 7.0) and np.all(y[..., dst_off + cout:] == 7.0)


# ---- conv_h4.h (conv_impl 10 / option h4): one workgroup per CU, 480 / 512-pixel tiles, 128-pixel wave tiles - round-3 experiment
@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,H,W,B,res", [
    (96, 96, 80, 80, 2, False),       # 40 x 12 tiles: 2 x 7 per image, last tile row partial (80 = 6 x 12 + 8)
    (96, 96, 80, 80, 2, True),        # residual
    (192, 192, 80, 80, 1, False),     # three chunks, two channel tiles
    (192, 256, 40, 40, 2, False),     # 40-wide map, 256 = 2 x 96 + 64: channel tail tile
    (64, 64, 32, 48, 3, True),        # 16 x 32 tiles, TC = 4, residual
    (96, 72, 36, 20, 2, False),       # tiles wider than the map; channel tail (72 of 96)
    (32, 96, 160, 160, 1, False),     # half chunk; 16 x 32 tiles on a 160-wide map
])
def test_h4_kernel_shapes_and_bit_identity_with_h2(cin, cout, H, W, B, res):
    dtype = "f16"
    rng = np.random.default_rng(cin * 11 + cout + H)
    x = q(rng.standard_normal((B, H, W, cin)).astype(np.float32), dtype)
    w = q((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), dtype)
    b = rng.standard_normal(cout).astype(np.float32)
    r = q(rng.standard_normal((B, H, W, cout)).astype(np.float32), dtype) if res else None
    y4 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=10)
    y2 = run_conv(dtype, [x], w, b, [(cin, 0, cin, 0)], 3, 1, True, r, B, H, W, impl=8)
    assert rel_err(y4, ref_conv(x, w, b, 1, True, r)) < TOL[dtype]
    assert np.array_equal(y4, y2)

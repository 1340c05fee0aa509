"""Lane-level numpy model of ``conv_igemm_kernel`` (manual_yolo_amd/csrc/conv_igemm.h).

There is no GPU in the build container, so the kernel's index arithmetic is mirrored here
statement by statement - block renumbering, staging rows/chunks, the flattened-K walker,
the swizzled LDS image, the MFMA fragment maps (cdna_hip_programming.md section 3) and the
epilogue map - and run against ``torch.nn.functional.conv2d`` on the CPU.  A mistake in any
of those shows up here before a GPU minute is spent.  (It cannot catch HIP-specific errors.)
"""
import numpy as np

TP = 4


def lds_slot(row, chunk):
    return row, chunk ^ ((row >> 1) & 7)


def emulate_conv(srcs, w_packed, bias, *, ksize, stride, B, Hin, Win, Hout, Wout, cout, CE, WC, TC,
                 act=False, res=None, nblk_check=True):
    """srcs: list of dict(arr=[B,h,w,ld] float array, ch_off, ch_cnt, up).  Returns [M, cout]."""
    WP = 4 // WC
    BM, BN = WP * TP * 16, WC * TC * 16
    XR, WR = BM // 32, (BN + 31) // 32
    M = B * Hout * Wout
    kpad = w_packed.shape[1]
    BK = 8 * CE
    nk = kpad // BK
    NB = (cout + BN - 1) // BN
    MB = (M + BM - 1) // BM
    nblk = MB * NB
    out = np.full((M, cout), np.nan, dtype=np.float64)
    seen = set()
    flat = [s["arr"].reshape(-1) for s in srcs]
    nsrc = len(srcs)
    s0 = srcs[0]
    s1 = srcs[1] if nsrc > 1 else srcs[0]
    for blk in range(nblk):
        q, r = nblk >> 3, nblk & 7
        xcd, slot = blk & 7, blk >> 3
        bid = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + slot
        assert 0 <= bid < nblk and bid not in seen
        seen.add(bid)
        mb, nb = bid // NB, bid % NB
        m0, n0 = mb * BM, nb * BN
        Xs = np.zeros((2, BM, 8, CE)); Ws = np.zeros((2, BN, 8, CE))
        # per-thread staging state
        tids = np.arange(256)
        c8, r0 = tids & 7, tids >> 3
        xoff0 = np.zeros((256, XR), np.int64); xoff1 = np.zeros((256, XR), np.int64)
        xmask = np.zeros((256, XR), np.int64)
        for i in range(XR):
            m = m0 + r0 + 32 * i
            vm = m < M
            mm = np.where(vm, m, 0)
            b = mm // (Hout * Wout); rem = mm - b * (Hout * Wout)
            ho = rem // Wout; wo = rem - ho * Wout
            if ksize == 3:
                hi0, wi0 = ho * stride - 1, wo * stride - 1
                xoff0[:, i] = ((b * s0["arr"].shape[1] + hi0) * s0["arr"].shape[2] + wi0) * s0["arr"].shape[3] + s0["ch_off"]
                msk = np.zeros(256, np.int64)
                for t in range(9):
                    hi, wi = hi0 + t // 3, wi0 + t % 3
                    ok = vm & (hi >= 0) & (hi < Hin) & (wi >= 0) & (wi < Win)
                    msk |= ok.astype(np.int64) << t
                xmask[:, i] = msk
            else:
                h0 = (ho >> 1) if s0["up"] else ho; w0 = (wo >> 1) if s0["up"] else wo
                xoff0[:, i] = ((b * s0["arr"].shape[1] + h0) * s0["arr"].shape[2] + w0) * s0["arr"].shape[3] + s0["ch_off"]
                h1 = (ho >> 1) if s1["up"] else ho; w1 = (wo >> 1) if s1["up"] else wo
                xoff1[:, i] = ((b * s1["arr"].shape[1] + h1) * s1["arr"].shape[2] + w1) * s1["arr"].shape[3] + s1["ch_off"]
                xmask[:, i] = vm.astype(np.int64)
        ct0 = s0["ch_cnt"] // CE
        tap = np.zeros(256, np.int64); coff = c8.copy()
        if ksize == 3:
            tap = c8 // ct0
            coff = c8 - tap * ct0

        def stage_load(ks):
            nonlocal tap, coff
            xreg = np.zeros((256, XR, CE)); wreg = np.zeros((256, WR, CE))
            for t in range(256):
                if ksize == 3:
                    ky, kx = tap[t] // 3, tap[t] % 3
                    toff = (ky * s0["arr"].shape[2] + kx) * s0["arr"].shape[3] + coff[t] * CE
                    for i in range(XR):
                        v = tap[t] < 9 and (xmask[t, i] >> tap[t]) & 1
                        if v:
                            o = xoff0[t, i] + toff
                            assert 0 <= o and o + CE <= flat[0].size
                            xreg[t, i] = flat[0][o:o + CE]
                else:
                    qq = ks * 8 + c8[t]
                    seg1 = ks * 8 >= ct0
                    cq = qq - ct0 if seg1 else qq
                    kv = cq < ((s1["ch_cnt"] // CE) if seg1 else ct0) and ((not seg1) or nsrc > 1)
                    toff = cq * CE
                    for i in range(XR):
                        if kv and xmask[t, i]:
                            o = (xoff1[t, i] if seg1 else xoff0[t, i]) + toff
                            f = flat[1 if (seg1 and nsrc > 1) else 0]
                            assert 0 <= o and o + CE <= f.size
                            xreg[t, i] = f[o:o + CE]
                for j in range(WR):
                    row = r0[t] + 32 * j; n = n0 + row
                    if row < BN and n < cout:
                        k0 = (ks * 8 + c8[t]) * CE
                        wreg[t, j] = w_packed[n, k0:k0 + CE]
            if ksize == 3:
                coff = coff + 8
                for t in range(256):
                    while coff[t] >= ct0:
                        coff[t] -= ct0; tap[t] += 1
            return xreg, wreg

        def stage_store(buf, xreg, wreg):
            for t in range(256):
                for i in range(XR):
                    rr, cc = lds_slot(r0[t] + 32 * i, c8[t]); Xs[buf, rr, cc] = xreg[t, i]
                for j in range(WR):
                    row = r0[t] + 32 * j
                    if row < BN:
                        rr, cc = lds_slot(row, c8[t]); Ws[buf, rr, cc] = wreg[t, j]

        acc = np.zeros((4, 64, TC, TP, 4))   # wave, lane, tc, tp, reg

        def compute(buf):
            for wave in range(4):
                wp, wc = wave // WC, wave % WC
                for kk in range(2):
                    for i in range(TC):
                        for j in range(TP):
                            A = np.zeros((16, 4 * CE)); Bm = np.zeros((4 * CE, 16))
                            for lane in range(64):
                                frow, fq = lane & 15, lane >> 4
                                rr, cc = lds_slot((wc * TC + i) * 16 + frow, kk * 4 + fq)
                                A[frow, fq * CE:(fq + 1) * CE] = Ws[buf, rr, cc]
                                rr, cc = lds_slot((wp * TP + j) * 16 + frow, kk * 4 + fq)
                                Bm[fq * CE:(fq + 1) * CE, frow] = Xs[buf, rr, cc]
                            D = A @ Bm
                            for lane in range(64):
                                for rg in range(4):
                                    acc[wave, lane, i, j, rg] += D[4 * (lane >> 4) + rg, lane & 15]

        xr, wr = stage_load(0); stage_store(0, xr, wr)
        for ks in range(nk):
            cur = ks & 1
            if ks + 1 < nk:
                xr, wr = stage_load(ks + 1)
            compute(cur)
            if ks + 1 < nk:
                stage_store(cur ^ 1, xr, wr)
        for wave in range(4):
            wp, wc = wave // WC, wave % WC
            for lane in range(64):
                frow, fq = lane & 15, lane >> 4
                for i in range(TC):
                    n = n0 + (wc * TC + i) * 16 + fq * 4
                    if n >= cout:
                        continue
                    for j in range(TP):
                        m = m0 + (wp * TP + j) * 16 + frow
                        if m >= M:
                            continue
                        for rg in range(4):
                            if n + rg < cout:
                                v = acc[wave, lane, i, j, rg] + bias[n + rg]
                                if act:
                                    v = v / (1 + np.exp(-v))
                                if res is not None:
                                    v += res[m, n + rg]
                                assert np.isnan(out[m, n + rg]), "output written twice"
                                out[m, n + rg] = v
    assert len(seen) == nblk
    assert not np.isnan(out).any(), "some outputs never written"
    return out

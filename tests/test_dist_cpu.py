"""CPU (gloo, world_size 2): the data-parallel shard + all-gather logic of manual_yolo_amd/dist.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from manual_yolo_amd.dist import DetectionGather, all_gather_detections, shard_bounds, unpad


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, tmp, n_total):
    # file rendezvous and file results: no TCP port to race for, no queue feeder threads (both were seen to fail now and then
    # on a loaded 8-core box)
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", init_method=f"file://{tmp}/rendezvous", rank=rank, world_size=world)
    lo, hi = shard_bounds(n_total, rank, world)
    bl = (n_total + world - 1) // world
    # fake per-frame detections: frame f has (f % 4) boxes whose first field is f
    dets = torch.zeros((bl, 5, 6)); counts = torch.zeros((bl,), dtype=torch.int32)
    for i, f in enumerate(range(lo, hi)):
        counts[i] = f % 4
        dets[i, : f % 4, 0] = float(f)
    gd, gc = all_gather_detections(dets, counts)
    # the fused one-message form: the "engine" writes into the payload views, two batches through the two slots
    g = DetectionGather(bl, 5, "cpu", depth=2)
    fused = []
    for step in range(3):
        od, oc, _ = g.out_buffers(step)
        od.copy_(dets + step); oc.copy_(counts)
        g.launch(step)
        fd, fc = g.wait(step)
        fused.append((fd.clone(), fc.clone()))
    torch.save((rank, gd.clone(), gc.clone(), fused), os.path.join(tmp, f"out{rank}.pt.tmp"))
    os.replace(os.path.join(tmp, f"out{rank}.pt.tmp"), os.path.join(tmp, f"out{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _run_world(world, n_total):
    import shutil
    import tempfile
    ctx = mp.get_context("spawn")
    tmp = tempfile.mkdtemp(prefix="miyolo_dist_")
    try:
        procs = [ctx.Process(target=_worker, args=(r, world, tmp, n_total)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
        if any(p.is_alive() for p in procs):
            for p in procs:
                p.kill()
            return None
        if any(p.exitcode != 0 for p in procs):
            return None
        return [torch.load(os.path.join(tmp, f"out{r}.pt"), weights_only=False) for r in range(world)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_all_gather_detections_world2():
    world, n_total = 2, 7
    got = _run_world(world, n_total) or _run_world(world, n_total)      # one retry: process start-up on a busy box
    assert got is not None, "the two gloo ranks did not finish"
    bl = 4
    for rank, gd, gc, fused in got:
        for step, (fd, fc) in enumerate(fused):            # one-message gather == two-collective gather
            assert torch.equal(fc, gc) and fd.shape == gd.shape
            for r in range(world):
                assert torch.equal(fd[r * bl:(r + 1) * bl], gd[r * bl:(r + 1) * bl] + step)
        assert gd.shape == (world * bl, 5, 6) and gc.shape == (world * bl,)
        frames = unpad(gd, gc)
        # rank r's shard sits at rows [r*bl, r*bl + shard_len)
        for r in range(world):
            lo, hi = shard_bounds(n_total, r, world)
            for i, f in enumerate(range(lo, hi)):
                rows = frames[r * bl + i]
                assert rows.shape[0] == f % 4 and bool((rows[:, 0] == f).all())
    assert torch.equal(got[0][1], got[1][1]) and torch.equal(got[0][2], got[1][2])


def test_detection_gather_world1_is_identity():
    g = DetectionGather(3, 4, "cpu")
    d, c, _ = g.out_buffers(0)
    d.fill_(2.0); c.copy_(torch.tensor([1, 0, 4], dtype=torch.int32))
    g.launch(0)
    gd, gc = g.wait(0)
    assert gd.shape == (3, 4, 6) and gc.tolist() == [1, 0, 4] and bool((gd == 2.0).all())

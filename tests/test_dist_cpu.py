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
    if n_total < 0 and rank == 1:          # test_failed_rank_is_reported: die before the collective, the peer blocks in it
        print("rank 1 gives up on purpose", file=__import__("sys").stderr, flush=True)
        os._exit(3)
    n_total = abs(n_total)
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


def _entry(rank, world, tmp, n_total):
    # each rank's stderr goes to a file so that a failure can be reported with its cause
    err = open(os.path.join(tmp, f"err{rank}.txt"), "w")
    os.dup2(err.fileno(), 2)
    _worker(rank, world, tmp, n_total)


def _run_world(world, n_total, timeout=600):
    """Start the ranks, wait for all of them, return their results.  NO retry: a rank that dies or hangs in the gather is what
    this test exists to show - the failure carries every rank's exit code (or "timed out") and the tail of its stderr.  The
    join timeout is generous because spawning two interpreters that import torch can take minutes on a loaded 8-core box."""
    import shutil
    import tempfile
    import time
    ctx = mp.get_context("spawn")
    tmp = tempfile.mkdtemp(prefix="miyolo_dist_")
    try:
        procs = [ctx.Process(target=_entry, args=(r, world, tmp, n_total)) for r in range(world)]
        for p in procs:
            p.start()
        t_end = time.monotonic() + timeout
        while any(p.is_alive() for p in procs) and time.monotonic() < t_end:
            if any(p.exitcode not in (None, 0) for p in procs):      # a dead rank leaves its peer in the collective: stop waiting
                time.sleep(2.0)
                break
            time.sleep(0.1)
        status = []
        for r, p in enumerate(procs):
            if p.is_alive():
                p.kill()
                p.join(10)
                status.append(f"rank {r}: timed out / killed while its peer had failed")
            else:
                status.append(f"rank {r}: exit code {p.exitcode}")
        if any(p.exitcode != 0 for p in procs):
            tails = []
            for r in range(world):
                try:
                    tails.append(f"--- rank {r} stderr tail ---\n" + open(os.path.join(tmp, f"err{r}.txt")).read()[-1500:])
                except OSError:
                    pass
            raise AssertionError("gloo world failed: " + "; ".join(status) + "\n" + "\n".join(tails))
        return [torch.load(os.path.join(tmp, f"out{r}.pt"), weights_only=False) for r in range(world)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_all_gather_detections_world2():
    world, n_total = 2, 7
    got = _run_world(world, n_total)
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


def test_failed_rank_is_reported():
    """A rank that dies leaves its peer blocked in the all-gather: the harness must fail with each rank's fate and stderr,
    not hang and not retry."""
    import pytest
    with pytest.raises(AssertionError) as e:
        _run_world(2, -7, timeout=120)
    msg = str(e.value)
    assert "rank 1: exit code 3" in msg and "gives up on purpose" in msg, msg


def test_detection_gather_world1_is_identity():
    g = DetectionGather(3, 4, "cpu")
    d, c, _ = g.out_buffers(0)
    d.fill_(2.0); c.copy_(torch.tensor([1, 0, 4], dtype=torch.int32))
    g.launch(0)
    gd, gc = g.wait(0)
    assert gd.shape == (3, 4, 6) and gc.tolist() == [1, 0, 4] and bool((gd == 2.0).all())

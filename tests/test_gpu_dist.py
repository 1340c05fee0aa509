"""-m gpu: the RCCL leg of the data-parallel path on ONE GPU - `torch.distributed` backend "nccl" (= RCCL on ROCm) with
world size 1: the same `all_gather_into_tensor` call, side stream, events and payload plumbing as with 8 ranks, only
without peers (8-GPU runs are the driver's).  The gloo world-2 twin is tests/test_dist_cpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from manual_yolo_amd.dist import DetectionGather, unpad
from manual_yolo_amd.engine import engine_from_weights
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_fused_gather_over_rccl_overlaps_and_matches(nccl_world1):
    sd, meta = synth_state_dict("detect", 64, "n", 0), synth_meta("detect", 64, "n")
    eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
    B, max_det = 4, 300
    frames = [torch.from_numpy(synth_frames(B, 160, 160, seed=s)).cuda() for s in (1, 2, 3)]
    want = [tuple(t.clone() for t in eng.detect(f, want_anchor=False)[:2]) for f in frames]
    g = DetectionGather(B, max_det, "cuda:0", depth=2, always_collective=True)
    assert g.nccl and g.side is not None
    got = []
    for k, f in enumerate(frames):                 # batch k+1 is enqueued while gather k runs on the side stream
        if k >= 2:
            got.append(tuple(t.clone() for t in g.wait(k - 2)))      # slot about to be overwritten
        # exactly bench.py's step: deferred wait, the gather's side stream waits for the detections by itself
        eng.detect(f, want_anchor=False, out=g.out_buffers(k), defer=True)
        g.launch(k, ready=eng.wait_outputs)
    for k in range(max(0, len(frames) - 2), len(frames)):
        got.append(tuple(t.clone() for t in g.wait(k)))
    torch.cuda.synchronize()
    for (wd, wc), (gd, gc) in zip(want, got):
        assert torch.equal(wd, gd) and torch.equal(wc, gc)
        assert sum(len(x) for x in unpad(gd.cpu(), gc.cpu())) == int(gc.sum())


def test_gather_behind_an_asynchronous_nms(nccl_world1):
    """Option nms_async: the detections of a call are produced on the library's internal stream; the gather's side stream must
    wait for them through the `ready` hook (Engine.wait_outputs) - the payload it sends is then the finished one."""
    sd, meta = synth_state_dict("detect", 64, "n", 0), synth_meta("detect", 64, "n")
    eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
    B, max_det = 4, 300
    frames = [torch.from_numpy(synth_frames(B, 160, 160, seed=s)).cuda() for s in (5, 6, 7, 8)]
    want = [tuple(t.clone() for t in eng.detect(f, conf=0.05, want_anchor=False)[:2]) for f in frames]
    eng.set_option("nms_async", 1)
    g = DetectionGather(B, max_det, "cuda:0", depth=2, always_collective=True)
    got = []
    for k, f in enumerate(frames):
        if k >= 2:
            got.append(tuple(t.clone() for t in g.wait(k - 2)))
        eng.detect(f, conf=0.05, want_anchor=False, out=g.out_buffers(k), defer=True)
        g.launch(k, ready=eng.wait_outputs)
    for k in range(len(frames) - 2, len(frames)):
        got.append(tuple(t.clone() for t in g.wait(k)))
    torch.cuda.synchronize()
    for (wd, wc), (gd, gc) in zip(want, got):
        assert torch.equal(wd, gd) and torch.equal(wc, gc)

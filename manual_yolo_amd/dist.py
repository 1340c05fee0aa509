"""Data-parallel sharding of frames across the GPUs of one node (one process per GPU).

The reference is single-process; frames are independent inside
``model(frame)``, so the batch dimension shards with replicated weights and ONE exchange
step: an all-gather of the fixed-shape, zero-padded per-frame detections
(``dets [B_local, max_det, 6] f32``, ``counts [B_local] i32``) - 461 KB per rank at B=64, a
latency-bound message on xGMI.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo"
is used by the CPU tests of this logic.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group
    when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MIYOLO_DIST_BACKEND=gloo lets the N>1 path be rehearsed with several ranks on ONE GPU
            # (RCCL refuses duplicate devices); production is nccl (= RCCL over xGMI)
            backend = os.environ.get("MIYOLO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``n_total`` frames for ``rank`` (sizes differ by <= 1)."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_detections(dets: torch.Tensor, counts: torch.Tensor, group=None,
                          out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Every rank gets every rank's padded detections: [world*B_local, max_det, 6], [world*B_local].
    Requires equal B_local on all ranks (pad the last shard)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dets, counts
    world = dist.get_world_size(group)
    if out is None:
        gd = torch.empty((world * dets.shape[0],) + tuple(dets.shape[1:]), dtype=dets.dtype, device=dets.device)
        gc = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    else:
        gd, gc = out
    if dets.is_cuda and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(gd, dets.contiguous(), group=group)
        dist.all_gather_into_tensor(gc, counts.contiguous(), group=group)
    else:  # gloo (CPU tests, single-GPU rehearsal)
        dist.all_gather(list(gd.chunk(world)), dets.contiguous(), group=group)
        dist.all_gather(list(gc.chunk(world)), counts.contiguous(), group=group)
    return gd, gc


class DetectionGather:
    """The exchange step of the data-parallel path as ONE message per rank, off the critical path (SURVEY.md 8e ii).

    Each rank owns `depth` payload buffers `[B_local*max_det*6 floats | B_local int32 counts]` (one contiguous fp32
    tensor; the counts are int32 bit patterns in its tail).  `out_buffers(slot)` hands the engine views of a payload, so
    `miyolo_detect` writes its outputs straight into the message - no packing copy.  `launch(slot)` issues a single
    `all_gather_into_tensor` (RCCL) on a SIDE stream that first waits for the compute stream's work so far; the
    compute stream goes straight on to the next batch, which writes the other payload.  `wait(slot)` makes the current
    stream wait for that gather and returns `(dets [world*B_local, max_det, 6], counts [world*B_local])` views of the
    gathered buffer.  461 KB per rank at B_local = 64: latency-bound on xGMI, hidden under ~8 ms of compute.
    With gloo (CPU tests, several ranks rehearsed on one GPU) the gather runs synchronously in `launch`."""

    def __init__(self, b_local: int, max_det: int, device, group=None, depth: int = 2, always_collective: bool = False):
        self.b, self.max_det, self.group, self.depth = b_local, max_det, group, depth
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.nd = b_local * max_det * 6
        self.n = self.nd + b_local
        self.device = torch.device(device)
        self.payload = [torch.zeros(self.n, dtype=torch.float32, device=self.device) for _ in range(depth)]
        self.gathered = [torch.zeros(self.world * self.n, dtype=torch.float32, device=self.device) for _ in range(depth)]
        # always_collective: run the collective even at world size 1 (exercises the RCCL code path on a one-GPU box)
        self.active = self.world > 1 or (always_collective and dist.is_initialized())
        self.nccl = self.active and self.device.type == "cuda" and dist.get_backend(group) == "nccl"
        self.side = torch.cuda.Stream(self.device) if self.nccl else None
        self.done = [None] * depth

    def out_buffers(self, slot: int):
        p = self.payload[slot % self.depth]
        return p[: self.nd].view(self.b, self.max_det, 6), p[self.nd:].view(torch.int32), None

    def launch(self, slot: int, ready=None):
        """`ready(stream)`: optional hook that orders `stream` behind whatever still produces the payload off the current
        stream (Engine.wait_outputs with option nms_async)."""
        k = slot % self.depth
        if not self.active:
            return
        if self.nccl:
            cur = torch.cuda.current_stream(self.device)
            self.side.wait_stream(cur)                       # the detections of this batch are complete
            if ready is not None:
                ready(self.side)
            with torch.cuda.stream(self.side):
                dist.all_gather_into_tensor(self.gathered[k], self.payload[k], group=self.group)
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.done[k] = ev
        else:
            if ready is not None:
                ready(None)
            dist.all_gather(list(self.gathered[k].chunk(self.world)), self.payload[k], group=self.group)

    def wait(self, slot: int):
        k = slot % self.depth
        if not self.active:
            d, c, _ = self.out_buffers(slot)
            return d, c
        if self.nccl and self.done[k] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.done[k])      # also orders a later overwrite of payload[k]
            self.done[k] = None
        g = self.gathered[k].view(self.world, self.n)
        return (g[:, : self.nd].reshape(self.world * self.b, self.max_det, 6),
                g[:, self.nd:].contiguous().view(torch.int32).reshape(self.world * self.b))


def unpad(dets: torch.Tensor, counts: torch.Tensor) -> List[torch.Tensor]:
    c = counts.tolist()
    return [dets[i, :c[i]] for i in range(dets.shape[0])]

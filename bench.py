#!/usr/bin/env python3
"""Headline benchmark: frames/s of yolov8m (nc=64) @ 640x640, batch 64 per GPU, NMS on-GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic frames already resident in
HBM: stem -> 82 MFMA convs -> SPPF pools -> Detect decode -> NMS -> (N>1) RCCL all-gather of the
padded detections.  Rank 0 prints ONE JSON line (contract in the task statement), with
`roofline` for the dominant kernel (per-launch HIP-event timings taken on the launch stream by
the engine) and `cpu_baseline` (the CPU oracle = restated Ultralytics CPU path, timed on this
host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--scale", default="m")
    ap.add_argument("--nc", type=int, default=64)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--workload", default="detect", choices=["detect", "classify"])
    ap.add_argument("--chunk", type=int, default=0, help="images per engine pass (0 = auto)")
    ap.add_argument("--conv-impl", type=int, default=-1, help="0 register-staged conv, 1 LDS-DMA ring, 2 ring + halo kernel, 3 persistent ring, 4 persistent halo, 5 warp-specialised, 6 half-size stages x 2 workgroups per CU (default: engine default)")
    ap.add_argument("--graph", type=int, default=-1, help="1: hipGraph replay of the step's launches (measured SLOWER: classifier 0.290 vs 0.230 ms per batch of 256, detect unchanged; default off)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (miyolo_set_option), repeatable: A/B timing of kernel choices")
    ap.add_argument("--ablate", type=int, default=0, help="timing experiments only (wrong results): see common.h")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--profile-out", default="")
    return ap.parse_args()


def cpu_baseline(args, sd, meta, frames_np):
    """Oracle (port of the Ultralytics CPU path) on the host cores, bounded sample."""
    from oracle.post_ref import non_max_suppression
    from oracle.yolo_ref import RefYolo
    # the GPU box gives one GPU's share of the host: 16 cores (more threads only oversubscribe)
    cores = min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    task = meta["task"]
    ref = RefYolo(sd, task, meta["nc"], meta["scale"], meta["bn_eps"], nc_quirk=meta.get("nc_quirk", True))
    bs = 4 if task == "detect" else 256
    x = torch.from_numpy(frames_np[:bs]).permute(0, 3, 1, 2).float() / 255
    def once():
        out = ref.forward(x)
        if task == "detect":
            non_max_suppression(out[0].numpy(), 0.25, 0.7)
    once()
    t0 = time.perf_counter(); n = 0
    while True:
        once(); n += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or n >= 50:
            break
    return {"value": round(bs * n / el, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x batch {bs} of the same synthetic {x.shape[2]}x{x.shape[3]} frames, torch-CPU fp32 "
                      f"restatement incl. NMS, {el:.1f} s"}


def main():
    args = parse()
    from manual_yolo_amd import dist as mdist
    from manual_yolo_amd.engine import engine_from_weights
    from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
    rank, world, local = mdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if os.environ.get("MIYOLO_FORCE_DEVICE"):          # rehearsal: several ranks on one GPU (with gloo)
        local = int(os.environ["MIYOLO_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    task = args.workload
    if task == "detect":
        sd, meta = synth_state_dict("detect", args.nc, args.scale, 0), synth_meta("detect", args.nc, args.scale)
        H = W = args.imgsz
        B = args.batch
    else:
        from manual_yolo_amd.ckpt import load_bundle
        sd, meta = load_bundle(os.path.join(ROOT, "tests", "golden", "rank_best.safetensors"))
        H = W = 64
        B = args.batch if args.batch != 64 else 256
    eng = engine_from_weights(sd, meta, args.dtype, local, bgr_input=False)
    if args.chunk:
        eng.set_option("max_chunk", args.chunk)
    if args.conv_impl >= 0:
        eng.set_option("conv_impl", args.conv_impl)
    if args.ablate:
        eng.set_option("ablate", args.ablate)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    use_graph = args.graph if args.graph >= 0 else 0
    if use_graph:
        eng.set_option("graph", 1)
    frames_np = synth_frames(B, H, W, seed=1 + rank)
    frames = torch.from_numpy(frames_np).to(dev)
    max_det = 300
    if task == "detect":
        out = (torch.empty((B, max_det, 6), dtype=torch.float32, device=dev),
               torch.empty((B,), dtype=torch.int32, device=dev), None)
        gout = (torch.empty((world * B, max_det, 6), dtype=torch.float32, device=dev),
                torch.empty((world * B,), dtype=torch.int32, device=dev)) if world > 1 else None

    def step():
        if task == "detect":
            d, c, _ = eng.detect(frames, 0.25, 0.7, False, max_det, None, want_anchor=False, out=out)
            if world > 1:
                mdist.all_gather_detections(d, c, out=gout)
        else:
            eng.classify(frames)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel: per-launch HIP events recorded by the engine
    roofline = None
    if not args.no_roofline:
        eng.set_option("profile", 1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        recs = eng.profile_read()
        eng.set_option("profile", 0)
        chunk = eng.chunk(B, H, W)
        per = {}
        perop = {}
        for op, cfg, ms in recs:
            o = eng.prog.ops[op]
            fl0, by0 = eng.op_work(op, min(chunk, B), H, W)
            e0 = perop.setdefault(op, {"name": o.name, "kind": o.kind, "k": o.ksize, "s": o.stride, "cin": o.cin, "cout": o.cout,
                                       "down": o.down_out, "cfg": cfg, "n": 0, "ms": 0.0, "flop": fl0, "bytes": by0})
            e0["n"] += 1; e0["ms"] += ms
        for op, cfg, ms in recs:
            k = ("%s<%s,k%d,wc%d,tc%d>" % ("conv_h2" if cfg >= 8000 else "conv_t2d" if cfg >= 7000 else "conv_dmh" if cfg >= 6000 else "conv_ws" if cfg >= 5000 else "conv_halop" if cfg >= 4000 else "conv_dmap" if cfg >= 3000 else "conv_halo" if cfg >= 2000 else "conv_dma" if cfg >= 1000 else "conv_igemm", args.dtype, (cfg // 100) % 10, (cfg // 10) % 10, cfg % 10)) if cfg else \
                {0: "stem", 2: "maxpool5", 3: "decode", 4: "cls_head"}.get(eng.prog.ops[op].kind, "op")
            fl, by = eng.op_work(op, min(chunk, B), H, W)
            e = per.setdefault(k, [0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
        dom = max(per.items(), key=lambda kv: kv[1][1])
        name, (n, ms, fl, by) = dom
        # HBM traffic per launch of that kernel: not measurable from inside this process - taken from the
        # committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json), same workload
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["kernels"]
            key = name.replace(",k", ",").replace(",wc", ",").replace(",tc", ",")
            if key in tj and (B, H, W, args.dtype, args.scale) == (64, 640, 640, "f16", "m"):
                traffic = round(tj[key]["traffic_bytes_per_launch"])
        except Exception:
            traffic = None
        total_ms = sum(v[1] for v in per.values())
        conv_fl = sum(v[2] for k, v in per.items() if k.startswith("conv")); conv_ms = sum(v[1] for k, v in per.items() if k.startswith("conv"))
        if fl > 0:
            ach = fl / (ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.dtype],
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic,
                        "launches": n, "avg_launch_ms": round(ms / n, 4), "flop_per_launch": fl / n,
                        "algorithmic_bytes_per_launch": round(by / n),
                        "share_of_step_kernel_time": round(ms / total_ms, 3),
                        "all_convs_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None}
        else:
            ach = by / (ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": n,
                        "avg_launch_ms": round(ms / n, 4)}
        if args.profile_out and rank == 0:
            with open(args.profile_out, "w") as f:
                agg = {k: {"launches": v[0], "ms": v[1], "flop": v[2], "bytes": v[3],
                           "tflops": (v[2] / (v[1] * 1e-3) / 1e12) if v[1] else 0,
                           "gbs": (v[3] / (v[1] * 1e-3) / 1e9) if v[1] else 0} for k, v in per.items()}
                layers = []
                for op in sorted(perop):
                    e0 = perop[op]
                    avg = e0["ms"] / e0["n"]
                    e0.update(avg_ms=avg, tflops=e0["flop"] / (avg * 1e-3) / 1e12 if avg else 0,
                              gbs=e0["bytes"] / (avg * 1e-3) / 1e9 if avg else 0)
                    layers.append(e0)
                json.dump({"by_kernel": agg, "by_op": layers}, f, indent=1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sd, meta, frames_np)

    if rank == 0:
        fl_step, by_step = eng.work(B, H, W)
        # layer-wise mixed roofline of the whole step, t_min = sum over ops of
        # max(flops / dense MFMA peak, compulsory bytes / HBM peak); model_roofline_frac = t_min / measured step
        t_min = 0.0
        for i in range(len(eng.prog.ops)):
            fl_i, by_i = eng.op_work(i, B, H, W)
            t_min += max(fl_i / (PEAK_TFLOPS[args.dtype] * 1e12), by_i / (HBM_PEAK_GBS * 1e9))
        line = {
            "metric": "frames/sec whole-node, yolov8m@640 batch=64; mAP delta vs CPU ref" if task == "detect"
                      else "images/sec, yolov8n-cls rank classifier 64x64",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"yolov8{args.scale} detect nc={args.nc} {H}x{W}, batch {B}/GPU, seeded random-init "
                                    f"weights, uint8 frames resident in HBM, NMS on-GPU (conf 0.25, iou 0.7, max_det 300)"
                                    + (", RCCL all-gather of detections" if world > 1 else ""))
                                   if task == "detect" else f"yolov8n-cls rank_classifier weights 64x64, batch {B}/GPU",
                       "global_batch": world * B, "parallelism": f"dp{world}", "hip_graph": bool(use_graph),
                       "gflop_per_frame": round(fl_step / B / 1e9, 3), "algorithmic_mb_per_frame": round(by_step / B / 1e6, 2),
                       "model_tflops": round(fl_step * world / (ms_per_step * 1e-3) / 1e12, 2),
                       "model_t_min_ms": round(t_min * 1e3, 4), "model_roofline_frac": round(t_min * 1e3 / ms_per_step, 4)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

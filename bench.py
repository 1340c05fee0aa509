#!/usr/bin/env python3
"""Headline benchmark: frames/s of yolov8m (nc=64) @ 640x640, batch 64 per GPU, NMS on-GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...;
     without torchrun, `python bench.py --gpus N` spawns the N ranks itself)

A "step" is one pass of the hot path over one batch of synthetic frames already resident in
HBM: stem -> 82 MFMA convs -> SPPF pools -> Detect decode -> NMS -> (N>1) one RCCL all-gather of
the padded detections, issued on a side stream so that it overlaps the next batch.  Rank 0 prints
ONE JSON line (contract in the task statement) with
  roofline      the dominant kernel (per-launch HIP-event timings taken on the launch stream by the engine),
  parity        the timed dtype's detections against the CPU oracle on a sample of the bench frames
                ("mAP delta vs CPU ref" of BASELINE.json's metric: oracle detections are the ground truth),
  exact_f32     a short timed loop of the exact-fp32 mode (the mode that meets north_star's identical-indices bar),
  classify_config2  BASELINE config 2 (the rank classifier, batch 256, f16): images/s, launches, HBM roofline, 63/67 check,
  cpu_baseline  the CPU oracle (restated Ultralytics CPU path) timed on this host's cores, SURVEY.md 8d protocol.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f8": 5000.0}     # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (default 64 detect / 256 classify)")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--scale", default="m")
    ap.add_argument("--nc", type=int, default=64)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "f8"])
    ap.add_argument("--workload", default="detect", choices=["detect", "classify"])
    ap.add_argument("--chunk", type=int, default=0, help="images per engine pass (0 = auto)")
    ap.add_argument("--conv-impl", type=int, default=-1, help="0 register-staged conv, 1 LDS-DMA ring, 3 engine default (persistent ring + halo-slab + 2-D-tile kernels), 7 / 8 force the 2-D-tile / halo-slab kernel where eligible")
    ap.add_argument("--graph", type=int, default=-1, help="1: hipGraph replay of the step's launches (measured no gain; default off)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (miyolo_set_option), repeatable: A/B timing of kernel choices")
    ap.add_argument("--ablate", type=int, default=0, help="timing experiments only (wrong results): see common.h")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true")
    ap.add_argument("--no-classify", action="store_true", help="skip the config-2 side block (classifier, batch 256)")
    ap.add_argument("--parity-frames", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=50.0)
    ap.add_argument("--profile-out", default="")
    ap.add_argument("--quant-cache", default="", help="fp8: .npz of the calibration (written if missing, read if present) - keeps the calibration pass out of a profiled run")
    ap.add_argument("--force-spawn", action="store_true", help="take the self-spawn path (one child per rank) even for --gpus 1")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process may use, WITHOUT touching the HIP/HSA runtime (an exec from a GPU-initialised process is fatal on this
    pool, and `torch.cuda.device_count()` falls through to hipGetDeviceCount on ROCm): the kfd topology in sysfs lists one
    node per agent, GPUs are the nodes with a non-zero `simd_count`; HIP_/ROCR_VISIBLE_DEVICES narrow it.  None: unknown."""
    import glob
    n = 0
    try:
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except (OSError, ValueError):
        return None
    if n == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args, timeout_s=1500.0):
    """`python bench.py --gpus N` without torchrun: start one child per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set),
    relay rank 0's JSON line, fail if any rank fails.  The parent imports neither torch nor the package and makes no HIP
    call (GPU count from sysfs); it polls ALL children: the first non-zero exit or the overall timeout ends the others (a
    rank that died would otherwise leave rank 0 waiting in the collective for ever)."""
    import socket
    import tempfile
    n = args.gpus
    have = visible_gpus()
    if have is not None and have < n and not os.environ.get("MIYOLO_FORCE_DEVICE"):
        print(f"[bench] --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--force-spawn"]
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else None
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out if r == 0 else subprocess.DEVNULL, text=True))
    t_end = time.monotonic() + timeout_s
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > t_end:
            failed = "timed out after %.0f s (ranks still running: %s)" % (timeout_s, [r for r, rc in enumerate(rcs) if rc is None])
            break
        time.sleep(0.2)
    if failed:
        for p in procs:                      # exactly the children started above, by PID
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"[bench] {failed}", file=sys.stderr)
    outs[0].seek(0)
    sys.stdout.write(outs[0].read())
    sys.stdout.flush()
    return 1 if failed else 0


# ----------------------------------------------------------------------------------------------- parity vs the oracle
def _iou_matrix(a, b):
    import numpy as np
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


def map50_95(dets, gts):
    """COCO-style mAP@[.5:.95] (101-point interpolation, per class, mean over classes with ground truth and over the
    ten IoU thresholds) of `dets` (list per image of [n,6] x1,y1,x2,y2,conf,cls) against `gts` (same format: the CPU
    oracle's detections are the ground truth).  Identical detections give exactly 1.0."""
    import numpy as np
    thr = np.arange(0.5, 0.96, 0.05)
    classes = sorted({int(c) for g in gts for c in g[:, 5]})
    aps = []
    for c in classes:
        recs = []                                     # (score, tp[10]) over all images
        npos = 0
        for d, g in zip(dets, gts):
            dc, gc = d[d[:, 5] == c], g[g[:, 5] == c]
            npos += len(gc)
            if len(dc) == 0:
                continue
            order = np.argsort(-dc[:, 4], kind="stable")
            dc = dc[order]
            tp = np.zeros((len(dc), len(thr)), bool)
            if len(gc):
                iou = _iou_matrix(dc[:, :4], gc[:, :4])
                for t, th in enumerate(thr):
                    used = np.zeros(len(gc), bool)
                    for i in range(len(dc)):
                        j = int(np.argmax(np.where(used, -1.0, iou[i])))
                        if not used[j] and iou[i, j] >= th:
                            used[j] = True
                            tp[i, t] = True
            recs += [(dc[i, 4], tp[i]) for i in range(len(dc))]
        if npos == 0:
            continue
        if not recs:
            aps.append(0.0)
            continue
        recs.sort(key=lambda r: -r[0])
        tpm = np.array([r[1] for r in recs], float)
        ctp = np.cumsum(tpm, 0); cfp = np.cumsum(1 - tpm, 0)
        rec = ctp / npos; prec = ctp / (ctp + cfp)
        ap_t = []
        grid = np.linspace(0, 1, 101)
        for t in range(len(thr)):
            mrec = np.concatenate(([0.0], rec[:, t], [1.0])); mpre = np.concatenate(([1.0], prec[:, t], [0.0]))
            mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
            # precision at each recall grid point = max precision at recall >= r (COCO); 0 beyond the reached recall
            idx = np.searchsorted(mrec, grid, side="left")
            ap_t.append(float(np.mean(mpre[np.minimum(idx, len(mpre) - 1)])))
        aps.append(float(np.mean(ap_t)))
    return float(np.mean(aps)) if aps else 1.0


def parity_block(eng, sd, meta, frames_np, nframes):
    """Detections of the timed engine vs the CPU oracle (restated Ultralytics CPU path, fp32) on `nframes` bench frames."""
    import numpy as np
    import torch
    from oracle.post_ref import non_max_suppression
    from oracle.yolo_ref import RefYolo
    n = min(nframes, len(frames_np))
    ref = RefYolo(sd, "detect", meta["nc"], meta["scale"], meta["bn_eps"], nc_quirk=meta.get("nc_quirk", True))
    x = torch.from_numpy(frames_np[:n]).permute(0, 3, 1, 2).float() / 255
    y = ref.forward(x)[0].numpy()
    outs, idxs = non_max_suppression(y, 0.25, 0.7)
    dets, counts, anchor = eng.detect(torch.from_numpy(frames_np[:n]).to(eng.device), 0.25, 0.7)
    dets, counts, anchor = dets.cpu().numpy(), counts.cpu().numpy(), anchor.cpu().numpy()
    got = [dets[b, :counts[b]] for b in range(n)]
    ident = 0; common = 0; total = 0; mb = 0.0; ms = 0.0
    for b in range(n):
        ga = anchor[b, :counts[b]]
        ident += int(len(ga) == len(idxs[b]) and np.array_equal(ga, idxs[b]))
        cm, gi, oi = np.intersect1d(ga, idxs[b], return_indices=True)
        common += len(cm); total += len(idxs[b])
        if len(cm):
            mb = max(mb, float(np.abs(got[b][gi, :4] - outs[b][oi, :4]).max()))
            ms = max(ms, float(np.abs(got[b][gi, 4] - outs[b][oi, 4]).max()))
    m = map50_95(got, outs)
    return {"frames": n, "reference": "CPU oracle (restated Ultralytics CPU path, fp32) on the same frames; its detections are the ground truth",
            "map50_95": round(m, 5), "map50_95_delta": round(1.0 - m, 5),
            "kept_index_agreement": round(common / max(total, 1), 5), "frames_with_identical_kept_indices": f"{ident}/{n}",
            "max_box_px": round(mb, 4), "max_score": round(ms, 6), "kept_boxes_reference": total}


# ----------------------------------------------------------------------------------------------- CPU baseline (8d protocol)
def cpu_baseline(args, sd, meta, frames_np):
    """the torch-CPU restatement (fp32, BN folded) on this host's cores; B = 1 and B = 32 (detect),
    3 warm-up + >= 10 timed iterations (bounded by --cpu-seconds), median; per-stage split."""
    import numpy as np
    import torch
    from oracle.post_ref import non_max_suppression
    from oracle.yolo_ref import RefYolo
    cores = min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16)     # one GPU's share of the host
    torch.set_num_threads(cores)
    task = meta["task"]
    ref = RefYolo(sd, task, meta["nc"], meta["scale"], meta["bn_eps"], nc_quirk=meta.get("nc_quirk", True))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    res = {}
    t_budget = time.perf_counter() + args.cpu_seconds
    for bs in ((1, 32) if task == "detect" else (1, 256)):
        bs = min(bs, len(frames_np))
        u8 = torch.from_numpy(frames_np[:bs])
        stages = {"preprocess": [], "inference": [], "postprocess": []}
        tot = []
        it = 0
        while it < 13 and (it < 5 or time.perf_counter() < t_budget):
            t0 = time.perf_counter()
            x = u8.permute(0, 3, 1, 2).float() / 255
            t1 = time.perf_counter()
            out = ref.forward(x)
            t2 = time.perf_counter()
            if task == "detect":
                non_max_suppression(out[0].numpy(), 0.25, 0.7)
            t3 = time.perf_counter()
            if it >= 3:                                   # 3 warm-up iterations
                stages["preprocess"].append(t1 - t0); stages["inference"].append(t2 - t1); stages["postprocess"].append(t3 - t2)
                tot.append(t3 - t0)
            it += 1
        med = float(np.median(tot))
        res[f"B{bs}"] = {"value": round(bs / med, 3), "iterations": len(tot),
                         "ms_per_frame": {k: round(float(np.median(v)) * 1e3 / bs, 3) for k, v in stages.items()}}
    big = res[sorted(res, key=lambda k: int(k[1:]))[-1]]
    return {"value": big["value"], "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": model,
            "sample": f"torch-CPU fp32 restatement incl. NMS on the same synthetic frames, median of >= 10 timed iterations after 3 warm-up "
                      f"(bounded by {args.cpu_seconds:.0f} s), per batch size", "by_batch": res}


# ----------------------------------------------------------------------------------------------- config 2 beside the headline
def classify_block(dev, local, steps=200):
    """BASELINE.json config 2 (yolov8n-cls rank classifier, 64 x 64, batch 256, f16, 1 GPU) as a side block of the headline
    line, outside its timed region: images/s, launches per call, the HBM roofline of SURVEY.md 8d (406 KB of compulsory
    fp16 traffic per image -> 19.6 M images/s at 8 TB/s) and the reference's known answer on the 67 validation crops."""
    import numpy as np
    import torch
    from manual_yolo_amd.ckpt import load_bundle
    from manual_yolo_amd.engine import engine_from_weights
    from manual_yolo_amd.synth import synth_frames
    g = os.path.join(ROOT, "tests", "golden")
    sd, meta = load_bundle(os.path.join(g, "rank_best.safetensors"))
    eng = engine_from_weights(sd, meta, "f16", local, bgr_input=False)
    B = 256
    x = torch.from_numpy(synth_frames(B, 64, 64, seed=0)).to(dev)
    for _ in range(10):
        eng.classify(x)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.classify(x)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / steps
    launches, lds = eng.classify_launches(64, 64)
    fl, by = eng.work(B, 64, 64)
    val = np.load(os.path.join(g, "rank_valid.npz"))
    _, p = eng.classify(torch.from_numpy(val["pre_u8"]).to(dev))
    top1 = int((p.argmax(1).cpu().numpy() == val["labels"]).sum())
    ips = B / ms * 1e3
    return {"workload": "yolov8n-cls rank_classifier weights (reference best.pt), 64x64, batch 256, f16, inputs resident in HBM",
            "value": round(ips, 1), "unit": "images/s", "ms_per_call": round(ms, 4), "steps": steps, "launches_per_call": launches,
            "lds_bytes_per_image": lds, "gflop_per_image": round(fl / B / 1e9, 5), "algorithmic_kb_per_image": round(by / B / 1e3, 1),
            "roofline": {"bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "algorithmic bytes (each layer's input once + output once + weights once) / wall time of the call; "
                                 "the one-launch kernel keeps activations in LDS, so its real HBM traffic is the input + weights only"},
            "top1_on_reference_valid_crops": f"{top1}/67", "reference_known_answer": "63/67 (runs/rank_classifier/results.csv:21)"}


# ----------------------------------------------------------------------------------------------- roofline from engine events
def kernel_name(cfg, kind, dtype):
    if not cfg:
        return {0: "stem", 2: "maxpool5", 3: "decode", 4: "cls_head"}.get(kind, "op")
    if cfg >= 10000:
        return "conv_pw<%s,nt%d>" % (dtype, cfg % 100)             # streaming 1x1 kernel (option pw, off by default)
    if cfg >= 9400:
        return "conv_stem2<%s,tc%d>" % (dtype, cfg % 10)           # the stem and the stride-2 conv behind it in one launch
    if cfg >= 9000:
        return "conv_bneck<%s,tc%d>" % (dtype, cfg % 10)           # two 3x3 convs of a narrow Bottleneck in one launch
    fam = ("conv_h4" if cfg >= 8000 and (cfg // 10) % 10 == 6 else "conv_h3" if cfg >= 8000 and (cfg // 10) % 10 == 5 else "conv_h2" if cfg >= 8000 else "conv_t2d" if cfg >= 7000 else "conv_dmh" if cfg >= 6000 else "conv_ws" if cfg >= 5000 else
           "conv_halop" if cfg >= 4000 else "conv_dmap" if cfg >= 3000 else "conv_halo" if cfg >= 2000 else "conv_dma" if cfg >= 1000 else "conv_igemm")
    return "%s<%s,k%d,wc%d,tc%d>" % (fam, dtype, (cfg // 100) % 10, (cfg // 10) % 10, cfg % 10)


def lookup_traffic(name, B, H, W, dtype):
    """HBM traffic per launch of kernel `name` (bench naming, e.g. conv_h2<f16,k3,wc4,tc6>) from the committed rocprofv3 --pmc
    passes of the SAME workload (profiles/r03_traffic*.json: FETCH_SIZE x 2 + WRITE_SIZE per launch, tools/pmc_traffic.py);
    (None, None) when no committed pass matches the workload.  It cannot be measured from inside this process."""
    import glob
    key = name.replace(",k", ",").replace(",wc", ",").replace(",tc", ",")
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_traffic*.json"))):
        try:
            tj = json.load(open(tf))
        except (OSError, ValueError):
            continue
        if tj.get("workload") != [B, H, W, dtype]:
            continue
        traffic = None
        if key in tj["kernels"]:
            traffic = round(tj["kernels"][key]["traffic_bytes_per_launch"])
        elif name.startswith("conv_h2<") or name.startswith("conv_h3<"):
            # rocprofv3 names the halo-slab kernels per tile geometry (conv_h2<dtype,tc,geo>): launch-weighted mean over them
            fam, tc = name[:7], name[name.index(",tc") + 3:-1]
            ks = [v for k, v in tj["kernels"].items() if k.startswith(f"{fam}<{dtype},{tc},")]
            if ks:
                traffic = round(sum(v["traffic_bytes_per_launch"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks))
        if traffic is not None:
            return traffic, f"profiles/{os.path.basename(tf)} (rocprofv3 --pmc passes of this workload, committed; not re-measured by this run)"
    return None, None


def roofline_block(eng, step, steps, B, H, W, dtype, dev, profile_out=""):
    import torch
    eng.set_option("profile", 1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    recs = eng.profile_read()
    eng.set_option("profile", 0)
    chunk = eng.chunk(B, H, W)
    per, perop = {}, {}
    recorded = {op for op, _, _ in recs}
    for op, cfg, ms in recs:
        o = eng.prog.ops[op]
        fl, by = eng.op_work(op, min(chunk, B), H, W)
        # ops absorbed into this launch (a fused Bottleneck's second conv, SPPF's 2nd and 3rd pool) have no record of their
        # own: their algorithmic work belongs to it.  The intermediate of a fused Bottleneck never touches HBM: its bytes
        # are not added (flops are)
        nxt = op + 1
        while nxt < len(eng.prog.ops) - 1 and nxt not in recorded and eng.prog.ops[nxt].kind in (1, 2):
            f2, b2 = eng.op_work(nxt, min(chunk, B), H, W)
            fl += f2
            if cfg < 9000:
                by += b2
            elif 9400 <= cfg < 9600:
                # stem + layer 1 (+ the 1x1 conv behind it, 95xx) in one launch: the map between two of its ops (one's output,
                # the next one's input) never touches HBM
                prev = eng.prog.ops[nxt - 1]
                es = {"f16": 2, "f32": 4, "f8": 1}[dtype]
                inner_map = min(chunk, B) * (H // prev.down_out) * (W // prev.down_out) * prev.cout * es
                by += b2 - 2 * inner_map
            nxt += 1
        e0 = perop.setdefault(op, {"name": o.name, "kind": o.kind, "k": o.ksize, "s": o.stride, "cin": o.cin, "cout": o.cout,
                                   "down": o.down_out, "cfg": cfg, "n": 0, "ms": 0.0, "flop": fl, "bytes": by})
        e0["n"] += 1; e0["ms"] += ms
        e = per.setdefault(kernel_name(cfg, o.kind, dtype), [0, 0.0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
    name, (n, ms, fl, by) = max(per.items(), key=lambda kv: kv[1][1])
    # HBM traffic per launch of that kernel: not measurable from inside this process - taken from the committed
    # rocprofv3 --pmc passes of the SAME workload (FETCH_SIZE x2 + WRITE_SIZE), with their provenance
    traffic, tsrc = lookup_traffic(name, B, H, W, dtype)
    total_ms = sum(v[1] for v in per.values())
    conv_fl = sum(v[2] for k, v in per.items() if k.startswith("conv")); conv_ms = sum(v[1] for k, v in per.items() if k.startswith("conv"))
    if fl > 0:
        ach = fl / (ms * 1e-3) / 1e12
        rl = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
              "frac": round(ach / PEAK_TFLOPS[dtype], 4), "traffic": traffic, "traffic_source": tsrc, "launches": n,
              "avg_launch_ms": round(ms / n, 4), "flop_per_launch": fl / n, "algorithmic_bytes_per_launch": round(by / n),
              "timing": "HIP events around every launch, ops in order on one stream (profile mode keeps the head chains on the caller's stream)",
              "share_of_step_kernel_time": round(ms / total_ms, 3),
              "all_convs_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None}
    else:
        ach = by / (ms * 1e-3) / 1e9
        rl = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": n, "avg_launch_ms": round(ms / n, 4)}
    if profile_out:
        with open(profile_out, "w") as f:
            agg = {k: {"launches": v[0], "ms": v[1], "flop": v[2], "bytes": v[3], "tflops": (v[2] / (v[1] * 1e-3) / 1e12) if v[1] else 0,
                       "gbs": (v[3] / (v[1] * 1e-3) / 1e9) if v[1] else 0} for k, v in per.items()}
            layers = []
            for op in sorted(perop):
                e0 = perop[op]
                avg = e0["ms"] / e0["n"]
                e0.update(avg_ms=avg, tflops=e0["flop"] / (avg * 1e-3) / 1e12 if avg else 0, gbs=e0["bytes"] / (avg * 1e-3) / 1e9 if avg else 0)
                layers.append(e0)
            json.dump({"by_kernel": agg, "by_op": layers}, f, indent=1)
    return rl


def model_roofline(eng, B, H, W, dtype, ms_per_step):
    """layer-wise mixed roofline, t_min = sum over ops of max(flops / dense MFMA peak, compulsory bytes / HBM peak)."""
    t_min = 0.0
    for i in range(len(eng.prog.ops)):
        fl_i, by_i = eng.op_work(i, B, H, W)
        t_min += max(fl_i / (PEAK_TFLOPS[dtype] * 1e12), by_i / (HBM_PEAK_GBS * 1e9))
    return t_min * 1e3, t_min * 1e3 / ms_per_step


def main():
    args = parse()
    if (args.gpus > 1 or args.force_spawn) and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import numpy as np  # noqa: F401
    import torch
    from manual_yolo_amd import dist as mdist
    from manual_yolo_amd.engine import engine_from_weights
    from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
    rank, world, local = mdist.init_from_env()
    if world != args.gpus and rank == 0:
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
        B = args.batch or 64
    else:
        from manual_yolo_amd.ckpt import load_bundle
        sd, meta = load_bundle(os.path.join(ROOT, "tests", "golden", "rank_best.safetensors"))
        H = W = 64
        B = args.batch or 256
    frames_np = synth_frames(B, H, W, seed=1 + rank)
    frames = torch.from_numpy(frames_np).to(dev)
    quant = None
    if args.dtype == "f8" and args.quant_cache and os.path.exists(args.quant_cache):
        from manual_yolo_amd.quant import load_spec
        quant = load_spec(args.quant_cache)
    eng = engine_from_weights(sd, meta, args.dtype, local, bgr_input=False, quant=quant)
    if args.dtype == "f8" and args.quant_cache and quant is None and rank == 0:
        from manual_yolo_amd.quant import save_spec
        save_spec(eng.quant, args.quant_cache)

    def tune(e):
        if args.chunk:
            e.set_option("max_chunk", args.chunk)
        if args.conv_impl >= 0:
            e.set_option("conv_impl", args.conv_impl)
        if args.ablate:
            e.set_option("ablate", args.ablate)
        for kv in args.opt:
            k, v = kv.split("=")
            e.set_option(k, int(v))
    tune(eng)
    use_graph = args.graph if args.graph >= 0 else 0
    if use_graph:
        eng.set_option("graph", 1)
    # --opt nms_async=1: the per-image NMS of step k on the library's internal stream beside step k+1's backbone; whoever
    # consumes a step's detections waits for them.  Measured 8 109 vs 8 262 FPS (its 128-KiB-LDS workgroups cannot share a CU
    # with the conv kernels' and only delay them), so it stays off
    nms_async = task == "detect" and "nms_async=1" in args.opt
    max_det = 300
    gather = mdist.DetectionGather(B, max_det, dev) if task == "detect" else None
    kstep = [0]

    def step():
        if task == "detect":
            k = kstep[0]; kstep[0] += 1
            if k >= gather.depth:
                gather.wait(k - gather.depth)          # the slot's previous gather is complete before it is overwritten
            eng.detect(frames, 0.25, 0.7, False, max_det, None, want_anchor=False, out=gather.out_buffers(k), defer=True)
            gather.launch(k, ready=eng.wait_outputs)   # one message, on the side stream: overlaps the next batch
        else:
            eng.classify(frames)

    def drain():
        if task == "detect":
            for k in range(max(0, kstep[0] - gather.depth), kstep[0]):
                gather.wait(k)

    def barrier():
        drain()
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
    t_enq = time.perf_counter() - t0                   # host time to enqueue the K steps (Python + ctypes + launches)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_fps = [B * args.steps / elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        rank_fps = [B * args.steps / float(x.item()) for x in allt]
        elapsed = max(float(x.item()) for x in allt)
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        roofline = roofline_block(eng, step, args.steps, B, H, W, args.dtype, dev, args.profile_out if rank == 0 else "")
        drain()

    parity = exact = cpu = None
    if rank == 0 and task == "detect" and not args.no_parity:
        parity = parity_block(eng, sd, meta, frames_np, args.parity_frames)
        parity["dtype"] = args.dtype
    if rank == 0 and world == 1 and task == "detect" and args.dtype != "f32" and not args.no_exact_f32:
        # the mode that meets north_star's "identical box indices after NMS": exact-fp32 MFMA chain, same workload
        e32 = engine_from_weights(sd, meta, "f32", local, bgr_input=False)
        tune(e32)
        out32 = (torch.empty((B, max_det, 6), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev), None)
        s32 = lambda: e32.detect(frames, 0.25, 0.7, False, max_det, None, want_anchor=False, out=out32)  # noqa: E731
        for _ in range(2):
            s32()
        torch.cuda.synchronize(dev)
        n32 = max(3, min(args.steps, 8))
        t1 = time.perf_counter()
        for _ in range(n32):
            s32()
        torch.cuda.synchronize(dev)
        ms32 = (time.perf_counter() - t1) * 1e3 / n32
        rl32 = roofline_block(e32, s32, 3, B, H, W, "f32", dev) if not args.no_roofline else None
        tmin32, frac32 = model_roofline(e32, B, H, W, "f32", ms32)
        exact = {"dtype": "f32", "value": round(B / ms32 * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms32, 3), "steps": n32,
                 "model_roofline_frac": round(frac32, 4), "roofline": rl32}
        if not args.no_parity:
            p32 = parity_block(e32, sd, meta, frames_np, min(args.parity_frames, 4))
            exact["parity"] = {k: p32[k] for k in ("frames", "map50_95_delta", "kept_index_agreement", "frames_with_identical_kept_indices", "max_box_px", "max_score")}
        del e32
    cls2 = None
    if rank == 0 and world == 1 and task == "detect" and not args.no_classify:
        cls2 = classify_block(dev, local)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sd, meta, frames_np)
        cpu["gpu_over_cpu"] = round(value / cpu["value"], 1)

    if rank == 0:
        fl_step, by_step = eng.work(B, H, W)
        t_min_ms, mfrac = model_roofline(eng, B, H, W, args.dtype, ms_per_step)
        line = {
            "metric": ("frames/sec whole-node, yolov8m@640 batch=64; mAP delta vs CPU ref" if (H, B, args.scale) == (640, 64, "m")
                       else f"frames/sec whole-node, yolov8{args.scale}@{H} batch={B}; mAP delta vs CPU ref") if task == "detect"
                      else "images/sec, yolov8n-cls rank classifier 64x64",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"yolov8{args.scale} detect nc={args.nc} {H}x{W}, batch {B}/GPU, seeded random-init "
                                    f"weights, uint8 frames resident in HBM, NMS on-GPU (conf 0.25, iou 0.7, max_det 300)"
                                    + (", one RCCL all-gather of the padded detections per step on a side stream" if world > 1 else ""))
                                   if task == "detect" else f"yolov8n-cls rank_classifier weights 64x64, batch {B}/GPU",
                       "global_batch": world * B, "parallelism": f"dp{world}", "hip_graph": bool(use_graph), "nms_async": bool(nms_async and not use_graph),
                       "gflop_per_frame": round(fl_step / B / 1e9, 3), "algorithmic_mb_per_frame": round(by_step / B / 1e6, 2),
                       "model_tflops": round(fl_step * world / (ms_per_step * 1e-3) / 1e12, 2),
                       "model_t_min_ms": round(t_min_ms, 4), "model_roofline_frac": round(mfrac, 4),
                       "host_enqueue_ms_per_step": round(t_enq * 1e3 / args.steps, 3),
                       "per_rank_fps": [round(x, 1) for x in rank_fps]},
            "roofline": roofline, "parity": parity, "exact_f32": exact, "classify_config2": cls2, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

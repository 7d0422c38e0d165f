#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: faces/sec at batch=256 (256x256 BackCamera) per GPU, 1/2/4/8-GPU weak scaling.

One "step" = one pass of the hot path over one batch per GPU: BackCamera BlazeFace network + SSD decode + sigmoid +
weighted NMS + letterbox removal on 256 synthetic 256x256 frames (configs[1] of BASELINE.json),
inputs already resident in HBM.  One process per GPU; with N > 1 the frozen .tflite is broadcast from rank 0 over
RCCL (torch.distributed "nccl"), frames are sharded 256/GPU, and no data-path collective exists (frames are independent).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the C oracle on the host cores — a *port*: the reference's TFLite path cannot be built here).
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

F32_PEAK_TFLOPS = 157.3  # dense f32 (MFMA = packed VALU FMA), /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def make_frames(batch, seed, size=256):
    """SURVEY.md §8d config 2 frame mix: 50% U(-1,1) noise, 50% face-bearing (committed man-face tensor with a seeded
    +-32 px roll and 0.8-1.2 gain) so that NMS sees real candidates."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    face = (gold["man_back_u8"].astype(np.float64) * 2.0 / 255.0 - 1.0).astype(np.float32)
    rs = np.random.RandomState(seed)
    x = np.empty((batch, size, size, 3), np.float32)
    for b in range(batch):
        if b % 2 == 0:
            x[b] = rs.uniform(-1, 1, (size, size, 3)).astype(np.float32)
        else:
            dy, dx = int(rs.randint(-32, 33)), int(rs.randint(-32, 33))
            x[b] = np.clip(np.roll(face, (dy, dx), axis=(0, 1)) * np.float32(rs.uniform(0.8, 1.2)), -1, 1)
    return x


def cpu_baseline(x_host, threads):
    """The oracle (CPU port of the same graph + glue) on the GPU box's host cores, bounded sample."""
    from oracle import pyoracle as po
    om = po.Model(os.path.join(ROOT, "models", "face_detection_back.tflite"))
    anchors = po.ssd_anchors(po.FD_BACK)
    n = min(len(x_host), max(threads, 64))
    om.run(x_host[: min(n, threads)], nthreads=threads)  # warm (page-in, thread pool)
    t0 = time.time()
    frames = 0
    while time.time() - t0 < 10.0:
        rb, rs = om.run(x_host[:n], nthreads=threads)
        for f in range(n):
            po.fd_postprocess(rb[f], rs[f], anchors, 256.0)
        frames += n
    dt = time.time() - t0
    return {"value": round(frames / dt, 1), "unit": "faces/s", "cores": threads, "kind": "port",
            "sample": "%d frames of the same batch (net + decode + NMS), C oracle, OpenMP over frames, %.1f s" % (frames, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=None)
    ap.add_argument("--lanes", type=int, default=None)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import rs_face_detection_tflite_amd as mi
    import ctypes as C

    # ---- weights: rank 0 reads the frozen .tflite, every other rank receives it over RCCL/xGMI (one-time, timed apart)
    from importlib import import_module
    mdist = import_module("rs_face_detection_tflite_amd.dist")
    path = os.path.join(ROOT, "models", "face_detection_back.tflite") if rank == 0 else "/nonexistent"
    torch.cuda.synchronize()
    t0 = time.time()
    model_bytes = mdist.broadcast_model_bytes(path, dist, device)
    torch.cuda.synchronize()
    bcast_ms = (time.time() - t0) * 1e3 if world > 1 else 0.0
    L = mi.lib()
    h = C.c_void_p()
    rc = L.mi_fd_create_from_bytes(int(mi.FaceDetectionModel.BackCamera), model_bytes, len(model_bytes), local_rank, C.byref(h))
    if rc != 0:
        raise SystemExit("mi_fd_create_from_bytes failed: %s" % L.mi_last_error().decode())
    model = mi.Model(handle=L.mi_fd_model(h), owner=True)
    if args.fuse is not None:
        model.set_option("fuse", args.fuse)
    if args.chunk is not None:
        model.set_option("chunk", args.chunk)
    if args.lanes is not None:
        model.set_option("lanes", args.lanes)

    B, cap = args.batch, 16
    x_host = make_frames(B, seed=rank)
    x = torch.from_numpy(x_host).to(device)
    out = torch.zeros((B, cap, 17), dtype=torch.float32, device=device)
    counts = torch.zeros((B,), dtype=torch.int32, device=device)
    stream = torch.cuda.Stream(device=device)
    sp = C.c_void_p(stream.cuda_stream)

    def step():
        rc = L.mi_fd_infer_tensor(h, C.c_void_p(x.data_ptr()), B, None, C.c_void_p(out.data_ptr()), cap,
                                  C.c_void_p(counts.data_ptr()), mi.MI_MEM_DEVICE, sp)
        if rc != 0:
            raise RuntimeError(L.mi_last_error().decode())

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dist, device)
    n_faces = int((counts > 0).sum().item())

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events between launches on the launch stream (eager replays of the
        # same plan on the same resident inputs), grouped by kernel symbol like rocprofv3 --stats does.
        recs = model.profile(x, reps=5)
        by = {}
        for r in recs:
            k = by.setdefault(r["kernel"], {"ms": 0.0, "bytes": 0.0, "macs": 0.0, "calls": 0})
            k["ms"] += r["ms"]; k["bytes"] += r["bytes"]; k["macs"] += r["macs"]; k["calls"] += 1
        dom = max(by, key=lambda k: by[k]["ms"])
        d = by[dom]
        achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(tpath):
            try:
                pm = json.load(open(tpath))
                if pm.get("workload") == "back256_b%d" % B and pm.get("kernel") == dom:
                    traffic = pm.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # Which roof binds the dominant kernel: its algorithmic bytes against HBM, or its algorithmic FLOPs against the dense
        # f32 rate (157.3 TFLOP/s: v_mfma_f32_* and v_pk_fma_f32 have the same peak on gfx950; the row-pipelined chain
        # kernels keep the intermediate rows of up to 4 layers in LDS, so their launches are FLOP-bound, not HBM-bound).
        tflops = 2 * d["macs"] / (d["ms"] * 1e-3) / 1e12
        hbm_frac, flop_frac = achieved / HBM_PEAK_GBS, tflops / F32_PEAK_TFLOPS
        common = {"kernel": dom, "launches_per_step": d["calls"], "avg_launch_ms": round(d["ms"] / d["calls"], 5),
                  "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"]), "algorithmic_flops_per_launch": round(2 * d["macs"] / d["calls"]),
                  "hbm_GBps": round(achieved, 1), "hbm_frac": round(hbm_frac, 4), "f32_TFLOPs": round(tflops, 2), "f32_frac": round(flop_frac, 4)}
        if hbm_frac >= flop_frac:
            roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4), "traffic": traffic}
        else:
            roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop_frac, 4), "traffic": traffic}
        roofline.update(common)
        roofline["net_event_ms"] = round(sum(r["ms"] for r in recs), 4)
        roofline["kernels"] = {k: {"ms": round(v["ms"], 4), "calls": v["calls"], "GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                   "TFLOPs": round(2 * v["macs"] / max(v["ms"], 1e-9) / 1e9, 2)} for k, v in by.items()}
        value = world * B * args.steps / elapsed
        result = {
            "metric": "faces/sec at batch=256 (256x256 back-camera) per GPU", "value": round(value, 1), "unit": "faces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BackCamera BlazeFace 256x256, batch=256 frames/GPU, net + SSD decode + sigmoid + weighted NMS "
                                   "(configs[1]); 50% noise / 50% face-bearing frames, inputs resident in HBM",
                       "global_batch": world * B, "frames_with_faces": n_faces, "parallelism": "frames sharded %d/GPU, no data-path collective" % B,
                       "weight_broadcast_ms": round(bcast_ms, 3), "plan": model.describe().splitlines()[0]},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:  # CPU baseline: rank 0 at N = 1 only
            threads = min(os.cpu_count() or 1, 64)
            result["cpu_baseline"] = cpu_baseline(x_host, threads)
    L.mi_fd_free(h)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

"""N>1 path on CPU: world_size-2 gloo run of the same sharding / weight-broadcast / timing logic bench.py uses with
RCCL on the GPUs (frames shard, the only exchange is the one-time model broadcast)."""
import hashlib
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, model_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import rs_face_detection_tflite_amd as mi
    from importlib import import_module
    d = import_module("rs_face_detection_tflite_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # rank 1 gets the bytes only through the broadcast (it is handed a path that does not exist)
    path = model_path("back") if rank == 0 else "/nonexistent/model.tflite"
    blob = d.broadcast_model_bytes(path, dist, dev)
    plan = mi.plan_describe(blob, 2)            # host-only lowering of the received bytes
    lo, hi = d.shard_range(513, world, rank)
    counts = torch.full((4,), rank, dtype=torch.int32)
    gathered = d.gather_counts(counts, dist, dev)
    t = d.max_over_ranks(1.0 + rank, dist, dev)
    q.put((rank, hashlib.md5(blob).hexdigest(), hashlib.md5(plan.encode()).hexdigest(), lo, hi, t, [int(g[0]) for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_broadcast_and_sharding():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = hashlib.md5(open(model_path("back"), "rb").read()).hexdigest()
    assert res[0][1] == res[1][1] == ref              # both ranks hold the frozen model bytes
    assert res[0][2] == res[1][2]                     # and lower them to the same launch plan
    assert (res[0][3], res[0][4]) == (0, 257) and (res[1][3], res[1][4]) == (257, 513)   # contiguous, disjoint, complete
    assert res[0][5] == res[1][5] == 2.0              # max over ranks
    assert res[0][6] == res[1][6] == [0, 1]


def test_shard_range_properties():
    import sys
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import rs_face_detection_tflite_amd  # noqa: F401
    d = import_module("rs_face_detection_tflite_amd.dist")
    for total in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [d.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no external launcher starts two rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1) and rank 0 prints the one JSON line with n_gpus = 2.  Here as a CPU rehearsal
    (--rehearse: gloo, the weight broadcast + host-side lowering + sharding + max-over-ranks timing, no kernels)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    for cfg, nfiles in ((2, 1), (5, 3)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "3", "--config", str(cfg)],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "torch imported in the launcher: False" in r.stderr      # the parent never loads torch (so it cannot touch HIP)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1                                     # exactly one JSON line, from rank 0
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["value"] is None and out["scaling"] == "weak"
        B = {2: 256, 5: 128}[cfg]
        assert out["config"]["shards"] == [[0, B], [B, 2 * B]] and out["config"]["plans_identical"] is True
        assert out["ms_per_step"] >= 2.0                           # max over ranks: rank 1 sleeps 2 ms per step
    # fewer GPUs visible than asked for (none here): refuse instead of running a smaller job under the label
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "visible" in r.stderr
        assert "torch imported in the launcher: False" in r.stderr or "[rank" not in r.stderr   # refused by sysfs count or by the ranks
    # a launcher that sets WORLD_SIZE differently from --gpus is an error, not a silent 1-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_world8_rehearsal_with_result_gather():
    """The launcher at the size of a whole node (8 ranks, BASELINE configs[3] and configs[4]): every rank receives the frozen graph(s)
    by broadcast and lowers them to the same plan, shards are contiguous / disjoint / complete, rank r would bind cuda:LOCAL_RANK = r
    (one distinct device per rank), the optional result collection of SURVEY.md 8e (`--gather`: dist.gather_counts, an all_gather
    of fixed-size records) returns the shards in rank order, and rank 0 prints one JSON line.  gloo on CPU — no run on more than one
    GPU exists."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    for cfg, B in ((2, 256), (5, 128)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse", "--gather", "--steps", "2", "--config", str(cfg)],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        out = json.loads(lines[0])
        c = out["config"]
        assert out["n_gpus"] == 8 and out["rehearsal"] is True and c["global_batch"] == 8 * B
        assert c["shards"] == [[k * B, (k + 1) * B] for k in range(8)]
        assert c["plans_identical"] is True
        assert c["rank_to_device"] == ["rank %d -> cuda:%d" % (k, k) for k in range(8)]
        assert c["gathered_counts_in_rank_order"] is True
        assert out["ms_per_step"] >= 8.0                            # max over ranks: rank 7 sleeps 8 ms per step


def test_launcher_parent_stays_clear_of_torch_and_hip():
    """The process that starts the ranks must not have initialised the GPU: it may not even import torch (whose device count can
    fall through to hipGetDeviceCount).  Importing bench.py and running its launcher (one rehearsal rank through --spawn) leaves
    torch, numpy and ctypes-loaded HIP libraries out of the parent."""
    import subprocess
    import sys
    code = ("import sys, argparse; sys.path.insert(0, %r); import bench; "
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules; "
            "n = bench.kfd_gpu_nodes(); assert n is None or n >= 0; "
            "a = argparse.Namespace(gpus=1, rehearse=True); "
            "rc = bench.launch_ranks(a, ['--gpus', '1', '--rehearse', '--spawn', '--steps', '2']); "
            "assert rc == 0, rc; assert 'torch' not in sys.modules and 'numpy' not in sys.modules; "
            "maps = open('/proc/self/maps').read(); assert 'libamdhip64' not in maps and 'libhsa-runtime' not in maps and 'libamd_smi' not in maps; "
            "print('clean')") % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr[-2000:]
    assert '"n_gpus": 1' in r.stdout


def test_secondary_configs_run_before_the_gpu_is_touched_and_flag_failures():
    """the child runs of configs 1 / 3 / 5 behind the headline line are started by bench.py's main() while the
    process has not imported torch nor made a HIP call, and a child that fails leaves an explicit error record (the line's top-level
    `secondary_errors` counts them).  Without a GPU every child fails at its device check (status 2): three error records, and the
    parent still has not loaded torch."""
    import subprocess, sys, json
    code = ("import sys, json; sys.path.insert(0, %r); import bench, argparse\n"
            "a = argparse.Namespace(steps=1, warmup=0)\n"
            "r = bench.run_secondary_configs(a)\n"
            "print(json.dumps({'r': r, 'torch': 'torch' in sys.modules}))\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-500:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["torch"] is False
    assert sorted(d["r"]) == ["config 1", "config 3", "config 5"]
    for v in d["r"].values():
        assert "error" in v and "status" in v["error"], v

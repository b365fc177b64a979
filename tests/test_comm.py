"""The native gather path (csrc/pt_comm.cpp: pt_comm_* / pt_gather_shards / pt_gather_finish) and bench.py's self-launching N-GPU mode.

No GPU: what happens without librccl.so (every call answers PT_ERR_UNAVAILABLE with a message -- no crash, no half-open RCCL group) and how
`bench.py --gpus N` fails on a host without devices (pt_create's message, not an argument error).
GPU (1 device): the single-process flavour (pt_comm_init_all + pt_comm_group_begin / end) end to end with ndev = 1, the root-only gather
buffer, and `bench.py --gpus 2` failing with pt_create's device-count message."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from vk_raytrace_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_MISSING = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from vk_raytrace_amd import capi, shard
L = capi.lib()
ident = (C.c_ubyte * 128)()
rc = L.pt_comm_get_unique_id(ident)
assert rc == capi.PT_ERR_UNAVAILABLE, rc
msg = L.pt_comm_last_error().decode()
assert "librccl" in msg and len(msg) > len("librccl.so not found: "), msg
comm = C.c_void_p()
assert L.pt_comm_init_rank(1, ident, 0, 0, C.byref(comm)) == capi.PT_ERR_UNAVAILABLE
assert L.pt_comm_init_all(1, None, C.byref(comm)) == capi.PT_ERR_UNAVAILABLE
assert L.pt_comm_group_begin() == capi.PT_ERR_UNAVAILABLE and L.pt_comm_group_end() == capi.PT_ERR_UNAVAILABLE
assert L.pt_comm_get_unique_id(None) == capi.PT_ERR_INVALID          # a bad argument stays distinguishable from a missing library
try:
    shard.NativeGather(0, 1, 0)
    raise SystemExit("NativeGather did not raise")
except capi.PtError as e:
    assert e.code == capi.PT_ERR_UNAVAILABLE and "librccl" in str(e), e
print("ok")
"""


def _last_json(text):
    """the bench line: the last line of stdout that is a JSON object (RCCL prints a version banner through C stdio)"""
    import json
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _visible_devices():
    """HIP devices this process can create contexts on, asked through libptmi itself (torch.cuda must not be initialised next to it: capi.lib())"""
    L, n = capi.lib(), 0
    while n < 64:
        ctx = C.c_void_p()
        if L.pt_create(n, C.byref(ctx)) != capi.PT_OK:
            break
        L.pt_destroy(ctx)
        n += 1
    return n


def test_missing_rccl_is_an_error_code_not_a_crash():
    env = dict(os.environ, PT_RCCL_LIB="/nonexistent/librccl.so.1")
    p = subprocess.run([sys.executable, "-c", _MISSING % ROOT], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.returncode, p.stdout, p.stderr)


def test_bench_self_launch_fails_with_the_device_message_without_devices():
    """`python bench.py --gpus 2` needs no launcher; here (no GPU) both ranks fail in pt_create and the parent reports it and stops."""
    if _visible_devices() > 0:
        pytest.skip("a GPU is present: covered by test_bench_gpus_2_on_a_one_gpu_box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "no HIP device available" in p.stderr and "exited with code" in p.stderr, p.stderr[-2000:]
    assert "must be launched through" not in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_on_a_one_gpu_box():
    if _visible_devices() != 1:
        pytest.skip("needs exactly one visible device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "device ordinal 1 out of range: 1 HIP device(s) visible" in p.stderr, p.stderr[-2000:]


@pytest.mark.gpu
def test_bench_two_processes_share_one_gpu():
    """The multi-process flow of bench.py for real -- two self-launched ranks, the torch-free rendezvous, the gather route agreed in a preflight, barriers
    around every timed window, the MAX of the ranks' times, counters summed over the ranks, and BOTH halves of the metric in the N > 1 line: the image
    gathered from the two ranks after the first window is bit-identical to the reference's shader compiled for the host (parity.l2 == 0).  Both ranks sit
    on device 0 (PT_BENCH_SAME_DEVICE=1); RCCL refuses two ranks on one device, so the preflight settles on the host route -- which is also what a run
    falls back to when RCCL cannot be brought up -- and the RCCL route is left to the one-rank tests below and to the driver's 8-GPU run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PT_BENCH_SAME_DEVICE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--width", "256", "--height", "160", "--tex-size", "32",
           "--tris", "2000", "--no-profile", "--no-interactive", "--cpu-seconds", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and len(line["repeats"]) == 2 and line["value"] > 0
    assert line["scaling"] == "strong" and line["launch"] == "one process per GPU" and line["gather"].startswith("host")
    assert len(line["ms_per_step_per_rank"]) == 2 and "preflight" in p.stderr
    par = line["parity"]
    assert par["l2"] == 0.0 and par["pixels_bit_identical"] == par["pixels"] > 0 and par["frames"] == 4 and "GATHERED from 2 ranks" in par["note"]
    assert line["cpu_baseline"]["value"] > 0
    one = subprocess.run(cmd[:3] + ["1"] + cmd[4:] + ["--no-cpu-baseline"], env={k: v for k, v in env.items() if k != "PT_BENCH_SAME_DEVICE"}, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = _last_json(one.stdout)
    for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests"):
        assert line["rays"][k] == ref["rays"][k], k   # the two shards together trace exactly the rays of the whole image
    assert abs(line["image_mean"] - ref["image_mean"]) < 1e-6


@pytest.mark.gpu
def test_bench_single_process_flavour():
    """`bench.py --single-process`: one process, N contexts, pt_comm_init_all, the shards gathered inside one RCCL group -- with N = 1 on this box (everything
    but the peer-to-peer transfers runs); the line says how it was launched and carries the parity of the gathered image.  `--gpus 2` falls back to this
    flavour when its self-launched ranks fail; on a one-GPU box the fallback then fails for the same reason and says so."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-process", "--steps", "3", "--warmup", "1", "--repeats", "2", "--width", "256", "--height", "160",
           "--tex-size", "32", "--tris", "2000", "--no-profile", "--no-interactive", "--cpu-seconds", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["launch"] == "single-process" and line["ranks_seen"] == 1 and line["gather"].startswith("rccl") and "RCCL" in p.stderr
    assert line["parity"]["l2"] == 0.0 and line["parity"]["pixels_bit_identical"] == line["parity"]["pixels"] > 0
    if _visible_devices() == 1:
        p = subprocess.run(cmd[:2] + ["--gpus", "2"] + cmd[5:], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode != 0 and "falling back to --single-process" in p.stderr and "device ordinal 1 out of range" in p.stderr, p.stderr[-2000:]


@pytest.mark.gpu
def test_single_process_gather_with_one_device():
    """pt_comm_init_all + pt_comm_group_begin / pt_gather_shards / pt_comm_group_end + pt_gather_finish: the flavour one process driving N GPUs
    uses, here with N = 1 (everything but the peer-to-peer transfers runs).  The gathered image equals the plain read-back."""
    from tests.common import Config, render_hip
    from vk_raytrace_amd import synth
    L = capi.lib()
    cfg = Config(synth.feature_box(tex_size=32), synth.procedural_sky(128, 64), 200, 120)
    img, r = render_hip(cfg, 3, return_obj=True)
    # the one-process-per-GPU flavour with a single rank first (what bench.py's N > 1 ranks do): unique id -> pt_comm_init_rank
    from vk_raytrace_amd import shard
    g = shard.NativeGather(0, 1, 0)
    assert g.ranks_seen() == 1
    got = g.gather(r)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))
    g.close()
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    rc = L.pt_comm_init_all(1, dev, C.byref(comm))
    assert rc == capi.PT_OK, L.pt_comm_last_error()
    n = C.c_int(0)
    assert L.pt_comm_count(comm, C.byref(n)) == capi.PT_OK and n.value == 1
    assert L.pt_gather_finish(r._ctx) == capi.PT_ERR_STATE           # nothing enqueued yet
    assert L.pt_gather_shards(r._ctx, comm, 1) == capi.PT_ERR_INVALID  # root out of range: rejected before anything is allocated or grouped
    assert L.pt_comm_group_begin() == capi.PT_OK
    assert L.pt_gather_shards(r._ctx, comm, 0) == capi.PT_OK
    assert L.pt_comm_group_end() == capi.PT_OK
    assert L.pt_gather_finish(r._ctx) == capi.PT_OK
    assert L.pt_gather_finish(r._ctx) == capi.PT_ERR_STATE           # consumed
    got = r.read_accum()
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))
    assert L.pt_comm_destroy(comm) == capi.PT_OK
    r.destroy()
    # two ranks that the context was not sharded for: the communicator check answers, and the RCCL group is closed again (a later call works)
    comm = C.c_void_p()
    assert L.pt_comm_init_all(1, dev, C.byref(comm)) == capi.PT_OK
    img2, r2 = render_hip(cfg, 1, shard=(1, 2), return_obj=True)
    assert L.pt_gather_shards(r2._ctx, comm, 0) == capi.PT_ERR_INVALID
    assert L.pt_comm_group_begin() == capi.PT_OK and L.pt_comm_group_end() == capi.PT_OK
    assert L.pt_comm_destroy(comm) == capi.PT_OK
    r2.destroy()


_GROUP = r"""
import sys
sys.path.insert(0, %r)
from vk_raytrace_amd.rendezvous import LocalGroup
rank, world = int(sys.argv[1]), int(sys.argv[2])
g = LocalGroup(rank, world, key=sys.argv[3], timeout=60)
g.barrier()
assert g.all_reduce([rank + 1.0, -rank], "max") == [float(world), 0.0]
assert g.all_reduce([rank + 1.0, 2.0], "sum") == [world * (world + 1) / 2.0, 2.0 * world]
blob = bytes(range(128)) if rank == 1 %% world else b"junk"
assert g.broadcast_bytes(blob, 1 %% world) == bytes(range(128))
parts = g.gather_bytes(bytes([rank]) * (rank + 1))   # the host route of bench.py's gather: every rank's shard on rank 0, in rank order
assert parts == ([bytes([r]) * (r + 1) for r in range(world)] if rank == 0 else None)
for _ in range(50):
    g.barrier()
g.close()
print("ok", rank)
"""


@pytest.mark.parametrize("world", [1, 3])
def test_local_group_control_plane(world, tmp_path):
    """vk_raytrace_amd/rendezvous.py: the torch-free control plane of bench.py's ranks (barrier, MAX / SUM all-reduce, byte broadcast, byte gather)."""
    procs = [subprocess.Popen([sys.executable, "-c", _GROUP % ROOT, str(r), str(world), f"test{os.getpid()}_{world}"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in reversed(range(world))]  # rank 0 (the server) starts last: the peers wait for it
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and out.startswith("ok"), (out, err)


def test_bench_watchdog_and_stdout_contract():
    """bench.py's guard around the RCCL calls of the preflight (a collective that never returns must not take the line with it) and the promise that the
    JSON line is the LAST thing on stdout (RCCL prints a version banner through C stdio, which a pipe holds back until exit)."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ok, res = bench.with_timeout(lambda: 41 + 1, 5, "quick")
    assert ok and res == 42
    ok, res = bench.with_timeout(lambda: 1 / 0, 5, "raises")
    assert not ok and "ZeroDivisionError" in res
    t0 = time.monotonic()
    ok, res = bench.with_timeout(lambda: time.sleep(30), 0.3, "pt_gather_shards")
    assert not ok and "pt_gather_shards: no answer" in res and time.monotonic() - t0 < 5
    code = ("import ctypes, json, os, sys\n"
            "libc = ctypes.CDLL(None)\n"
            "libc.printf(b'RCCL version : banner held back by C stdio\\n')\n"      # what librccl does at init
            "ctypes.CDLL(None).fflush(None)\n"                                      # bench.py, before the line
            "print(json.dumps({'value': 1})); sys.stdout.flush()\n"
            "os.dup2(os.open(os.devnull, os.O_WRONLY), 1)\n"                        # bench.py, after the line
            "libc.printf(b'late noise\\n')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == '{"value": 1}', (out.stdout, out.stderr)

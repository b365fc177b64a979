"""N > 1 path on CPU: two gloo ranks shard the image tiles exactly like the GPU ranks do
(vk_raytrace_amd/shard.py), render their own pixels, gather on rank 0 and assemble.  The result must be
bit-identical to a single-process render: seeds depend on the global pixel index only
(shaders/pathtrace.comp:97).  Two flavours of "render": the oracle standing in for the HIP path, and the
PRODUCT's own source -- the per-path functions the HIP kernels are made of, compiled for the host
(tests/test_trace_host.py host_render) with the device's tile / slot layout of the rank's shard."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from tests import orc
    from tests.common import Config
    from vk_raytrace_amd import synth, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 100, 70   # not a multiple of the tile size: partial edge tiles
    cfg = Config(synth.feature_box(tex_size=32), synth.procedural_sky(64, 32), W, H)
    o = orc.Oracle(threads=2)
    o.set_scene(cfg.scene); integ, _ = o.set_env(cfg.env); o.set_camera(cfg.camera); o.set_sunsky(cfg.sunsky)
    st = cfg.state(integ)
    acc = np.zeros((H, W, 4), np.float32)
    ids = shard.local_pixel_ids(W, H, rank, world)
    for f in range(3):
        st.frame = f
        o.render_frame(st, acc, ids)
    t = torch.from_numpy(acc)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)          # the single collective of the path
    if rank == 0:
        full = shard.assemble_rowmajor([b.numpy() for b in bufs], W, H)
        ref = np.zeros((H, W, 4), np.float32)
        for f in range(3):
            st.frame = f
            o.render_frame(st, ref)
        np.save(out_path, np.stack([full, ref]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_is_bit_identical(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    full, ref = np.load(out)
    assert np.array_equal(full, ref)
    assert full[..., 3].min() == 1.0     # every pixel was written by exactly one rank


def _worker_product(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    from tests.common import Config, render_oracle
    from tests.test_trace_host import host_render
    from vk_raytrace_amd import synth, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 100, 70   # not a multiple of the tile size: partial edge tiles
    cfg = Config(synth.feature_box(tex_size=32), synth.procedural_sky(64, 32), W, H, depth=5)
    acc = host_render(cfg, 3, two=rank % 2, shard=(rank, world))   # (one rank on the flat, one on the two-level structure: same pixels)
    t = torch.from_numpy(acc)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)          # the single collective of the path
    if rank == 0:
        full = shard.assemble_rowmajor([b.numpy() for b in bufs], W, H)
        ids = [shard.local_pixel_ids(W, H, r, world) for r in range(world)]
        own = np.zeros(W * H, bool); own[ids[0]] = True
        foreign = bufs[0].numpy().reshape(-1, 4)[~own]
        np.save(out_path, np.stack([full, render_oracle(cfg, 3), np.broadcast_to(np.float32(np.abs(foreign).max()), full.shape)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_of_the_products_host_build(tmp_path):
    import torch.multiprocessing as mp
    from tests.test_trace_host import harness
    harness()   # build once, before the ranks race for it
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker_product, args=(2, _free_port(), out), nprocs=2, join=True)
    full, ref, foreign = np.load(out)
    assert np.array_equal(full.view(np.uint32), ref.view(np.uint32))   # == the single-process oracle, bit for bit
    assert foreign.max() == 0.0                                        # a rank writes nothing outside its own tiles

"""N > 1 path on CPU: two gloo ranks shard the image tiles exactly like the GPU ranks do
(vk_raytrace_amd/shard.py), render their own pixels (here with the oracle standing in for the HIP path),
gather on rank 0 and assemble.  The result must be bit-identical to a single-process render: seeds depend
on the global pixel index only (shaders/pathtrace.comp:97)."""
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

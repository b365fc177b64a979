"""Image-tile sharding across the GPUs of one node (no reference counterpart).

One process per GPU.  The scene, BVH, textures and environment are replicated; the image is cut into
PT_TILE x PT_TILE tiles and rank r renders the tiles with (tx + ty) % nranks == r (diagonal
interleave: sky and interior tiles are spread evenly).  Seeds depend on the global pixel index only
(shaders/pathtrace.comp:97), so every pixel is bit-identical to a single-GPU render.  Nothing is
exchanged while rendering; at the end ONE gather of the accumulated framebuffer shards goes to rank 0
(RCCL over xGMI, natively behind the C ABI: each peer's shard travels on its own point-to-point link).
"""
import numpy as np

from . import host_device as hd

TILE = hd.TILE


def tiles_of_rank(width, height, rank, nranks):
    """Global tile ids owned by `rank`, in increasing order (the layout of pt_local_shard)."""
    tx_n = (width + TILE - 1) // TILE
    ty_n = (height + TILE - 1) // TILE
    return [ty * tx_n + tx for ty in range(ty_n) for tx in range(tx_n) if (tx + ty) % nranks == rank]


def max_tiles_per_rank(width, height, nranks):
    return max(len(tiles_of_rank(width, height, r, nranks)) for r in range(nranks))


def local_pixel_ids(width, height, rank, nranks):
    """Row-major pixel ids (y*width + x) of the pixels `rank` owns."""
    tx_n = (width + TILE - 1) // TILE
    ids = []
    for t in tiles_of_rank(width, height, rank, nranks):
        tx, ty = t % tx_n, t // tx_n
        xs = np.arange(tx * TILE, min((tx + 1) * TILE, width))
        ys = np.arange(ty * TILE, min((ty + 1) * TILE, height))
        ids.append((ys[:, None] * width + xs[None, :]).reshape(-1))
    return np.concatenate(ids).astype(np.uint32) if ids else np.zeros(0, np.uint32)


class NativeGather:
    """The gather through libptmi's own RCCL path (pt_comm_* / pt_gather_shards / pt_gather_finish, csrc/pt_comm.cpp) -- what a C++ host
    calls.  The group (vk_raytrace_amd/rendezvous.py) is used for nothing but handing rank 0's ncclUniqueId to the other processes; no
    torch is involved (rendezvous.py explains why it must not be)."""

    def __init__(self, rank, nranks, device_ordinal, group=None):
        """group: a rendezvous.LocalGroup (or anything with broadcast_bytes(data, src)) that reaches every rank; only needed for nranks > 1"""
        import ctypes as C
        import struct
        from . import capi
        self._C, self._capi, self._lib = C, capi, capi.lib()
        self.rank, self.nranks = rank, nranks
        self.comm = None
        if nranks > 1 and group is None:
            raise ValueError("more than one rank needs a group to distribute the ncclUniqueId")
        ident = (C.c_ubyte * 128)()
        rc = self._lib.pt_comm_get_unique_id(ident) if rank == 0 else capi.PT_OK
        # rank 0's status travels with the id: if it failed (librccl.so missing, ...) EVERY rank raises instead of waiting for the others
        msg = struct.pack("<i", rc) + bytes(ident) + (self._lib.pt_comm_last_error() if rc != capi.PT_OK else b"")
        if nranks > 1:
            msg = group.broadcast_bytes(msg, 0)
        rc, raw, why = struct.unpack("<i", msg[:4])[0], msg[4:132], msg[132:].decode(errors="replace")
        if rc != capi.PT_OK:
            raise capi.PtError(rc, f"pt_comm_get_unique_id on rank 0: {why}")
        ident = (C.c_ubyte * 128).from_buffer_copy(raw)
        comm = C.c_void_p()
        rc = self._lib.pt_comm_init_rank(nranks, ident, rank, device_ordinal, C.byref(comm))
        if rc != capi.PT_OK:
            raise capi.PtError(rc, "pt_comm_init_rank: " + self._lib.pt_comm_last_error().decode())
        self.comm = comm

    def ranks_seen(self):
        """ncclCommCount of the communicator: the number of processes that really met."""
        n = self._C.c_int(0)
        rc = self._lib.pt_comm_count(self.comm, self._C.byref(n))
        if rc != self._capi.PT_OK:
            raise self._capi.PtError(rc, "pt_comm_count: " + self._lib.pt_comm_last_error().decode())
        return n.value

    def gather(self, renderer):
        """Full RGBA32F image on rank 0, None elsewhere."""
        renderer._check(self._lib.pt_gather_shards(renderer._ctx, self.comm, 0))
        if self.rank != 0:
            renderer.synchronize()
            return None
        renderer._check(self._lib.pt_gather_finish(renderer._ctx))
        return renderer.read_accum()

    def close(self):
        if self.comm:
            self._lib.pt_comm_destroy(self.comm)
            self.comm = None


def assemble_rowmajor(shards, width, height):
    """Host-side assembly used on the CPU (gloo) path: shards[r] is rank r's row-major image in which
    only its own pixels are valid."""
    out = np.zeros((height, width, 4), np.float32)
    flat = out.reshape(-1, 4)
    n = len(shards)
    for r, s in enumerate(shards):
        ids = local_pixel_ids(width, height, r, n)
        flat[ids] = np.asarray(s, np.float32).reshape(-1, 4)[ids]
    return out

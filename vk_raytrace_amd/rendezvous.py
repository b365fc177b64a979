"""Control plane of a one-node multi-GPU run without torch: the ranks of one launcher (torch.distributed.run, or bench.py's own self-launch)
meet on a Unix-domain socket and get a barrier, small all-reduces and a byte broadcast -- everything bench.py needs around the timed region
and for handing out the ncclUniqueId of libptmi's native RCCL gather.

Why not torch.distributed: merely importing the PyTorch-ROCm wheel loads its bundled HIP / HSA / RCCL libraries into the process, after which the
system RCCL that libptmi opens (linked, like libptmi itself, against /opt/rocm's HIP runtime) fails to initialise ("unhandled cuda error",
measured: gpurun_out/r03f).  A rank therefore imports nothing of torch; the launcher may still be torchrun -- it only sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, which is all that is read here.  The data path never passes through this module:
the one collective of the path is pt_gather_shards (RCCL over xGMI, csrc/pt_comm.cpp).
"""
import os
import socket
import struct
import tempfile
import time


class LocalGroup:
    """Ranks 0 .. world-1 of one node.  `key` distinguishes concurrent jobs (default: MASTER_PORT and the launcher's PID, which every rank of a
    launcher shares).  Rank 0 serves; every collective is one message from each peer to rank 0 and one reply."""

    def __init__(self, rank, world, key=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        if key is None:
            key = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self.path = os.path.join(self._private_dir(), f"{key}.sock")
        self.peers = {}
        self.sock = None
        if self.world == 1:
            return
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            try:
                os.unlink(self.path)  # inside a directory only this user can write: a leftover of an earlier job of ours, nothing else
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                srv.bind(self.path)
                srv.listen(self.world)
                srv.settimeout(timeout)
                while len(self.peers) < self.world - 1:
                    c, _ = srv.accept()
                    c.settimeout(timeout)
                    if not self._same_user(c):
                        c.close()
                        continue
                    r = struct.unpack("<i", self._recv(c, 4))[0]
                    if not (0 < r < self.world) or r in self.peers:
                        c.close()
                        raise ConnectionError(f"rendezvous: a peer announced rank {r} (world {self.world}, already seen: {sorted(self.peers)})")
                    self.peers[r] = c
            finally:
                srv.close()
                try:
                    os.unlink(self.path)
                except FileNotFoundError:
                    pass
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank {self.rank}: rank 0 never opened {self.path}")
                    time.sleep(0.01)
            s.settimeout(timeout)
            s.sendall(struct.pack("<i", self.rank))
            self.sock = s

    @staticmethod
    def _private_dir():
        """$XDG_RUNTIME_DIR (per user, 0700 by specification) or a 0700 directory of this user under the temp dir: nobody else can pre-bind or
        replace the socket path"""
        base = os.environ.get("XDG_RUNTIME_DIR")
        if base and os.path.isdir(base) and os.access(base, os.W_OK):
            return base
        d = os.path.join(tempfile.gettempdir(), f"ptmi_rdzv_{os.getuid()}")
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(d)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError(f"{d} exists and is not a private directory of uid {os.getuid()}")
        return d

    @staticmethod
    def _same_user(conn):
        """SO_PEERCRED: the connecting process runs under this user's uid (Linux)"""
        try:
            pid, uid, gid = struct.unpack("3i", conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
            return uid == os.getuid()
        except (OSError, AttributeError):
            return True  # no peer credentials on this platform: the private directory is the protection

    @staticmethod
    def _recv(s, n):
        buf = b""
        while len(buf) < n:
            part = s.recv(n - len(buf))
            if not part:
                raise ConnectionError("a rank left the group")
            buf += part
        return buf

    def _exchange(self, payload, combine):
        """every rank contributes `payload` (bytes); rank 0 combines the world's payloads (list in rank order) into the reply everyone gets"""
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [None] * (self.world - 1)
            for r, c in self.peers.items():
                n = struct.unpack("<I", self._recv(c, 4))[0]
                parts[r] = self._recv(c, n)
            out = combine(parts)
            for c in self.peers.values():
                c.sendall(struct.pack("<I", len(out)) + out)
            return out
        self.sock.sendall(struct.pack("<I", len(payload)) + payload)
        n = struct.unpack("<I", self._recv(self.sock, 4))[0]
        return self._recv(self.sock, n)

    def barrier(self):
        self._exchange(b"", lambda parts: b"")

    def all_reduce(self, values, op="sum"):
        """element-wise over the ranks' equally long lists of floats; op: "sum" | "max" """
        vals = [float(v) for v in values]
        fmt = f"<{len(vals)}d"

        def combine(parts):
            rows = [struct.unpack(fmt, p) for p in parts]
            f = max if op == "max" else sum
            return struct.pack(fmt, *[f(col) for col in zip(*rows)])
        return list(struct.unpack(fmt, self._exchange(struct.pack(fmt, *vals), combine)))

    def broadcast_bytes(self, data, src=0):
        """rank `src`'s bytes on every rank (the others pass anything)"""
        return self._exchange(bytes(data) if self.rank == src else b"", lambda parts: parts[src])

    def gather_bytes(self, data):
        """every rank's bytes on rank 0, as a list in rank order (None on the other ranks)"""
        got = []

        def combine(parts):
            got.extend(parts)
            return b""
        self._exchange(bytes(data), combine)
        return got if self.rank == 0 else None

    def close(self):
        for c in self.peers.values():
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = {}, None

"""Ensemble axis across GPUs: partition independent problem instances over ranks, share the
sensing matrix (or its thin factors) with ONE broadcast at setup, gather the per-instance results
at the end.  The reference runs the same ensemble as a sequential Python
loop (tramp/experiments/multiple_experiments.py:30-49).

Instances never interact, so there is no per-iteration collective and no cross-GPU reduction.
One process per GPU; the collectives are RCCL behind the C ABI (csrc/comm.hip:
tramp_comm_init / tramp_bcast_matrix / tramp_gather_scalars / tramp_comm_max) - no PyTorch.
The 128-byte RCCL unique id travels over a small TCP star on MASTER_ADDR (`Rendezvous`), which
also backs `SocketComm`: the same API on host arrays, used by the world_size-2 CPU tests.
"""
import ctypes as C
import os
import socket
import struct
import subprocess
import sys
import time

import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [start, stop) of `total` instances owned by `rank` (sizes differ by <= 1)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def instance_seeds(base_seed, start, stop):
    "Per-instance RNG seeds: an instance draws the same data whatever the number of ranks."
    return [int(base_seed) + 1000003 * i for i in range(start, stop)]


# ------------------------------------------------------------------ rendezvous (host side)
_MAGIC = b"TRAMPRDV1"
_PORT_STEPS = (17, 117, 217, 317, 417, 517)


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class Rendezvous:
    """TCP star on rank 0.  `exchange(payload)` is an all-gather of byte strings and the only
    primitive; every rank must call it the same number of times in the same order.

    The port is MASTER_PORT + a small offset (the launcher's own store may sit on MASTER_PORT
    itself); if that is taken rank 0 moves to the next candidate and the clients find it through
    the handshake."""

    def __init__(self, rank, world, addr="127.0.0.1", base_port=29500, timeout=600.0):
        self.rank, self.world = int(rank), int(world)
        self.conns = {}
        self.sock = None
        if self.world == 1:
            return
        deadline = time.time() + timeout
        ports = [base_port + s for s in _PORT_STEPS]
        if self.rank == 0:
            srv = None
            for p in ports:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr, p))
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise RuntimeError(f"rendezvous: no free port among {ports} on {addr}")
            srv.listen(self.world)
            srv.settimeout(1.0)
            while len(self.conns) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: ranks missing: %s" %
                                       sorted(set(range(1, self.world)) - set(self.conns)))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(5.0)
                    hello = _recv_exact(c, len(_MAGIC) + 8)
                    r, w = struct.unpack("<ii", hello[len(_MAGIC):])
                    if hello[:len(_MAGIC)] != _MAGIC or w != self.world or not 0 < r < self.world or r in self.conns:
                        c.close()
                        continue
                    c.sendall(b"OK")
                    c.settimeout(None)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.conns[r] = c
                except (OSError, ConnectionError, struct.error):
                    c.close()
            srv.close()
        else:
            while self.sock is None:
                for p in ports:
                    try:
                        s = socket.create_connection((addr, p), timeout=2.0)
                        s.settimeout(5.0)
                        s.sendall(_MAGIC + struct.pack("<ii", self.rank, self.world))
                        if _recv_exact(s, 2) == b"OK":
                            s.settimeout(None)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.sock = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        pass
                if self.sock is None:
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: rank {self.rank} found no rank 0 on {addr}:{ports}")
                    time.sleep(0.2)

    def exchange(self, payload=b""):
        "all-gather: returns [payload of rank 0, ..., payload of rank world-1]"
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [None] * (self.world - 1)
            for r, c in self.conns.items():
                parts[r] = _recv_msg(c)
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for c in self.conns.values():
                _send_msg(c, blob)
            return parts
        _send_msg(self.sock, payload)
        blob = _recv_msg(self.sock)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def close(self):
        for c in self.conns.values():
            c.close()
        self.conns = {}
        if self.sock is not None:
            self.sock.close()
            self.sock = None


def env_world():
    "(rank, local_rank, world, addr, port) from the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*)"
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", 0))),
            int(os.environ.get("WORLD_SIZE", 1)), os.environ.get("MASTER_ADDR", "127.0.0.1"),
            int(os.environ.get("MASTER_PORT", 29500)))


# ------------------------------------------------------------------ communicators
class SocketComm:
    """The communicator API on host arrays over the rendezvous sockets (CPU tests; no GPU)."""
    backend = "socket"

    def __init__(self, rdv):
        self.rdv = rdv
        self.rank, self.world = rdv.rank, rdv.world

    def broadcast(self, arr, root=0):
        "in-place broadcast of a NumPy array; returns seconds"
        t0 = time.perf_counter()
        data = self.rdv.exchange(arr.tobytes() if self.rank == root else b"")[root]
        if self.rank != root:
            arr[...] = np.frombuffer(data, dtype=arr.dtype).reshape(arr.shape)
        return time.perf_counter() - t0

    def gather_scalars(self, values):
        "every rank gets [world, count] (count doubles per rank)"
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        parts = self.rdv.exchange(v.tobytes())
        return np.stack([np.frombuffer(p, dtype=np.float64) for p in parts])

    def max(self, x):
        return float(self.gather_scalars([float(x)]).max())

    def barrier(self):
        self.rdv.exchange(b"")

    def close(self):
        self.rdv.close()


class RcclComm:
    """RCCL communicator of one device context (C ABI, csrc/comm.hip).  Device arrays in, plain pointers out."""
    backend = "rccl"

    def __init__(self, ctx, rdv):
        self.ctx, self.rdv = ctx, rdv
        self.rank, self.world = rdv.rank, rdv.world
        uid = (C.c_ubyte * 128)()
        failure = None
        if self.rank == 0:
            try:
                ctx._call("tramp_comm_unique_id", uid)
            except Exception as e:              # noqa: BLE001  (the exchange below still happens: ranks stay in step)
                failure = e
        blob = rdv.exchange(bytes(uid) if (self.rank == 0 and failure is None) else b"")[0]
        if failure is not None:
            raise failure
        if len(blob) != 128:
            raise RuntimeError("rank 0 could not create an RCCL unique id")
        uid = (C.c_ubyte * 128).from_buffer_copy(blob)
        h = C.c_void_p()
        self._h = None
        ctx._call("tramp_comm_init", self.rank, self.world, uid, C.byref(h))
        self._h = h

    def broadcast(self, darr, root=0):
        "in-place broadcast of a DeviceArray over xGMI; returns seconds (stream-synchronised)"
        self.ctx.sync()
        t0 = time.perf_counter()
        self.ctx._call("tramp_bcast_matrix", self._h, darr.ptr, darr.nbytes, root)
        self.ctx.sync()
        return time.perf_counter() - t0

    def gather_scalars(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        send = self.ctx.array(v, np.float64)
        recv = self.ctx.empty((self.world, v.size), np.float64)
        self.ctx._call("tramp_gather_scalars", self._h, send.ptr, v.size, recv.ptr)
        return recv.get()

    def max(self, x):
        v = C.c_double(float(x))
        self.ctx._call("tramp_comm_max", self._h, C.byref(v))
        return v.value

    def barrier(self):
        self.ctx._call("tramp_comm_barrier", self._h)

    def close(self):
        self.abort()
        self.rdv.close()

    def abort(self):
        "destroy the communicator ONCE and forget the handle; the rendezvous stays open (socket fallback uses it)"
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.ctx._call("tramp_comm_destroy", h)
            except Exception:                   # noqa: BLE001
                pass

    def __del__(self):
        try:
            h, self._h = getattr(self, "_h", None), None
            if h and self.ctx.handle:
                self.ctx.lib.tramp_comm_destroy(self.ctx.handle, h)
        except Exception:
            pass


def init_comm(ctx=None, backend="rccl", timeout=600.0):
    """Communicator for this process from the launcher's environment (RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT).  backend "rccl" needs a device context, "socket" is the CPU stub."""
    rank, _, world, addr, port = env_world()
    rdv = Rendezvous(rank, world, addr, port, timeout=timeout)
    if backend == "socket":
        return SocketComm(rdv)
    if ctx is None:
        raise ValueError("the RCCL communicator needs a device context")
    return RcclComm(ctx, rdv)


def init_comm_or_sockets(ctx, timeout=600.0, log=None):
    """RCCL communicator if EVERY rank can build one, else the socket communicator on the same rendezvous
    (host values only: barrier / max / gather - no device broadcast; the caller then computes the shared
    factors on every rank instead of receiving them).  The ranks agree on the outcome over the sockets, in
    two steps: (1) every rank probes the library (tramp_comm_available: dlopen + symbols, no collective) and
    the flags are exchanged BEFORE anyone enters ncclCommInitRank - a rank without RCCL can therefore not
    leave the others waiting in the bootstrap; (2) after the init the ranks exchange its outcome, and a rank
    whose communicator did come up drops it (RcclComm.abort) when another rank's did not."""
    rank, _, world, addr, port = env_world()
    rdv = Rendezvous(rank, world, addr, port, timeout=timeout)

    def fall_back(errs, stage):
        if log:
            log("RCCL unavailable on %d of %d ranks at %s (%s): socket communicator, factors computed on every rank"
                % (sum(1 for e in errs if e), world, stage, next(e for e in errs if e).decode(errors="replace")))
        return SocketComm(rdv)

    err = b""
    try:
        ctx._call("tramp_comm_available")
    except Exception as e:                      # noqa: BLE001  (library missing, a symbol missing)
        err = repr(e).encode()
    errs = rdv.exchange(err)
    if any(errs):
        return fall_back(errs, "probe")
    comm, err = None, b""
    try:
        comm = RcclComm(ctx, rdv)
    except Exception as e:                      # noqa: BLE001  (IPC refused, duplicate device, ...)
        err = repr(e).encode()
    errs = rdv.exchange(err)
    if not any(errs):
        return comm
    if comm is not None:
        comm.abort()
    return fall_back(errs, "init")


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(world, argv, env=None, port=None, grace=30.0):
    """Start `world` copies of `argv` (one per GPU) with the launcher environment set, wait for them,
    return the largest exit code.  The caller must not have initialised the GPU: the children do."""
    port = port or free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    # a rank that dies leaves the others waiting in a collective: once one has failed, the rest get a grace
    # period and are then stopped (these very processes, by handle), so that the launcher always returns
    rc, failed_at = 0, None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > grace:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc

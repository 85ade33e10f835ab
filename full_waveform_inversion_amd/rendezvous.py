"""Control plane of a one-node, one-process-per-GPU job: rendezvous, barrier, tiny reductions.

Standard library only (no torch, no MPI).  The data path never goes through here: gradients are summed
by RCCL on the device (``fwi_allreduce_gradient``).  What a launcher such as ``torch.distributed.run``
provides is RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; this module turns that into
the three things the shot loop needs from a control plane -- hand the 128-byte ``ncclUniqueId`` from rank 0
to everyone, a barrier, and a reduction of a few doubles -- the role the reference's
``multiprocessing.Pipe`` gather plays (full_waveform_inversion.py:816-848).

Topology: a star.  Rank 0 listens, every other rank connects and says who it is.  MASTER_PORT itself may
be held by the launcher (torchrun's agent keeps its own store there), and jobs on one node are commonly
given ADJACENT MASTER_PORTs, so the control port is looked for well away from it: rank 0 takes the first
free port of a 64-port window derived from MASTER_PORT (``control_port_base``: 20000 + a multiple of 64
below 65536, so that neighbouring MASTER_PORTs get disjoint windows and none of them contains a
MASTER_PORT of the usual 29500+ range), and the other ranks probe that window.  ``FWI_RDZV_PORT`` names
the control port explicitly (a window of one).  The hello carries MASTER_PORT and the world size as a job
token, so a listener of another job that landed in the same window is skipped rather than joined.
Messages are capped at 64 MiB (``MAX_MESSAGE_BYTES``): nothing the control plane carries comes near it,
and a peer that announces more is a protocol error, not an allocation.
"""
from __future__ import annotations

import os
import socket
import struct
import time

_MAGIC = b"FWIRDZV1"
_PORT_SPAN = 64
_HELLO = struct.Struct("!8sIII")   # magic, token (MASTER_PORT), world, rank
_LEN = struct.Struct("!Q")
MAX_MESSAGE_BYTES = 64 << 20


def control_port_base(master_port):
    """First port of the 64-port control window of the job whose launcher holds ``master_port``."""
    return 20000 + ((int(master_port) * 64) % (9472 - _PORT_SPAN))  # 20000 .. 29407: below the 29500+ launchers


class RendezvousError(RuntimeError):
    pass


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise RendezvousError("peer closed the control connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(_LEN.pack(len(payload)) + payload)


def _recv_msg(sock):
    (n,) = _LEN.unpack(_recv_exact(sock, _LEN.size))
    if n > MAX_MESSAGE_BYTES:
        raise RendezvousError("peer announced a %d-byte control message (limit %d)" % (n, MAX_MESSAGE_BYTES))
    return _recv_exact(sock, n)


class Rendezvous:
    """``Rendezvous.from_env()`` in every rank; then ``broadcast`` / ``barrier`` / ``allreduce``.

    Collective calls must be made by all ranks in the same order (as with any communicator).
    """

    def __init__(self, rank, world, addr="127.0.0.1", port=29500, timeout=300.0):
        if not (0 <= rank < world):
            raise ValueError("rank %d outside world of %d" % (rank, world))
        self.rank, self.world = int(rank), int(world)
        self.addr, self.port, self.timeout = addr, int(port), float(timeout)
        self._peers = []      # rank 0: sockets of ranks 1 .. world-1, by rank
        self._up = None       # other ranks: socket to rank 0
        self._listener = None
        explicit = os.environ.get("FWI_RDZV_PORT")
        self._ports = [int(explicit)] if explicit else list(range(control_port_base(self.port),
                                                                   control_port_base(self.port) + _PORT_SPAN))
        if self.world > 1:
            if self.rank == 0:
                self._serve()
            else:
                self._join()

    @classmethod
    def from_env(cls, timeout=300.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")),
                   timeout)

    # -- set-up -------------------------------------------------------------------------------------
    def _serve(self):
        ls = None
        for p in self._ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((self.addr, p))
                s.listen(self.world + 8)
                ls = s
                break
            except OSError:
                s.close()
        if ls is None:
            raise RendezvousError("no free control port in %d..%d" % (self._ports[0], self._ports[-1]))
        self._listener = ls
        peers = {}
        deadline = time.monotonic() + self.timeout
        while len(peers) < self.world - 1:
            ls.settimeout(max(0.1, deadline - time.monotonic()))
            try:
                c, _ = ls.accept()
            except socket.timeout:
                raise RendezvousError("only %d of %d ranks joined within %.0f s"
                                      % (len(peers) + 1, self.world, self.timeout)) from None
            try:
                c.settimeout(10.0)
                magic, token, world, rank = _HELLO.unpack(_recv_exact(c, _HELLO.size))
                ok = (magic == _MAGIC and token == self.port and world == self.world
                      and 0 < rank < self.world and rank not in peers)
                c.sendall(b"\x01" if ok else b"\x00")
                if not ok:
                    c.close()
                    continue
            except (OSError, RendezvousError, struct.error):
                c.close()
                continue
            c.settimeout(self.timeout)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            peers[rank] = c
        self._peers = [peers[r] for r in range(1, self.world)]

    def _join(self):
        deadline = time.monotonic() + self.timeout
        hello = _HELLO.pack(_MAGIC, self.port, self.world, self.rank)
        while time.monotonic() < deadline:
            for p in self._ports:
                try:
                    s = socket.create_connection((self.addr, p), timeout=2.0)
                except OSError:
                    continue
                try:
                    s.settimeout(10.0)
                    s.sendall(hello)
                    if _recv_exact(s, 1) == b"\x01":
                        s.settimeout(self.timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._up = s
                        return
                except (OSError, RendezvousError):
                    pass
                s.close()
            time.sleep(0.05)
        raise RendezvousError("rank %d found no rank 0 on %s:%d..%d within %.0f s"
                              % (self.rank, self.addr, self._ports[0], self._ports[-1], self.timeout))

    def set_timeout(self, seconds):
        """Bound how long a collective waits for a peer from now on (an optional phase of a job can then fail
        in seconds instead of holding the other ranks for the set-up timeout)."""
        self.timeout = float(seconds)
        for s in self._peers + [self._up]:
            if s is not None:
                s.settimeout(self.timeout)

    def close(self):
        for s in self._peers + [self._up, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._up, self._listener = [], None, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- collectives --------------------------------------------------------------------------------
    def broadcast(self, payload=None):
        """Bytes from rank 0 to every rank (the argument is ignored on the other ranks)."""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for s in self._peers:
                _send_msg(s, payload)
            return payload
        return _recv_msg(self._up)

    def gather(self, payload):
        """Rank 0 gets the list of every rank's bytes (by rank); the others get None."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [_recv_msg(s) for s in self._peers]
        _send_msg(self._up, payload)
        return None

    def barrier(self):
        self.broadcast(b"".join(self.gather(b"") or []))

    def allreduce(self, values, op="sum"):
        """Element-wise sum / max / min of a short list of doubles over the ranks; every rank gets it."""
        vals = [float(v) for v in values]
        fn = {"sum": sum, "max": max, "min": min}[op]
        fmt = "!%dd" % len(vals)
        rows = self.gather(struct.pack(fmt, *vals))
        out = None
        if rows is not None:
            cols = zip(*(struct.unpack(fmt, r) for r in rows))
            out = struct.pack(fmt, *(fn(c) for c in cols))
        return list(struct.unpack(fmt, self.broadcast(out)))

    def allreduce_array(self, a):
        """Sum of a (small) float64 NumPy array over the ranks, in rank order (bit-reproducible): the host
        path of the CPU multi-process tests.  Production gradients are summed by RCCL on the device."""
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.float64)
        rows = self.gather(a.tobytes())
        out = None
        if rows is not None:
            acc = np.frombuffer(rows[0], np.float64).copy()
            for r in rows[1:]:
                acc += np.frombuffer(r, np.float64)
            out = acc.tobytes()
        return np.frombuffer(self.broadcast(out), np.float64).reshape(a.shape).copy()

"""A small RPC transport with the part of the `zerorpc` API the reference's two scripts use
(scripts/reactive_tamp.py:89-94 planner side, scripts/sim.py:29-49 world side):

    server = Server(obj); server.bind("tcp://0.0.0.0:4242"); server.run()
    client = Client(); client.connect("tcp://127.0.0.1:4242"); reply = client.any_method(*args)

so that, when the real zerorpc (ZeroMQ + gevent) is not installed, `compat.install()` can register this
module under the name `zerorpc` and the unchanged scripts run as two processes.  Wire format: one TCP
connection per client, length-prefixed msgpack frames -- request [method, args], reply [error, result];
`bytes` payloads (the scripts ship `torch.save` blobs, utils/data_transfer.py:4-12) travel as msgpack
bin.  Like zerorpc's server on gevent, calls are served one at a time in the order they arrive
(reactive_tamp.py's command() is never re-entered).  Only public methods of the served object are
callable.

Security posture (this is a lab transport, like the zerorpc it stands in for -- no authentication):
  * the transport itself never unpickles: frames are msgpack.  The `bytes` payloads the scripts exchange ARE
    `torch.save` blobs, i.e. pickles; `compat.bytes_to_torch` loads them with `weights_only=True` (tensors and
    plain containers only -- no arbitrary object construction);
  * "tcp://0.0.0.0:port" / "tcp://*:port" (what the reference's scripts bind) listens on 127.0.0.1 ONLY unless
    M3P2I_RPC_BIND_ALL=1 is set: planner and world run on one machine in every documented use;
  * a frame must be a 2-element list [str, list]; anything else closes that client, never the server;
  * a client that stalls in the middle of a frame is dropped after CLIENT_TIMEOUT seconds;
  * frames above MAX_FRAME (64 MiB; the largest real payload is a [K, nA, 13] root-state blob: 2.3 MB at K = 4000)
    are refused before any allocation.
"""
from __future__ import annotations

import os
import select
import socket
import struct

import msgpack

_HDR = struct.Struct("!I")
MAX_FRAME = 64 << 20
CLIENT_TIMEOUT = 10.0      # seconds a half-sent frame may stall before the client is dropped


class RemoteError(RuntimeError):
    """An exception raised by the served object, re-raised on the client (zerorpc.RemoteError)."""


class LostRemote(RuntimeError):
    """The peer went away (zerorpc.LostRemote)."""


def _endpoint(ep: str):
    if not ep.startswith("tcp://"):
        raise ValueError(f"only tcp:// endpoints are supported, got {ep!r}")
    host, _, port = ep[len("tcp://"):].rpartition(":")
    return ("" if host in ("*", "0.0.0.0") else host), int(port)


def _bind_host(host: str) -> str:
    """Wildcard endpoints listen on the loopback interface unless the operator opts in to all interfaces.  That differs
    from the reference's zerorpc scripts (which bind every interface): it is said at bind time, once per process."""
    global _warned_narrowed
    if host == "":
        if os.environ.get("M3P2I_RPC_BIND_ALL") == "1":
            return ""
        if not _warned_narrowed:
            _warned_narrowed = True
            import sys
            print("m3p2i_aip_amd.rpc: wildcard endpoint bound to 127.0.0.1 only (a sim / planner split across two hosts "
                  "gets connection-refused); set M3P2I_RPC_BIND_ALL=1 to listen on all interfaces as zerorpc does",
                  file=sys.stderr)
        return "127.0.0.1"
    return host


_warned_narrowed = False


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise LostRemote("connection closed by the peer")
        buf += chunk
    return bytes(buf)


def _send(sock, obj):
    payload = msgpack.packb(obj, use_bin_type=True)
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > MAX_FRAME:
        raise LostRemote(f"frame of {n} bytes refused")
    return msgpack.unpackb(_recv_exact(sock, n), raw=False)


class Server:
    def __init__(self, methods=None, name=None, context=None, pool_size=None, heartbeat=None):
        self._target = methods
        self._listen = []
        self._clients = []
        self._running = False

    def bind(self, endpoint):
        host, port = _endpoint(endpoint)
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind((_bind_host(host), port))
        s.listen(8)
        self._listen.append(s)
        return s.getsockname()[1]

    def _call(self, name, args):
        if not isinstance(name, str) or name.startswith("_"):
            raise AttributeError(f"no such method: {name!r}")
        fn = getattr(self._target, name)
        if not callable(fn):
            raise AttributeError(f"{name!r} is not callable")
        return fn(*args)

    def _drop(self, s):
        if s in self._clients:
            self._clients.remove(s)
        try:
            s.close()
        except OSError:
            pass

    def serve_once(self, timeout=None):
        """Wait for activity (up to `timeout` seconds) and serve what arrived; returns the number of calls."""
        try:
            ready, _, _ = select.select(self._listen + self._clients, [], [], timeout)
        except (OSError, ValueError):      # close() from another thread while waiting
            return 0
        served = 0
        for s in ready:
            if s in self._listen:
                c, _ = s.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(CLIENT_TIMEOUT)        # a stalled half-frame must not block the (single-threaded) server
                self._clients.append(c)
                continue
            try:
                frame = _recv(s)
                if not (isinstance(frame, (list, tuple)) and len(frame) == 2 and isinstance(frame[0], str)
                        and isinstance(frame[1], (list, tuple))):
                    raise ValueError("malformed request frame (want [method: str, args: list])")
                name, args = frame
            except (LostRemote, OSError, ValueError, TypeError, msgpack.exceptions.UnpackException,
                    msgpack.exceptions.ExtraData):   # (socket.timeout and ConnectionError are OSErrors)
                self._drop(s)
                continue
            try:
                reply = [None, self._call(name, args)]
            except Exception as e:  # the served object's exception goes back to the caller
                reply = [f"{type(e).__name__}: {e}", None]
            try:
                _send(s, reply)
            except (ConnectionError, OSError):
                self._drop(s)
            served += 1
        return served

    def run(self):
        self._running = True
        while self._running:
            self.serve_once(0.5)

    def stop(self):
        self._running = False

    def close(self):
        self.stop()
        for s in self._listen + self._clients:
            s.close()
        self._listen, self._clients = [], []


class Client:
    def __init__(self, connect_to=None, context=None, timeout=30, heartbeat=None, passive_heartbeat=False):
        self._sock = None
        self._timeout = timeout
        if connect_to:
            self.connect(connect_to)

    def connect(self, endpoint, resolve=True):
        host, port = _endpoint(endpoint)
        self._sock = socket.create_connection((host or "127.0.0.1", port), timeout=self._timeout)
        self._sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def close(self):
        if self._sock is not None:
            self._sock.close()
            self._sock = None

    def __call__(self, method, *args):
        if self._sock is None:
            raise LostRemote("not connected")
        _send(self._sock, [method, list(args)])
        err, result = _recv(self._sock)
        if err is not None:
            raise RemoteError(err)
        return result

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *args: self(name, *args)

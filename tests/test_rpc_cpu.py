"""The zerorpc stand-in (m3p2i_aip_amd/rpc.py): the calls scripts/reactive_tamp.py:89-94 and
scripts/sim.py:29-49 make, over a real TCP connection (server in a thread), CPU only."""
import threading

import pytest

torch = pytest.importorskip("torch")


class _Planner:
    """what reactive_tamp.py serves: methods taking / returning torch.save blobs"""

    def __init__(self):
        self.calls = 0
        self._secret = 1

    def run_tamp(self, dof_bytes, root_bytes):
        from m3p2i_aip_amd.compat import bytes_to_torch, torch_to_bytes
        self.calls += 1
        return torch_to_bytes(bytes_to_torch(dof_bytes)[0, :2] + bytes_to_torch(root_bytes).sum())

    def get_suction(self):
        from m3p2i_aip_amd.compat import torch_to_bytes
        return torch_to_bytes(self.calls % 2 == 0)

    def boom(self):
        raise ValueError("planner side failed")


def test_server_client_round_trip_like_the_reference_scripts():
    from m3p2i_aip_amd import compat
    compat.install(force_standins=True)
    import zerorpc                                    # the stand-in registered under the reference's import name
    from m3p2i_aip.utils.data_transfer import bytes_to_torch, torch_to_bytes
    obj = _Planner()
    server = zerorpc.Server(obj)
    port = server.bind("tcp://127.0.0.1:0")
    th = threading.Thread(target=server.run, daemon=True)
    th.start()
    try:
        planner = zerorpc.Client()
        planner.connect(f"tcp://127.0.0.1:{port}")
        dof, root = torch.arange(8.0).view(2, 4), torch.ones(2, 3, 13)
        for i in range(3):                            # sim.py:41-49
            action = bytes_to_torch(planner.run_tamp(torch_to_bytes(dof), torch_to_bytes(root)))
            assert torch.equal(action, dof[0, :2] + 78.0)
            assert bytes_to_torch(planner.get_suction()) == ((i + 1) % 2 == 0)
        assert obj.calls == 3
        with pytest.raises(zerorpc.RemoteError, match="planner side failed"):
            planner.boom()
        with pytest.raises(zerorpc.RemoteError):      # private attributes are not callable from outside
            planner("_secret")
        with pytest.raises(zerorpc.RemoteError):
            planner.no_such_method()
        big = torch.randn(300000)                     # a frame larger than one recv
        other = zerorpc.Client(f"tcp://127.0.0.1:{port}")   # a second client (viewer + world in the reference)
        out = bytes_to_torch(other.run_tamp(torch_to_bytes(big.view(1, -1)), torch_to_bytes(torch.zeros(1))))
        assert torch.equal(out, big[:2])
        other.close()
        planner.close()
    finally:
        server.close()
        th.join(timeout=5)


def test_bad_endpoint_and_lost_remote():
    from m3p2i_aip_amd import rpc
    with pytest.raises(ValueError):
        rpc.Server(object()).bind("ipc:///tmp/x")
    c = rpc.Client()
    with pytest.raises(rpc.LostRemote):
        c.anything()

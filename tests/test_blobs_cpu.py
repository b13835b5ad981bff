"""The tensors a tick exchanges in the reference's wire format (torch.save archives over zerorpc, utils/data_transfer.py:4-12)
through m3p2i_aip_amd.blobs: same archives, read and written by torch itself on the other side."""
import io
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _torch_save(t):
    b = io.BytesIO()
    torch.save(t, b)
    return b.getvalue()


def _torch_load(b):
    return torch.load(io.BytesIO(b), weights_only=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int64, torch.int32, torch.uint8, torch.bool, torch.float16])
@pytest.mark.parametrize("shape", [(1, 18), (1, 7, 13), (30, 2), (9,), (1,), (20, 9), (4000, 2)])
def test_patched_archives_are_torch_archives(dtype, shape):
    from m3p2i_aip_amd.blobs import TensorBlobCodec
    c = TensorBlobCodec()
    rng = np.random.default_rng(3)
    for trial in range(3):
        t = torch.from_numpy(rng.integers(0, 2 if dtype == torch.bool else 100, shape)).to(dtype)
        if dtype.is_floating_point:
            t = t / 7 - 3
            if trial == 1:
                t.view(-1)[0] = float("nan"); t.view(-1)[-1] = float("-inf")
        b = c.save(t)
        back = _torch_load(b)                                     # torch's own reader accepts what the codec wrote ...
        assert back.dtype == dtype and tuple(back.shape) == shape and torch.equal(back.view(torch.uint8), t.contiguous().view(torch.uint8))
        assert len(b) == len(_torch_save(t))
        mine = c.load(_torch_save(t))                             # ... and the codec reads what torch's own writer wrote
        assert mine.dtype == dtype and torch.equal(mine.view(torch.uint8), t.contiguous().view(torch.uint8))
    assert c.fast_saves == 3 and c.slow_saves == 0 and c.fast_loads == 3 and c.slow_loads == 0


def test_everything_else_takes_torchs_own_path():
    from m3p2i_aip_amd.blobs import TensorBlobCodec
    c = TensorBlobCodec()
    base = torch.arange(40.0).reshape(4, 10)
    cases = [base[:, :3],                      # not contiguous
             base[1],                          # a view into a larger storage
             torch.ones(3, requires_grad=True),
             torch.zeros(0),                   # empty
             3.5, 7, "reach", [torch.ones(2), 1], {"a": torch.ones(2)}]
    for x in cases:
        got = c.load(c.save(x))
        want = _torch_load(_torch_save(x))
        if torch.is_tensor(want):
            assert torch.equal(got, want) and got.dtype == want.dtype and got.shape == want.shape
        elif isinstance(want, list):
            assert torch.equal(got[0], want[0]) and got[1] == want[1]
        elif isinstance(want, dict):
            assert torch.equal(got["a"], want["a"])
        else:
            assert got == want
    assert c.fast_saves == 0 and c.slow_saves == len(cases)
    # the suction flag (a bool per tick, reactive_tamp.py:83-85): one archive per value, recognised byte for byte
    for v in (True, False, None, True):
        b = c.save(v)
        assert b == _torch_save(v) and c.load(b) is v and c.load(_torch_save(v)) is v
    assert c.slow_saves == len(cases)
    # a first blob of a kind this process has never written is read by torch and its archive learnt; the next one is lifted out
    c2 = TensorBlobCodec()
    for i in range(3):
        t = torch.full((1, 18), float(i))
        assert torch.equal(c2.load(_torch_save(t)), t)
    assert (c2.slow_loads, c2.fast_loads) == (1, 2)


def test_a_damaged_archive_is_not_lifted_out():
    """An archive whose payload does not match its CRC, or that differs from the known one outside the payload, is left to
    torch's own reader: whatever that does with it (torch.load does not verify CRCs; it may raise on a damaged header) is what
    the caller gets, exactly as without the codec."""
    from m3p2i_aip_amd.blobs import TensorBlobCodec
    c = TensorBlobCodec()
    t = torch.arange(18.0).reshape(1, 18)
    a = c._archive(t.dtype, t.shape, t.device)
    for where in (a.data_off + 5, 10, 40, 700):
        b = bytearray(c.save(t))
        b[where] ^= 0x40
        try:
            want = _torch_load(bytes(b))
        except Exception as e:
            want = type(e)
        try:
            got = c.load(bytes(b))
        except Exception as e:
            got = type(e)
        if torch.is_tensor(want):
            assert torch.is_tensor(got) and torch.equal(got.view(torch.uint8), want.view(torch.uint8))
        else:
            assert got == want
    assert c.fast_loads == 0


def test_compat_data_transfer_uses_it_and_is_faster():
    from m3p2i_aip_amd import compat
    from m3p2i_aip_amd.blobs import CODEC
    t = torch.randn(1, 18)
    for _ in range(5):
        assert torch.equal(compat.bytes_to_torch(compat.torch_to_bytes(t)), t)
    assert compat.bytes_to_torch(compat.torch_to_bytes(True)) is True
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        compat.bytes_to_torch(compat.torch_to_bytes(t))
    fast = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        _torch_load(_torch_save(t))
    slow = (time.perf_counter() - t0) / n
    print(f"round trip of a [1, 18] f32 tensor: codec {fast * 1e6:.1f} us, torch.save + torch.load {slow * 1e6:.1f} us")
    assert CODEC.fast_saves >= n and fast < 0.5 * slow

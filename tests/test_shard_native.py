"""libsonde_rccl.so (include/sonde_shard.h, csrc/shard_rccl.cpp): the node-level scatter / gather of SURVEY 8(e) in native
code.  CPU: the library loads, exports every declared symbol, and its range arithmetic equals the Python sharder's.
GPU (one device here; the N-GPU run is the driver's): a communicator of one rank scatters to / gathers from itself through
the same grouped ncclSend / ncclRecv calls the N-rank case makes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sdrpp_radiosonde_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_and_range_arithmetic():
    L = shard.NativeShard.lib()
    hdr = open(os.path.join(ROOT, "include", "sonde_shard.h")).read()
    declared = set(re.findall(r"\b(sonde_shard_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) == 10
    for s in declared:
        assert hasattr(L, s), s
    for n in (0, 1, 7, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard.NativeShard.channel_range(n, r, world) for r in range(world)]
            assert spans == [shard.channel_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    bad = C.c_void_p()
    assert L.sonde_shard_create(None, 1, 0, 0, C.byref(bad)) != 0 and b"bad argument" in L.sonde_shard_last_error()


@pytest.mark.gpu
def test_single_rank_scatter_gather_on_device():
    import torch
    ns = shard.NativeShard(0, rank=0, world=1)
    x = torch.randn((6, 4096, 2), device="cuda:0")
    got = ns.scatter_iq(x, x.shape, root=0)
    torch.cuda.synchronize()
    assert torch.equal(got, x) and got.data_ptr() != x.data_ptr()
    back = ns.gather_bytes(got, root=0)
    torch.cuda.synchronize()
    assert back.shape == (1, x.numel() * 4) and torch.equal(back.view(-1).view(torch.float32).reshape(x.shape), x)
    ns.close()


@pytest.mark.gpu
def test_single_rank_scatter_rows_layouts():
    """sonde_shard_scatter_rows through NativeShard.scatter_rows: the destination layout follows the source layout, passed alike on
    every rank: rows on the decoder's recommended stride land on that stride (one send per peer), rows back to back
    land back to back (one send per peer of exactly the shard's bytes), any other stride lands on the recommended stride row by row."""
    import torch
    from sdrpp_radiosonde_amd.batch import row_stride, strided_rows
    ns = shard.NativeShard(0, rank=0, world=1)
    n = 8 * 2048 + 2048                                   # 144 KiB rows -> 256 KiB stride
    x = torch.randn((5, n, 2), device="cuda:0")
    got = ns.scatter_rows(x, 5, n, root=0)                                     # back to back
    torch.cuda.synchronize()
    assert got.shape == x.shape and got.stride(0) == 2 * n and torch.equal(got, x) and got.data_ptr() != x.data_ptr()
    xs = strided_rows(x)
    got = ns.scatter_rows(xs, 5, n, root=0, src_stride=row_stride(n))         # on the recommended stride
    torch.cuda.synchronize()
    assert got.stride(0) == 2 * row_stride(n) and torch.equal(got, x)
    xo = strided_rows(x, n + 4096)
    got = ns.scatter_rows(xo, 5, n, root=0, src_stride=n + 4096)              # any other stride
    torch.cuda.synchronize()
    assert got.stride(0) == 2 * row_stride(n) and torch.equal(got, x)
    with pytest.raises(AssertionError):
        ns.scatter_rows(xs, 5, n, root=0)                                      # the root's tensor contradicts the stride every rank was told
    ns.close()

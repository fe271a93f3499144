"""N>1 path on CPU: world_size-2 gloo run of the channel sharding (scatter of IQ blocks from the
ingest rank, per-rank decode, gather of frames).  The per-rank decoder here is the CPU oracle --
the product's HIP path needs a GPU -- so this covers the sharding logic the GPU ranks use."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdrpp_radiosonde_amd.shard import channel_range


def test_channel_range_partitions():
    for n, w in ((65536, 8), (1024, 3), (7, 8), (10, 4)):
        spans = [channel_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdrpp_radiosonde_amd import synth
    from sdrpp_radiosonde_amd.shard import scatter_iq, gather_frames
    import oracle_lib
    C, n = 3, 2048 * 34
    full = None
    if rank == 0:
        full = synth.make_rs41_batch(C * world, n, seed=55, ebn0_db=25.0).iq
    mine = scatter_iq(full, C, n, "cpu", src=0)
    lo, hi = channel_range(C * world, rank, world)
    fr = oracle_lib.batch_run(0, mine.numpy())
    fr["channel"] += lo                      # local -> global channel ids
    allf = gather_frames(fr, dst=0)
    if rank == 0:
        ref = oracle_lib.batch_run(0, full.numpy())
        assert len(ref) >= C * world
        assert allf.tobytes() == ref.tobytes()
        open(os.path.join(tmp, "ok"), "w").write(str(len(ref)))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_decode_gather_world2(tmp_path, oracle):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) >= 6

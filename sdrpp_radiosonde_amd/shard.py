"""Channel sharding across the GPUs of one node.

Channels are independent units: rank r of W owns the contiguous range
[r*C/W, (r+1)*C/W).  The only exchange step the path has is the ingest scatter of IQ blocks from
the rank that holds them (RCCL grouped send/recv over xGMI on GPUs, gloo on CPU in the tests) and
the small gather of decoded frames; there is no all-reduce anywhere on the data path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def channel_range(n_channels: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced partition; the first (n % world) ranks get one extra channel."""
    base, extra = divmod(n_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_iq(full: torch.Tensor | None, channels_per_rank: int, n_samples: int, device, src: int = 0,
               group=None) -> torch.Tensor:
    """Scatter [W*C, n, 2] float32 IQ held by `src` so that every rank gets its [C, n, 2] shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = torch.empty((channels_per_rank, n_samples, 2), dtype=torch.float32, device=device)
    chunks = None
    if rank == src:
        assert full is not None and full.shape[0] == world * channels_per_rank
        chunks = [c.contiguous() for c in full.chunk(world, dim=0)]
    dist.scatter(out, scatter_list=chunks, src=src, group=group)
    return out


def gather_frames(frames: np.ndarray, dst: int = 0, group=None) -> np.ndarray | None:
    """Gather per-rank frame records (structured array, channel ids already global) on `dst`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    objs = [None] * world if rank == dst else None
    dist.gather_object(frames, objs, dst=dst, group=group)
    if rank != dst:
        return None
    allf = np.concatenate(objs)
    return allf[np.lexsort((allf["bitpos"], allf["channel"]))]


# ---------------------------------------------------------------- native path (libsonde_rccl.so, include/sonde_shard.h)
class NativeShard:
    """The node-level scatter / gather in native code: grouped ncclSend / ncclRecv over xGMI (csrc/shard_rccl.cpp).
    The 128-byte communicator id travels through the torch.distributed group that launched the ranks (any backend)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes as C
            import os
            from . import _lib as main
            main.load()
            path = os.path.join(main.PKG_DIR, "libsonde_rccl.so")
            if not os.path.exists(path):                 # built on demand: only multi-GPU hosts need RCCL (csrc/Makefile `rccl`)
                import subprocess
                subprocess.check_call(["make", "-s", "-C", os.path.join(main.PKG_DIR, "csrc"), "rccl"])
            L = C.CDLL(path)
            vp = C.c_void_p
            L.sonde_shard_unique_id.argtypes = [vp]
            L.sonde_shard_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
            L.sonde_shard_destroy.argtypes = [vp]
            L.sonde_shard_destroy.restype = None
            L.sonde_shard_range.argtypes = [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            L.sonde_shard_range.restype = None
            L.sonde_shard_scatter.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
            L.sonde_shard_gather.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
            L.sonde_shard_scatter_rows.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int, vp]
            L.sonde_shard_last_error.restype = C.c_char_p
            cls._lib = L
        return cls._lib

    def __init__(self, device: int, group=None, rank: int | None = None, world: int | None = None):
        import ctypes as C
        L = self.lib()
        if world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.rank, self.world, self.device = rank, world, device
        ident = C.create_string_buffer(128)
        if rank == 0 and L.sonde_shard_unique_id(ident) != 0:
            raise RuntimeError(L.sonde_shard_last_error().decode())
        if world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        if L.sonde_shard_create(ident, world, rank, device, C.byref(h)) != 0:
            raise RuntimeError(L.sonde_shard_last_error().decode())
        self.h = h

    @staticmethod
    def channel_range(n_channels: int, rank: int, world: int) -> tuple[int, int]:
        import ctypes as C
        f, c = C.c_uint32(), C.c_uint32()
        NativeShard.lib().sonde_shard_range(n_channels, world, rank, C.byref(f), C.byref(c))
        return f.value, f.value + c.value

    def scatter_iq(self, full: torch.Tensor | None, shard_shape, root: int = 0) -> torch.Tensor:
        """root holds `full` = world consecutive shards ([world * C, n, 2] float32); returns this rank's [C, n, 2]."""
        import ctypes as C
        out = torch.empty(tuple(shard_shape), dtype=torch.float32, device=f"cuda:{self.device}")
        nbytes = out.numel() * 4
        if self.rank == root:
            assert full is not None and full.is_contiguous() and full.numel() * 4 == nbytes * self.world
        st = torch.cuda.current_stream(out.device).cuda_stream
        if self.lib().sonde_shard_scatter(self.h, C.c_void_p(full.data_ptr() if self.rank == root else 0), C.c_void_p(out.data_ptr()),
                                          nbytes, root, C.c_void_p(st)) != 0:
            raise RuntimeError(self.lib().sonde_shard_last_error().decode())
        return out

    def scatter_rows(self, full: torch.Tensor | None, n_rows_total: int, n_samples: int, root: int = 0, src_stride: int | None = None) -> torch.Tensor:
        """root holds `full` = [n_rows_total, n, 2] float32, rows `src_stride` samples apart (default: back to back); every rank gets
        the rows of its channel range (sonde_shard_range: unequal when the count does not divide) as a [count, n, 2] view of what
        SondeBatch.submit takes.  src_stride must be passed ALIKE ON EVERY RANK (only the root can read it off `full`; ADVICE r4: a
        rank guessing it could pick the other transfer shape and hang): it decides the shape of the transfer as in the node-level host
        (include/sonde_node.h): rows on the decoder's recommended stride -> one send per peer, rows land on that stride; rows back to
        back -> one send per peer of exactly the shard's bytes, rows land back to back; any other stride -> one send per row into
        rows on the recommended stride."""
        import ctypes as C
        from .batch import row_stride
        lo, hi = self.channel_range(n_rows_total, self.rank, self.world)
        reco = row_stride(n_samples, iq=True)
        src_stride = int(src_stride or n_samples)
        dst_stride = n_samples if src_stride == n_samples else reco
        buf = torch.empty((hi - lo, dst_stride, 2), dtype=torch.float32, device=f"cuda:{self.device}")
        if self.rank == root:
            assert full is not None and full.shape[0] == n_rows_total and full.shape[1] == n_samples and full.stride(1) == 2
            assert full.stride(0) == 2 * src_stride, "src_stride (passed alike on every rank) must be the root tensor's row stride"
        st = torch.cuda.current_stream(buf.device).cuda_stream
        if self.lib().sonde_shard_scatter_rows(self.h, C.c_void_p(full.data_ptr() if self.rank == root else 0), src_stride * 8,
                                               C.c_void_p(buf.data_ptr()), dst_stride * 8, n_samples * 8, n_rows_total, root, C.c_void_p(st)) != 0:
            raise RuntimeError(self.lib().sonde_shard_last_error().decode())
        return buf[:, :n_samples]

    def gather_bytes(self, part: torch.Tensor, root: int = 0) -> torch.Tensor | None:
        """Every rank contributes the same number of bytes (a padded frame block); root gets [world, nbytes] uint8."""
        import ctypes as C
        part = part.contiguous().view(torch.uint8).reshape(-1)
        out = torch.empty((self.world, part.numel()), dtype=torch.uint8, device=part.device) if self.rank == root else None
        st = torch.cuda.current_stream(part.device).cuda_stream
        if self.lib().sonde_shard_gather(self.h, C.c_void_p(part.data_ptr()), part.numel(), C.c_void_p(out.data_ptr() if out is not None else 0),
                                         root, C.c_void_p(st)) != 0:
            raise RuntimeError(self.lib().sonde_shard_last_error().decode())
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib().sonde_shard_destroy(self.h)
            self.h = None


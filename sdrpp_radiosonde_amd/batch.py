"""Python face of the B0 batch API (include/sonde_abi.h): thin, no compute -- every call goes
through the C ABI of libsonde_mi355.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FRAME_DTYPE, INPUT_IQ, INPUT_IQ8, INPUT_IQ16, INPUT_REAL, TILE


class SondeError(RuntimeError):
    pass


class SondeBatch:
    """Many independent 48 kS/s channels on one GPU; one HIP workgroup per channel.

    submit() takes a device tensor (torch, any object with data_ptr()) shaped
    [C, n, 2] float32 (IQ) or [C, n] float32 (discriminator samples), n % 2048 == 0.
    """

    def __init__(self, n_channels: int, max_samples: int, *, types=None, input_kind: int = INPUT_IQ, device: int = 0, flags: int = 0, time_slices: int = 0):
        self.L = _lib.load()
        self.n_channels = int(n_channels)
        self.max_samples = int(max_samples)
        self.input_kind = input_kind
        self.device = int(device)
        cfg = _lib.SondeBatchConfig()
        cfg.n_channels = self.n_channels
        self._types = None
        if types is not None:
            self._types = np.ascontiguousarray(types, dtype=np.uint8)
            assert self._types.shape == (self.n_channels,)
            cfg.types = self._types.ctypes.data_as(C.POINTER(C.c_uint8))
        cfg.max_samples = self.max_samples
        cfg.flags = flags
        cfg.time_slices = time_slices          # 0: the library's choice (SondeBatchConfig.time_slices)
        cfg.input_kind = input_kind
        cfg.device = device
        h = C.c_void_p()
        if self.L.sonde_batch_create(C.byref(cfg), C.byref(h)) != 0:
            raise SondeError(_lib.last_error())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.sonde_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc < 0:
            raise SondeError(_lib.last_error())
        return rc

    def submit(self, samples, stream: int | None = None):
        shape = tuple(samples.shape)
        n = shape[1]
        if shape[0] != self.n_channels:
            raise SondeError("first dimension must be n_channels")
        is_iq = self.input_kind in (INPUT_IQ, INPUT_IQ16, INPUT_IQ8)
        if is_iq and (len(shape) != 3 or shape[2] != 2):
            raise SondeError("IQ input must be [C, n, 2] (float32; int16 for INPUT_IQ16, int8 for INPUT_IQ8)")
        if not is_iq and len(shape) != 2:
            raise SondeError("real input must be [C, n] float32")
        # the C side sees only a pointer: check what it cannot (dtype, device, inner layout)
        dt = str(getattr(samples, "dtype", ""))
        want = {INPUT_IQ16: "int16", INPUT_IQ8: "int8"}.get(self.input_kind, "float32")
        if not dt.endswith(want) or dt.endswith("u" + want):
            raise SondeError(f"samples must be {want}, got {dt}")
        dev = getattr(samples, "device", None)
        if dev is None or getattr(dev, "type", "") != "cuda":
            raise SondeError("samples must be a device (HIP) tensor; use submit_host() for host memory")
        if dev.index is not None and dev.index != self.device:
            raise SondeError(f"samples live on device {dev.index}, the batch on device {self.device}")
        st = tuple(samples.stride())
        if (is_iq and st[1:] != (2, 1)) or (not is_iq and st[1] != 1):
            raise SondeError("samples must be contiguous inside a channel (only the channel stride may be padded)")
        stride = samples.stride(0) // (2 if is_iq else 1)
        # keep the device buffers of the last two submits alive (Python-side lifetime only): with SONDE_FLAG_LATE_JOIN / _PIPELINE the
        # buffer of submit t may be read until submit t + 1 is queued (include/sonde_abi.h); with default flags stream order covers it
        self._keep = (samples, getattr(self, "_keep", (None, None))[0])
        self._chk(self.L.sonde_batch_submit(self.h, C.c_void_p(samples.data_ptr()), n, stride, C.c_void_p(stream or 0)))

    def submit_host(self, samples: np.ndarray):
        want = {INPUT_IQ16: np.int16, INPUT_IQ8: np.int8}.get(self.input_kind, np.float32)
        if want is not np.float32 and np.asarray(samples).dtype != want:
            raise SondeError(f"samples must be {np.dtype(want).name} for this input kind, got {np.asarray(samples).dtype} (no silent conversion)")
        samples = np.ascontiguousarray(samples, dtype=want)
        n = samples.shape[1]
        self._chk(self.L.sonde_batch_submit_host(self.h, samples.ctypes.data_as(C.c_void_p), n, n))

    def wait_input(self, stream: int | None = None):
        """Device-side: work queued on `stream` after this call may overwrite the last submit's sample buffer (sonde_batch_wait_input)."""
        self._chk(self.L.sonde_batch_wait_input(self.h, C.c_void_p(stream or 0)))

    def sync(self) -> int:
        return self._chk(self.L.sonde_batch_sync(self.h))

    def frames(self) -> np.ndarray:
        n = self.sync()
        out = np.zeros(n, dtype=FRAME_DTYPE)
        if n:
            got = self._chk(self.L.sonde_batch_frames(self.h, out.ctypes.data_as(C.c_void_p), n))
            out = out[:got]
        return out

    def ticket(self) -> int:
        """Number of the last submit (1-based)."""
        return int(self.L.sonde_batch_ticket(self.h))

    def frames_of(self, ticket: int) -> np.ndarray:
        """Frames of submit `ticket` (one of the last two): waits for that submit only, not for later ones."""
        n = self._chk(self.L.sonde_batch_frames_of(self.h, ticket, None, 0))       # count first: the buffer is sized from it
        out = np.zeros(n, dtype=FRAME_DTYPE)
        if n:
            out = out[:self._chk(self.L.sonde_batch_frames_of(self.h, ticket, out.ctypes.data_as(C.c_void_p), n))]
        return out

    def overflow(self) -> int:
        return self._chk(self.L.sonde_batch_overflow(self.h))

    def poll(self, cap: int = 4096):
        """SondeData fragments of the last submit with their channels: list of (channel, SondeData)."""
        out = (_lib.SondeData * cap)()
        chan = (C.c_uint32 * cap)()
        res = []
        while True:
            n = self._chk(self.L.sonde_batch_poll(self.h, out, chan, cap))
            if n == 0:
                return res
            for i in range(n):
                d = _lib.SondeData()
                C.memmove(C.byref(d), C.byref(out[i]), C.sizeof(d))
                res.append((int(chan[i]), d))

    def kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.sonde_batch_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def class_ms(self) -> dict:
        """Mixed batches: {class index: average ms of that demodulator class's kernel alone} over the timed submits
        (sonde_batch_class_ms); {} for a batch of one class."""
        v = (C.c_float * 4)()
        self.L.sonde_batch_class_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        n = self._chk(self.L.sonde_batch_class_ms(self.h, v))
        return {k: float(v[k]) for k in range(4) if n > 0 and v[k] >= 0.0}

    def launch_info(self) -> dict:
        """{'units': launch units per submit, 'join': 0 at every submit (default) / 1 one submit late (FLAG_LATE_JOIN) / 2 never (FLAG_PIPELINE)} (sonde_batch_launch_info)"""
        u, j = C.c_uint32(), C.c_int32()
        self.L.sonde_batch_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        self._chk(self.L.sonde_batch_launch_info(self.h, C.byref(u), C.byref(j)))
        return {"units": int(u.value), "join": int(j.value)}

    def set_timing(self, every_n: int):
        """Record kernel-timing events on every n-th submit only (0: never); the next submit is timed."""
        self._chk(self.L.sonde_batch_set_timing(self.h, int(every_n)))

    def nbits(self, channel: int) -> int:
        return int(self.L.sonde_batch_nbits(self.h, channel))

    def read_bits(self, channel: int, start: int, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=np.uint8)
        self._chk(self.L.sonde_batch_read_bits(self.h, channel, start, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def state(self, channel: int) -> dict:
        t, p = C.c_int64(), C.c_int32()
        b, a, y = C.c_float(), C.c_float(), C.c_float()
        self._chk(self.L.sonde_batch_read_state(self.h, channel, C.byref(t), C.byref(p), C.byref(b), C.byref(a), C.byref(y)))
        # afc_u: the newest AFC state (SPEC 3.0b); `yprev` is the key's name of rounds 1-3 (the same value), kept for the oracle binding's sake
        return dict(t_next=t.value, period=p.value, bias=b.value, amp=a.value, afc_u=y.value, yprev=y.value)


def row_stride(n_samples: int, iq: bool = True, kind: int | None = None) -> int:
    """Channel stride (in samples) the library recommends for rows of n_samples: the next power of two in bytes (HBM channel
    spread of rows streamed side by side; include/sonde_abi.h sonde_row_stride).  kind: an INPUT_* value (overrides iq)."""
    if kind is None:
        kind = INPUT_IQ if iq else INPUT_REAL
    return int(_lib.load().sonde_row_stride(int(n_samples), kind))


def strided_rows(x, stride: int | None = None):
    """A copy of the device tensor x = [C, n, 2] (or [C, n]) whose rows lie `stride` samples apart (default: row_stride), as a
    view of the padded allocation: what submit() takes."""
    import torch
    n = x.shape[1]
    st = stride or row_stride(n, kind={torch.int16: INPUT_IQ16, torch.int8: INPUT_IQ8}.get(x.dtype, INPUT_IQ) if x.dim() == 3 else INPUT_REAL)
    if st == n:
        return x
    buf = torch.empty((x.shape[0], st) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    buf[:, :n] = x
    return buf[:, :n]


class SondeChannelizer:
    """Wideband front-end (BASELINE config 4): n_streams x 10 MS/s complex IQ -> 512 bins each (20 kS/s, one phase sample per
    step) -> per-bin decode; every stage is one launch over all streams.  submit() takes [samples_per_submit, 2] (one stream) or
    [n_streams, samples_per_submit, 2].  Bins carry the 12 kS/s sondes (RS41, DFM, iMS-100, MRZ-N1) and, unfused, the AFSK
    sondes; an M10 channel (50 kHz wide) does not fit a 19.5 kHz bin: use SondeVfo for those."""

    def __init__(self, types=None, blocks_per_submit: int = 1, device: int = 0, n_streams: int = 1, fused: bool | None = None, overlap: bool | None = None,
                 dual: bool = False, input_kind: int = INPUT_IQ):
        """dual: both stackings of every stream (SPEC 3.5c): 1024 channels per stream, 1024 p + k = even bin k (centre k x 19531.25 Hz),
        1024 p + 512 + k = odd bin k (centre (k + 1/2) x 19531.25 Hz): every carrier lies within 4.9 kHz of a bin centre."""
        self.L = _lib.load()
        self._types = None
        self.n_streams = int(n_streams)
        self.dual = bool(dual)
        self.n_channels = (1024 if dual else 512) * self.n_streams
        tp = None
        if types is not None:
            self._types = np.ascontiguousarray(types, dtype=np.uint8)
            assert self._types.shape == (self.n_channels,)
            tp = self._types.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        create = self.L.sonde_chan_create_dual if dual else self.L.sonde_chan_create_multi
        if create(tp, blocks_per_submit, self.n_streams, device, C.byref(h)) != 0:
            raise SondeError(_lib.last_error() or "sonde_chan_create failed")
        self.h = h
        # fused (default where possible): discriminator + resampler inside the decoder kernel; fused=False keeps the 48 kS/s rows
        # (read()); fused=None leaves the library's choice alone
        self.fused = bool(self.L.sonde_chan_set_fused(self.h, 1 if fused else 0)) if fused is not None else bool(self.L.sonde_chan_set_fused(self.h, -1))
        # overlap (an option, off by default): filter bank of submit k+1 beside the decoder of submit k, on internal streams
        self.overlap = bool(self.L.sonde_chan_set_overlap(self.h, 1)) if overlap else False
        # input_kind: INPUT_IQ (complex64 blocks), INPUT_IQ16 (int16 I, Q pairs) or INPUT_IQ8 (int8 pairs): the receiver's own format
        self.input_kind = int(self.L.sonde_chan_set_input(self.h, input_kind))
        if self.input_kind != input_kind:
            raise SondeError("sonde_chan_set_input: input kind not accepted")
        self.samples_per_submit = int(self.L.sonde_chan_samples_per_submit(self.h))
        self.n_steps = self.samples_per_submit // 500
        self.batch = SondeBatch.__new__(SondeBatch)          # borrowed view of the embedded 512-channel batch
        self.batch.L = self.L
        self.batch.h = C.c_void_p(self.L.sonde_chan_batch(self.h))
        self.batch.n_channels = self.n_channels
        self.batch.close = lambda: None

    def submit(self, iq, stream: int | None = None):
        assert tuple(iq.shape) in ((self.samples_per_submit, 2), (self.n_streams, self.samples_per_submit, 2)) and iq.is_contiguous()
        assert self.n_streams == 1 or iq.dim() == 3
        want = {INPUT_IQ16: "int16", INPUT_IQ8: "int8"}.get(self.input_kind, "float32")
        if not str(iq.dtype).endswith(want) or str(iq.dtype).endswith("u" + want):
            raise SondeError(f"wideband block must be {want}, got {iq.dtype}")
        self._keep = iq
        if self.L.sonde_chan_submit(self.h, C.c_void_p(iq.data_ptr()), self.samples_per_submit, C.c_void_p(stream or 0)) != 0:
            raise SondeError(_lib.last_error() or "sonde_chan_submit failed")

    def frames(self) -> np.ndarray:
        return SondeBatch.frames(self.batch)

    def kernel_ms(self):
        """(filter bank, discriminator + resampler, demodulator, framers) average ms over the timed submits."""
        v = [C.c_float() for _ in range(4)]
        if self.L.sonde_chan_kernel_ms(self.h, *[C.byref(x) for x in v]) != 0:
            raise SondeError(_lib.last_error() or "sonde_chan_kernel_ms failed")
        return tuple(x.value for x in v)

    def read(self):
        """(phases [bins, n_steps] in quadrants, 48 kS/s rows [bins, n_steps * 12 / 5] or None in fused mode) of the last submit."""
        bins = np.zeros((self.n_channels, self.n_steps), dtype=np.float32)
        out48 = None if self.fused else np.zeros((self.n_channels, self.n_steps * 12 // 5), dtype=np.float32)
        if self.L.sonde_chan_read(self.h, bins.ctypes.data_as(C.c_void_p), out48.ctypes.data_as(C.c_void_p) if out48 is not None else None) != 0:
            raise SondeError("sonde_chan_read failed")
        return bins, out48

    def close(self):
        if getattr(self, "h", None):
            self.batch.h = None
            self.L.sonde_chan_destroy(self.h)
            self.h = None

    __del__ = close


def get_taps(sonde_type: int) -> np.ndarray:
    out = np.zeros((32, 32), dtype=np.float32)
    if _lib.load().sonde_get_taps(sonde_type, out.ctypes.data_as(C.c_void_p)) != 0:
        raise SondeError(_lib.last_error())
    return out


# the reference's VFO bandwidth = sample rate per sonde type (/root/reference/src/main.hpp:44-52)
VFO_RATE = {0: 10000, 1: 15000, 2: 20000, 3: 50000, 4: 20000, 5: 20000, 6: 20000}


class SondeVfo:
    """VFO front-end (SURVEY 8 a1 + a2): IQ at the sonde type's VFO rate -> discriminator -> rational resampler -> 48 kS/s
    FM audio rows for a SondeBatch created with INPUT_REAL (/root/reference/src/main.cpp:55-60)."""

    def __init__(self, n_channels: int, rate_in: int, max_in: int, device: int = 0):
        self.L = _lib.load()
        h = C.c_void_p()
        if self.L.sonde_vfo_create(n_channels, rate_in, max_in, device, C.byref(h)) != 0:
            raise SondeError(_lib.last_error() or "sonde_vfo_create failed")
        self.h = h
        self.n_channels, self.rate_in, self.max_in = n_channels, rate_in, max_in
        up, down = C.c_int(), C.c_int()
        self.L.sonde_vfo_ratio(rate_in, C.byref(up), C.byref(down))
        self.up, self.down = up.value, down.value

    def out_samples(self, n_in: int) -> int:
        return int(self.L.sonde_vfo_out_samples(self.h, n_in))

    def process(self, iq, out=None, stream: int | None = None):
        """iq: CUDA float32 tensor [C, n_in, 2] (rows may be strided); returns the [C, n_out] float32 rows."""
        import torch
        assert iq.is_cuda and iq.dtype == torch.float32 and iq.shape[0] == self.n_channels and iq.shape[2] == 2 and iq.stride(1) == 2
        n_in = iq.shape[1]
        n_out = self.out_samples(n_in)
        if out is None:
            out = torch.empty((self.n_channels, n_out), dtype=torch.float32, device=iq.device)
        if stream is None:
            stream = torch.cuda.current_stream(iq.device).cuda_stream
        if self.L.sonde_vfo_process(self.h, C.c_void_p(iq.data_ptr()), n_in, iq.stride(0) // 2, C.c_void_p(out.data_ptr()), out.stride(0),
                                    C.c_void_p(stream)) != 0:
            raise SondeError(_lib.last_error() or "sonde_vfo_process failed")
        return out

    def taps(self) -> np.ndarray:
        g = np.zeros((self.up, 16), dtype=np.float32)
        if self.L.sonde_vfo_taps(self.rate_in, g.ctypes.data_as(C.c_void_p)) != 0:
            raise SondeError(_lib.last_error() or "sonde_vfo_taps failed")
        return g

    def close(self):
        if getattr(self, "h", None):
            self.L.sonde_vfo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


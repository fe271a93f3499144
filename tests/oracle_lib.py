"""ctypes binding of the CPU oracle (oracle/libsonde_oracle.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
FRAME_MAX = 528


class OrFrame(C.Structure):
    _fields_ = [("channel", C.c_uint32), ("type", C.c_uint32), ("len", C.c_int32), ("nerr", C.c_int32 * 2),
                ("flags", C.c_uint32), ("bitpos", C.c_uint64), ("data", C.c_uint8 * FRAME_MAX)]


FRAME_DTYPE = np.dtype([("channel", "<u4"), ("type", "<u4"), ("len", "<i4"), ("nerr", "<i4", (2,)),
                        ("flags", "<u4"), ("bitpos", "<u8"), ("data", "u1", (FRAME_MAX,))])
assert FRAME_DTYPE.itemsize == C.sizeof(OrFrame), (FRAME_DTYPE.itemsize, C.sizeof(OrFrame))

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "libsonde_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    L.or_recip.restype = C.c_float
    L.or_recip.argtypes = [C.c_float]
    L.or_atan2.restype = C.c_float
    L.or_atan2.argtypes = [C.c_float, C.c_float]
    L.or_discriminate.argtypes = [f32p, C.c_size_t, f32p, f32p]
    L.or_modem.restype = C.c_void_p
    L.or_make_taps.argtypes = [C.c_void_p, f32p]
    L.or_demod_new.restype = C.c_void_p
    L.or_demod_new.argtypes = [C.c_int]
    L.or_demod_free.argtypes = [C.c_void_p]
    L.or_demod_feed.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_int]
    L.or_demod_nbits.restype = C.c_uint64
    L.or_demod_nbits.argtypes = [C.c_void_p]
    L.or_demod_getbits.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, u8p]
    L.or_demod_state.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), f32p, f32p, f32p]
    L.or_gf256_mul.restype = C.c_uint8
    L.or_gf256_mul.argtypes = [C.c_uint8, C.c_uint8]
    L.or_rs255_decode.restype = C.c_int
    L.or_rs255_decode.argtypes = [u8p, C.c_int]
    L.or_rs255_encode.argtypes = [u8p, C.c_int]
    L.or_crc16_ccitt.restype = C.c_uint16
    L.or_crc16_ccitt.argtypes = [u8p, C.c_size_t]
    L.or_channel_new.restype = C.c_void_p
    L.or_channel_new.argtypes = [C.c_int, C.c_uint32]
    L.or_channel_free.argtypes = [C.c_void_p]
    L.or_channel_feed.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_int]
    L.or_channel_nframes.restype = C.c_size_t
    L.or_channel_nframes.argtypes = [C.c_void_p]
    L.or_channel_frame.restype = C.POINTER(OrFrame)
    L.or_channel_frame.argtypes = [C.c_void_p, C.c_size_t]
    L.or_channel_demod.restype = C.c_void_p
    L.or_channel_demod.argtypes = [C.c_void_p]
    L.or_batch_run.restype = C.c_size_t
    L.or_batch_run.argtypes = [C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    L.or_chan_new.restype = C.c_void_p
    L.or_chan_new_odd.restype = C.c_void_p
    L.or_chan_twist.argtypes = [f32p]
    L.or_chan_ramp.restype = C.c_float
    L.or_chan_ramp.argtypes = [C.c_size_t]
    L.or_chan_free.argtypes = [C.c_void_p]
    L.or_chan_block.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, f32p]
    L.or_chan_block2.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, f32p, C.c_void_p, f32p]
    L.or_chan_composite_taps.argtypes = [f32p, C.c_int, f32p]
    L.or_chan_proto.argtypes = [f32p]
    L.or_chan_twiddles.argtypes = [f32p]
    L.or_chan_resamp_taps.argtypes = [f32p]
    L.or_vfo_new.restype = C.c_void_p
    L.or_vfo_new.argtypes = [C.c_int]
    L.or_vfo_free.argtypes = [C.c_void_p]
    L.or_vfo_process.restype = C.c_size_t
    L.or_vfo_process.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p]
    L.or_vfo_ratio.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.or_resamp_taps.argtypes = [C.c_int, C.c_double, C.c_double, f32p]
    L.or_fft512.argtypes = [f32p, f32p, f32p]
    L.or_dewpt.restype = C.c_float
    L.or_dewpt.argtypes = [C.c_float, C.c_float]
    L.or_altitude_to_pressure.restype = C.c_float
    L.or_altitude_to_pressure.argtypes = [C.c_float]
    L.or_rs41_temp.restype = C.c_float
    L.or_rs41_temp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, f32p, f32p]
    L.or_rs41_rh.restype = C.c_float
    L.or_rs41_rh.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
    L.or_dfm_temp.restype = C.c_float
    L.or_dfm_temp.argtypes = [C.c_float, C.c_float, C.c_float]
    L.or_rs41_pressure_d.restype = C.c_double
    L.or_rs41_pressure_d.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, f32p]
    L.or_ozone_mpa_d.restype = C.c_double
    L.or_ozone_mpa_d.argtypes = [C.c_double, C.c_double]
    L.or_m10_temp_d.restype = C.c_double
    L.or_m10_temp_d.argtypes = [C.c_uint, C.c_uint]
    L.or_m10_rh_d.restype = C.c_double
    L.or_m10_rh_d.argtypes = [C.c_uint32, C.c_uint32, C.c_double]
    L.or_m20_temp_d.restype = C.c_double
    L.or_m20_temp_d.argtypes = [C.c_uint]
    L.or_ims100_temp_d.restype = C.c_double
    L.or_ims100_temp_d.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_double]
    L.or_modem_set_decim.argtypes = [C.c_int, C.c_int]
    L.or_afsk_table.argtypes = [f32p]
    L.or_imet_crc.restype = C.c_uint16
    L.or_imet_crc.argtypes = [C.POINTER(C.c_uint8), C.c_size_t]
    L.or_yard_batch_run.restype = C.c_size_t
    L.or_yard_batch_run.argtypes = [C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t]
    L.or_gf256_init()
    _lib = L
    return L


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u8ptr(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def batch_run(sonde_type: int, iq: np.ndarray, nthreads: int = 1, cap_per_channel: int = 0) -> np.ndarray:
    """iq: [C, n, 2] float32.  Returns structured array of frames (FRAME_DTYPE), ordered by channel then time."""
    L = lib()
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    nch, n = iq.shape[0], iq.shape[1]
    cap = nch * (cap_per_channel or (n // 4096 + 8))      # shortest frame: DFM, 560 chips = 5376 samples
    out = np.zeros(cap, dtype=FRAME_DTYPE)
    total = L.or_batch_run(sonde_type, fptr(iq.reshape(-1)), nch, n, nthreads, out.ctypes.data, cap)
    assert total <= cap
    return out[:total]


def yard_run(sonde_type: int, iq: np.ndarray, nthreads: int = 1, cutoff_rel: float = 0.0, loop_bw: float = 0.0, cap_per_channel: int = 0) -> np.ndarray:
    """The conventional yardstick receiver (oracle/or_yardstick.c) over iq = [C, n, 2]: frames in (channel, time) order."""
    L = lib()
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    nch, n = iq.shape[0], iq.shape[1]
    cap = nch * (cap_per_channel or (n // 4096 + 8))
    out = np.zeros(cap, dtype=FRAME_DTYPE)
    total = L.or_yard_batch_run(sonde_type, fptr(iq.reshape(-1)), nch, n, nthreads, cutoff_rel, loop_bw, out.ctypes.data, cap)
    assert total <= cap
    return out[:total]


class Channel:
    """One oracle channel with streaming feed."""

    def __init__(self, sonde_type: int = 0, channel: int = 0):
        self.L = lib()
        self.h = self.L.or_channel_new(sonde_type, channel)

    def feed(self, x: np.ndarray, is_iq: bool = True):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        n = x.size // 2 if is_iq else x.size
        self.L.or_channel_feed(self.h, fptr(x), n, 1 if is_iq else 0)

    def feed_decimated(self, z: np.ndarray, dec: int):
        """Real input already decimated by the type's factor `dec` (SPEC 3.5b: the channelizer's composite filter)."""
        z = np.ascontiguousarray(z, dtype=np.float32).reshape(-1)
        self.L.or_channel_feed(self.h, fptr(z), z.size * dec, 2)

    def frames(self) -> np.ndarray:
        n = self.L.or_channel_nframes(self.h)
        out = np.zeros(n, dtype=FRAME_DTYPE)
        for i in range(n):
            C.memmove(out.ctypes.data + i * FRAME_DTYPE.itemsize, self.L.or_channel_frame(self.h, i), FRAME_DTYPE.itemsize)
        return out

    def bits(self) -> np.ndarray:
        d = self.L.or_channel_demod(self.h)
        n = self.L.or_demod_nbits(d)
        out = np.zeros(n, dtype=np.uint8)
        if n:
            self.L.or_demod_getbits(d, 0, n, u8ptr(out))
        return out

    def state(self):
        d = self.L.or_channel_demod(self.h)
        t, p = C.c_int64(), C.c_int32()
        b, a, y = C.c_float(), C.c_float(), C.c_float()
        self.L.or_demod_state(d, C.byref(t), C.byref(p), C.byref(b), C.byref(a), C.byref(y))
        return dict(t_next=t.value, period=p.value, bias=b.value, amp=a.value, afc_u=y.value, yprev=y.value)

    def __del__(self):
        try:
            self.L.or_channel_free(self.h)
        except Exception:
            pass


class Vfo:
    """One channel of the oracle's VFO front-end (or_chan.c or_vfo_*): IQ at the VFO rate -> 48 kS/s FM audio."""

    def __init__(self, rate_in: int):
        self.L = lib()
        self.h = self.L.or_vfo_new(rate_in)
        assert self.h, rate_in
        up, down, fc = C.c_int(), C.c_int(), C.c_int()
        self.L.or_vfo_ratio(rate_in, C.byref(up), C.byref(down), C.byref(fc))
        self.up, self.down, self.cutoff = up.value, down.value, fc.value
        self.rate_in = rate_in

    def process(self, iq: np.ndarray) -> np.ndarray:
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        n_in = iq.shape[0]
        out = np.zeros(n_in * self.up // self.down, dtype=np.float32)
        n = self.L.or_vfo_process(self.h, fptr(iq), n_in, fptr(out))
        assert n == out.shape[0]
        return out

    def taps(self) -> np.ndarray:
        g = np.zeros((self.up, 16), dtype=np.float32)
        self.L.or_resamp_taps(self.up, float(self.rate_in) * self.up, float(self.cutoff), fptr(g))
        return g

    def __del__(self):
        if getattr(self, "h", None):
            self.L.or_vfo_free(self.h)
            self.h = None


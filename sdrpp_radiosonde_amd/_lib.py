"""ctypes binding of libsonde_mi355.so (include/sonde_abi.h).

The HIP library is the product: if it is missing or cannot be loaded this module raises --
there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# SONDE_MI355_LIB: developer override used for A/B timing of kernel variants (still a HIP build of this tree)
LIB_PATH = os.environ.get("SONDE_MI355_LIB") or os.path.join(PKG_DIR, "libsonde_mi355.so")

TILE = 2048
FRAME_MAX = 528
(RS41, DFM09, IMS100, M10, IMET4, C50, MRZN1) = range(7)
INPUT_IQ, INPUT_REAL, INPUT_IQ16, INPUT_IQ8 = 0, 1, 2, 3
PROCEED, PARSED = 0, 1
DATA_SEQ, DATA_POS, DATA_SPEED, DATA_TIME, DATA_PTU, DATA_SERIAL, DATA_SHUTDOWN, DATA_OZONE = (1 << i for i in range(8))


class SondeFrame(C.Structure):
    _fields_ = [("channel", C.c_uint32), ("type", C.c_uint32), ("len", C.c_int32), ("nerr", C.c_int32 * 2),
                ("flags", C.c_uint32), ("bitpos", C.c_uint64), ("data", C.c_uint8 * FRAME_MAX)]


FRAME_DTYPE = np.dtype([("channel", "<u4"), ("type", "<u4"), ("len", "<i4"), ("nerr", "<i4", (2,)),
                        ("flags", "<u4"), ("bitpos", "<u8"), ("data", "u1", (FRAME_MAX,))])
assert FRAME_DTYPE.itemsize == C.sizeof(SondeFrame)


class SondeData(C.Structure):
    _fields_ = [("fields", C.c_int), ("seq", C.c_int), ("lat", C.c_float), ("lon", C.c_float), ("alt", C.c_float),
                ("speed", C.c_float), ("heading", C.c_float), ("climb", C.c_float), ("time", C.c_int64),
                ("calib_percent", C.c_float), ("temp", C.c_float), ("rh", C.c_float), ("pressure", C.c_float),
                ("serial", C.c_char * 32), ("shutdown", C.c_int), ("o3_mpa", C.c_float)]


class SondeBatchConfig(C.Structure):
    _fields_ = [("n_channels", C.c_uint32), ("types", C.POINTER(C.c_uint8)), ("max_samples", C.c_uint32),
                ("input_kind", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32), ("launch_units", C.c_uint32),
                ("struct_size", C.c_uint32), ("time_slices", C.c_uint32)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(SondeBatchConfig)      # SONDE_BATCH_CONFIG_INIT: sonde_batch_create refuses any other value


FLAG_WIDE = 1            # one decimation step less for every GFSK sonde (SONDE_FLAG_WIDE)
FLAG_RS41_WIDE = FLAG_WIDE
FLAG_SPLIT_FEC = 2
FLAG_PIPELINE = 4        # launch units never joined into the caller's stream (SONDE_FLAG_PIPELINE; opt-in)
FLAG_JOIN = 16           # accepted and ignored: joining at every submit is the default since round 6 (SONDE_FLAG_JOIN)
FLAG_LATE_JOIN = 32      # launch units joined into the caller's stream ONE SUBMIT LATE (SONDE_FLAG_LATE_JOIN; opt-in: round 5's default)
FLAG_WIDE_AUTO = 8       # SONDE_FLAG_WIDE for the types whose reference channel is >= 20 kHz only (iMS-100, MRZ-N1, M10)


# every symbol include/sonde_abi.h declares; tests check the .so exports all of them
ABI_SYMBOLS = [
    "sonde_batch_create", "sonde_batch_destroy", "sonde_batch_submit", "sonde_batch_submit_host",
    "sonde_row_stride", "sonde_sample_bytes", "sonde_batch_wait_input", "sonde_batch_sync", "sonde_batch_frames", "sonde_batch_frames_of", "sonde_batch_ticket", "sonde_batch_overflow", "sonde_batch_kernel_ms", "sonde_batch_set_timing", "sonde_batch_class_ms", "sonde_batch_read_bits",
    "sonde_batch_launch_info", "sonde_batch_nbits", "sonde_batch_read_state", "sonde_batch_test_rs255", "sonde_batch_poll", "sonde_get_taps", "sonde_get_afsk_table", "sonde_parse_frame",
    "sonde_parser_create", "sonde_parser_feed", "sonde_parser_destroy", "sonde_rs41_temp", "sonde_rs41_rh", "sonde_dfm_temp", "sonde_rs41_pressure", "sonde_ozone_mpa",
    "sonde_m10_temp", "sonde_m10_rh", "sonde_m20_temp", "sonde_ims100_temp",
    "sonde_last_error", "sonde_version", "sonde_hbm_read_probe", "sonde_dewpt", "sonde_altitude_to_pressure",
    "sonde_gpx_open", "sonde_gpx_close", "sonde_gpx_start_track", "sonde_gpx_stop_track", "sonde_gpx_add_point",
    "sonde_ptu_open", "sonde_ptu_close", "sonde_ptu_add_point",
    "sonde_chan_create", "sonde_chan_create_multi", "sonde_chan_create_dual", "sonde_chan_streams", "sonde_chan_channels", "sonde_chan_set_fused", "sonde_chan_set_input", "sonde_chan_set_overlap", "sonde_chan_destroy", "sonde_chan_samples_per_submit", "sonde_chan_submit", "sonde_chan_batch",
    "sonde_chan_read", "sonde_chan_tables", "sonde_chan_kernel_ms",
    "sonde_vfo_create", "sonde_vfo_destroy", "sonde_vfo_ratio", "sonde_vfo_out_samples", "sonde_vfo_process", "sonde_vfo_process_host", "sonde_vfo_taps",
] + [f"{x}_{fn}" for x in ("rs41", "dfm09", "ims100", "m10", "imet4", "c50", "mrzn1")
     for fn in ("decoder_init", "decoder_deinit", "decode")]

_lib = None


def load() -> C.CDLL:
    """Load libsonde_mi355.so; raises if it was not built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: build the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU); there is no
        # CPU fallback -- if this fails, so does everything that needs the library
        import subprocess
        csrc = os.path.join(os.path.dirname(LIB_PATH), "csrc")
        try:
            subprocess.check_call(["make", "-s", "-C", csrc])
        except (OSError, subprocess.CalledProcessError) as e:
            raise RuntimeError(f"{LIB_PATH} is missing and `make -C {csrc}` failed ({e}); there is no CPU fallback") from e
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sonde_hbm_read_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    L.sonde_hbm_read_probe.restype = C.c_int
    L.sonde_last_error.restype = C.c_char_p
    L.sonde_version.restype = C.c_char_p
    L.sonde_batch_create.argtypes = [C.POINTER(SondeBatchConfig), C.POINTER(vp)]
    L.sonde_batch_destroy.argtypes = [vp]
    L.sonde_batch_destroy.restype = None
    L.sonde_batch_submit.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    L.sonde_batch_submit_host.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.sonde_batch_sync.argtypes = [vp]
    L.sonde_batch_sync.restype = C.c_long
    if hasattr(L, "sonde_batch_wait_input"):              # absent only in older A/B builds loaded through SONDE_MI355_LIB
        L.sonde_batch_wait_input.argtypes = [vp, vp]
    L.sonde_batch_frames.argtypes = [vp, vp, C.c_size_t]
    L.sonde_batch_frames.restype = C.c_long
    if hasattr(L, "sonde_batch_frames_of"):
        L.sonde_batch_frames_of.argtypes = [vp, C.c_uint64, vp, C.c_size_t]
        L.sonde_batch_frames_of.restype = C.c_long
        L.sonde_batch_ticket.argtypes = [vp]
        L.sonde_batch_ticket.restype = C.c_uint64
        L.sonde_batch_overflow.argtypes = [vp]
        L.sonde_batch_overflow.restype = C.c_long
    L.sonde_batch_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    if hasattr(L, "sonde_batch_set_timing"):          # absent only in older A/B builds loaded through SONDE_MI355_LIB
        L.sonde_batch_set_timing.argtypes = [vp, C.c_int]
    L.sonde_batch_read_bits.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_size_t, vp]
    L.sonde_batch_nbits.argtypes = [vp, C.c_uint32]
    L.sonde_batch_nbits.restype = C.c_uint64
    L.sonde_batch_read_state.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    if hasattr(L, "sonde_batch_test_rs255"):
        L.sonde_batch_test_rs255.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.sonde_get_taps.argtypes = [C.c_int, vp]
    L.sonde_parse_frame.argtypes = [vp, C.POINTER(SondeData), C.c_int]
    L.sonde_batch_poll.argtypes = [vp, C.POINTER(SondeData), C.POINTER(C.c_uint32), C.c_size_t]
    L.sonde_batch_poll.restype = C.c_long
    L.sonde_parser_create.argtypes = [C.c_int]
    L.sonde_parser_create.restype = vp
    L.sonde_parser_feed.argtypes = [vp, vp, C.POINTER(SondeData), C.c_int]
    L.sonde_parser_feed.restype = C.c_int
    L.sonde_parser_destroy.argtypes = [vp]
    L.sonde_parser_destroy.restype = None
    L.sonde_rs41_pressure.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_float)]
    L.sonde_rs41_pressure.restype = C.c_float
    L.sonde_ozone_mpa.argtypes = [C.c_float, C.c_float]
    L.sonde_ozone_mpa.restype = C.c_float
    L.sonde_m10_temp.argtypes = [C.c_uint, C.c_uint]
    L.sonde_m10_temp.restype = C.c_float
    L.sonde_m10_rh.argtypes = [C.c_uint32, C.c_uint32, C.c_float]
    L.sonde_m10_rh.restype = C.c_float
    L.sonde_m20_temp.argtypes = [C.c_uint]
    L.sonde_m20_temp.restype = C.c_float
    L.sonde_ims100_temp.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float]
    L.sonde_ims100_temp.restype = C.c_float
    L.sonde_rs41_temp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.sonde_rs41_temp.restype = C.c_float
    L.sonde_rs41_rh.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
    L.sonde_rs41_rh.restype = C.c_float
    L.sonde_dfm_temp.argtypes = [C.c_float, C.c_float, C.c_float]
    L.sonde_dfm_temp.restype = C.c_float
    L.sonde_dewpt.restype = C.c_float
    L.sonde_dewpt.argtypes = [C.c_float, C.c_float]
    L.sonde_altitude_to_pressure.restype = C.c_float
    L.sonde_altitude_to_pressure.argtypes = [C.c_float]
    L.sonde_chan_create.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp)]
    L.sonde_chan_create_multi.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
    L.sonde_chan_set_fused.argtypes = [vp, C.c_int]
    L.sonde_chan_set_overlap.argtypes = [vp, C.c_int]
    L.sonde_row_stride.argtypes = [C.c_size_t, C.c_int]
    L.sonde_row_stride.restype = C.c_size_t
    L.sonde_sample_bytes.argtypes = [C.c_int]
    L.sonde_sample_bytes.restype = C.c_size_t
    L.sonde_chan_streams.argtypes = [vp]
    L.sonde_chan_streams.restype = C.c_uint32
    L.sonde_chan_destroy.argtypes = [vp]
    L.sonde_chan_destroy.restype = None
    L.sonde_chan_samples_per_submit.argtypes = [vp]
    L.sonde_chan_samples_per_submit.restype = C.c_uint32
    L.sonde_chan_submit.argtypes = [vp, vp, C.c_size_t, vp]
    L.sonde_chan_kernel_ms.argtypes = [vp] + [C.POINTER(C.c_float)] * 4
    L.sonde_chan_batch.argtypes = [vp]
    L.sonde_chan_batch.restype = vp
    L.sonde_chan_read.argtypes = [vp, vp, vp]
    L.sonde_chan_tables.argtypes = [vp, vp, vp]
    if hasattr(L, "sonde_vfo_create"):
        L.sonde_vfo_create.argtypes = [C.c_uint32, C.c_int, C.c_size_t, C.c_int, C.POINTER(vp)]
        L.sonde_vfo_destroy.argtypes = [vp]
        L.sonde_vfo_destroy.restype = None
        L.sonde_vfo_ratio.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sonde_vfo_out_samples.argtypes = [vp, C.c_size_t]
        L.sonde_vfo_out_samples.restype = C.c_size_t
        L.sonde_vfo_process.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp]
        L.sonde_vfo_process_host.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.sonde_vfo_taps.argtypes = [C.c_int, vp]
    f = C.c_float
    L.sonde_gpx_open.restype = vp
    L.sonde_gpx_open.argtypes = [C.c_char_p]
    L.sonde_gpx_close.argtypes = [vp]
    L.sonde_gpx_start_track.argtypes = [vp, C.c_char_p]
    L.sonde_gpx_stop_track.argtypes = [vp]
    L.sonde_gpx_add_point.argtypes = [vp, C.c_long, f, f, f, f, f]
    L.sonde_ptu_open.restype = vp
    L.sonde_ptu_open.argtypes = [C.c_char_p]
    L.sonde_ptu_close.argtypes = [vp]
    L.sonde_ptu_add_point.argtypes = [vp, C.c_long, f, f, f, f, f, f, f, f, f, f, C.c_char_p]
    for x in ("rs41", "dfm09", "ims100", "m10", "imet4", "c50", "mrzn1"):
        getattr(L, f"{x}_decoder_init").argtypes = [C.c_int]
        getattr(L, f"{x}_decoder_init").restype = vp
        getattr(L, f"{x}_decoder_deinit").argtypes = [vp]
        getattr(L, f"{x}_decoder_deinit").restype = None
        getattr(L, f"{x}_decode").argtypes = [vp, C.POINTER(SondeData), vp, C.c_size_t]
        getattr(L, f"{x}_decode").restype = C.c_int
    _lib = L
    return L


def last_error() -> str:
    return load().sonde_last_error().decode()

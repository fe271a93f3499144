"""A scripted session for the GPX / CSV sinks, runnable against the reference's classes (oracle/_ref)
and against the product's (libsonde_mi355.so); returns the file content after every operation."""
import ctypes as C
import os
import tempfile

NAN = float("nan")
GPX_OPS = [
    ("point", 1700000000, 47.5, 8.25, 1000.0, 12.0, 90.0),      # before any track: dropped
    ("stop",),                                                    # nothing open
    ("start", "S12 45"),                                          # space is not isgraph: ignored
    ("start", "S1234567"),
    ("start", "S1234567"),                                        # repeat: no-op
    ("point", 1700000000, 47.5, 8.25, 1000.0, 12.0, 90.0),
    ("point", 1700000000, 47.6, 8.26, 1005.0, 12.0, 90.0),       # same time: dropped
    ("point", 1700000001, 47.5, 8.25, 1000.0, 12.0, 90.0),       # same position: dropped
    ("point", 1700000001, NAN, 8.25, 1000.0, 12.0, 90.0),
    ("point", 1700000002, 0.0, 0.0, 0.0, 1.0, 2.0),              # all-zero position: dropped
    ("point", 1700000003, 47.500123, 8.250456, 1015.5, 11.25, 271.125),
    ("point", 1700003604, -33.865143, 151.2099, 35000.25, 0.0, 359.99),
    ("start", "T7654321"),                                        # new name closes the old track
    ("point", 1700003605, 10.0, -20.0, 1.5, 3.0, 4.0),
    ("stop",),
    ("point", 1700003606, 11.0, -21.0, 2.5, 3.0, 4.0),           # no open track: dropped
    ("start", "U1"),
    ("point", 86399, 0.000001, 0.0, 0.0, 0.0, 0.0),
]
PTU_OPS = [
    (1700000000, -50.04, 30.05, -59.7688, 193.3049, 47.5001234, 8.2504567, 12000.44, 12.04, 90.06, 5.05, ""),
    (1700000001, 10.0, 50.0, 0.1, 900.0, -33.8651432, 151.2099, 35000.25, 0.0, 359.99, -3.14, "O3=3.14mPa"),
    (0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, "x,y"),
]
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_lib():
    L = C.CDLL(os.path.join(_ROOT, "oracle", "_ref", "libref_sinks.so"))
    f = C.c_float
    for n in ("ref_gpx_new", "ref_ptu_new"):
        getattr(L, n).restype = C.c_void_p
    L.ref_gpx_free.argtypes = L.ref_gpx_deinit.argtypes = L.ref_gpx_stop_track.argtypes = [C.c_void_p]
    L.ref_gpx_init.argtypes = L.ref_gpx_start_track.argtypes = [C.c_void_p, C.c_char_p]
    L.ref_gpx_add_point.argtypes = [C.c_void_p, C.c_long, f, f, f, f, f]
    L.ref_ptu_free.argtypes = L.ref_ptu_deinit.argtypes = [C.c_void_p]
    L.ref_ptu_init.argtypes = [C.c_void_p, C.c_char_p]
    L.ref_ptu_add_point.argtypes = [C.c_void_p, C.c_long, f, f, f, f, f, f, f, f, f, f, C.c_char_p]
    return L


def _read(path):
    with open(path, "rb") as fh:
        return fh.read().decode("latin-1")


def run_gpx(L, kind):
    path = tempfile.mktemp(suffix=".gpx")
    snaps = []
    if kind == "ref":
        g = L.ref_gpx_new()
        assert L.ref_gpx_init(g, path.encode())
        start, stop, point = L.ref_gpx_start_track, L.ref_gpx_stop_track, L.ref_gpx_add_point
    else:
        g = L.sonde_gpx_open(path.encode())
        assert g
        start, stop, point = L.sonde_gpx_start_track, L.sonde_gpx_stop_track, L.sonde_gpx_add_point
    snaps.append(_read(path))
    for op in GPX_OPS:
        if op[0] == "start":
            start(g, op[1].encode())
        elif op[0] == "stop":
            stop(g)
        else:
            point(g, *op[1:])
        snaps.append(_read(path))
    if kind == "ref":
        L.ref_gpx_deinit(g)
        L.ref_gpx_free(g)
    else:
        L.sonde_gpx_close(g)
    snaps.append(_read(path))
    os.unlink(path)
    return snaps


def run_ptu(L, kind):
    path = tempfile.mktemp(suffix=".csv")
    snaps = []
    if kind == "ref":
        p = L.ref_ptu_new()
        assert L.ref_ptu_init(p, path.encode())
        add = L.ref_ptu_add_point
    else:
        p = L.sonde_ptu_open(path.encode())
        assert p
        add = L.sonde_ptu_add_point
    for op in PTU_OPS:
        add(p, *op[:-1], op[-1].encode())
        snaps.append(_read(path))
    if kind == "ref":
        L.ref_ptu_deinit(p)
        L.ref_ptu_free(p)
    else:
        L.sonde_ptu_close(p)
    snaps.append(_read(path))
    os.unlink(path)
    return snaps

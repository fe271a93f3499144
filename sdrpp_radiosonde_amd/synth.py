"""Synthetic radiosonde signal generator (test/bench input, not part of the decode path).

Builds valid RS41-SG frames (header, RS(255,231) parity, frame type, CRC16'd
subframes, XOR whitening -- SURVEY.md Appendix B.2), serialises them LSB-first
and modulates them as Gaussian-filtered 2-FSK complex baseband IQ at 48 kS/s
(SURVEY.md section 8d "Synthetic signal").  The Reed-Solomon *encoder* here is an
independent numpy implementation; the decoders (oracle C, HIP) never share
code with it, so generator -> decoder round trips are a real cross-check.

The modulator runs on any torch device: CPU for the parity tests, the GPU for
bench.py's large batches.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

FS = 48000  # /root/reference/src/main.cpp:16  OUT_SAMPLE_RATE

# ---------------------------------------------------------------- GF(2^8) / RS(255,231)
_GF_EXP = np.zeros(512, dtype=np.int32)
_GF_LOG = np.zeros(256, dtype=np.int32)
_x = 1
for _i in range(255):
    _GF_EXP[_i] = _x
    _GF_LOG[_x] = _i
    _x <<= 1
    if _x & 0x100:
        _x ^= 0x11D
_GF_EXP[255:510] = _GF_EXP[0:255]


def gf_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.int32)
    b = np.asarray(b, dtype=np.int32)
    out = _GF_EXP[(_GF_LOG[a] + _GF_LOG[b]) % 255]
    return np.where((a == 0) | (b == 0), 0, out).astype(np.int32)


def _rs_generator(nroots: int = 24) -> np.ndarray:
    g = np.array([1], dtype=np.int32)
    for j in range(nroots):
        # multiply by (x + alpha^j); little-endian coefficients
        shifted = np.concatenate([[0], g])
        scaled = np.concatenate([gf_mul(g, _GF_EXP[j]), [0]])
        g = shifted ^ scaled
    return g.astype(np.int32)


_RS_G = _rs_generator()


def rs_parity(msg: np.ndarray) -> np.ndarray:
    """msg: [F, k] bytes (coefficient of x^(24+i) is msg[:, i]).  Returns [F, 24] parity
    (coefficients of x^0..x^23) such that the codeword has roots alpha^0..alpha^23."""
    msg = np.asarray(msg, dtype=np.int32)
    F, k = msg.shape
    rem = np.zeros((F, 24), dtype=np.int32)
    for i in range(k - 1, -1, -1):
        fb = msg[:, i] ^ rem[:, 23]
        new = np.zeros_like(rem)
        new[:, 1:] = rem[:, :-1]
        new ^= gf_mul(fb[:, None], _RS_G[None, :24])
        rem = new
    return rem


def crc16_ccitt(data: np.ndarray) -> np.ndarray:
    """data: [F, n] bytes -> [F] CRC16-CCITT (0x1021, init 0xFFFF)."""
    data = np.asarray(data, dtype=np.uint32)
    crc = np.full(data.shape[0], 0xFFFF, dtype=np.uint32)
    for i in range(data.shape[1]):
        crc ^= data[:, i] << 8
        for _ in range(8):
            hi = (crc & 0x8000) != 0
            crc = (crc << 1) & 0xFFFF
            crc = np.where(hi, crc ^ 0x1021, crc)
    return crc


# ---------------------------------------------------------------- RS41 frame builder
RS41_HEADER_AIR = np.array([0x10, 0xB6, 0xCA, 0x11, 0x22, 0x96, 0x12, 0xF8], dtype=np.uint8)
RS41_MASK = np.array([
    0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
    0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
    0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
    0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1,
], dtype=np.uint8)
RS41_STD_LEN = 320
RS41_EXT_LEN = 518
# (type, payload length) of the standard RS41-SG frame; 57 + sum(len+4) = 320
RS41_SUBFRAMES_STD = [(0x79, 0x28), (0x7A, 0x2A), (0x7C, 0x1E), (0x7D, 0x59), (0x7B, 0x15), (0x76, 0x11)]
# extended frame carries an XDATA (0x7E) block and a longer pad; 57 + sum(len+4) = 518
RS41_SUBFRAMES_EXT = [(0x79, 0x28), (0x7A, 0x2A), (0x7C, 0x1E), (0x7D, 0x59), (0x7B, 0x15), (0x7E, 0x3C), (0x76, 0x97)]


def _put_le(buf: np.ndarray, off: int, val: np.ndarray, nbytes: int) -> None:
    v = np.asarray(val).astype(np.int64)
    for b in range(nbytes):
        buf[:, off + b] = (v >> (8 * b)) & 0xFF


# ---- RS41 sensor model for the generator (inverse of the parser's conversion; values of a typical RS41-SG)
RS41_RF1, RS41_RF2 = 750.0, 1100.0
RS41_CO1 = (-243.911, 0.187654, 8.2e-06)
RS41_CALT1 = (1.0, 0.0, 0.0)
RS41_CALH0 = 45.0
RS41_F1, RS41_F2 = 133000, 190000           # reference counts (temperature and humidity channels alike)


# RS41-SGP pressure sensor model: P(hPa) = sum_j sum_k cfP[4j+k] * a0^j * T^k with a0 = cfP[24] / fp, fp the normalised
# count.  Only the j = 0, 1 / k = 0, 1 terms are used here: P = 1100 - 1000 a0 + 0.02 T.
RS41_CFP = np.zeros(25)
RS41_CFP[0], RS41_CFP[4], RS41_CFP[1], RS41_CFP[24] = 1100.0, -1000.0, 0.02, 0.5
RS41_CFP_SLOT = (0, 4, 8, 12, 16, 20, 24, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11)      # calibration word i -> matrix entry
RS41_TPRESS = 21.5                                                                # pressure sensor temperature, deg C


def rs41_true_pressure(frame_idx):
    """hPa the SGP variant of the generator encodes for frame k (0 for the sensor-less RS41-SG)."""
    alt = 1000.0 + 5.0 * np.asarray(frame_idx, dtype=np.float64)
    return 1013.25 * np.exp(-alt / 8000.0)


def rs41_pressure_counts(P):
    a0 = (1100.0 + 0.02 * RS41_TPRESS - np.asarray(P)) / 1000.0
    fp = RS41_CFP[24] / a0
    return np.rint(RS41_F1 + fp * (RS41_F2 - RS41_F1)).astype(np.int64)


def xdata_ozone_string(cell_ua: float, tpump_c: float) -> bytes:
    """ASCII hex XDATA of an OIF411 ozone interface: type 05, number 01, pump temperature, cell current, battery,
    pump current, external voltage."""
    return b"0501%04X%05X%02X%03X%02X" % (int(round(tpump_c * 100)) & 0xFFFF, int(round(cell_ua * 1e4)), 125, 95, 0)


def ozone_true(frame_idx):
    """(cell current uA, pump temperature C) the generator's ozonesonde reports in frame k."""
    k = np.asarray(frame_idx, dtype=np.float64)
    return 1.5 + 0.01 * k, 25.0 - 0.02 * k


def rs41_calibration_memory(channel: int) -> np.ndarray:
    """The 816-byte calibration table of a sonde (51 fragments of 16 bytes); only the words the PTU
    conversion reads are meaningful, the rest is a channel-dependent pattern."""
    mem = ((np.arange(51 * 16) * 37 + 11 * (channel % 251)) & 0xFF).astype(np.uint8)
    def putf(off, vals):
        mem[off: off + 4 * len(vals)] = np.frombuffer(np.asarray(vals, dtype="<f4").tobytes(), dtype=np.uint8)
    putf(0x3D, [RS41_RF1]); putf(0x41, [RS41_RF2])
    putf(0x4D, RS41_CO1); putf(0x59, RS41_CALT1)
    putf(0x75, [RS41_CALH0, 0.0])
    putf(0x125, RS41_CO1); putf(0x131, RS41_CALT1)
    putf(0x25E, [RS41_CFP[k] for k in RS41_CFP_SLOT])                       # pressure polynomial (RS41-SGP)
    kill = 0xFFFF if channel % 2 == 0 else 3600 + channel % 60000          # burst-kill countdown (s); 0xFFFF = not armed
    mem[0x316], mem[0x317] = kill & 0xFF, kill >> 8
    return mem


def rs41_true_ptu(channel_ids, frame_idx):
    """Temperature (deg C) and relative humidity (%) the generator encodes for (channel, frame)."""
    alt = 1000.0 + 5.0 * np.asarray(frame_idx, dtype=np.float64)
    T = 15.0 - 0.0065 * alt - 0.01 * (np.asarray(channel_ids) % 100)
    RH = 30.0 + 40.0 * np.exp(-alt / 8000.0)
    return T, RH


def rs41_ptu_counts(T, RH):
    """Sensor counts that the conversion of sonde_rs41_temp / sonde_rs41_rh maps back to (T, RH)."""
    p0, p1, p2 = RS41_CO1
    R = (-p1 + np.sqrt(p1 * p1 - 4.0 * p2 * (p0 - T))) / (2.0 * p2)        # T = p0 + p1 R + p2 R^2
    g = (RS41_F2 - RS41_F1) / (RS41_RF2 - RS41_RF1)
    Rb = (RS41_F1 * RS41_RF2 - RS41_F2 * RS41_RF1) / (RS41_F2 - RS41_F1)
    fT = np.rint((R + Rb) * g).astype(np.int64)
    rh_raw = RH + T / 5.5                                                  # undo the temperature term (T >= -25)
    fh = (rh_raw / 100.0 + 7.5) / (350.0 / RS41_CALH0)
    fH = np.rint(RS41_F1 + fh * (RS41_F2 - RS41_F1)).astype(np.int64)
    return fT, fH


def rs41_build_frames(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray, extended: bool = False, sgp: bool = False) -> np.ndarray:
    """Return unscrambled RS41 frames [F, len] (uint8) for the (channel, frame number) pairs.

    Telemetry values are deterministic functions of (seed, channel, frame) so any
    frame can be regenerated alone; they are physically plausible (ECEF position
    near 47N 8E climbing at 5 m/s) so the field parsers have something to chew on.
    """
    channel_ids = np.asarray(channel_ids, dtype=np.int64)
    frame_idx = np.asarray(frame_idx, dtype=np.int64)
    F = channel_ids.shape[0]
    flen = RS41_EXT_LEN if extended else RS41_STD_LEN
    layout = RS41_SUBFRAMES_EXT if extended else RS41_SUBFRAMES_STD
    fr = np.zeros((F, flen), dtype=np.uint8)
    fr[:, 0:8] = RS41_HEADER_AIR ^ RS41_MASK[0:8]
    fr[:, 56] = 0xF0 if extended else 0x0F

    rng = np.random.Generator(np.random.Philox(key=seed & 0xFFFFFFFFFFFFFFFF))
    # per-frame random filler, reproducible from (seed, channel, frame): hash into a Philox counter
    filler = np.zeros((F, flen), dtype=np.uint8)
    for i in range(F):
        g = np.random.Generator(np.random.Philox(key=(seed ^ (int(channel_ids[i]) << 20) ^ int(frame_idx[i])) & 0xFFFFFFFFFFFFFFFF))
        filler[i] = g.integers(0, 256, size=flen, dtype=np.uint8)
    del rng

    off = 57
    for (stype, slen) in layout:
        fr[:, off] = stype
        fr[:, off + 1] = slen
        body = fr[:, off + 2: off + 2 + slen]
        body[:] = filler[:, off + 2: off + 2 + slen]
        if stype == 0x79:  # status: frame number, serial, calibration fragment
            seq = 1000 + frame_idx
            _put_le(body, 0, seq, 2)
            for i in range(F):
                body[i, 2:10] = np.frombuffer(("S%07d" % (int(channel_ids[i]) % 10000000)).encode(), dtype=np.uint8)
            body[:, 23] = (seq % 51).astype(np.uint8)  # calibration fragment index
            for i in range(F):
                k = int(seq[i] % 51)
                body[i, 24:40] = rs41_calibration_memory(int(channel_ids[i]))[16 * k: 16 * k + 16]
        elif stype == 0x7A:  # measurements: 12 x 24-bit counts
            T, RH = rs41_true_ptu(channel_ids, frame_idx)
            fT, fH = rs41_ptu_counts(T, RH)
            if sgp:     # RS41-SGP: pressure sensor counts + the sensor's temperature (i16, 0.01 C) at body offset 38
                pm = (rs41_pressure_counts(rs41_true_pressure(frame_idx)), np.full(F, RS41_F1), np.full(F, RS41_F2))
                _put_le(body, 38, np.full(F, int(round(RS41_TPRESS * 100))), 2)
            else:
                pm = (np.zeros(F), np.zeros(F), np.zeros(F))
            for k, v in enumerate((fT, np.full(F, RS41_F1), np.full(F, RS41_F2),
                                   fH, np.full(F, RS41_F1), np.full(F, RS41_F2),
                                   fT, np.full(F, RS41_F1), np.full(F, RS41_F2)) + pm):
                _put_le(body, 3 * k, v, 3)
        elif stype == 0x7C:  # GPS info: week, ms of week
            _put_le(body, 0, np.full(F, 2200), 2)
            _put_le(body, 2, (frame_idx * 1000 + 123456000) % 604800000, 4)
        elif stype == 0x7B:  # GPS position: ECEF cm, velocity cm/s
            lat = math.radians(47.0) + 1e-5 * channel_ids
            lon = math.radians(8.0) + 1e-6 * frame_idx
            alt = 1000.0 + 5.0 * frame_idx
            a, e2 = 6378137.0, 6.69437999014e-3
            Nn = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
            X = (Nn + alt) * np.cos(lat) * np.cos(lon)
            Y = (Nn + alt) * np.cos(lat) * np.sin(lon)
            Z = (Nn * (1 - e2) + alt) * np.sin(lat)
            _put_le(body, 0, np.round(X * 100).astype(np.int64) & 0xFFFFFFFF, 4)
            _put_le(body, 4, np.round(Y * 100).astype(np.int64) & 0xFFFFFFFF, 4)
            _put_le(body, 8, np.round(Z * 100).astype(np.int64) & 0xFFFFFFFF, 4)
            up = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=1)
            east = np.stack([-np.sin(lon), np.cos(lon), np.zeros(F)], axis=1)
            v = 5.0 * up + 12.0 * east
            for k in range(3):
                _put_le(body, 12 + 2 * k, np.round(v[:, k] * 100).astype(np.int64) & 0xFFFF, 2)
            body[:, 18] = 9  # sats
        elif stype == 0x7E:  # XDATA: instrument-chain byte, ASCII hex string of an ozone interface, zero padding
            body[:] = 0
            body[:, 0] = 1
            cur, tp = ozone_true(frame_idx)
            for i in range(F):
                xs = np.frombuffer(xdata_ozone_string(float(cur[i]), float(tp[i])), dtype=np.uint8)
                body[i, 1: 1 + len(xs)] = xs
        elif stype == 0x76:
            body[:] = 0
        crc = crc16_ccitt(body)
        fr[:, off + 2 + slen] = crc & 0xFF
        fr[:, off + 3 + slen] = (crc >> 8) & 0xFF
        off += slen + 4
    assert off == flen, (off, flen)

    for c in range(2):
        msg = fr[:, 56 + c::2]
        fr[:, 8 + 24 * c: 32 + 24 * c] = rs_parity(msg).astype(np.uint8)
    return fr


def rs41_scramble(frames: np.ndarray) -> np.ndarray:
    n = frames.shape[1]
    mask = np.tile(RS41_MASK, (n + 63) // 64)[:n]
    return frames ^ mask[None, :]


def bytes_to_bits_lsb(b: np.ndarray) -> np.ndarray:
    """[F, n] bytes -> [F, 8n] bits, least significant bit of each byte first (RS41 air order)."""
    return np.unpackbits(b[..., None], axis=-1, bitorder="little").reshape(b.shape[0], -1)


@dataclass
class SynthBatch:
    iq: torch.Tensor            # [C, n, 2] float32 (I, Q) on `device`
    frames: list                # per channel: list of (bit_offset, unscrambled frame bytes np.uint8)
    bits: np.ndarray            # [C, nbits] transmitted bit stream
    cfo_hz: np.ndarray
    tau: np.ndarray
    amp: np.ndarray


def rs41_bitstreams(seed: int, channels: np.ndarray, nbits: int, extended: bool = False, preamble_bytes: int = 40, sgp: bool = False):
    """Continuous on-air bit streams: random lead-in of alternating bits, then
    [40-byte alternating preamble | whitened frame] back to back."""
    channels = np.asarray(channels, dtype=np.int64)
    C = channels.shape[0]
    flen = RS41_EXT_LEN if extended else RS41_STD_LEN
    stride = 8 * (flen + preamble_bytes)
    nfr = nbits // stride + 2
    rng = np.random.Generator(np.random.Philox(key=(seed * 7919 + 17) & 0xFFFFFFFFFFFFFFFF))
    lead = rng.integers(64, stride, size=C)
    ch_rep = np.repeat(channels, nfr)
    fi_rep = np.tile(np.arange(nfr), C)
    frames = rs41_build_frames(seed, ch_rep, fi_rep, extended, sgp).reshape(C, nfr, flen)
    air = rs41_scramble(frames.reshape(C * nfr, flen))
    fbits = bytes_to_bits_lsb(air).reshape(C, nfr, 8 * flen)
    alt = (np.arange(stride + 8 * preamble_bytes) & 1).astype(np.uint8)
    bits = np.zeros((C, nbits + stride * 2), dtype=np.uint8)
    out_frames = []
    for c in range(C):
        pos = int(lead[c])
        bits[c, :pos] = alt[:pos]
        lst = []
        for f in range(nfr):
            if pos >= nbits:
                break
            bits[c, pos: pos + 8 * preamble_bytes] = alt[: 8 * preamble_bytes]
            pos += 8 * preamble_bytes
            bits[c, pos: pos + 8 * flen] = fbits[c, f]
            if pos + 8 * flen <= nbits:
                lst.append((pos, frames[c, f].copy()))
            pos += 8 * flen
        out_frames.append(lst)
    return bits[:, :nbits], out_frames


def rs41_cyclic_bitstreams(seed: int, channels: np.ndarray, n_frames: int = 32, preamble_bytes: int = 64):
    """A bit stream per channel that can be repeated end to end without a seam: n_frames x [preamble | whitened frame],
    rotated by a per-channel offset that lies inside the first preamble, so that the wrap point falls between two frames.
    Returns (bits [C, n_frames * 8 * (320 + preamble_bytes)], per channel [(bit offset, unscrambled frame)])."""
    channels = np.asarray(channels, dtype=np.int64)
    C = channels.shape[0]
    flen = RS41_STD_LEN
    stride = 8 * (flen + preamble_bytes)
    rng = np.random.Generator(np.random.Philox(key=(seed * 7919 + 23) & 0xFFFFFFFFFFFFFFFF))
    lead = rng.integers(64, 8 * preamble_bytes + 1, size=C)               # bits of preamble in front of the first frame
    frames = rs41_build_frames(seed, np.repeat(channels, n_frames), np.tile(np.arange(n_frames), C)).reshape(C, n_frames, flen)
    fbits = bytes_to_bits_lsb(rs41_scramble(frames.reshape(C * n_frames, flen))).reshape(C, n_frames, 8 * flen)
    bits = np.zeros((C, n_frames * stride), dtype=np.uint8)
    bits[:, 1::2] = 1                                                     # alternating preamble everywhere ...
    out_frames = []
    for c in range(C):
        lst = []
        for f in range(n_frames):
            pos = int(lead[c]) + f * stride
            bits[c, pos: pos + 8 * flen] = fbits[c, f]                    # ... the frames on top
            lst.append((pos, frames[c, f].copy()))
        out_frames.append(lst)
    return bits, out_frames


def make_rs41_cyclic(n_channels: int, n_block: int, n_blocks: int = 5, *, seed: int = 1, ebn0_db: float = 30.0,
                     device: str | torch.device = "cpu", first_channel: int = 0, **mod_kw) -> SynthBatch:
    """n_blocks consecutive blocks of n_block samples of ONE continuous RS41 signal per channel whose bit stream repeats
    after exactly n_blocks blocks: cycling through the blocks feeds a decoder a stream without junk frames at the wrap
    (with the default 5 x 196 608 samples = 98 304 symbols = 32 frames of 320 + 64 bytes).  iq: [C, n_blocks * n_block, 2]."""
    baud = 4800.0
    total = n_blocks * n_block
    nb = total * baud / FS
    assert abs(nb - round(nb)) < 1e-9 and int(round(nb)) % (8 * 384) == 0, "the cycle must hold a whole number of frame periods"
    channels = np.arange(first_channel, first_channel + n_channels)
    bits, frames = rs41_cyclic_bitstreams(seed, channels, n_frames=int(round(nb)) // (8 * 384))
    ext = np.concatenate([bits, bits[:, :32]], axis=1)                       # the modulator looks two symbols ahead
    iq, cfo, tau, amp = gfsk_modulate(ext, total, baud, seed=seed + first_channel, ebn0_db=ebn0_db, device=device, **mod_kw)
    return SynthBatch(iq=iq, frames=frames, bits=bits, cfo_hz=cfo, tau=tau, amp=amp)


def gfsk_modulate(bits: np.ndarray, n_samples: int, baud: float, *, seed: int = 0, ebn0_db: float = 30.0,
                  h: float = 1.0, bt: float = 0.5, cfo_max_hz: float = 500.0, amp_range=(0.25, 1.0),
                  device: str | torch.device = "cpu", chunk: int = 256, invert: bool = False, fs: float = FS):
    """bits: [C, nbits] -> IQ [C, n_samples, 2] float32.  Per channel: CFO ~ U(-cfo_max, cfo_max),
    timing offset ~ U(0, 1) symbol, amplitude ~ U(amp_range), complex AWGN at Eb/N0."""
    C, nbits = bits.shape
    sps = fs / baud
    assert nbits >= int(n_samples / sps) + 4, "need more bits"
    rng = np.random.Generator(np.random.Philox(key=(seed * 104729 + 5) & 0xFFFFFFFFFFFFFFFF))
    cfo = rng.uniform(-cfo_max_hz, cfo_max_hz, size=C)
    tau = rng.uniform(0.0, 1.0, size=C)
    amp = rng.uniform(amp_range[0], amp_range[1], size=C)
    dev = h * baud / 2.0
    sigma_t = math.sqrt(math.log(2.0)) / (2.0 * math.pi * bt)  # in symbols
    k_erf = 1.0 / (sigma_t * math.sqrt(2.0))
    out = torch.empty((C, n_samples, 2), dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 2654435761 % (2 ** 63))
    n = torch.arange(n_samples, device=device, dtype=torch.float64)
    for c0 in range(0, C, chunk):
        c1 = min(C, c0 + chunk)
        nrz = torch.from_numpy(bits[c0:c1].astype(np.float32) * 2.0 - 1.0).to(device)
        if invert:
            nrz = -nrz
        u = n[None, :] / sps - torch.from_numpy(tau[c0:c1]).to(device)[:, None]  # symbol units
        k0 = torch.floor(u).to(torch.int64)
        f = torch.zeros_like(u)
        for j in range(-2, 3):
            k = k0 + j
            valid = (k >= 0) & (k < nbits)
            a = torch.gather(nrz, 1, k.clamp(0, nbits - 1)).to(torch.float64) * valid
            x = u - k.to(torch.float64) - 0.5
            f += a * 0.5 * (torch.erf(k_erf * (x + 0.5)) - torch.erf(k_erf * (x - 0.5)))
        f = f * dev + torch.from_numpy(cfo[c0:c1]).to(device)[:, None]
        ph = torch.cumsum(f, dim=1) * (2.0 * math.pi / fs)
        a_t = torch.from_numpy(amp[c0:c1]).to(device)[:, None]
        sig = a_t * math.sqrt(sps / (2.0 * 10.0 ** (ebn0_db / 10.0)))
        noise = torch.randn((c1 - c0, n_samples, 2), generator=gen, device=device, dtype=torch.float32)
        out[c0:c1, :, 0] = (a_t * torch.cos(ph)).to(torch.float32) + sig.to(torch.float32) * noise[:, :, 0]
        out[c0:c1, :, 1] = (a_t * torch.sin(ph)).to(torch.float32) + sig.to(torch.float32) * noise[:, :, 1]
    return out, cfo, tau, amp


def make_rs41_cyclic_block(n_channels: int, n_block: int, n_blocks: int, k: int, *, seed: int = 1, ebn0_db: float = 30.0,
                           device: str | torch.device = "cpu", h: float = 1.0, bt: float = 0.5, cfo_max_hz: float = 500.0,
                           amp_range=(0.25, 1.0), chunk: int = 128, fs: float = FS) -> torch.Tensor:
    """Block k (of n_blocks) of the seamless RS41 signal make_rs41_cyclic describes, generated ALONE: [C, n_block, 2] float32.
    For ingest nodes that hold one block of many channels at a time (bench.py --gpus N: generate -> scatter -> free).
    The phase at the block's first sample comes from the closed-form integral of the Gaussian frequency pulses (a prefix sum
    of the NRZ symbols plus the erf-integral of the few pulses that straddle the boundary), the phase inside the block from
    the same per-sample frequency the whole-signal modulator integrates; the noise generator is keyed by (seed, k, chunk).
    Not sample-identical to make_rs41_cyclic (the discrete and the continuous integral differ by a fraction of one sample's
    phase step at each block boundary), the same signal to a decoder."""
    baud = 4800.0
    total = n_blocks * n_block
    nb = total * baud / fs
    assert abs(nb - round(nb)) < 1e-9 and int(round(nb)) % (8 * 384) == 0, "the cycle must hold a whole number of frame periods"
    nb = int(round(nb))
    C = n_channels
    sps = fs / baud
    start = k * n_block
    rng = np.random.Generator(np.random.Philox(key=(seed * 104729 + 5) & 0xFFFFFFFFFFFFFFFF))
    cfo = rng.uniform(-cfo_max_hz, cfo_max_hz, size=C)
    tau = rng.uniform(0.0, 1.0, size=C)
    amp = rng.uniform(amp_range[0], amp_range[1], size=C)
    dev_hz = h * baud / 2.0
    sigma_t = math.sqrt(math.log(2.0)) / (2.0 * math.pi * bt)
    k_erf = 1.0 / (sigma_t * math.sqrt(2.0))
    out = torch.empty((C, n_block, 2), dtype=torch.float32, device=device)
    n = torch.arange(start, start + n_block, device=device, dtype=torch.float64)

    def H(z):                                                    # integral of erf(k_erf z)
        return z * torch.erf(k_erf * z) + torch.exp(-(k_erf * z) ** 2) / (k_erf * math.sqrt(math.pi))

    for c0 in range(0, C, chunk):
        c1 = min(C, c0 + chunk)
        bits, _ = rs41_cyclic_bitstreams(seed + 7 * (c0 // chunk), np.arange(c0, c1), n_frames=nb // (8 * 384))
        nrz_np = bits.astype(np.float64) * 2.0 - 1.0
        nrz = torch.from_numpy(nrz_np).to(device)                                     # [c, nb]
        prefix = torch.from_numpy(np.concatenate([np.zeros((c1 - c0, 1)), np.cumsum(nrz_np, axis=1)], axis=1)).to(device)
        tau_t = torch.from_numpy(tau[c0:c1]).to(device)[:, None]
        # ---- phase at the block's first sample: sum over the symbols of this cycle that started before it
        us = start / sps - tau_t                                                      # [c, 1] symbol units
        kb = torch.floor(us).to(torch.int64) - 4                                      # symbols <= kb lie wholly before the block
        acc = torch.gather(prefix, 1, (kb + 1).clamp(0, nb))                          # their NRZ sum (0 if none)
        for j in range(1, 10):
            kk = kb + j
            valid = (kk >= 0) & (kk < nb)
            a = torch.gather(nrz, 1, kk.clamp(0, nb - 1)) * valid
            x = us - kk.to(torch.float64) - 0.5
            acc = acc + a * (0.5 * (H(x + 0.5) - H(x - 0.5)) + 0.5)
        ph0 = (torch.from_numpy(cfo[c0:c1]).to(device)[:, None] * start + dev_hz * sps * acc) * (2.0 * math.pi / fs)
        # ---- inside the block: the modulator's per-sample frequency (symbols looked up cyclically)
        u = n[None, :] / sps - tau_t
        k0 = torch.floor(u).to(torch.int64)
        f = torch.zeros_like(u)
        for j in range(-2, 3):
            kk = k0 + j
            valid = kk >= 0                                                            # the cycle's first pass has no past
            a = torch.gather(nrz, 1, torch.remainder(kk, nb)) * valid
            x = u - kk.to(torch.float64) - 0.5
            f += a * 0.5 * (torch.erf(k_erf * (x + 0.5)) - torch.erf(k_erf * (x - 0.5)))
        f = f * dev_hz + torch.from_numpy(cfo[c0:c1]).to(device)[:, None]
        ph = ph0 + (torch.cumsum(f, dim=1) - f) * (2.0 * math.pi / fs)                  # phase AT sample i: everything before it
        a_t = torch.from_numpy(amp[c0:c1]).to(device)[:, None]
        sig = (a_t * math.sqrt(sps / (2.0 * 10.0 ** (ebn0_db / 10.0)))).to(torch.float32)
        gen = torch.Generator(device=device)
        gen.manual_seed((seed * 2654435761 + 1000003 * k + c0) % (2 ** 63))
        noise = torch.randn((c1 - c0, n_block, 2), generator=gen, device=device, dtype=torch.float32)
        out[c0:c1, :, 0] = (a_t * torch.cos(ph)).to(torch.float32) + sig * noise[:, :, 0]
        out[c0:c1, :, 1] = (a_t * torch.sin(ph)).to(torch.float32) + sig * noise[:, :, 1]
    return out


def make_rs41_batch(n_channels: int, n_samples: int, *, seed: int = 1, ebn0_db: float = 30.0,
                    device: str | torch.device = "cpu", first_channel: int = 0, extended: bool = False,
                    invert: bool = False, sgp: bool = False, **mod_kw) -> SynthBatch:
    baud = 4800.0
    nbits = int(n_samples * baud / mod_kw.get("fs", FS)) + 16       # fs=...: IQ at another rate (the VFO front-end's input)
    channels = np.arange(first_channel, first_channel + n_channels)
    bits, frames = rs41_bitstreams(seed, channels, nbits, extended, sgp=sgp)
    iq, cfo, tau, amp = gfsk_modulate(bits, n_samples, baud, seed=seed + first_channel, ebn0_db=ebn0_db,
                                      device=device, invert=invert, **mod_kw)
    return SynthBatch(iq=iq, frames=frames, bits=bits, cfo_hz=cfo, tau=tau, amp=amp)


# ================================================================ Manchester / biphase sondes
# Protocol facts: SURVEY.md Appendix B.3-B.5 ([RECALL]); the oracle (oracle/or_framers.c) and the HIP
# framers hold the same tables, the *encoders* below are independent implementations.

def manchester(bits: np.ndarray) -> np.ndarray:
    """[F, n] bits -> [F, 2n] chips, 1 -> 10, 0 -> 01."""
    b = np.asarray(bits, dtype=np.uint8)
    return np.stack([b, 1 - b], axis=-1).reshape(b.shape[0], -1)


def _bits_msb(vals: np.ndarray, nbits: int) -> np.ndarray:
    """[F] ints -> [F, nbits] bits, most significant first."""
    v = np.asarray(vals, dtype=np.int64)
    return ((v[:, None] >> np.arange(nbits - 1, -1, -1)[None, :]) & 1).astype(np.uint8)


# ---------------------------------------------------------------- DFM06/09/17
DFM_SYNC16 = 0x45CF
DFM_FRAME_CHIPS = 560


def hamming84_encode(nib: np.ndarray) -> np.ndarray:
    """nibble (c0..c3 = bits 3..0) -> 8-bit codeword c0..c7 (c0 = MSB), parity rows 0x78/0xB4/0xD2/0xE1."""
    n = np.asarray(nib, dtype=np.int64) & 0xF
    c0, c1, c2, c3 = (n >> 3) & 1, (n >> 2) & 1, (n >> 1) & 1, n & 1
    c4, c5, c6, c7 = c1 ^ c2 ^ c3, c0 ^ c2 ^ c3, c0 ^ c1 ^ c3, c0 ^ c1 ^ c2
    return ((c0 << 7) | (c1 << 6) | (c2 << 5) | (c3 << 4) | (c4 << 3) | (c5 << 2) | (c6 << 1) | c7).astype(np.uint8)


def dfm_true_temp(channel_ids, frame_idx):
    return 12.0 - 0.03 * np.asarray(frame_idx, dtype=np.float64) - 0.01 * (np.asarray(channel_ids) % 50)


DFM_F1, DFM_F2 = 1000.0, 45000.0          # reference channel readings (f1: offset, f2: gain reference, Rf = 220k)


def dfm_meas_counts(T):
    """Thermistor channel reading that sonde_dfm_temp maps back to T (deg C), with the fixed reference readings."""
    B0, T0, R0, Rf = 3260.0, 25.0 + 273.15, 5.0e3, 220.0e3
    R = R0 * np.exp(B0 * (1.0 / (np.asarray(T) + 273.15) - 1.0 / T0))
    return DFM_F1 + R * (DFM_F2 / Rf), DFM_F1, DFM_F2


def dfm_fl24(val):
    """Encode positive values as mantissa(20 bits) / 2^exp(4 bits), the largest exponent that keeps the mantissa in range."""
    val = np.asarray(val, dtype=np.float64)
    e = np.clip(np.floor(np.log2((2 ** 20 - 1) / np.maximum(val, 1e-9))), 0, 15).astype(np.int64)
    m = np.minimum(np.round(val * 2.0 ** e), 2 ** 20 - 1).astype(np.int64)
    return (e << 20) | m


def dfm_build_frames(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray):
    """Returns (codewords [F,33] uint8, air bits [F,280])."""
    ch = np.asarray(channel_ids, dtype=np.int64)
    fi = np.asarray(frame_idx, dtype=np.int64)
    F = ch.shape[0]
    nib = np.zeros((F, 33), dtype=np.int64)
    # CONF: 7 nibbles: channel id nibble + 6 nibbles of (pseudo) sensor data
    nib[:, 0] = fi % 7
    conf = (seed * 2654435761 + ch * 40503 + fi * 9973) & 0xFFFFFF
    # measurement channels as 24-bit floats (20-bit mantissa / 2^exponent): 0 = thermistor, 3 and 4 = references
    f0, f3, f4 = dfm_meas_counts(dfm_true_temp(ch, fi))
    conf = np.where(fi % 7 == 0, dfm_fl24(f0), conf)
    conf = np.where(fi % 7 == 3, dfm_fl24(np.full(F, f3)), conf)
    conf = np.where(fi % 7 == 4, dfm_fl24(np.full(F, f4)), conf)
    for k in range(6):
        nib[:, 1 + k] = (conf >> (4 * (5 - k))) & 0xF
    # DAT1/DAT2: 6 bytes + id nibble; ids cycle 0,1 | 2,3 | 4,8
    ids = np.array([[0, 1], [2, 3], [4, 8]])[fi % 3]
    lat = np.round((47.0 + 1e-3 * ch) * 1e7).astype(np.int64)
    lon = np.round((8.0 + 1e-5 * fi) * 1e7).astype(np.int64)
    alt = np.round((1000.0 + 5.0 * fi) * 100).astype(np.int64)
    for blk in range(2):
        pid = ids[:, blk]
        payload = np.zeros(F, dtype=np.int64)           # 48 bits
        payload = np.where(pid == 0, (fi & 0xFF) << 16, payload)                                # frame counter: 8 bits at bit 24
        payload = np.where(pid == 1, (fi * 1000 + 123000) % 60000 & 0xFFFF, payload)            # UTC ms of minute: last 16 bits
        payload = np.where(pid == 2, ((lat & 0xFFFFFFFF) << 16) | 1200, payload)               # lat 1e-7, hor. speed cm/s
        payload = np.where(pid == 3, ((lon & 0xFFFFFFFF) << 16) | 9000, payload)               # lon 1e-7, heading 0.01 deg
        payload = np.where(pid == 4, ((alt & 0xFFFFFFFF) << 16) | 500, payload)                # alt cm, climb cm/s
        payload = np.where(pid == 8, (2024 << 36) | (6 << 32) | (15 << 27) | (12 << 22) | ((fi % 60) << 16), payload)
        for k in range(12):
            nib[:, 7 + 13 * blk + k] = (payload >> (4 * (11 - k))) & 0xF
        nib[:, 7 + 13 * blk + 12] = pid
    cw = hamming84_encode(nib)
    bits = np.zeros((F, 280), dtype=np.uint8)
    bits[:, :16] = _bits_msb(np.full(F, DFM_SYNC16), 16)
    off = 16
    for (o, n) in ((0, 7), (7, 13), (20, 13)):
        blk = cw[:, o: o + n]                                  # [F, n]
        planes = ((blk[:, None, :] >> (7 - np.arange(8))[None, :, None]) & 1)   # [F, 8, n]: bit j of codeword i
        bits[:, off: off + 8 * n] = planes.reshape(F, 8 * n)
        off += 8 * n
    return cw, bits


# ---------------------------------------------------------------- M10
M10_SYNC_CHIPS = np.array([int(c) for c in "10011001100110010100110010011001"], dtype=np.uint8)
M10_FRAME_BYTES = 101


def m10_checksum(frames: np.ndarray, n: int = 99) -> np.ndarray:
    """Meteomodem 16-bit checksum over frames[:, :n] (vectorised over frames)."""
    cs = np.zeros(frames.shape[0], dtype=np.int64)
    for i in range(n):
        b = frames[:, i].astype(np.int64)
        c1 = cs & 0xFF
        b = ((b >> 1) | ((b & 1) << 7)) & 0xFF
        b ^= (b >> 2) & 0xFF
        t6 = (cs & 1) ^ ((cs >> 2) & 1) ^ ((cs >> 4) & 1)
        t7 = ((cs >> 1) & 1) ^ ((cs >> 3) & 1) ^ ((cs >> 5) & 1)
        t = (cs & 0x3F) | (t6 << 6) | (t7 << 7)
        s = (cs >> 7) & 0xFF
        s ^= (s >> 2) & 0xFF
        cs = ((c1 << 8) | (b ^ t ^ s)) & 0xFFFF
    return cs


def _put_be(buf, off, val, nbytes):
    v = np.asarray(val).astype(np.int64)
    for b in range(nbytes):
        buf[:, off + b] = (v >> (8 * (nbytes - 1 - b))) & 0xFF


def m10_true_ptu(channel_ids, frame_idx):
    T = 14.0 - 0.02 * np.asarray(frame_idx, dtype=np.float64) - 0.01 * (np.asarray(channel_ids) % 40)
    RH = 35.0 + 0.1 * (np.asarray(frame_idx) % 100)
    return T, RH


def m10_temp_adc(T, scale: int = 1):
    """ADC reading the M10 thermistor model maps back to T (range `scale`): inverse of sonde_m10_temp."""
    p0, p1, p2, p3 = 1.07303516e-03, 2.41296733e-04, 2.26744154e-06, 6.52855181e-08
    Rs, Rp = (12.1e3, 36.5e3, 475.0e3)[scale], (1.0e20, 330.0e3, 2000.0e3)[scale]
    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    lnr = np.zeros_like(T)
    for i, t in enumerate(T):                     # solve the cubic in ln R by bisection (monotone)
        lo, hi = 2.0, 20.0
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            if 1.0 / (p0 + p1 * mid + p2 * mid ** 2 + p3 * mid ** 3) - 273.15 > t:
                lo = mid
            else:
                hi = mid
        lnr[i] = 0.5 * (lo + hi)
    R = np.exp(lnr)
    x = Rs / R + Rs / Rp
    return np.rint(4095.0 / (1.0 + x)).astype(np.int64)


def m10_rh_counts(RH, T, ref: int = 100000):
    q = (np.asarray(RH) - (20.0 - np.asarray(T)) * 0.03) * 0.002 + 0.8955
    return np.rint(q * ref).astype(np.int64), ref


def m20_true_temp(channel_ids, frame_idx):
    return 9.0 - 0.02 * np.asarray(frame_idx, dtype=np.float64) - 0.01 * (np.asarray(channel_ids) % 40)


def m20_temp_adc(T):
    R = 15.0e3 * np.exp(3450.0 * (1.0 / (np.asarray(T, dtype=np.float64) + 273.15) - 1.0 / 273.15))
    return np.rint(4095.0 * R / (R + 22.1e3)).astype(np.int64)


M20_FRAME_BYTES = 70


def m20_build_frames(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray) -> np.ndarray:
    """M20: 70-byte frame (length byte 0x45, type 0x20), the same rolling checksum over all but the last two bytes.
    Returned padded to the 101-byte window of the M10 framer (the pad is what the transmitter idles behind the frame)."""
    ch = np.asarray(channel_ids, dtype=np.int64)
    fi = np.asarray(frame_idx, dtype=np.int64)
    F = ch.shape[0]
    fr = np.zeros((F, M10_FRAME_BYTES), dtype=np.uint8)
    g = np.random.Generator(np.random.Philox(key=(seed * 37 + 11) & 0xFFFFFFFFFFFFFFFF))
    fr[:, :M20_FRAME_BYTES] = g.integers(0, 256, size=(F, M20_FRAME_BYTES), dtype=np.uint8)
    fr[:, M20_FRAME_BYTES:] = 0xAA                                                       # idle pattern behind the frame
    fr[:, 0], fr[:, 1] = 0x45, 0x20
    adc = m20_temp_adc(m20_true_temp(ch, fi))
    fr[:, 0x04], fr[:, 0x05] = adc & 0xFF, (adc >> 8) & 0x0F
    _put_be(fr, 0x08, np.round((1000.0 + 5.0 * fi) * 100).astype(np.int64) & 0xFFFFFF, 3)   # altitude, cm
    _put_be(fr, 0x0B, np.full(F, 1200), 2)                                               # vE, 0.01 m/s
    _put_be(fr, 0x0D, np.zeros(F, dtype=np.int64), 2)                                    # vN
    _put_be(fr, 0x0F, (fi + 123456) % 604800, 3)                                         # GPS time of week, s
    _put_be(fr, 0x18, np.full(F, 500), 2)                                                # vU
    _put_be(fr, 0x1A, np.full(F, 2200), 2)                                               # GPS week
    _put_be(fr, 0x1C, np.round((47.0 + 1e-3 * ch) * 1e6).astype(np.int64) & 0xFFFFFFFF, 4)
    _put_be(fr, 0x20, np.round((8.0 + 1e-5 * fi) * 1e6).astype(np.int64) & 0xFFFFFFFF, 4)
    cs = m10_checksum(fr, M20_FRAME_BYTES - 2)
    fr[:, M20_FRAME_BYTES - 2], fr[:, M20_FRAME_BYTES - 1] = (cs >> 8) & 0xFF, cs & 0xFF
    return fr


def m10_build_frames(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray) -> np.ndarray:
    ch = np.asarray(channel_ids, dtype=np.int64)
    fi = np.asarray(frame_idx, dtype=np.int64)
    F = ch.shape[0]
    fr = np.zeros((F, M10_FRAME_BYTES), dtype=np.uint8)
    g = np.random.Generator(np.random.Philox(key=(seed * 31 + 7) & 0xFFFFFFFFFFFFFFFF))
    fr[:] = g.integers(0, 256, size=fr.shape, dtype=np.uint8)
    fr[:, 0], fr[:, 1], fr[:, 2] = 0x64, 0x9F, 0x20
    _put_be(fr, 0x04, np.round(12.0 * 200).astype(np.int64) & 0xFFFF + 0 * ch, 2)     # vE  (1/200 m/s)
    _put_be(fr, 0x06, np.zeros(F, dtype=np.int64), 2)                                   # vN
    _put_be(fr, 0x08, np.full(F, 5 * 200), 2)                                           # vU
    _put_be(fr, 0x0A, (fi * 1000 + 123456000) % 604800000, 4)                           # GPS time of week, ms
    _put_be(fr, 0x0E, np.round((47.0 + 1e-3 * ch) * (2 ** 32 / 360.0)).astype(np.int64) & 0xFFFFFFFF, 4)
    _put_be(fr, 0x12, np.round((8.0 + 1e-5 * fi) * (2 ** 32 / 360.0)).astype(np.int64) & 0xFFFFFFFF, 4)
    _put_be(fr, 0x16, np.round((1000.0 + 5.0 * fi) * 1000).astype(np.int64) & 0xFFFFFFFF, 4)   # mm
    _put_be(fr, 0x20, np.full(F, 2200), 2)                                              # GPS week
    T, RH = m10_true_ptu(ch, fi)                                                         # thermistor + humidity captures
    adc = m10_temp_adc(T, 1) + 0xA000
    fr[:, 0x3E] = 1
    fr[:, 0x3F], fr[:, 0x40] = adc & 0xFF, (adc >> 8) & 0xFF
    sen, ref = m10_rh_counts(RH, T)
    for b in range(3):
        fr[:, 0x32 + b] = (ref >> (8 * b)) & 0xFF
        fr[:, 0x35 + b] = (sen >> (8 * b)) & 0xFF
    cs = m10_checksum(fr)
    fr[:, 99], fr[:, 100] = (cs >> 8) & 0xFF, cs & 0xFF
    return fr


# ---------------------------------------------------------------- iMS-100 / RS-11G
IMS_SYNC24 = 0x049DCE
IMS_NBLK, IMS_BLK_BITS, IMS_DATA_BYTES = 12, 46, 51
BCH_G = 0x1539


def bch_parity(data34: int) -> int:
    r = data34 << 12
    for i in range(45, 11, -1):
        if r >> i & 1:
            r ^= BCH_G << (i - 12)
    return r & 0xFFF


IMS_CAL = (-70.0, 3.0e-3, 1.0e-8)          # thermistor polynomial c0 + c1 f + c2 f^2 of the generator's sonde


def ims_true_ptu(channel_ids, frame_idx):
    """(temperature C as the parser reconstructs it from the integer count, humidity %) of frame k."""
    f = ims_temp_count(channel_ids, frame_idx).astype(np.float64)
    return IMS_CAL[0] + IMS_CAL[1] * f + IMS_CAL[2] * f * f, 45.0 + 0.05 * (np.asarray(frame_idx) % 200)


def ims_temp_count(channel_ids, frame_idx):
    return (27000 - 40 * np.asarray(frame_idx, dtype=np.int64) - 10 * (np.asarray(channel_ids) % 50)).astype(np.int64)


def ims_words(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray) -> np.ndarray:
    """The 24 sixteen-bit words of an iMS-100 frame (layout: parse.cpp feed_ims100)."""
    ch = np.asarray(channel_ids, dtype=np.int64)
    fi = np.asarray(frame_idx, dtype=np.int64)
    F = ch.shape[0]
    g = np.random.Generator(np.random.Philox(key=(seed * 131 + 3) & 0xFFFFFFFFFFFFFFFF))
    w = g.integers(0, 65536, size=(F, 24)).astype(np.int64)

    def put32(col, v):
        v = np.asarray(v).astype(np.int64) & 0xFFFFFFFF
        w[:, col], w[:, col + 1] = v >> 16, v & 0xFFFF
    w[:, 0] = fi & 0xFFFF
    cal = np.frombuffer(np.asarray(IMS_CAL, dtype=">f4").tobytes(), dtype=">u4").astype(np.int64)
    calw = np.where(fi % 4 == 0, 5000000 + ch, cal[np.clip(fi % 4 - 1, 0, 2)])
    put32(1, calw)
    _, RH = ims_true_ptu(ch, fi)
    w[:, 3] = ims_temp_count(ch, fi)
    w[:, 4] = np.round(RH * 100).astype(np.int64)
    put32(5, (fi * 500 + 123456000) % 604800000)
    w[:, 7] = 2200
    put32(8, np.round((47.0 + 1e-3 * ch) * 1e6))
    put32(10, np.round((8.0 + 1e-5 * fi) * 1e6))
    put32(12, np.round((1000.0 + 2.5 * fi) * 100))
    w[:, 14], w[:, 15], w[:, 16] = 1200, 9000, 500
    return w


def ims_build_frames(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray):
    """Returns (data bytes [F,51] as the FEC stage delivers them, air bits [F, 576]).  408 data bits = 24 groups of
    a 16-bit word (MSB first) + one parity bit that makes the number of ones in the 17 bits odd."""
    w = ims_words(seed, channel_ids, frame_idx)
    F = w.shape[0]
    wb = ((w[:, :, None] >> np.arange(15, -1, -1)[None, None, :]) & 1).astype(np.uint8)      # [F, 24, 16]
    par = (1 - wb.sum(axis=2) % 2).astype(np.uint8)                                           # odd overall parity
    dbits = np.concatenate([wb, par[:, :, None]], axis=2).reshape(F, 408)
    data = np.packbits(dbits, axis=1)
    bits = np.zeros((F, 24 + IMS_NBLK * IMS_BLK_BITS), dtype=np.uint8)
    bits[:, :24] = _bits_msb(np.full(F, IMS_SYNC24), 24)
    for f in range(F):
        for b in range(IMS_NBLK):
            d = 0
            for k in range(34):
                d = (d << 1) | int(dbits[f, 34 * b + k])
            blk = (d << 12) | bch_parity(d)
            bits[f, 24 + 46 * b: 24 + 46 * (b + 1)] = [(blk >> (45 - k)) & 1 for k in range(46)]
    return data, bits


# ---------------------------------------------------------------- MRZ-N1
MRZ_HEADER = (0xAA, 0xBF, 0x35)
MRZ_FRAME_BYTES = 45


def crc16_modbus(data: np.ndarray) -> np.ndarray:
    """[F, n] bytes -> [F] CRC16, reflected polynomial 0xA001, init 0xFFFF."""
    data = np.asarray(data, dtype=np.uint32)
    crc = np.full(data.shape[0], 0xFFFF, dtype=np.uint32)
    for i in range(data.shape[1]):
        crc ^= data[:, i]
        for _ in range(8):
            lsb = (crc & 1) != 0
            crc >>= 1
            crc = np.where(lsb, crc ^ 0xA001, crc)
    return crc


def mrz_true_temp(channel_ids, frame_idx):
    return 11.0 - 0.02 * np.asarray(frame_idx, dtype=np.float64) - 0.01 * (np.asarray(channel_ids) % 40)


def mrz_build_frames(seed: int, channel_ids: np.ndarray, frame_idx: np.ndarray) -> np.ndarray:
    """MRZ-N1 payload frames [F, 45] (layout: parse.cpp feed_mrzn1), CRC included."""
    ch = np.asarray(channel_ids, dtype=np.int64)
    fi = np.asarray(frame_idx, dtype=np.int64)
    F = ch.shape[0]
    fr = np.zeros((F, MRZ_FRAME_BYTES), dtype=np.uint8)
    g = np.random.Generator(np.random.Philox(key=(seed * 41 + 5) & 0xFFFFFFFFFFFFFFFF))
    fr[:, 34:43] = g.integers(0, 256, size=(F, 9), dtype=np.uint8)
    _put_le(fr, 0, fi & 0xFFFF, 2)
    tod = (fi + 45296) % 86400                                              # 12:34:56 + k s
    fr[:, 2], fr[:, 3], fr[:, 4] = tod // 3600, (tod // 60) % 60, tod % 60
    fr[:, 5], fr[:, 6], fr[:, 7] = 15, 6, 24
    lat = np.radians(47.0 + 1e-3 * ch)
    lon = np.radians(8.0 + 1e-5 * fi)
    alt = 1000.0 + 5.0 * fi
    a, e2 = 6378137.0, 6.69437999014e-3
    Nn = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    xyz = [(Nn + alt) * np.cos(lat) * np.cos(lon), (Nn + alt) * np.cos(lat) * np.sin(lon), (Nn * (1 - e2) + alt) * np.sin(lat)]
    for k in range(3):
        _put_le(fr, 8 + 4 * k, np.round(xyz[k] * 100).astype(np.int64) & 0xFFFFFFFF, 4)
    up = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=1)
    east = np.stack([-np.sin(lon), np.cos(lon), np.zeros(F)], axis=1)
    v = 5.0 * up + 12.0 * east
    for k in range(3):
        _put_le(fr, 20 + 2 * k, np.round(v[:, k] * 100).astype(np.int64) & 0xFFFF, 2)
    fr[:, 26] = 9
    _put_le(fr, 27, np.round(mrz_true_temp(ch, fi) * 100).astype(np.int64) & 0xFFFF, 2)
    fr[:, 29] = fi % 4
    _put_le(fr, 30, np.where(fi % 4 == 0, 7000000 + ch, 0x12345678 + fi), 4)
    crc = crc16_modbus(fr[:, :43])
    fr[:, 43], fr[:, 44] = crc & 0xFF, (crc >> 8) & 0xFF
    return fr


def biphase_s(bits: np.ndarray) -> np.ndarray:
    """[n] bits -> [2n] chips: transition at every bit boundary, extra mid-bit transition for a 0
    (so that a 1 reads as two equal chips).  Polarity-free by construction."""
    out = np.zeros(2 * len(bits), dtype=np.uint8)
    lvl = 0
    for i, b in enumerate(bits):
        lvl ^= 1
        out[2 * i] = lvl
        if not b:
            lvl ^= 1
        out[2 * i + 1] = lvl
    return out


# ---------------------------------------------------------------- chip streams + batches for any type
SONDE_BAUD = {0: 4800.0, 1: 5000.0, 2: 4800.0, 3: 9600.0, 6: 4800.0}     # on-air symbol (chip) rates


def chip_streams(sonde_type: int, seed: int, channels: np.ndarray, nchips: int, m20: bool = False):
    """Continuous on-air chip streams [C, nchips] plus, per channel, the list of
    (chip offset of the sync, expected decoded frame bytes)."""
    channels = np.asarray(channels, dtype=np.int64)
    C = channels.shape[0]
    rng = np.random.Generator(np.random.Philox(key=(seed * 7919 + 17 + sonde_type) & 0xFFFFFFFFFFFFFFFF))
    if sonde_type == 1:      # DFM: continuous frames
        flen = DFM_FRAME_CHIPS
        gap = 0
    elif sonde_type == 3:    # M10: bursts with idle gap
        flen = 32 + 16 * M10_FRAME_BYTES
        gap = 752
    elif sonde_type == 2:    # iMS-100
        flen = 2 * (24 + IMS_NBLK * IMS_BLK_BITS)
        gap = 96
    elif sonde_type == 6:    # MRZ-N1: one frame per second
        flen = 48 + 16 * MRZ_FRAME_BYTES
        gap = 4800 - flen
    else:
        raise ValueError("use rs41_bitstreams for RS41")
    stride = flen + gap
    nfr = nchips // stride + 2
    lead = rng.integers(64, 64 + stride, size=C)
    ch_rep = np.repeat(channels, nfr)
    fi_rep = np.tile(np.arange(nfr), C)
    if sonde_type == 1:
        expect, bits = dfm_build_frames(seed, ch_rep, fi_rep)
        chips = manchester(bits)
    elif sonde_type == 3:
        expect = (m20_build_frames if m20 else m10_build_frames)(seed, ch_rep, fi_rep)
        chips = np.concatenate([np.tile(M10_SYNC_CHIPS, (C * nfr, 1)), manchester(np.unpackbits(expect, axis=1))], axis=1)
    elif sonde_type == 6:
        expect = mrz_build_frames(seed, ch_rep, fi_rep)
        hdr = np.tile(np.unpackbits(np.array(MRZ_HEADER, dtype=np.uint8)), (C * nfr, 1))
        chips = manchester(np.concatenate([hdr, np.unpackbits(expect, axis=1)], axis=1))
    else:
        expect, bits = ims_build_frames(seed, ch_rep, fi_rep)
        chips = np.stack([biphase_s(b) for b in bits])
    chips = chips.reshape(C, nfr, flen)
    expect = expect.reshape(C, nfr, -1)
    idle = (np.arange(nchips + 2 * stride) & 1).astype(np.uint8)
    out = np.zeros((C, nchips + 2 * stride), dtype=np.uint8)
    frames = []
    for c in range(C):
        pos = int(lead[c])
        out[c, :pos] = idle[:pos]
        lst = []
        for f in range(nfr):
            if pos >= nchips:
                break
            out[c, pos: pos + flen] = chips[c, f]
            if pos + flen <= nchips:
                lst.append((pos, expect[c, f].copy()))
            pos += flen
            out[c, pos: pos + gap] = idle[:gap]
            pos += gap
        frames.append(lst)
    return out[:, :nchips], frames


def make_batch(sonde_type: int, n_channels: int, n_samples: int, *, seed: int = 1, ebn0_db: float = 30.0,
               device: str | torch.device = "cpu", first_channel: int = 0, invert: bool = False, m20: bool = False,
               **mod_kw) -> SynthBatch:
    """Synthetic batch of any supported sonde type (0 RS41, 1 DFM09, 2 iMS-100, 3 M10 (m20=True: M20 frames), 4 iMet-4,
    5 SRS-C50, 6 MRZ-N1)."""
    if sonde_type == 4:
        return make_imet_batch(n_channels, n_samples, seed=seed, snr_db=ebn0_db, device=device, first_channel=first_channel)
    if sonde_type == 5:
        return make_c50_batch(n_channels, n_samples, seed=seed, snr_db=ebn0_db, device=device, first_channel=first_channel)
    if sonde_type == 0:
        return make_rs41_batch(n_channels, n_samples, seed=seed, ebn0_db=ebn0_db, device=device,
                               first_channel=first_channel, invert=invert, **mod_kw)
    baud = SONDE_BAUD[sonde_type]
    nchips = int(n_samples * baud / mod_kw.get("fs", FS)) + 16
    channels = np.arange(first_channel, first_channel + n_channels)
    chips, frames = chip_streams(sonde_type, seed, channels, nchips, m20=m20)
    iq, cfo, tau, amp = gfsk_modulate(chips, n_samples, baud, seed=seed + first_channel + 1000 * sonde_type,
                                      ebn0_db=ebn0_db, device=device, invert=invert, **mod_kw)
    return SynthBatch(iq=iq, frames=frames, bits=chips, cfo_hz=cfo, tau=tau, amp=amp)


# ================================================================ iMet-1 / iMet-4 (Bell-202 AFSK, asynchronous characters)
# Protocol facts: public iMet notes ([RECALL]); independent of the decoders' code.
IMET_BAUD = 1200.0
IMET_MARK_HZ, IMET_SPACE_HZ = 1200.0, 2200.0


def imet_crc(data: np.ndarray) -> int:
    crc = 0x1D0F
    for b in data.tolist():
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def imet_true_values(channel: int, k: int):
    """What packet pair k of a channel encodes: (pressure hPa, temperature C, humidity %, lat, lon, alt m, h, m, s)."""
    alt = 1200.0 + 5.0 * k
    return (1013.25 * math.exp(-alt / 8000.0), 15.0 - 0.0065 * alt, 40.0 + (channel % 30), 47.0 + 1e-3 * channel,
            8.0 + 1e-4 * k, alt, 12, (k // 60) % 60, k % 60)


def imet_build_packets(channel: int, k: int, xdata: bool = False):
    """[PTU packet, GPS packet (, XDATA packet of an ECC ozonesonde)] of second k (uint8 arrays, CRC included)."""
    P, T, U, lat, lon, alt, hh, mm, ss = imet_true_values(channel, k)
    ptu = bytearray([0x01, 0x01])
    ptu += int(k & 0xFFFF).to_bytes(2, "little")
    ptu += int(round(P * 100)).to_bytes(3, "little")
    ptu += int(round(T * 100)).to_bytes(2, "little", signed=True)
    ptu += int(round(U * 100)).to_bytes(2, "little")
    ptu += bytes([52])                                       # 5.2 V
    gps = bytearray([0x01, 0x02])
    gps += np.asarray([lat, lon], dtype="<f4").tobytes()
    gps += int(round(alt) + 5000).to_bytes(2, "little")
    gps += bytes([9, hh, mm, ss])
    pkts = [ptu, gps]
    if xdata:   # 01 03 len | 01 (ECC ozonesonde) index | cell current u16 BE (0.001 uA) | pump temperature i16 BE (0.01 C) | mA | 0.1 V
        cur, tp = ozone_true(k)
        xd = bytearray([0x01, 0x03, 8, 0x01, 0x00])
        xd += int(round(float(cur) * 1000)).to_bytes(2, "big")
        xd += int(round(float(tp) * 100)).to_bytes(2, "big", signed=True)
        xd += bytes([95, 125])
        pkts.append(xd)
    out = []
    for pkt in pkts:
        a = np.frombuffer(bytes(pkt), dtype=np.uint8)
        c = imet_crc(a)
        out.append(np.concatenate([a, np.array([c >> 8, c & 0xFF], dtype=np.uint8)]))
    return out


def uart_bits(pkt: np.ndarray) -> np.ndarray:
    """8N1 characters, LSB first: start 0, data, stop 1."""
    b = np.zeros((len(pkt), 10), dtype=np.uint8)
    b[:, 9] = 1
    for m in range(8):
        b[:, 1 + m] = (pkt >> m) & 1
    return b.reshape(-1)


def imet_bitstreams(seed: int, channels: np.ndarray, nbits: int, xdata: bool = False):
    """Per channel: idle marks, then PTU and GPS packets separated by short idle gaps.
    Returns (bits [C, nbits], per channel list of (bit offset of the first start bit, packet bytes))."""
    channels = np.asarray(channels, dtype=np.int64)
    rng = np.random.Generator(np.random.Philox(key=(seed * 7907 + 29) & 0xFFFFFFFFFFFFFFFF))
    bits = np.ones((len(channels), nbits + 2048), dtype=np.uint8)
    frames = []
    for ci, c in enumerate(channels):
        pos = int(rng.integers(40, 200))
        lst, k = [], 0
        while pos < nbits:
            for pkt in imet_build_packets(int(c), k, xdata):
                ub = uart_bits(pkt)
                bits[ci, pos: pos + len(ub)] = ub
                if pos + len(ub) <= nbits:
                    lst.append((pos, pkt))
                pos += len(ub) + int(rng.integers(8, 48))
            k += 1
        frames.append(lst)
    return bits[:, :nbits], frames


def afsk_modulate(bits: np.ndarray, n_samples: int, *, seed: int = 0, snr_db: float = 30.0, fm_dev_hz: float = 3000.0,
                  cfo_max_hz: float = 500.0, amp_range=(0.25, 1.0), device: str | torch.device = "cpu", fs: float = FS,
                  baud: float = 1200.0, mark_hz: float = 1200.0, space_hz: float = 2200.0):
    """bits (1 = mark) -> audio tones (phase-continuous 1200/2200 Hz at 1200 Bd) -> FM -> IQ [C, n_samples, 2].
    snr_db: carrier-to-noise ratio in the full fs bandwidth."""
    C, nbits = bits.shape
    sps = fs / baud
    assert nbits >= int(n_samples / sps) + 4
    rng = np.random.Generator(np.random.Philox(key=(seed * 104723 + 7) & 0xFFFFFFFFFFFFFFFF))
    cfo = rng.uniform(-cfo_max_hz, cfo_max_hz, size=C)
    tau = rng.uniform(0.0, 1.0, size=C)
    amp = rng.uniform(amp_range[0], amp_range[1], size=C)
    gen = torch.Generator(device=device)
    gen.manual_seed((seed * 31 + 3) & 0x7FFFFFFF)
    n = torch.arange(n_samples, device=device, dtype=torch.float64)
    out = torch.empty((C, n_samples, 2), dtype=torch.float32, device=device)
    for c in range(C):
        idx = torch.clamp(torch.floor(n / sps - tau[c]).to(torch.int64), 0, nbits - 1)
        b = torch.from_numpy(bits[c].astype(np.float64)).to(device)[idx]
        f_tone = b * mark_hz + (1.0 - b) * space_hz
        audio = torch.cos(torch.cumsum(f_tone, dim=0) * (2.0 * math.pi / fs))
        ph = torch.cumsum(fm_dev_hz * audio + cfo[c], dim=0) * (2.0 * math.pi / fs)
        sig = amp[c] / math.sqrt(2.0 * 10.0 ** (snr_db / 10.0))
        noise = torch.randn((n_samples, 2), generator=gen, device=device, dtype=torch.float32)
        out[c, :, 0] = (amp[c] * torch.cos(ph)).to(torch.float32) + float(sig) * noise[:, 0]
        out[c, :, 1] = (amp[c] * torch.sin(ph)).to(torch.float32) + float(sig) * noise[:, 1]
    return out, cfo, tau, amp


def make_imet_batch(n_channels: int, n_samples: int, *, seed: int = 1, snr_db: float = 30.0,
                    device: str | torch.device = "cpu", first_channel: int = 0, xdata: bool = False, **mod_kw) -> SynthBatch:
    nbits = int(n_samples * IMET_BAUD / mod_kw.get("fs", FS)) + 16
    channels = np.arange(first_channel, first_channel + n_channels)
    bits, frames = imet_bitstreams(seed, channels, nbits, xdata)
    iq, cfo, tau, amp = afsk_modulate(bits, n_samples, seed=seed + first_channel, snr_db=snr_db, device=device, **mod_kw)
    return SynthBatch(iq=iq, frames=frames, bits=bits, cfo_hz=cfo, tau=tau, amp=amp)


# ================================================================ wideband scene for the channelizer (config 4)
WB_FS = 10_000_000.0
WB_BINS = 512


def make_wideband_rs41(bins_active, n_samples: int, *, seed: int = 1, ebn0_db: float = 30.0,
                       device: str | torch.device = "cpu", offset_hz: float = 0.0):
    """RS41 transmitters at the centres of the given channelizer bins (spacing 19531.25 Hz; each within +-300 Hz of it, plus
    offset_hz for all of them), summed into one 10 MS/s complex stream [n_samples, 2].
    Returns (iq, {bin: [(bit offset, frame bytes), ...]})."""
    bins_active = list(bins_active)
    baud = 4800.0
    nbits = int(n_samples * baud / WB_FS) + 16
    chans = np.array(bins_active)
    bits, frames = rs41_bitstreams(seed, chans, nbits)
    total = torch.zeros((n_samples, 2), dtype=torch.float32, device=device)
    n = torch.arange(n_samples, device=device, dtype=torch.float64)
    for i, k in enumerate(bins_active):
        # noise is added once per transmitter at its own Eb/N0 (white over the full 10 MHz)
        iq, _, _, _ = gfsk_modulate(bits[i: i + 1], n_samples, baud, seed=seed + 31 * i, ebn0_db=ebn0_db, device=device,
                                    cfo_max_hz=300.0, amp_range=(0.5, 0.9), fs=WB_FS)
        ph = (2.0 * math.pi * (k / WB_BINS + offset_hz / WB_FS)) * n
        c, s_ = torch.cos(ph).to(torch.float32), torch.sin(ph).to(torch.float32)
        total[:, 0] += iq[0, :, 0] * c - iq[0, :, 1] * s_
        total[:, 1] += iq[0, :, 0] * s_ + iq[0, :, 1] * c
    return total, {k: frames[i] for i, k in enumerate(bins_active)}


# ================================================================ SRS-C50 (AFSK 2400 Bd, one value per packet)
C50_BAUD, C50_MARK_HZ, C50_SPACE_HZ = 2400.0, 4700.0, 2900.0


def c50_true_temp(channel: int, k: int) -> float:
    return 13.0 - 0.05 * k - 0.01 * (channel % 40)


def c50_packet(ptype: int, value: int) -> np.ndarray:
    body = bytes([ptype]) + int(value & 0xFFFFFFFF).to_bytes(4, "big")
    c1 = c2 = 0
    for b in body:
        c1 = (c1 + b) & 0xFF
        c2 = (c2 + c1) & 0xFF
    return np.frombuffer(bytes([0x00, 0xFF]) + body + bytes([c1, c2]), dtype=np.uint8)


def c50_build_packets(channel: int, k: int):
    """The seven packets of second k: id, date, time, latitude, longitude, altitude, temperature."""
    tod = (k + 45296) % 86400
    t32 = int(np.frombuffer(np.float32(c50_true_temp(channel, k)).tobytes(), dtype="<u4")[0])
    return [c50_packet(0x10, 3000000 + channel), c50_packet(0x14, 150624),
            c50_packet(0x15, (tod // 3600) * 10000 + ((tod // 60) % 60) * 100 + tod % 60),
            c50_packet(0x16, int(round((47.0 + 1e-3 * channel) * 1e6))), c50_packet(0x17, int(round((8.0 + 1e-4 * k) * 1e6))),
            c50_packet(0x18, int(round((1200.0 + 5.0 * k) * 100))), c50_packet(0x03, t32)]


def c50_bitstreams(seed: int, channels: np.ndarray, nbits: int):
    channels = np.asarray(channels, dtype=np.int64)
    rng = np.random.Generator(np.random.Philox(key=(seed * 7927 + 31) & 0xFFFFFFFFFFFFFFFF))
    bits = np.ones((len(channels), nbits + 4096), dtype=np.uint8)
    frames = []
    for ci, c in enumerate(channels):
        pos = int(rng.integers(40, 200))
        lst, k = [], 0
        while pos < nbits:
            for pkt in c50_build_packets(int(c), k):
                ub = uart_bits(pkt)
                bits[ci, pos: pos + len(ub)] = ub
                if pos + len(ub) <= nbits:
                    lst.append((pos, pkt))
                pos += len(ub) + int(rng.integers(4, 40))      # short idle gaps: the transmitter sends packets back to back
            k += 1
        frames.append(lst)
    return bits[:, :nbits], frames


def make_c50_batch(n_channels: int, n_samples: int, *, seed: int = 1, snr_db: float = 30.0,
                   device: str | torch.device = "cpu", first_channel: int = 0, **mod_kw) -> SynthBatch:
    nbits = int(n_samples * C50_BAUD / FS) + 16
    channels = np.arange(first_channel, first_channel + n_channels)
    bits, frames = c50_bitstreams(seed, channels, nbits)
    iq, cfo, tau, amp = afsk_modulate(bits, n_samples, seed=seed + first_channel, snr_db=snr_db, device=device, baud=C50_BAUD,
                                      mark_hz=C50_MARK_HZ, space_hz=C50_SPACE_HZ, fm_dev_hz=4000.0, **mod_kw)
    return SynthBatch(iq=iq, frames=frames, bits=bits, cfo_hz=cfo, tau=tau, amp=amp)

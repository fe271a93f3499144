"""Scoring of decoded frame records against the frames a generator transmitted (tests/golden/tx_*.npz).
Shares nothing with either decoder: a record counts when its FEC / check is clean and its FEC-covered bytes equal the bytes of
a transmitted frame of the same channel."""
import numpy as np

NAMES = ["rs41", "dfm09", "ims100", "m10", "imet4", "c50", "mrzn1"]


def clean_mask(t, fr):
    if t == 0:
        return (fr["nerr"] >= 0).all(axis=1)          # both RS(255,231) codewords decoded
    if t in (1, 2):
        return fr["nerr"][:, 1] == 0                   # no uncorrectable Hamming / BCH block
    return fr["nerr"][:, 0] == 0                       # checksum / CRC ok


def score(t, fr, tx, txch, txlen):
    """-> (transmitted frames that came back clean, clean records that match no transmitted frame)"""
    lo = 8 if t == 0 else 0                            # RS41: the 8 sync bytes are outside the code
    sent = {}
    for i in range(len(tx)):
        sent.setdefault(int(txch[i]), {})[bytes(tx[i, lo: txlen[i]])] = i
    hit, alien = set(), 0
    for f in fr[clean_mask(t, fr)]:
        key = bytes(f["data"][lo: f["len"]])
        i = sent.get(int(f["channel"]), {}).get(key)
        if i is None:
            alien += 1
        else:
            hit.add(i)
    return len(hit), alien

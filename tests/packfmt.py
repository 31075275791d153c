"""Test-side numpy restatement of the packed batch format (include/topsicle_hip.h): 16 bases per uint32 word, base j in
bits [2j, 2j+1], code = (ASCII >> 1) & 3; uint16 invalid mask per word; reads padded with zero words to 64-base quads."""
import numpy as np

from topsicle_amd import hiplib

_VALID = np.zeros(256, bool)
_VALID[list(b"ACGTacgt")] = True


def np_pack(bases: np.ndarray, offsets: np.ndarray):
    n = len(offsets) - 1
    lens = np.diff(offsets)
    words = ((lens + 63) // 64) * 4
    desc = np.zeros(n, hiplib.DESC_DTYPE)
    desc["word_off"][1:] = np.cumsum(words)[:-1]
    desc["len"] = lens
    nw = int(words.sum())
    seq2 = np.zeros(nw, np.uint32)
    inv = np.zeros(nw, np.uint16)
    sh2 = (2 * np.arange(16)).astype(np.uint32)
    sh1 = np.arange(16).astype(np.uint32)
    for i in range(n):
        L = int(lens[i])
        if L == 0:
            continue
        b = np.zeros(int(words[i]) * 16, np.uint8)
        seg = bases[offsets[i]:offsets[i + 1]]
        b[:L] = seg
        code = ((b >> 1) & 3).astype(np.uint32)
        code[L:] = 0
        bad = np.zeros(len(b), np.uint32)
        bad[:L] = ~_VALID[seg]
        w0 = int(desc["word_off"][i])
        seq2[w0:w0 + words[i]] = (code.reshape(-1, 16) << sh2).sum(axis=1, dtype=np.uint32)
        inv[w0:w0 + words[i]] = (bad.reshape(-1, 16) << sh1).sum(axis=1, dtype=np.uint32).astype(np.uint16)
        if bad.any():
            desc["flags"][i] = hiplib.RD_HAS_INVALID
    return seq2, inv, desc

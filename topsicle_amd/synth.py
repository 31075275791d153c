"""Seeded synthetic long reads for benchmarks and size-independent parity properties.

Generator of SURVEY.md section 8(d): every read starts with a telomere tract of length
T ~ U[tract_min, tract_max] made of exact repeats of the motif from a random phase, followed
by i.i.d. uniform ACGT up to a fixed read length; per-base substitution / insertion / deletion
errors are then applied (ONT-like 3 % / 2 % / 2 %, HiFi-like 0.1 % / 0.05 % / 0.05 %), and with
probability 0.5 the read is reverse-complemented (telomere at the 3' end, which the reference
reports as tail 'reverse').  `telomeric_fraction` < 1 makes the remaining reads pure random
sequence (the step-1-dominated regime).
"""
from __future__ import annotations

import numpy as np

ONT = (0.03, 0.02, 0.02)
HIFI = (0.001, 0.0005, 0.0005)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGT")] = list(b"TGCA")


def make_reads(n_reads: int, read_len: int, motif: str, seed: int, errors=ONT,
               tract_min: int = 1000, tract_max: int = 8000, telomeric_fraction: float = 1.0):
    """Returns (bases u8[n_reads*read_len], offsets i64[n_reads+1], truth dict)."""
    rng = np.random.default_rng(seed)
    sub, ins, dele = errors
    m = np.frombuffer(motif.upper().encode(), dtype=np.uint8)
    lgen = read_len + max(64, int(read_len * (dele + 0.01)) + 8 * int(np.sqrt(read_len * (ins + dele) + 1)))
    tract = rng.integers(tract_min, tract_max + 1, n_reads)
    tract = np.minimum(tract, read_len)
    telomeric = rng.random(n_reads) < telomeric_fraction
    tract = np.where(telomeric, tract, 0)
    phase = rng.integers(0, len(m), n_reads)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    tiled = np.tile(m, read_len // len(m) + 3)
    chunk = max(1, min(n_reads, (64 << 20) // lgen))          # ~64 MB of bases at a time
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        nr = hi - lo
        flat = _ACGT[rng.integers(0, 4, (nr, lgen), dtype=np.uint8)]
        for i in range(nr):
            t_i = int(tract[lo + i])
            if t_i:
                flat[i, :t_i] = tiled[phase[lo + i]:phase[lo + i] + t_i]
        flat = flat.reshape(-1)
        n = flat.size
        # substitutions
        k = rng.binomial(n, sub)
        if k:
            flat[rng.integers(0, n, k)] = _ACGT[rng.integers(0, 4, k, dtype=np.uint8)]
        # deletions and insertions on the concatenated stream; read starts are re-mapped
        starts = np.arange(nr, dtype=np.int64) * lgen
        kd = rng.binomial(n, dele)
        dpos = np.unique(rng.integers(0, n, kd)) if kd else np.zeros(0, np.int64)
        ki = rng.binomial(n, ins)
        ipos = np.sort(rng.integers(0, n, ki)) if ki else np.zeros(0, np.int64)
        keep = np.ones(n, dtype=bool)
        keep[dpos] = False
        # insertion indices are given in pre-deletion coordinates -> shift by deletions before them
        ipos_after = ipos - np.searchsorted(dpos, ipos, side="left")
        flat = flat[keep]
        flat = np.insert(flat, ipos_after, _ACGT[rng.integers(0, 4, len(ipos_after), dtype=np.uint8)])
        new_starts = starts - np.searchsorted(dpos, starts, side="left") + np.searchsorted(ipos, starts, side="left")
        ends = np.append(new_starts[1:], flat.size)
        for i in range(nr):
            seg = flat[new_starts[i]:min(ends[i], new_starts[i] + read_len)]
            if seg.size < read_len:
                seg = np.concatenate([seg, _ACGT[rng.integers(0, 4, read_len - seg.size, dtype=np.uint8)]])
            out[lo + i] = seg
    rev = rng.random(n_reads) < 0.5
    idx = np.nonzero(rev)[0]
    for i in idx:
        out[i] = _COMP[out[i, ::-1]]
    offsets = np.arange(n_reads + 1, dtype=np.int64) * read_len
    truth = dict(tract=tract, reverse=rev, telomeric=telomeric)
    return out.reshape(-1), offsets, truth


def split_reads(bases: np.ndarray, offsets: np.ndarray) -> list[str]:
    raw = bases.tobytes()
    return [raw[offsets[i]:offsets[i + 1]].decode("ascii") for i in range(len(offsets) - 1)]


def make_ragged_reads(n_reads: int, motif: str, seed: int, errors=ONT, len_mu: float = 9.3, len_sigma: float = 0.8,
                      min_len: int = 60, max_len: int = 60000, n_frac: float = 0.0, lower_frac: float = 0.0,
                      tract_min: int = 1000, tract_max: int = 8000, telomeric_fraction: float = 1.0):
    """Reads of log-normal length (what a real ONT file looks like: the reference's demo data span 1.6 - 48 kb) from the same
    generator: every read is cut out of a `max_len` read of make_reads -- its first L bases, or its last L when it was reverse-
    complemented, so the telomere stays at the read's end.  `n_frac` of the bases become N, `lower_frac` of the READS are
    written in lower case.  Returns (bases u8, offsets i64[n+1], truth) like make_reads."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    lens = np.clip(np.exp(rng.normal(len_mu, len_sigma, n_reads)), min_len, max_len).astype(np.int64)
    full, off, truth = make_reads(n_reads, max_len, motif, seed, errors, tract_min, tract_max, telomeric_fraction)
    full = full.reshape(n_reads, max_len)
    offsets = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    out = np.empty(int(offsets[-1]), dtype=np.uint8)
    for i in range(n_reads):
        L = int(lens[i])
        out[offsets[i]:offsets[i + 1]] = full[i, max_len - L:] if truth["reverse"][i] else full[i, :L]
    if n_frac > 0:
        k = rng.binomial(out.size, n_frac)
        out[rng.integers(0, out.size, k)] = ord("N")
    if lower_frac > 0:
        for i in np.nonzero(rng.random(n_reads) < lower_frac)[0]:
            seg = out[offsets[i]:offsets[i + 1]]
            seg |= 0x20                                   # ASCII lower case (N -> n as well)
    truth = dict(truth, length=lens)
    return out, offsets, truth

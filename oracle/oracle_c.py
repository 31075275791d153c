"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/_build/liboracle.so (oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def available() -> bool:
    try:
        lib()
        return True
    except Exception:
        return False


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "oracle.c")
        if not os.path.exists(SO) or (os.path.exists(src) and os.path.getmtime(SO) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-s", "-C", HERE])
        L = C.CDLL(SO)
        L.orc_window_count.restype = C.c_int64
        L.orc_window_count.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_window_counts.restype = C.c_int64
        L.orc_window_counts.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_trc_counts.restype = None
        L.orc_trc_counts.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_binseg_l2.restype = C.c_int
        L.orc_binseg_l2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_binseg_l2_y.restype = C.c_int
        L.orc_binseg_l2_y.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_nonoverlap_count.restype = C.c_int
        L.orc_nonoverlap_count.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_batch.restype = C.c_int64
        L.orc_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                C.POINTER(C.c_double)]
        L.orc_batch_ck.restype = C.c_int64
        L.orc_batch_ck.argtypes = L.orc_batch.argtypes + [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _pats(patterns):
    return "".join(patterns).encode(), len(patterns), len(patterns[0])


def trc_counts(seq: str, patterns, no_bp=1000):
    blob, P, k = _pats(patterns)
    cs = np.zeros(P, np.int32)
    ce = np.zeros(P, np.int32)
    b = seq.encode()
    lib().orc_trc_counts(b, len(b), blob, P, k, no_bp, cs.ctypes.data, ce.ctypes.data)
    return cs.tolist(), ce.tolist()


def window_counts(seq: str, tail: str, patterns, W, s, t, M):
    blob, P, k = _pats(patterns)
    b = seq.encode()
    n = lib().orc_window_count(len(b), W, s, t, M)
    sums = np.zeros(max(n, 1), np.int32)
    raw = np.zeros(max(n, 1) * P, np.uint8)
    got = lib().orc_window_counts(b, len(b), 0 if tail == "forward" else 1, blob, P, k, W, s, t, M, sums.ctypes.data, raw.ctypes.data)
    assert got == n
    return sums[:n], raw[: n * P].reshape(n, P)


def binseg_l2(sums, n_patterns, jump=5, min_size=2):
    s = np.ascontiguousarray(sums, np.int32)
    g = C.c_double(0)
    b = lib().orc_binseg_l2(s.ctypes.data, len(s), n_patterns, jump, min_size, C.byref(g))
    return (None if b < 0 else b), g.value


def binseg_l2_y(y, jump=5, min_size=2):
    a = np.ascontiguousarray(y, np.float64)
    g = C.c_double(0)
    b = lib().orc_binseg_l2_y(a.ctypes.data, len(a), jump, min_size, C.byref(g))
    return (None if b < 0 else b), g.value


CK_MUL = 0x9E3779B97F4A7C15


def checksum_weights(n: int) -> np.ndarray:
    """CK_MUL^i mod 2^64 for i < n (oracle.c: ck_i32 / ck_u8)."""
    w = np.empty(max(n, 1), np.uint64)
    w[0] = 1
    if n > 1:
        with np.errstate(over="ignore"):
            w[1:] = np.cumprod(np.full(n - 1, CK_MUL, np.uint64))
    return w


def checksums(values: np.ndarray, starts: np.ndarray) -> np.ndarray:
    """Per-segment checksum sum((v + 1) * CK_MUL^i) mod 2^64 of `values` cut at `starts` (n + 1 offsets), vectorised: what
    oracle.c computes per read (ck_i32 over window sums, ck_u8 over raw rows); 0 for an empty segment."""
    starts = np.asarray(starts, np.int64)
    n = len(starts) - 1
    lens = np.diff(starts)
    out = np.zeros(n, np.uint64)
    if n == 0 or starts[-1] == starts[0]:
        return out
    v = np.asarray(values)[starts[0]:starts[-1]].astype(np.uint64) + np.uint64(1)
    idx = np.arange(len(v), dtype=np.int64) - np.repeat(starts[:-1] - starts[0], lens)
    with np.errstate(over="ignore"):
        prod = v * checksum_weights(int(lens.max()))[idx]
        nz = np.nonzero(lens)[0]
        out[nz] = np.add.reduceat(prod, (starts[:-1] - starts[0])[nz])
    return out


def batch_ck(bases: np.ndarray, offsets: np.ndarray, patterns, motif_len, no_bp, min_len, cutoff, W, s, t, M,
             both_tails=False, threads=1, want_raw=False):
    """batch() plus per-read checksums of the window sums (and of the raw rows): (out[n,7], ck[n,2] uint64)."""
    blob, P, k = _pats(patterns)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n = len(offsets) - 1
    out = np.zeros((n, 7), np.int32)
    ck = np.zeros((n, 2), np.uint64)
    el = C.c_double(0)
    done = lib().orc_batch_ck(bases.ctypes.data, offsets.ctypes.data, n, blob, P, k, motif_len, no_bp, min_len, cutoff,
                              W, s, t, M, 1 if both_tails else 0, threads, 0.0, out.ctypes.data, C.byref(el), ck.ctypes.data,
                              1 if want_raw else 0)
    assert done == n
    return out, ck


def batch(bases: np.ndarray, offsets: np.ndarray, patterns, motif_len, no_bp, min_len, cutoff, W, s, t, M,
          both_tails=True, threads=1, budget_s=0.0):
    """Per-read pipeline over a batch; returns (out[n,7], reads_done, seconds)."""
    blob, P, k = _pats(patterns)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n = len(offsets) - 1
    out = np.zeros((n, 7), np.int32)
    el = C.c_double(0)
    done = lib().orc_batch(bases.ctypes.data, offsets.ctypes.data, n, blob, P, k, motif_len, no_bp, min_len, cutoff,
                           W, s, t, M, 1 if both_tails else 0, threads, budget_s, out.ctypes.data, C.byref(el))
    return out, int(done), el.value


def usable_cores():
    """CPU parallelism this process can really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes
    show 256 CPUs but run under `cpu.max = 1600000 100000`, i.e. 16 CPUs' worth of time)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:                                            # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = max(1, min(n, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def timed_baseline(seqs, patterns, motif_len, prm, budget_s=15.0, cutoff=0.7):
    """bench.py's cpu_baseline leg: the C port on all host cores, bounded in time."""
    enc = [s.encode() for s in seqs]
    offsets = np.zeros(len(enc) + 1, np.int64)
    np.cumsum([len(e) for e in enc], out=offsets[1:])
    bases = np.frombuffer(b"".join(enc), np.uint8)
    cores = usable_cores()
    out, done, el = batch(bases, offsets, patterns, motif_len, prm.no_bp, prm.min_len, cutoff, prm.window, prm.slide,
                          prm.trimfirst, prm.maxlen, both_tails=True, threads=cores, budget_s=budget_s)
    nb = int(offsets[min(done, len(enc))])
    return dict(value=nb / el, unit="bases/s", cores=cores, kind="port", reads_per_s=done / el,
                sample=f"{done} reads of the same batch through oracle/oracle.c (C restatement of allsteps.py: string windows, "
                       f"per-pattern literal search, numpy-order float64 Binseg; both tails scanned like the reference, single "
                       f"pass instead of the reference's per-read file re-parse), {cores} threads = usable CPUs "
                       f"(affinity {len(os.sched_getaffinity(0))}, cgroup quota applied), {el:.1f} s")

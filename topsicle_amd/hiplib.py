"""ctypes binding of libtopsicle_hip.so (C ABI: include/topsicle_hip.h).

This is the only way the package reaches the GPU.  There is deliberately NO CPU fallback:
if the shared library is missing, cannot be loaded, or no MI355X is visible, every entry
point raises `TopsicleHipError` -- results never silently come from anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

LIB_NAME = "libtopsicle_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

# flags (TPS_F_*)
F_STEP1, F_WINDOWS, F_BINSEG, F_STORE_SUMS, F_STORE_RAW, F_TAILS_IN = 1, 2, 4, 8, 16, 32
MAX_K, MAX_PATTERNS, MAX_SLOTS = 15, 31, 16

EXPORTS = [
    "tps_abi_version", "tps_device_count", "tps_ctx_create", "tps_ctx_destroy", "tps_last_error",
    "tps_set_patterns", "tps_batch_upload", "tps_batch_upload_packed", "tps_batch_share", "tps_host_alloc", "tps_host_free",
    "tps_batch_download_packed", "tps_batch_kmer_followers", "tps_batch_set_tails", "tps_batch_scan", "tps_sync",
    "tps_batch_results", "tps_batch_window_offsets", "tps_batch_window_sums", "tps_batch_window_raw",
    "tps_batch_raw_to_fd", "tps_batch_trc_counts", "tps_trc_counts", "tps_window_counts", "tps_binseg_l2", "tps_binseg_l2_ties", "tps_batch_read_sums", "tps_window_count",
    "tps_kernel_time_ms", "tps_kernel_time_reset", "tps_device_info", "tps_batch_kernel_info", "tps_ctx_debug_option", "tps_debug_stamps_get",
]


class TopsicleHipError(RuntimeError):
    pass


class Params(C.Structure):
    """struct tps_params -- names follow the reference CLI (Topsicle/main.py:319-334)."""
    _fields_ = [("no_bp", C.c_int32), ("min_len", C.c_int32), ("min_count", C.c_int32),
                ("window", C.c_int32), ("slide", C.c_int32), ("trimfirst", C.c_int32),
                ("maxlen", C.c_int32), ("jump", C.c_int32), ("min_size", C.c_int32),
                ("flags", C.c_uint32)]


DESC_DTYPE = np.dtype([("word_off", "<i8"), ("len", "<i4"), ("flags", "<u4")], align=True)      # struct tps_read_desc
assert DESC_DTYPE.itemsize == 16
RD_HAS_INVALID = 1

RESULT_DTYPE = np.dtype([("best_start", "<i4"), ("best_start_idx", "<i4"), ("best_end", "<i4"),
                         ("best_end_idx", "<i4"), ("tail", "<i4"), ("pass", "<i4"), ("n_win", "<i4"),
                         ("bkp", "<i4"), ("gain", "<f8"), ("flags", "<u4"), ("reserved", "<u4")], align=True)
assert RESULT_DTYPE.itemsize == 48
RES_TIE = 1            # TPS_RES_TIE: the exact tournament decided the change-point (candidates within float64 noise of each other)


def make_params(no_bp=1000, min_len=0, min_count=-1, window=100, slide=6, trimfirst=100, maxlen=20000,
                jump=5, min_size=2, flags=F_STEP1 | F_WINDOWS | F_BINSEG) -> Params:
    return Params(no_bp, min_len, min_count, window, slide, trimfirst, maxlen, jump, min_size, flags)


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libtopsicle_hip.so (built in-tree by `__graft_entry__.build()` / `make`)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("TOPSICLE_HIP_LIB") or LIB_PATH      # TOPSICLE_HIP_LIB: e.g. the diagnostics build with phase stamps
    if not os.path.exists(p):
        raise TopsicleHipError(f"{p} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
                               "topsicle_amd has no CPU fallback.")
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise TopsicleHipError(f"cannot load {p}: {e}") from e
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    proto = {
        "tps_abi_version": (C.c_int, []),
        "tps_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "tps_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "tps_ctx_destroy": (C.c_int, [vp]),
        "tps_last_error": (C.c_char_p, []),
        "tps_set_patterns": (C.c_int, [vp, C.c_char_p, i32, i32]),
        "tps_batch_upload": (C.c_int, [vp, i32, vp, vp, i64]),
        "tps_batch_upload_packed": (C.c_int, [vp, i32, vp, vp, vp, i64, i64]),
        "tps_batch_share": (C.c_int, [vp, i32, vp, i32]),
        "tps_host_alloc": (C.c_int, [vp, i64, C.POINTER(vp)]),
        "tps_host_free": (C.c_int, [vp, vp]),
        "tps_batch_download_packed": (C.c_int, [vp, i32, vp, vp, vp, i64, i64, C.POINTER(i64)]),
        "tps_batch_kmer_followers": (C.c_int, [vp, i32, i32, i32, i32, i32, i32, vp, i64, vp, i64]),
        "tps_batch_set_tails": (C.c_int, [vp, i32, vp]),
        "tps_batch_scan": (C.c_int, [vp, i32, C.POINTER(Params)]),
        "tps_sync": (C.c_int, [vp]),
        "tps_batch_results": (C.c_int, [vp, i32, vp, i64]),
        "tps_batch_window_offsets": (C.c_int, [vp, i32, vp, i64]),
        "tps_batch_window_sums": (C.c_int, [vp, i32, vp, i64]),
        "tps_batch_window_raw": (C.c_int, [vp, i32, vp, i64]),
        "tps_batch_raw_to_fd": (C.c_int, [vp, i32, vp, i64, C.c_int, i64, vp, C.POINTER(C.c_uint32), C.POINTER(i64)]),
        "tps_batch_trc_counts": (C.c_int, [vp, i32, vp, vp, i64]),
        "tps_trc_counts": (C.c_int, [vp, vp, vp, i64, i32, vp, vp]),
        "tps_window_counts": (C.c_int, [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp]),
        "tps_binseg_l2": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, vp, vp]),
        "tps_binseg_l2_ties": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp]),
        "tps_batch_read_sums": (C.c_int, [vp, i32, i64, vp, i64]),
        "tps_window_count": (i64, [i64, i32, i32, i32, i32]),
        "tps_kernel_time_ms": (C.c_int, [vp, C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "tps_kernel_time_reset": (C.c_int, [vp]),
        "tps_device_info": (C.c_int, [vp, C.c_char_p, i32]),
        "tps_batch_kernel_info": (C.c_int, [vp, i32, C.c_char_p, i32]),
        "tps_ctx_debug_option": (C.c_int, [vp, C.c_char_p, i64]),
        "tps_debug_stamps_get": (C.c_int, [vp, i32, vp, i64]),
    }
    for name, (res, args) in proto.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TopsicleHipError(f"{p} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def debug_options_from_env() -> dict:
    """$TOPSICLE_HIP_DEBUG = "key=value,key" -> {key: int}: the one environment variable through which diagnostics reach the
    library (tps_ctx_debug_option; the library itself reads no environment).  Unset in normal use."""
    out = {}
    for item in os.environ.get("TOPSICLE_HIP_DEBUG", "").split(","):
        item = item.strip()
        if item:
            key, _, val = item.partition("=")
            out[key.strip()] = int(val) if val.strip() else 1
    return out


_crc_cb = None


def _crc32_callback():
    """Address of libtopsicle_io.so's tps_crc32 (the checksum tps_batch_raw_to_fd runs over what it writes)."""
    global _crc_cb
    if _crc_cb is None:
        from . import seqio
        lib = seqio._load_io()
        if lib is None:
            raise TopsicleHipError("libtopsicle_io.so is not built (its tps_crc32 checksums the raw-count archive)")
        _crc_cb = C.cast(lib.tps_crc32, C.c_void_p)
    return _crc_cb


def pack_reads(seqs) -> tuple[np.ndarray, np.ndarray]:
    """Concatenate read strings/bytes into (bases u8, offsets i64[n+1])."""
    bs = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        np.cumsum([len(b) for b in bs], out=offsets[1:])
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), offsets


def window_count(read_len: int, window: int, slide: int, trimfirst: int, maxlen: int) -> int:
    """Windows seq_cut_windows yields for one read (allsteps.py:219, 263-271)."""
    ns = min(read_len, maxlen) - trimfirst
    if window < 1 or slide < 1 or ns < window:
        return 0
    return (ns - window) // slide + 1


def binseg_l2_float64(y, jump: int = 5, min_size: int = 2):
    """`rpt.Binseg(model="l2").fit(y).predict(n_bkps=1)` in ruptures' own float64 arithmetic (allsteps.py:310-311; ruptures
    1.1.9: CostL2.error = signal[a:b].var(axis=0).sum() * (b - a), candidates range(0, n, jump) with both sides >= min_size,
    max() over (gain, bkp) tuples).  HOST arithmetic on purpose, and only ever run on reads the device flagged TPS_RES_TIE:
    there two or more candidates are within float64 rounding noise of the best gain, upstream's answer is decided by that noise
    (np.tile([12, 13], 200): 5, where the exact rule says 395), and the only way to give upstream's answer is to repeat
    upstream's arithmetic.  Returns the split index or -1."""
    sig = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    n = sig.shape[0]
    if (n // jump) < 1 or (-(-min_size // jump)) * jump + min_size > n:
        return -1

    def cost(a, b):
        return sig[a:b].var(axis=0).sum() * (b - a)

    whole = cost(0, n)
    best = None
    for b in range(0, n, jump):
        if b >= min_size and n - b >= min_size:
            cand = (whole - cost(0, b) - cost(b, n), b)
            if best is None or cand > best:
                best = cand
    return -1 if best is None else int(best[1])


def resolve_ties(engine, slot: int, res: np.ndarray, n_patterns: int, jump: int = 5, min_size: int = 2) -> int:
    """Reads of `res` whose change-point the device decided by its exact tournament (flags & RES_TIE) get ruptures' float64
    answer instead: their S_w come down (engine.read_sums), y = S_w / P, binseg_l2_float64.  In place; returns how many."""
    if "flags" not in res.dtype.names:
        return 0
    idx = np.nonzero((res["flags"] & RES_TIE) != 0)[0]
    for i in idx:
        s_w = engine.read_sums(slot, int(i), int(res["n_win"][i]))
        res["bkp"][i] = binseg_l2_float64(np.asarray(s_w, dtype=np.float64) / n_patterns, jump, min_size)
    return len(idx)


class HipScanner:
    """One context on one MI355X.  Not thread-safe; use one per host thread and device."""

    def __init__(self, device: int = 0, lib_path: str | None = None):
        self.lib = load_library(lib_path)
        n = C.c_int(0)
        rc = self.lib.tps_device_count(C.byref(n))
        if rc != 0 or n.value <= 0:
            raise TopsicleHipError("no MI355X / HIP device visible: " + self._err() + " (topsicle_amd has no CPU fallback)")
        h = C.c_void_p()
        self._check(self.lib.tps_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.patterns: list[str] = []
        self._n = {}
        self._lib_path = lib_path
        self._helpers: list["HipScanner"] = []
        for key, value in debug_options_from_env().items():
            self.debug_option(key, value)

    def debug_option(self, key: str, value: int = 1):
        """tps_ctx_debug_option: diagnostics / tests (event_stride, no_events, force_generic, spans_per_tile, force_pair, so_order,
        stamps, file_order, no_stride, no_inline_rows).  Applies to this context and to the helper contexts it makes afterwards."""
        self._check(self.lib.tps_ctx_debug_option(self._h, key.encode(), int(value)))
        self._debug = getattr(self, "_debug", {})
        self._debug[key] = int(value)
        for h in self._helpers:
            h.debug_option(key, value)

    def stamps(self, slot: int) -> np.ndarray:
        """uint64[n, 16] phase clocks of the slot's last scan (a -DTPS_STAMPS build with the option "stamps" on)."""
        n = self._n[slot]
        out = np.zeros((n, 16), dtype=np.uint64)
        self._check(self.lib.tps_debug_stamps_get(self._h, slot, _ptr(out), n))
        return out

    def helper(self, j: int) -> "HipScanner":
        """The j-th helper context on this context's device (made on first use, closed with this one): with several pattern
        tables per batch (`--telophrase 4 5 6`) every table beyond the first is scanned by a helper that BORROWS this context's
        resident batch (share) and keeps its own table, outputs and stream -- the k passes overlap on the GPU."""
        while len(self._helpers) <= j:
            h = HipScanner(self.device, self._lib_path)
            for key, value in getattr(self, "_debug", {}).items():
                h.debug_option(key, value)
            self._helpers.append(h)
        return self._helpers[j]

    # -- plumbing
    def _err(self) -> str:
        m = self.lib.tps_last_error()
        return m.decode("utf-8", "replace") if m else ""

    def _check(self, rc: int):
        if rc != 0:
            raise TopsicleHipError(f"libtopsicle_hip error {rc}: {self._err()}")

    def close(self):
        for h in getattr(self, "_helpers", []):      # (they borrow this context's batches: they go first)
            h.close()
        self._helpers = []
        if getattr(self, "_h", None):
            self.lib.tps_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self.lib.tps_device_info(self._h, buf, 256))
        return buf.value.decode()

    # -- pattern table
    def kernel_info(self, slot: int) -> str:
        """'<kernel name> lds=<bytes per workgroup> wgs_per_cu=<n>' of the slot's last scan."""
        buf = C.create_string_buffer(256)
        self._check(self.lib.tps_batch_kernel_info(self._h, slot, buf, 256))
        return buf.value.decode()

    def set_patterns(self, patterns: list[str]):
        if not patterns:
            raise TopsicleHipError("empty pattern list")
        k = len(patterns[0])
        if any(len(p) != k for p in patterns):
            raise TopsicleHipError("all patterns of one table must have the same length")
        blob = "".join(patterns).encode("ascii")
        self._check(self.lib.tps_set_patterns(self._h, blob, len(patterns), k))
        self.patterns = list(patterns)

    # -- resident batches
    def upload(self, slot: int, bases: np.ndarray, offsets: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        self._check(self.lib.tps_batch_upload(self._h, slot, _ptr(bases), _ptr(offsets), n))
        self._n[slot] = n

    def upload_packed(self, slot: int, seq2: np.ndarray, inv, desc: np.ndarray):
        """A batch the host has packed already (2 bits per base; include/topsicle_hip.h).  Arrays that live in buffers
        from `host_alloc` are copied asynchronously: keep them unchanged until `sync()` / a result download."""
        assert seq2.dtype == np.uint32 and desc.dtype == DESC_DTYPE and (inv is None or inv.dtype == np.uint16)
        n, nw = len(desc), len(seq2)
        assert inv is None or len(inv) == nw
        self._check(self.lib.tps_batch_upload_packed(self._h, slot, _ptr(seq2), _ptr(inv), _ptr(desc), n, nw))
        self._n[slot] = n

    def share(self, slot: int, src: "HipScanner", src_slot: int):
        """This context's `slot` = the resident batch of `src`'s `src_slot` (same device, no copy): one context per pattern
        table scans one batch at the same time (tps_batch_share)."""
        self._check(self.lib.tps_batch_share(self._h, slot, src._h, src_slot))
        self._n[slot] = src._n[src_slot]

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Pinned host memory as a uint8 array (freed with the context or by `host_free`)."""
        p = C.c_void_p()
        self._check(self.lib.tps_host_alloc(self._h, int(nbytes), C.byref(p)))
        buf = (C.c_uint8 * max(int(nbytes), 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8, count=int(nbytes))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self.lib.tps_host_free(self._h, C.c_void_p(p)))

    def download_packed(self, slot: int):
        """(seq2, inv, desc) of the slot's resident packed batch (tests / diagnostics)."""
        n = self._n[slot]
        nw = C.c_int64(0)
        self._check(self.lib.tps_batch_download_packed(self._h, slot, None, None, None, n, 0, C.byref(nw)))
        seq2 = np.zeros(nw.value, np.uint32)
        inv = np.zeros(nw.value, np.uint16)
        desc = np.zeros(n, DESC_DTYPE)
        self._check(self.lib.tps_batch_download_packed(self._h, slot, _ptr(seq2), _ptr(inv), _ptr(desc), n, nw.value, None))
        return seq2, inv, desc

    def kmer_followers(self, slot: int, n_fwd: int, follow: int, lo: int = 100, hi: int = 2000, min_len: int = 0, want_hist: bool = True):
        """patterns_vs_match_heatmap's counting (descriptive_plot.py:259-291) on the resident batch:
        (picks uint32[n, 2, n_fwd, pw], hist int64[2, n_fwd, 4**follow + 1] or None)."""
        n = self._n[slot]
        pw = (hi - lo + 31) // 32
        picks = np.zeros((n, 2, n_fwd, pw), dtype=np.uint32)
        hist = np.zeros((2, n_fwd, 4 ** follow + 1), dtype=np.int64) if want_hist else None
        self._check(self.lib.tps_batch_kmer_followers(self._h, slot, n_fwd, follow, lo, hi, min_len, _ptr(picks), picks.size,
                                                      _ptr(hist), 0 if hist is None else hist.size))
        return picks, hist

    def set_tails(self, slot: int, tails: np.ndarray):
        tails = np.ascontiguousarray(tails, dtype=np.uint8)
        assert len(tails) == self._n[slot]
        self._check(self.lib.tps_batch_set_tails(self._h, slot, _ptr(tails)))

    def scan(self, slot: int, prm: Params):
        self._check(self.lib.tps_batch_scan(self._h, slot, C.byref(prm)))

    def sync(self):
        self._check(self.lib.tps_sync(self._h))

    def results(self, slot: int) -> np.ndarray:
        n = self._n[slot]
        out = np.zeros(n, dtype=RESULT_DTYPE)
        self._check(self.lib.tps_batch_results(self._h, slot, _ptr(out), n))
        return out

    def window_offsets(self, slot: int) -> np.ndarray:
        n = self._n[slot]
        out = np.zeros(n + 1, dtype=np.int64)
        self._check(self.lib.tps_batch_window_offsets(self._h, slot, _ptr(out), n + 1))
        return out

    def window_sums(self, slot: int) -> tuple[np.ndarray, np.ndarray]:
        off = self.window_offsets(slot)
        out = np.zeros(int(off[-1]), dtype=np.int32)
        self._check(self.lib.tps_batch_window_sums(self._h, slot, _ptr(out), len(out)))
        return out, off

    def read_sums(self, slot: int, read: int, n_win: int) -> np.ndarray:
        """S_w of one read of the slot's last scan (any scan that ran the window step)."""
        out = np.zeros(int(n_win), dtype=np.int32)
        self._check(self.lib.tps_batch_read_sums(self._h, slot, int(read), _ptr(out), len(out)))
        return out

    def window_raw(self, slot: int) -> tuple[np.ndarray, np.ndarray]:
        off = self.window_offsets(slot)
        p = len(self.patterns)
        out = np.zeros(int(off[-1]) * p, dtype=np.uint8)
        self._check(self.lib.tps_batch_window_raw(self._h, slot, _ptr(out), len(out)))
        return out.reshape(-1, p), off

    def raw_to_fd(self, slot: int, reads: np.ndarray, fd: int, file_off: int) -> tuple[int, int]:
        """The raw rows of `reads` (ascending read indices of the slot's batch) of the last scan, back to back at byte `file_off`
        of `fd`: device -> pinned pieces -> pwritev inside the library (tps_batch_raw_to_fd).  Returns (bytes written, their
        CRC-32 -- libtopsicle_io.so's carry-less-multiplication CRC runs over the pieces as they are written)."""
        reads = np.ascontiguousarray(reads, dtype=np.int64)
        crc, nbytes = C.c_uint32(0), C.c_int64(0)
        self._check(self.lib.tps_batch_raw_to_fd(self._h, slot, _ptr(reads), len(reads), int(fd), int(file_off), _crc32_callback(),
                                                 C.byref(crc), C.byref(nbytes)))
        return nbytes.value, crc.value

    def batch_trc_counts(self, slot: int) -> tuple[np.ndarray, np.ndarray]:
        n, p = self._n[slot], len(self.patterns)
        cs = np.zeros((n, p), dtype=np.int32)
        ce = np.zeros((n, p), dtype=np.int32)
        self._check(self.lib.tps_batch_trc_counts(self._h, slot, _ptr(cs), _ptr(ce), n))
        return cs, ce

    # -- one-shot calls
    def trc_counts(self, bases, offsets, no_bp: int = 1000):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n, p = len(offsets) - 1, len(self.patterns)
        cs = np.zeros((n, p), dtype=np.int32)
        ce = np.zeros((n, p), dtype=np.int32)
        self._check(self.lib.tps_trc_counts(self._h, _ptr(bases), _ptr(offsets), n, no_bp, _ptr(cs), _ptr(ce)))
        return cs, ce

    def window_counts(self, bases, offsets, tails, window, slide, trimfirst, maxlen, raw=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        tails = np.ascontiguousarray(tails, dtype=np.uint8)
        n, p = len(offsets) - 1, len(self.patterns)
        lens = np.diff(offsets)
        nw = np.array([window_count(int(x), window, slide, trimfirst, maxlen) for x in lens], dtype=np.int64)
        win_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(nw, out=win_off[1:])
        sums = np.zeros(int(win_off[-1]), dtype=np.int32)
        rawbuf = np.zeros(int(win_off[-1]) * p, dtype=np.uint8) if raw else None
        self._check(self.lib.tps_window_counts(self._h, _ptr(bases), _ptr(offsets), _ptr(tails), n, window, slide,
                                               trimfirst, maxlen, _ptr(win_off), _ptr(sums), _ptr(rawbuf)))
        return sums, win_off, (rawbuf.reshape(-1, p) if raw else None)

    def binseg_l2(self, sums, win_off, n_patterns, jump=5, min_size=2):
        sums = np.ascontiguousarray(sums, dtype=np.int32)
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        n = len(win_off) - 1
        bkp = np.full(n, -1, dtype=np.int32)
        gain = np.zeros(n, dtype=np.float64)
        tie = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.tps_binseg_l2_ties(self._h, _ptr(sums), _ptr(win_off), n, n_patterns, jump, min_size, _ptr(bkp), _ptr(gain), _ptr(tie)))
        for i in np.nonzero(tie)[0]:                   # float64 cannot separate the best candidates: ruptures' own arithmetic decides
            bkp[i] = binseg_l2_float64(sums[win_off[i]:win_off[i + 1]].astype(np.float64) / n_patterns, jump, min_size)
        return bkp, gain

    # -- measurement
    def kernel_time_ms(self) -> tuple[int, float, float]:
        n, tot, mean = C.c_int32(0), C.c_double(0), C.c_double(0)
        self._check(self.lib.tps_kernel_time_ms(self._h, C.byref(n), C.byref(tot), C.byref(mean)))
        return n.value, tot.value, mean.value

    def kernel_time_reset(self):
        self._check(self.lib.tps_kernel_time_reset(self._h))

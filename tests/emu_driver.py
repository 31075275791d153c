"""Test helper: builds and drives tests/emu (host emulation of the HIP kernel source)."""
import ctypes as C
import os
import subprocess

import numpy as np

from topsicle_amd import hiplib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "emu_scan.cpp")
DEPS = [SRC] + [os.path.join(HERE, "..", "topsicle_amd", "csrc", f) for f in ("tps_device.h", "tps_plan.h", "tps_pack.h")] + \
       [os.path.join(HERE, "..", "include", "topsicle_hip.h")]


def build(asan=False):
    out = os.path.join(HERE, "emu", "_build", "libtps_emu_asan.so" if asan else "libtps_emu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-shared", "-fPIC"]
    if asan:
        cmd += ["-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined"]      # (-O2 -g takes three times as long to compile)
    subprocess.check_call(cmd + ["-o", out, SRC])
    return out


_lib = None
# planner knobs of the next scan / plan calls (tps::PlanKnobs; tests set them with monkeypatch.setitem) and "val_off": the LDS layout of
# a batch without invalid letters
KNOBS = {"force_pair": 0, "so_order": 0, "val_off": 0, "raw_m": 0}


def _knobs(L):
    L.emu_set_knobs(int(KNOBS["force_pair"]), int(KNOBS["so_order"]), int(KNOBS["val_off"]))
    L.emu_set_raw_m(int(KNOBS.get("raw_m", 0)))


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build(asan=bool(os.environ.get("TPS_EMU_ASAN"))))      # TPS_EMU_ASAN=1: the -fsanitize=address,undefined build
        _lib.emu_last_error.restype = C.c_char_p
        _lib.emu_scan.restype = C.c_int
        _lib.emu_binseg.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def scan(patterns, seqs, prm, tails=None, spans_pref=0, lds_budget=160 * 1024, base_shift=0, force_generic=0):
    """Returns dict(results, c_start, c_end, win_off, sums, raw)."""
    L = lib()
    bases, offsets = hiplib.pack_reads(seqs)
    n, P, k = len(seqs), len(patterns), len(patterns[0])
    res = np.zeros(n, dtype=hiplib.RESULT_DTYPE)
    cs = np.zeros((n, P), np.int32)
    ce = np.zeros((n, P), np.int32)
    lens = np.diff(offsets)
    nw = [hiplib.window_count(int(x), prm.window, prm.slide, prm.trimfirst, prm.maxlen) for x in lens]
    tot = int(sum(nw))
    win_off = np.zeros(n + 1, np.int64)
    sums = np.zeros(max(tot, 1), np.int32)
    raw = np.zeros(max(tot * P, 1), np.uint8)
    t = None if tails is None else np.ascontiguousarray(tails, dtype=np.uint8)
    _knobs(L)
    rc = L.emu_scan("".join(patterns).encode(), P, k, _p(bases), _p(offsets), C.c_int64(n), _p(t), C.byref(prm),
                    spans_pref, lds_budget, base_shift, force_generic, _p(res), _p(cs), _p(ce), _p(win_off), _p(sums), _p(raw))
    if rc != 0:
        raise RuntimeError(f"emu_scan rc={rc}: {L.emu_last_error().decode()}")
    return dict(results=res, c_start=cs, c_end=ce, win_off=win_off, sums=sums[:tot], raw=raw[:tot * P].reshape(-1, P))


def followers(patterns, seqs, n_fwd, follow, lo=100, hi=2000, min_len=0):
    """(picks uint32[n, 2, n_fwd, pw], hist int64[2, n_fwd, 4**follow + 1]) like HipScanner.kmer_followers."""
    L = lib()
    bases, offsets = hiplib.pack_reads(seqs)
    n, P, k = len(seqs), len(patterns), len(patterns[0])
    pw = (hi - lo + 31) // 32
    picks = np.zeros((n, 2, n_fwd, pw), np.uint32)
    hist = np.zeros((2, n_fwd, 4 ** follow + 1), np.uint64)
    rc = L.emu_followers("".join(patterns).encode(), P, k, _p(bases), _p(offsets), C.c_int64(n), n_fwd, follow, lo, hi, min_len, _p(picks), _p(hist))
    if rc != 0:
        raise RuntimeError(f"emu_followers rc={rc}: {L.emu_last_error().decode()}")
    return picks, hist.astype(np.int64)


def binseg(sums, win_off, n_patterns, jump=5, min_size=2, want_tie=False):
    L = lib()
    sums = np.ascontiguousarray(sums, np.int32)
    win_off = np.ascontiguousarray(win_off, np.int64)
    n = len(win_off) - 1
    bkp = np.zeros(n, np.int32)
    gain = np.zeros(n, np.float64)
    tie = np.zeros(n, np.uint8)
    L.emu_binseg(_p(sums), _p(win_off), C.c_int64(n), n_patterns, jump, min_size, _p(bkp), _p(gain), _p(tie))
    return (bkp, gain, tie) if want_tie else (bkp, gain)


def plan(k, P, prm, max_nwin, spans_pref=0, lds_budget=160 * 1024, force_generic=0):
    out = (C.c_int32 * 10)()
    _knobs(lib())
    rc = lib().emu_plan(k, P, C.byref(prm), C.c_int64(max_nwin), spans_pref, lds_budget, force_generic, out)
    if rc != 0:
        raise RuntimeError(lib().emu_last_error().decode())
    keys = ["spans_per_tile", "span_dw", "blk_log2", "q", "r", "lw", "seq_dw", "lds_bytes", "variant", "rec_rs"]
    return dict(zip(keys, list(out)))


def plan_table(patterns, prm, max_nwin):
    """Plan of a scan with a real pattern table: dict(variant, pp_d, lds_bytes, pair_n)."""
    out = (C.c_int32 * 6)()
    k = len(patterns[0])
    _knobs(lib())
    rc = lib().emu_plan_table("".join(patterns).encode(), len(patterns), k, C.byref(prm), C.c_int64(max_nwin), out)
    if rc != 0:
        raise RuntimeError(lib().emu_last_error().decode())
    return dict(zip(["variant", "pp_d", "lds_bytes", "pair_n", "tile_full", "tw"], list(out)))


def dispatch_order(n_win, passes):
    """tps::plan_dispatch_order (csrc/tps_plan.h) as the library calls it: the read each wave slot of a launch takes, or an empty
    array when the batch keeps file order."""
    n_win = np.ascontiguousarray(n_win, dtype=np.int64)
    passes = np.ascontiguousarray(passes, dtype=np.uint8)
    out = np.zeros(max(len(n_win), 1), dtype=np.int32)
    L = lib()
    L.emu_dispatch_order.restype = C.c_int64
    m = L.emu_dispatch_order(_p(n_win), _p(passes), C.c_int64(len(n_win)), _p(out))
    return out[:m].copy()


def stride_base(patterns, prm, max_len=20000):
    """tps::stride_base (csrc/tps_plan.h): the base slide whose fused kernel a scan at prm.slide runs on, keeping every m-th window; 0 = none."""
    k = len(patterns[0])
    return int(lib().emu_stride_base("".join(patterns).encode(), len(patterns), k, C.byref(prm), C.c_int64(max_len)))

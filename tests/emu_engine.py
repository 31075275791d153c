"""Test helper: an engine with HipScanner's interface backed by the host emulation of the kernel
source (tests/emu).  Lets the host-side logic (allsteps mirror, batch driver, CLI) be tested in a
container without a GPU.  Never used by the product."""
import numpy as np

import emu_driver as emu
from topsicle_amd import hiplib


class EmuEngine:
    device = -1

    def __init__(self):
        self.patterns = []
        self.slots = {}

    def device_info(self):
        return "host emulation of tps_device.h (tests only)"

    def close(self):
        pass

    # -- several pattern tables over one batch (HipScanner.helper / share): one emulated context per table
    def helper(self, j):
        hs = self.__dict__.setdefault("_helpers", [])
        while len(hs) <= j:
            hs.append(EmuEngine())
        return hs[j]

    def share(self, slot, src, src_slot):
        if src is self:
            raise hiplib.TopsicleHipError("the batch must come from another context")
        if src_slot not in src.slots:
            raise hiplib.TopsicleHipError("no batch uploaded in the source slot")
        s = src.slots[src_slot]
        self.slots[slot] = dict(bases=s["bases"], offsets=s["offsets"], tails=None, out=None)     # (borrowed: the same arrays)

    def set_patterns(self, patterns):
        k = len(patterns[0])
        if k > hiplib.MAX_K or len(patterns) > hiplib.MAX_PATTERNS or any(set(p.upper()) - set("ACGT") for p in patterns):
            raise hiplib.TopsicleHipError("pattern table not supported")
        self.patterns = list(patterns)

    def upload(self, slot, bases, offsets):
        self.slots[slot] = dict(bases=np.array(bases, np.uint8), offsets=np.array(offsets, np.int64), tails=None, out=None)

    def upload_packed(self, slot, seq2, inv, desc):
        """Unpack to ASCII (the emulation packs again itself): codes 0..3 = A, C, T, G; flagged positions become N."""
        letters = np.frombuffer(b"ACTG", np.uint8)
        seqs = []
        for d in desc:
            w0, L = int(d["word_off"]), int(d["len"])
            nw = (L + 15) // 16
            w = np.asarray(seq2[w0:w0 + nw], np.uint32)
            codes = ((w[:, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(-1)[:L]
            s = letters[codes].copy()
            if inv is not None and (int(d["flags"]) & 1):
                v = np.asarray(inv[w0:w0 + nw], np.uint16)
                bad = ((v[:, None] >> np.arange(16, dtype=np.uint16)) & 1).reshape(-1)[:L].astype(bool)
                s[bad] = ord("N")
            seqs.append(s.tobytes())
        bases, offsets = hiplib.pack_reads(seqs)
        self.upload(slot, bases, offsets)

    def kmer_followers(self, slot, n_fwd, follow, lo=100, hi=2000, min_len=0, want_hist=True):
        picks, hist = emu.followers(self.patterns, self._seqs(self.slots[slot]), n_fwd, follow, lo, hi, min_len)
        return picks, (hist if want_hist else None)

    def set_tails(self, slot, tails):
        self.slots[slot]["tails"] = np.array(tails, np.uint8)

    def _seqs(self, s):
        raw = s["bases"].tobytes()
        o = s["offsets"]
        return [raw[o[i]:o[i + 1]].decode("latin1") for i in range(len(o) - 1)]

    def scan(self, slot, prm):
        s = self.slots[slot]
        p = hiplib.Params.from_buffer_copy(prm)
        s["out"] = emu.scan(self.patterns, self._seqs(s), p, tails=s["tails"])
        s["flags"] = p.flags

    def sync(self):
        pass

    def results(self, slot):
        return self.slots[slot]["out"]["results"]

    def window_offsets(self, slot):
        return self.slots[slot]["out"]["win_off"]

    def window_sums(self, slot):
        o = self.slots[slot]["out"]
        return o["sums"], o["win_off"]

    def read_sums(self, slot, read, n_win):
        o = self.slots[slot]["out"]
        lo = int(o["win_off"][read])
        return np.array(o["sums"][lo:lo + int(n_win)], np.int32)

    def window_raw(self, slot):
        o = self.slots[slot]["out"]
        return o["raw"], o["win_off"]

    def raw_to_fd(self, slot, reads, fd, file_off):
        """HipScanner.raw_to_fd for the emulated context: the same bytes at the same place, (bytes, crc32)."""
        import os
        import zlib
        o = self.slots[slot]["out"]
        wo, raw = o["win_off"], np.ascontiguousarray(o["raw"], np.uint8)
        blob = b"".join(raw[int(wo[i]):int(wo[i + 1])].tobytes() for i in reads)
        os.pwrite(fd, blob, file_off)
        return len(blob), zlib.crc32(blob)

    def batch_trc_counts(self, slot):
        o = self.slots[slot]["out"]
        return o["c_start"], o["c_end"]

    def trc_counts(self, bases, offsets, no_bp=1000):
        self.upload(99, bases, offsets)
        self.scan(99, hiplib.make_params(no_bp=no_bp, flags=hiplib.F_STEP1))
        return self.batch_trc_counts(99)

    def window_counts(self, bases, offsets, tails, window, slide, trimfirst, maxlen, raw=False):
        self.upload(99, bases, offsets)
        self.set_tails(99, tails)
        flags = hiplib.F_WINDOWS | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | (hiplib.F_STORE_RAW if raw else 0)
        self.scan(99, hiplib.make_params(no_bp=0, window=window, slide=slide, trimfirst=trimfirst, maxlen=maxlen, flags=flags))
        o = self.slots[99]["out"]
        return o["sums"], o["win_off"], (o["raw"] if raw else None)

    def binseg_l2(self, sums, win_off, n_patterns, jump=5, min_size=2):
        bkp, gain, tie = emu.binseg(sums, win_off, n_patterns, jump, min_size, want_tie=True)
        sums = np.asarray(sums)
        for i in np.nonzero(tie)[0]:                   # like HipScanner.binseg_l2
            bkp[i] = hiplib.binseg_l2_float64(sums[win_off[i]:win_off[i + 1]].astype(np.float64) / n_patterns, jump, min_size)
        return bkp, gain

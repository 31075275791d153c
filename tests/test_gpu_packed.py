"""The packed batch format (include/topsicle_hip.h, csrc/tps_pack.h) on the GPU: the device pack kernel behind
tps_batch_upload against a numpy restatement of the format, and host-packed uploads (tps_batch_upload_packed, pinned
and ordinary memory) against the ASCII path -- same resident words, same scan results."""
import numpy as np
import pytest

import topsicle_oracle as orc
from topsicle_amd import hiplib, synth
from packfmt import np_pack

pytestmark = pytest.mark.gpu
FULL = hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS


@pytest.fixture(scope="module")
def sc():
    s = hiplib.HipScanner(0)
    yield s
    s.close()


def _odd_reads(rng):
    seqs = []
    for L in [0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1000, 1023, 4097, 20011]:
        b = rng.integers(0, 4, L)
        s = np.frombuffer(b"ACGT", np.uint8)[b].copy()
        if L > 20:
            pos = rng.integers(0, L, max(1, L // 40))
            s[pos] = np.frombuffer(b"NnacgtRY-*", np.uint8)[rng.integers(0, 10, pos.size)]
        seqs.append(s.tobytes())
    seqs.append(b"acgtacgtacgtnnnnACGT" * 77)
    seqs.append(b"A" * 333)
    return seqs


def test_device_pack_kernel_equals_format_restatement(sc):
    rng = np.random.default_rng(5)
    bases, offsets = hiplib.pack_reads(_odd_reads(rng))
    sc.upload(0, bases, offsets)
    seq2, inv, desc = sc.download_packed(0)
    w_seq2, w_inv, w_desc = np_pack(bases, offsets)
    assert np.array_equal(desc["word_off"], w_desc["word_off"]) and np.array_equal(desc["len"], w_desc["len"])
    assert np.array_equal(desc["flags"], w_desc["flags"])
    assert np.array_equal(seq2, w_seq2) and np.array_equal(inv, w_inv)
    # a big batch too (every thread / loop shape of the kernel)
    b2, o2, _ = synth.make_reads(3000, 15000, "CCCTAA", seed=3)
    b2 = b2.copy()
    pos = rng.integers(0, b2.size, b2.size // 500)
    b2[pos] = np.frombuffer(b"Nnacgt", np.uint8)[rng.integers(0, 6, pos.size)]
    sc.upload(1, b2, o2)
    seq2, inv, desc = sc.download_packed(1)
    w_seq2, w_inv, w_desc = np_pack(b2, o2)
    assert np.array_equal(seq2, w_seq2) and np.array_equal(inv, w_inv) and np.array_equal(desc, w_desc)


@pytest.mark.parametrize("pinned", [False, True])
def test_host_packed_upload_scans_like_ascii_upload(sc, pinned):
    motif, k = "CCCTAA", 5
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    rng = np.random.default_rng(11)
    bases, offsets, _ = synth.make_reads(600, 9000, motif, seed=21, tract_min=300, tract_max=4000)
    bases = bases.copy()
    pos = rng.integers(0, bases.size, bases.size // 300)
    bases[pos] = np.frombuffer(b"NnacgtRY", np.uint8)[rng.integers(0, 8, pos.size)]
    prm = hiplib.make_params(min_len=1000, min_count=-1, flags=FULL | hiplib.F_STORE_RAW)
    sc.upload(2, bases, offsets)
    sc.scan(2, prm)
    sc.sync()
    res_a = sc.results(2).copy()
    sums_a, _ = sc.window_sums(2)
    raw_a, _ = sc.window_raw(2)
    seq2, inv, desc = np_pack(bases, offsets)
    if pinned:
        buf = sc.host_alloc(seq2.nbytes + inv.nbytes + desc.nbytes + 64)
        a = buf[: seq2.nbytes].view(np.uint32)
        o = (seq2.nbytes + 15) & ~15
        b = buf[o: o + inv.nbytes].view(np.uint16)
        o2 = (o + inv.nbytes + 15) & ~15
        d = buf[o2: o2 + desc.nbytes].view(hiplib.DESC_DTYPE)
        a[:], b[:], d[:] = seq2, inv, desc
        seq2, inv, desc = a, b, d
    sc.upload_packed(3, seq2, inv, desc)
    sc.scan(3, prm)
    sc.sync()
    res_p = sc.results(3)
    sums_p, _ = sc.window_sums(3)
    raw_p, _ = sc.window_raw(3)
    assert np.array_equal(res_a, res_p) and np.array_equal(sums_a, sums_p) and np.array_equal(raw_a, raw_p)
    if pinned:
        sc.host_free(buf)
    # spot check against the oracle
    off = sc.window_offsets(3)
    for i in range(0, 600, 61):
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        tail = ["forward", "reverse"][int(res_p["tail"][i])]
        _, counts = orc.window_count_matrix(seq, tail, pats, 100, 6, 100, 20000)
        assert np.array_equal(raw_p[off[i]:off[i + 1]], counts)


def test_packed_upload_without_inv_array_and_bad_descriptors(sc):
    sc.set_patterns(orc.kmer_table("CCCTAA", 4))
    bases, offsets, _ = synth.make_reads(50, 12000, "CCCTAA", seed=2)
    seq2, inv, desc = np_pack(bases, offsets)
    assert not inv.any()
    prm = hiplib.make_params(min_len=1000, min_count=-1, flags=FULL)
    sc.upload(4, bases, offsets)
    sc.scan(4, prm)
    sc.sync()
    want = sc.results(4).copy()
    sc.upload_packed(5, seq2, None, desc)                 # clean batch: no inv array at all
    sc.scan(5, prm)
    sc.sync()
    assert np.array_equal(sc.results(5), want)
    bad = desc.copy()
    bad["word_off"][3] += 2                               # not on a quad boundary
    with pytest.raises(hiplib.TopsicleHipError):
        sc.upload_packed(5, seq2, None, bad)
    bad = desc.copy()
    bad["len"][-1] += 4000                                # runs past the packed words
    with pytest.raises(hiplib.TopsicleHipError):
        sc.upload_packed(5, seq2, None, bad)
    bad = desc.copy()
    bad["flags"][0] = hiplib.RD_HAS_INVALID               # flagged, but no inv array
    with pytest.raises(hiplib.TopsicleHipError):
        sc.upload_packed(5, seq2, None, bad)

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


def test_clean_and_dirty_batches_alternate_in_one_slot(sc):
    """A batch without non-ACGT letters is scanned without the invalid-mask staging area (one more workgroup per CU, the
    staged bases inside row[]); the LDS layout is planned per uploaded batch.  Clean / dirty / clean batches through the
    same slot and context, ASCII and host-packed uploads, every read against the oracle."""
    motif, k, slide = "CCCTAA", 4, 6
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    prm = hiplib.make_params(no_bp=1000, min_len=3000, min_count=1, window=100, slide=slide, trimfirst=100, maxlen=20000,
                             flags=FULL)
    rng = np.random.default_rng(23)
    infos = []
    for step, dirty in enumerate([False, True, False, True, False]):
        b, o, _ = synth.make_reads(96, 9000, motif, seed=500 + step)
        b = b.copy()
        if dirty:
            pos = rng.integers(0, b.size, b.size // 300)
            b[pos] = np.frombuffer(b"Nn-RY", np.uint8)[rng.integers(0, 5, pos.size)]
        if step % 2:
            seq2, inv, desc = np_pack(b, o)
            sc.upload_packed(0, seq2, inv if dirty else None, desc)
        else:
            sc.upload(0, b, o)
        sc.scan(0, prm)
        sc.sync()
        infos.append(sc.kernel_info(0) if hasattr(sc, "kernel_info") else "")
        res = sc.results(0)
        sums, win_off = sc.window_sums(0)
        raw = b.tobytes()
        for i in range(0, 96, 5):
            seq = raw[o[i]:o[i + 1]].decode("latin1")
            cs, ce = orc.trc_counts(seq, pats)
            assert res["best_start"][i] == max(cs) and res["best_end"][i] == max(ce), (step, i)
            if res["pass"][i]:
                tail = ["forward", "reverse"][int(res["tail"][i])]
                _, counts = orc.window_count_matrix(seq, tail, pats, 100, slide, 100, 20000)
                lo, hi = win_off[i], win_off[i + 1]
                assert np.array_equal(sums[lo:hi], counts.sum(axis=1)), (step, i)
                assert res["bkp"][i] == orc.binseg_l2_exact(counts.sum(axis=1)), (step, i)
    if infos[0]:
        assert infos[0] != infos[1] and infos[0] == infos[2], infos        # the plan follows the batch (LDS bytes per workgroup differ)

"""oracle/ref_mirror.py (the reference-shaped pure-Python CPU baseline of bench.py) against the reference's own shipped
result: Topsicle_demo/telolengths_all.csv (17 reads, --pattern CCCTAAA, k = 5, window 100, slide 6, trimfirst 100)."""
import csv
import gzip
import os
import shutil

import ref_mirror


def test_process_file_reproduces_the_reference_csv(gold_dir, tmp_path):
    fq = tmp_path / "demo.fastq"
    with gzip.open(os.path.join(gold_dir, "demo_col0.fastq.gz"), "rb") as src, open(fq, "wb") as dst:
        shutil.copyfileobj(src, dst)
    rows = ref_mirror.process_file(str(fq), "CCCTAAA", 5, 9000, 0.7, 100, 6, 100, 20000)
    gold = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    assert len(rows) == len(gold) == 17
    for (rid, _tail, trc, boundary), g in zip(rows, gold):
        assert rid == g[3]
        assert f"{trc:.3f}" == g[2]
        assert boundary == int(g[4])


def test_pool_over_files(gold_dir, tmp_path):
    paths = []
    for i in range(2):
        fq = tmp_path / f"demo{i}.fastq"
        with gzip.open(os.path.join(gold_dir, "demo_col0.fastq.gz"), "rb") as src, open(fq, "wb") as dst:
            shutil.copyfileobj(src, dst)
        paths.append(str(fq))
    wall, n_pass, per_file = ref_mirror.timed_pool(paths, "CCCTAAA", 5, 9000, 0.7, 100, 6, 100, 20000, 2)
    assert n_pass == 34 and len(per_file) == 2 and wall > 0

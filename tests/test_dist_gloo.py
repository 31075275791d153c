"""N>1 path on CPU: two gloo ranks shard the demo reads, scan them and gather on rank 0."""
import csv
import json
import os
import subprocess
import sys

from topsicle_amd import dist


def test_shard_by_bases_is_a_partition():
    lens = [5, 100, 7, 30, 30, 1, 80, 2]
    for w in (1, 2, 3, 8, 11):
        sh = dist.shard_by_bases(lens, w)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(w - 1))
    assert dist.shard_by_bases([], 3) == [(0, 0)] * 3


def test_two_rank_gloo_run(tmp_path, gold_dir):
    out = tmp_path / "rows.json"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "tests", "dist_worker.py"), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = json.load(open(out))
    gold = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    assert got["world"] == 2 and got["max"] == 2.0 and got["total_reads"] == 44.0
    assert [(r[0], r[1], r[2]) for r in got["rows"]] == [(g[3], g[2], int(g[4])) for g in gold]
    (a0, a1), (b0, b1) = got["shards"]
    assert a0 == 0 and a1 == b0 and b1 == 44 and 0 < a1 < 44

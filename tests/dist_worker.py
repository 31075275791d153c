"""world_size-2 worker for tests/test_dist_gloo.py: each rank scans its shard of the demo reads
with the emulated engine, rank 0 gathers the rows and writes them as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]

import gzip  # noqa: E402

from emu_engine import EmuEngine  # noqa: E402
from topsicle_amd import allsteps, batch, dist, hiplib, seqio  # noqa: E402


def main():
    out_path = sys.argv[1]
    g = dist.Group()
    recs = list(seqio.read_records(os.path.join(ROOT, "tests", "golden", "demo_col0.fastq.gz")))
    lo, hi = dist.shard_by_bases([len(r.seq) for r in recs], g.world)[g.rank]
    pats = allsteps.patterns_to_search("CCCTAAA", 5)
    ratio = 1000 / 7
    prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, ratio, 1000), window=100, slide=6,
                             trimfirst=100, maxlen=20000)
    pool = batch.EnginePool([EmuEngine()], pats)
    rows = []
    g.barrier()
    for rs, res, _s, _r, _w in pool.scan_stream(iter(recs[lo:hi]), prm):
        for rec, r in zip(rs, res):
            if r["pass"]:
                best = int(r["best_start"] if r["tail"] == 0 else r["best_end"])
                rows.append([rec.id, f"{best / ratio:.3f}", int(r["bkp"]) * 6 + 100])
    g.barrier()
    t = g.max(float(g.rank + 1))
    tot = g.sum(float(hi - lo))
    gathered = g.gather_objects(rows)
    if g.rank == 0:
        flat = [row for part in gathered for row in part]
        json.dump(dict(rows=flat, max=t, total_reads=tot, world=g.world,
                       shards=dist.shard_by_bases([len(r.seq) for r in recs], g.world)), open(out_path, "w"))
    g.close()


if __name__ == "__main__":
    main()

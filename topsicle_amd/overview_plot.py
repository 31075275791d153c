"""`python -m topsicle_amd.overview_plot`: the exploratory overview of a run's telomeric reads -- where the motif sits in each
read, and which bases follow each of its k-mers -- as a thin driver over the GPU entry points.

Stages (the upstream script of the same name filters at a fixed TRC cutoff of 0.7 and then plots, overview_plot.py:63):
  1. one batched pass of the TRC step (kernel step 1) over every input file picks the reads to look at;
  2. their k-mer / follower counts come from tps_batch_kmer_followers (descriptive_plot.pattern_matches);
  3. matplotlib always draws descriptive_plot_<i>.png (the scatter of whole-motif hits, upstream overview_plot.py:92; host side,
     at most 41 reads) and, with --recfindingpattern, heatmap_<i>.png; --rawcount keeps the heat map's rows as CSV.
Nothing is written to a temporary FASTA: the selected records stay in memory.
"""
from __future__ import annotations

import argparse
import os

from . import allsteps, batch, descriptive_plot as dp, hiplib, seqio

TRC_CUTOFF = 0.7


def telomeric_records(path, pattern, k, min_len, engines):
    """Records of `path` that pass the TRC filter (> 0.7 over 1000 bases of either end, longer than min_len)."""
    prm = hiplib.make_params(no_bp=1000, min_len=min_len, flags=hiplib.F_STEP1,
                             min_count=allsteps.min_count_for_cutoff(TRC_CUTOFF, 1000 / len(pattern), 1000))
    pool = batch.EnginePool(engines, allsteps.patterns_to_search(pattern, k))
    keep = []
    for pb, res, _s, _r, _w in pool.scan_file(path, prm):
        keep += [pb.record(int(i)) for i in res["pass"].nonzero()[0]]
    return keep


def input_files(where):
    if not os.path.isdir(where):
        return [where]
    return [os.path.join(root, f) for root, _d, files in os.walk(where) for f in files]


def run(args, engines=None):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    os.makedirs(args.outputDir, exist_ok=True)
    phrases = args.telophrase or [len(args.pattern) - 2]
    own = engines is None
    engines = engines or [hiplib.HipScanner(args.device)]
    allsteps.set_engine(engines[0])
    try:
        shown = 0
        for path in input_files(args.inputDir):
            recs = telomeric_records(path, args.pattern, phrases[0], args.minSeqLength, engines)
            if not recs:
                continue
            shown += 1
            print(f"{path}: {len(recs)} reads with TRC > {TRC_CUTOFF}")
            dp.descriptive_plot_records(recs, os.path.basename(path).split(".")[0], args.pattern, args.minSeqLength)
            plt.savefig(os.path.join(args.outputDir, f"descriptive_plot_{shown}.png"), format="png", dpi=300)
            plt.close()
            if not args.recfindingpattern:
                continue
            for k in phrases:
                table = dp.heatmap_from_records(recs, os.path.basename(path).split(".")[0], args.pattern, k, args.minSeqLength)
                plt.savefig(os.path.join(args.outputDir, f"heatmap_{shown}.png"), format="png", dpi=300)
                plt.close()
                if args.rawcount:
                    table.to_csv(os.path.join(args.outputDir, f"heatmap_rawcount_{shown}.csv"), index=False)
        print(f"plots are in {args.outputDir}")
    finally:
        allsteps.set_engine(None)
        if own:
            for e in engines:
                e.close()


def build_parser():
    ap = argparse.ArgumentParser(description="Overview plots of telomeric reads (MI355X build)")
    ap.add_argument("--inputDir", "-i", required=True, help="input file or directory")
    ap.add_argument("--outputDir", "-o", required=True, help="output directory")
    ap.add_argument("--pattern", required=True, help="telomere repeat (5' to 3')")
    ap.add_argument("--minSeqLength", type=int, default=9000)
    ap.add_argument("--telophrase", nargs="+", type=int, help="k-mer length(s); default len(pattern) - 2")
    ap.add_argument("--recfindingpattern", action="store_true", help="also draw the k-mer / following-bases heat map")
    ap.add_argument("--rawcount", action="store_true", help="keep the heat map's rows as heatmap_rawcount_<i>.csv")
    ap.add_argument("--device", type=int, default=0)
    return ap


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()

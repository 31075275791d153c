"""Mirror of the reference's exploratory script overview_plot.py (repository root upstream; it is not an
installed entry point there either): TRC filter -> descriptive plot -> optional k-mer / following-bases
heatmap.  Same flags, same output file names.

    python -m topsicle_amd.overview_plot --inputDir reads.fastq.gz --outputDir out --pattern CCCTAA \
        [--recfindingpattern] [--rawcount] [--telophrase 4] [--minSeqLength 9000]

The filter (step 1 over every read, overview_plot.py:61-64) runs on the GPU through
`allsteps.patternTRC_count`; the plots look at the few reads that pass.

Differences from upstream, on purpose: the temporary filtered file is written INSIDE --outputDir, one per input
file (upstream concatenates `outputDir + "temp_reads_in_heatmap.fasta"` without a separator and reuses that
one name for every input, overview_plot.py:68-70), and it carries the input's real extension.
"""
from __future__ import annotations

import argparse
import datetime
import os

from . import allsteps, seqio
from .descriptive_plot import descriptive_plot, patterns_vs_match_heatmap


def tprint(*args):
    now = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    print(f"[{now}]", " ".join(str(a) for a in args))


def plot_running(args):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    filenames = []
    os.makedirs(args.outputDir, exist_ok=True)
    if os.path.isdir(args.inputDir):
        for root, _dirs, files in os.walk(args.inputDir):
            filenames += [os.path.join(root, f) for f in files]
    else:
        filenames.append(args.inputDir)

    if args.telophrase is None:
        telo_phrases = [len(args.pattern) - 2]
        tprint(f"No telophrase provided, use kmer: {telo_phrases}")
    else:
        telo_phrases = args.telophrase if isinstance(args.telophrase, list) else [args.telophrase]

    filtered_files = []
    for n, seq_loc in enumerate(filenames, start=1):
        trc_results = allsteps.patternTRC_count(seq_loc, telopattern=args.pattern, read_length=args.minSeqLength,
                                                kmer=telo_phrases[0], no_bp=1000, cutoff=0.7)
        if not trc_results:
            continue
        keep = {row[0] for row in trc_results}
        fmt = seqio.check_file_type(seq_loc) or "fasta"
        filtered = os.path.join(args.outputDir, f"temp_reads_in_heatmap_{n}.{fmt}")
        with open(filtered, "w") as out:
            for rec in seqio.read_records(seq_loc):
                if rec.id in keep:
                    seqio.write_record(out, rec, fmt)
        filtered_files.append(filtered)

    print("Loaded all data, start plotting")
    for i, seq_loc in enumerate(filtered_files, start=1):
        print(f"Descriptive plot on: {seq_loc}")
        descriptive_plot(seq_loc, pattern=args.pattern, minSeqLength=args.minSeqLength)
        plt.savefig(f"{args.outputDir}/descriptive_plot_{i}.png", format="png", dpi=300)
        plt.close()
    print(f"Descriptive plot is in here: {args.outputDir}")

    if args.recfindingpattern:
        for i, seq_loc in enumerate(filtered_files, start=1):
            for phrase in telo_phrases:
                print(f"Heatmap on {seq_loc}")
                heatmap = patterns_vs_match_heatmap(seq_loc, args.pattern, phrase, args.minSeqLength)
                plt.savefig(f"{args.outputDir}/heatmap_{i}.png", format="png", dpi=300)
                plt.close()
                if args.rawcount and heatmap is not None:
                    csv_path = f"{args.outputDir}/heatmap_rawcount_{i}.csv"
                    print(f"Saving raw count of heatmap to {csv_path}")
                    heatmap.to_csv(csv_path, index=False)
    print(f"Heatmap is in here: {args.outputDir}")

    for f in filtered_files:
        if os.path.exists(f):
            os.remove(f)
            print("clean up temp files")
    return "plotted the plot"


def main(argv=None):
    parser = argparse.ArgumentParser(description="Command line input handling for run_analysis function")
    parser.add_argument("--inputDir", type=str, help="Path to the input folder directory")
    parser.add_argument("--outputDir", type=str, help="Path to the output folder directory")
    parser.add_argument("--pattern", metavar="CHAR", type=str, required=True,
                        help="Required, Telomere repeat sequence (in 5' to 3' orientation). For e.g., in human use CCCTAA")
    parser.add_argument("--minSeqLength", type=int, help="Minimum of long read sequence, default = 9kbp", default=9000)
    parser.add_argument("--telophrase", nargs="+", type=int,
                        help="Length of telomere k-mer to search. By default will use telomere k-mer length minus 2")
    parser.add_argument("--recfindingpattern", action="store_true", help="Optional, use this to plot the heatmap of patterns vs match")
    parser.add_argument("--rawcount", action="store_true", help="Optional, save raw count results to CSV for flexibility of plotting")
    plot_running(parser.parse_args(argv))


if __name__ == "__main__":
    main()

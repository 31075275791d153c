# CLI on a synthetic file at slides 10 and 3 (the default kernels widened in round 4): the two-pass route writes the same telolengths_all.csv as the one-pass route
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from topsicle_amd import synth, e2e
b, o, _ = synth.make_reads(3000, 22000, "CCCTAA", seed=77, telomeric_fraction=0.2)
os.makedirs("gpurun_out/s10", exist_ok=True)
e2e.write_fastq("gpurun_out/s10/reads.fastq", b, o)
PY
for tp in on off; do for s in 10 3; do
python3 -m topsicle_amd.main -i gpurun_out/s10/reads.fastq -o gpurun_out/s10/out_${tp}_$s --pattern CCCTAA --slide $s --twopass $tp > gpurun_out/s10/log_${tp}_$s.txt 2>&1
done; done
for s in 10 3; do cmp gpurun_out/s10/out_on_$s/telolengths_all.csv gpurun_out/s10/out_off_$s/telolengths_all.csv && echo "slide $s: two-pass == one-pass, rows: $(wc -l < gpurun_out/s10/out_on_$s/telolengths_all.csv)"; done
python3 - <<'PY'
# oracle spot check of slide 10 boundaries
import csv, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "oracle"))
import topsicle_oracle as orc
from topsicle_amd import seqio
recs = {r.id: r.seq for r in seqio.read_records("gpurun_out/s10/reads.fastq")}
rows = list(csv.DictReader(open("gpurun_out/s10/out_on_10/telolengths_all.csv")))
print(rows[0])
PY
rm -rf gpurun_out/s10/reads.fastq gpurun_out/s10/out_*

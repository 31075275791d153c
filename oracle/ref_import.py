"""TEST INFRASTRUCTURE ONLY -- imports the reference's own allsteps.py in the BUILD container.

`/root/reference/Topsicle/allsteps.py` imports Bio, ruptures and seaborn at module top
(allsteps.py:14-19, 28-31).  None of the three is installed here, so this module seeds
`sys.modules` with inert stand-ins *before* importing the reference:

  * `Bio.SeqIO.parse` -- a minimal FASTA/FASTQ record reader (ids = first token of the header,
    exactly what Biopython exposes as `record.id`).  It touches parsing only.
  * `seaborn`         -- `color_palette`/`set_style` no-ops (styling only).
  * `ruptures`        -- `Binseg(model="l2").fit(y).predict(pen, n_bkps)` which RECORDS the `y`
    vector the reference hands over (so goldens hold the reference's own y) and answers with
    the from-definition restatement in `topsicle_oracle.binseg_l2_numpy` (ruptures 1.1.9 is an
    un-vendored third-party dependency: requirements.txt:7, setup.py:14).

Nothing in here travels to the GPU box and nothing outside `oracle/gen_golden.py` and the
`-m "not gpu"` cross-check tests (which skip when /root/reference is absent) imports it.
"""
import gzip
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TOPSICLE_REFERENCE", "/root/reference")

CAPTURED_Y = []          # every y vector handed to the ruptures stand-in, in call order


class _Seq(str):
    """str with the few Bio.Seq behaviours allsteps.py relies on (slicing, upper, len)."""

    def __getitem__(self, item):
        return _Seq(str.__getitem__(self, item))

    def upper(self):
        return _Seq(str.upper(self))


class _Record:
    def __init__(self, rid, seq, desc="", qual=None):
        self.id = rid
        self.name = rid
        self.description = desc
        self.seq = _Seq(seq)
        self.qual = qual

    def __len__(self):
        return len(self.seq)


def _parse(handle, fmt):
    if fmt == "fasta":
        rid, desc, chunks = None, "", []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if rid is not None:
                    yield _Record(rid, "".join(chunks), desc)
                desc = line[1:]
                rid = desc.split()[0] if desc.split() else ""
                chunks = []
            elif rid is not None:
                chunks.append(line.strip())
        if rid is not None:
            yield _Record(rid, "".join(chunks), desc)
    elif fmt == "fastq":
        while True:
            head = handle.readline()
            if not head:
                return
            head = head.rstrip("\r\n")
            if not head:
                continue
            seq = handle.readline().rstrip("\r\n")
            handle.readline()
            qual = handle.readline().rstrip("\r\n")
            desc = head[1:]
            yield _Record(desc.split()[0] if desc.split() else "", seq, desc, qual)
    else:
        raise ValueError(fmt)


def _write(records, handle, fmt):
    """Bio.SeqIO.write's text for the two formats the reference writes (main.py:83-86): FASTA with the sequence wrapped at 60
    columns, four-line FASTQ; the title is the description when it begins with the id (Biopython's FastaWriter / FastqPhredWriter
    -- which is every record that came out of SeqIO.parse)."""
    recs = [records] if isinstance(records, _Record) else list(records)
    for r in recs:
        title = r.description if (r.description and r.description.split(None, 1)[:1] == [r.id]) else \
            (r.id + (" " + r.description if r.description and r.description != r.id else ""))
        s = str(r.seq)
        if fmt == "fasta":
            handle.write(">" + title + "\n")
            for i in range(0, len(s), 60):
                handle.write(s[i:i + 60] + "\n")
        elif fmt == "fastq":
            handle.write("@" + title + "\n" + s + "\n+\n" + (r.qual or "") + "\n")
        else:
            raise ValueError(fmt)
    return len(recs)


def _install_standins():
    if "Topsicle.allsteps" in sys.modules:
        return
    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqio.parse = _parse
    seqio.write = _write
    qualio = types.ModuleType("Bio.SeqIO.QualityIO")
    qualio.FastqGeneralIterator = lambda handle: iter(())
    seqio.QualityIO = qualio
    bio.SeqIO = seqio
    sys.modules.update({"Bio": bio, "Bio.SeqIO": seqio, "Bio.SeqIO.QualityIO": qualio})

    sns = types.ModuleType("seaborn")
    sns.color_palette = lambda *a, **k: [(0, 0, 0)] * 30
    sns.set_style = lambda *a, **k: None

    def _heatmap(*a, **k):                     # styling only: hand back the current axes like seaborn does
        import matplotlib.pyplot as plt
        return plt.gca()
    sns.heatmap = _heatmap
    sys.modules["seaborn"] = sns

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import topsicle_oracle as orc

    rpt = types.ModuleType("ruptures")

    class Binseg:
        def __init__(self, model="l2", jump=5, min_size=2, **kw):
            assert model == "l2"
            self.jump, self.min_size = jump, min_size

        def fit(self, signal):
            import numpy as np
            self.y = np.asarray(signal, dtype=float).copy()
            CAPTURED_Y.append(self.y)
            return self

        def predict(self, pen=None, n_bkps=None):
            assert n_bkps == 1
            bkp, _gain = orc.binseg_l2_numpy(self.y, self.jump, self.min_size)
            if bkp is None:
                raise RuntimeError("BadSegmentationParameters")
            return [bkp, len(self.y)]

    rpt.Binseg = Binseg
    sys.modules["ruptures"] = rpt

    import matplotlib
    matplotlib.use("Agg")


def load_reference_allsteps():
    """Return the reference's `Topsicle.allsteps` module (its own code, unchanged)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise FileNotFoundError(REFERENCE_ROOT)
    _install_standins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    # import the submodule directly: Topsicle/__init__.py star-imports descriptive_plot too,
    # which is harmless with the stand-ins in place.
    return importlib.import_module("Topsicle.allsteps")


def load_reference_main():
    """The reference's `Topsicle.main` module (its own code, unchanged): main() reads sys.argv, analysis_run(args) forks its
    Pool over the input files (the stand-ins are inherited by the children)."""
    load_reference_allsteps()
    import importlib
    return importlib.import_module("Topsicle.main")


def run_reference_main(argv):
    """Topsicle/main.py:main() on `argv` in this process: returns the SystemExit code (None when it returned)."""
    import contextlib
    import io
    m = load_reference_main()
    old = sys.argv
    sys.argv = ["topsicle"] + list(argv)
    if hasattr(m.tprint, "logfile"):
        del m.tprint.logfile
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m.main()
    except SystemExit as e:
        return e.code if e.code is not None else 0
    finally:
        sys.argv = old
        import matplotlib.pyplot as plt
        plt.close("all")
    return None


def load_reference_descriptive_plot():
    """The reference's `Topsicle.descriptive_plot` module (its own code, unchanged; plots go to the Agg backend)."""
    load_reference_allsteps()
    import importlib
    return importlib.import_module("Topsicle.descriptive_plot")


def read_records(path):
    """All (id, sequence) pairs of a FASTA/FASTQ(.gz) file through the same stand-in parser."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8") as h:
        first = h.read(1)
    fmt = "fastq" if first == "@" else "fasta"
    with opener(path, "rt", encoding="utf-8") as h:
        return [(r.id, str(r.seq)) for r in _parse(h, fmt)]

"""topsicle_amd -- MI355X-native telomere k-mer scanner and boundary caller.

Drop-in for the hot path of Topsicle (Topsicle/allsteps.py): per-read TRC counts, sliding-window
telomere k-mer counts and the single-split change-point call run as hand-written HIP kernels
(gfx950) behind the reference's own Python function signatures (`topsicle_amd.allsteps`) and
`topsicle` command line (`topsicle_amd.main`).  There is no CPU fallback.
"""
__version__ = "0.1.0"

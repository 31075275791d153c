"""Multi-process plumbing for one-process-per-GPU runs (bench.py --gpus N, sharded CLI jobs).

The hot path has no exchange step: reads are independent, so ranks only need (i) a disjoint
shard of the reads, (ii) a barrier around the timed region and (iii) a small host-side gather of
per-read rows.  torch.distributed with the gloo backend does that over 127.0.0.1; no tensor
ever lives on a GPU here and no RCCL collective is issued.
"""
from __future__ import annotations

import os


class Group:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self._dist = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the container hostname may not resolve
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj):
        """List of every rank's object on rank 0 (None elsewhere)."""
        if self._dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self._dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None


def shard_by_bases(lengths, world: int):
    """Contiguous shards of reads with ~equal total bases: list of (lo, hi) per rank."""
    import numpy as np
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return [(0, 0)] * world
    cum = np.cumsum(lengths)
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]

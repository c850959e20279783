"""Host-side rendezvous for one-process-per-GPU runs (bench.py, launched by torch.distributed.run).

torch.distributed (gloo) is used ONLY to hand the RCCL unique id from rank 0 to the other ranks, to
barrier around the timed region and to take the max over ranks; the data path (all-reduce over xGMI)
is RCCL inside libnanollama_hip.so on the engine's own HIP stream.
"""
from __future__ import annotations

import os
from typing import Callable, Optional


class Rendezvous:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._dist = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    def broadcast_bytes(self, make: Callable[[], bytes]) -> Optional[bytes]:
        """Rank 0 calls make(); every rank returns the same bytes."""
        if not self._dist:
            return make()
        box = [make() if self.rank == 0 else None]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self) -> None:
        if self._dist:
            self._dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self._dist:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self) -> None:
        if self._dist:
            self._dist.destroy_process_group()
            self._dist = None

"""Host tensors over a torch.distributed process group (gloo): the transport of the CPU tests (tests/test_slabs_gloo.py)
and of bench.py's one-GPU rehearsal (--transport host).  Test infrastructure: the product package imports no framework;
its exchange is RCCL inside libvoxcarve (vc_allgather)."""
import numpy as np

from voxcarve.slabs import merge_rank_entries, merge_rank_lists


class TorchTransport:
    """Variable-length all-gather of host records over a torch.distributed group (gloo)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self._gathered = None

    def allgather_records(self, local_records):
        parts = self._parts(local_records)
        self._gathered = merge_rank_lists(parts)
        counts = np.array([p.size for p in parts], dtype=np.uint64)
        return counts, int(self._gathered.size)

    def allgather_entries(self, local_entries):
        """Host exchange of the compact form: returns all ranks' entries [M, 2] for engine.expand_entries."""
        return merge_rank_entries(self._parts(np.ascontiguousarray(local_entries, dtype=np.uint64).ravel()))

    def _parts(self, local_u64):
        torch, dist = self._torch, self._dist
        world = dist.get_world_size(self._group)
        local = np.ascontiguousarray(local_u64, dtype=np.uint64)
        counts_t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts_t, torch.tensor([local.size], dtype=torch.int64), group=self._group)
        counts = np.array([int(t.item()) for t in counts_t], dtype=np.uint64)
        cap = int(counts.max()) if counts.size else 0
        send = torch.zeros(max(cap, 1), dtype=torch.int64)
        send[:local.size] = torch.from_numpy(local.view(np.int64).copy())
        recv = [torch.zeros(max(cap, 1), dtype=torch.int64) for _ in range(world)]
        dist.all_gather(recv, send, group=self._group)
        return [recv[r][:int(counts[r])].numpy().view(np.uint64) for r in range(world)]

    def fetch(self):
        return self._gathered

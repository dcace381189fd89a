"""Multi-GPU carve: block-split of the grid along z and the survivor all-gather.

Every voxel is independent (reference voxel_reconstruction.py:105-122 loops over them),
so the grid shards with no exchange until the end: rank r of G owns the z-slab
iz in [floor(r*nz/G), floor((r+1)*nz/G)), a contiguous range of the linear voxel index
(i = iz*nx*ny + ...), and the rank-ordered concatenation of the per-rank ascending survivor
lists IS the reference's output order (assignment.py:121-133).  One process per GPU; the
one collective is a variable-length all-gather of 8-byte survivor records, done by RCCL
over xGMI inside libvoxcarve (vc_allgather).  The reference has no distributed code.

A ``Transport`` moves the records between ranks:
  RcclTransport   device-to-device via the engine's RCCL communicator (the product path)
  TorchTransport  host tensors via a torch.distributed process group (gloo) -- exercises the
                  same split / merge logic on CPU-only machines (tests)
"""
import numpy as np


def slab_range(nz, n_ranks, rank):
    """z-range [z0, z1) of rank r: floor split, empty slabs allowed when n_ranks > nz."""
    if not (0 <= rank < n_ranks):
        raise ValueError("rank %d not in [0,%d)" % (rank, n_ranks))
    return (rank * nz) // n_ranks, ((rank + 1) * nz) // n_ranks


def slab_index_range(grid, n_ranks, rank):
    """Linear-index range [i0, i1) of rank r's slab."""
    nx, ny, nz = grid
    z0, z1 = slab_range(nz, n_ranks, rank)
    return z0 * nx * ny, z1 * nx * ny


def balanced_bounds(weights, chunk, nz, n_ranks):
    """Slab boundaries [b_0 = 0, ..., b_G = nz] at multiples of ``chunk`` layers so that every rank gets about
    the same share of ``weights`` (cost of each chunk of z-layers: the hull is not spread evenly over z, and
    the hierarchical kernel's time follows the undecided words, not the voxel count).  Contiguous and
    ascending, so the rank-ordered concatenation stays the reference order.  Deterministic in its inputs:
    every rank must pass the same weights (share them with a max-reduction)."""
    w = np.maximum(np.asarray(weights, dtype=np.float64), 0.0)
    nchunks = (nz + chunk - 1) // chunk
    if w.size != nchunks:
        raise ValueError("%d weights for %d chunks" % (w.size, nchunks))
    if n_ranks < 1:
        raise ValueError("n_ranks %d" % n_ranks)
    total = float(w.sum())
    if total <= 0.0:
        return [(r * nz) // n_ranks for r in range(n_ranks + 1)]
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for r in range(1, n_ranks):
        target = total * r / n_ranks
        k = int(np.searchsorted(cum, target))              # first chunk edge at or past the target
        if k > 0 and target - cum[k - 1] < cum[min(k, nchunks)] - target:
            k -= 1                                          # the nearer edge
        k = max(k, 0)
        z = min(k * chunk, nz)
        bounds.append(max(z, bounds[-1]))
    bounds.append(nz)
    return bounds


def measure_chunk_cost(engine, nz, chunk, mode="lut", repeats=5, reduce_max=None, **carve_kwargs):
    """Kernel time of the carve on every chunk of ``chunk`` z-layers (current masks of slot 0), in ms.
    ``reduce_max`` (e.g. engine.comm_max) makes the figures identical on every rank."""
    out = []
    for z in range(0, nz, chunk):
        engine.set_slab(z, min(z + chunk, nz))
        if mode == "lut":
            engine.build_lut()
        # records kept: with a communicator attached a records=False step is a collective call, and this
        # loop must not depend on what the other ranks are doing (only the kernel time is read)
        for _ in range(2):
            engine.carve(mode=mode, **carve_kwargs)
        engine.timing(reset=True)
        for _ in range(repeats):
            engine.carve(mode=mode, **carve_kwargs)
        tm = engine.timing()
        out.append(tm["carve_ms_sum"] / max(1, tm["carve_launches"]))
    floor = 0.6 * min(out)                                  # most of an empty chunk's time is launch + start-up
    out = [max(t - floor, 1e-4) for t in out]
    if reduce_max is not None:
        out = [reduce_max(t) for t in out]
    return out


def merge_rank_lists(per_rank_records):
    """Concatenate per-rank record arrays in rank order and verify global ascending order."""
    parts = [np.ascontiguousarray(p, dtype=np.uint64) for p in per_rank_records]
    out = np.concatenate(parts) if parts else np.empty(0, np.uint64)
    idx = out.astype(np.uint32)
    if idx.size > 1 and not np.all(idx[1:] > idx[:-1]):
        raise RuntimeError("gathered survivor list is not strictly ascending: slabs overlap or are misordered")
    return out


def merge_rank_entries(per_rank_entries):
    """Concatenate per-rank {bits, base} entry arrays ([M_r, 2] u64, vc_pack_entries) in rank order and verify
    that the words ascend -- the form vc_expand_entries takes."""
    parts = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 2) for p in per_rank_entries]
    out = np.concatenate(parts) if parts else np.empty((0, 2), np.uint64)
    base = out[:, 1].astype(np.int64)
    if base.size > 1 and not np.all(base[1:] >= base[:-1] + 64):
        raise RuntimeError("gathered occupancy words overlap or are misordered")
    return out


class RcclTransport:
    """All-gather on the device through the engine's communicator (vc_comm_init / vc_allgather)."""

    def __init__(self, engine, n_ranks, rank, uid):
        self.engine = engine
        engine.comm_init(n_ranks, rank, uid)

    def allgather_records(self, local_records_unused=None):
        counts, total = self.engine.allgather()
        return counts, total

    def fetch(self):
        return self.engine.fetch_gathered()


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


def file_rendezvous(rank, payload=None, tag="uid", timeout=120.0):
    """One-shot broadcast from rank 0 through the node-local filesystem (one process per GPU on ONE
    node).  Used to hand the RCCL unique id to the other ranks without importing a framework: a
    process that uses vc_comm_* must not load a second ROCm runtime (torch wheels bundle their own
    libhsa / librccl, and RCCL then resolves HSA from the wrong copy).  The directory is keyed by the
    launcher's PID (identical for all ranks of one launch, unique per launch) and MASTER_PORT."""
    import os
    import time
    d = os.path.join("/tmp", "voxcarve_rdzv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
    path = os.path.join(d, tag)
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)
        return payload
    t_end = time.time() + timeout
    while time.time() < t_end:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if data:
                return data
        except FileNotFoundError:
            pass
        time.sleep(0.002)
    raise TimeoutError("rendezvous file %s did not appear" % path)


def file_rendezvous_cleanup(rank):
    import os
    import shutil
    if rank == 0:
        shutil.rmtree(os.path.join("/tmp", "voxcarve_rdzv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))),
                      ignore_errors=True)


class ShmTransport:
    """Last-resort exchange through /dev/shm files (one node): every rank writes its records, waits for
    the others' files, reads them in rank order.  Slow (device -> host -> shared memory -> host) but free
    of any library; bench.py falls back to it only if the RCCL communicator cannot be created, and says so."""

    def __init__(self, n_ranks, rank):
        import os
        self.n_ranks, self.rank = n_ranks, rank
        self.dir = os.path.join("/dev/shm", "voxcarve_xchg_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
        os.makedirs(self.dir, exist_ok=True)
        self.round = 0
        self._gathered = None

    def _path(self, rnd, r, kind):
        import os
        return os.path.join(self.dir, "%s_%d_%d.npy" % (kind, rnd, r))

    def _wait(self, path, timeout=120.0):
        import os
        import time
        t_end = time.time() + timeout
        while not os.path.exists(path):
            if time.time() > t_end:
                raise TimeoutError("exchange file %s did not appear" % path)
            time.sleep(0.0005)

    def allgather_records(self, local_records):
        parts = self._parts(local_records)
        self._gathered = merge_rank_lists(parts)
        counts = np.array([p.size for p in parts], dtype=np.uint64)
        return counts, int(counts.sum())

    def allgather_entries(self, local_entries):
        """Host exchange of the compact form: returns all ranks' entries [M, 2] for engine.expand_entries."""
        return merge_rank_entries(self._parts(np.ascontiguousarray(local_entries, dtype=np.uint64).ravel()))

    def _parts(self, local_u64):
        import os
        rnd = self.round
        self.round += 1
        local = np.ascontiguousarray(local_u64, dtype=np.uint64)
        tmp = self._path(rnd, self.rank, "tmp")
        np.save(tmp, local)
        os.replace(tmp, self._path(rnd, self.rank, "rec"))
        parts = []
        for r in range(self.n_ranks):
            path = self._path(rnd, r, "rec")
            self._wait(path)
            parts.append(local if r == self.rank else np.load(path))
        # everyone has read round rnd once all ranks have posted their "done" marker; then remove own file
        open(self._path(rnd, self.rank, "done"), "w").close()
        for r in range(self.n_ranks):
            self._wait(self._path(rnd, r, "done"))
        if rnd >= 1:                                        # files of the round before are no longer needed by anyone
            for kind in ("rec", "done"):
                try:
                    os.remove(self._path(rnd - 1, self.rank, kind))
                except OSError:
                    pass
        return parts

    def barrier(self):
        self._parts(np.empty(0, np.uint64))

    def max(self, x):
        import os
        rnd = self.round
        self.round += 1
        tmp = self._path(rnd, self.rank, "tmpv")
        np.save(tmp, np.array([x], dtype=np.float64))
        os.replace(tmp, self._path(rnd, self.rank, "val"))
        vals = []
        for r in range(self.n_ranks):
            path = self._path(rnd, r, "val")
            self._wait(path)
            vals.append(float(np.load(path)[0]))
        return max(vals)

    def fetch(self):
        return self._gathered

    def close(self):
        """Collective: every rank says goodbye; rank 0 removes the directory once all have."""
        import os
        import shutil
        open(os.path.join(self.dir, "bye_%d" % self.rank), "w").close()
        if self.rank == 0:
            for r in range(self.n_ranks):
                self._wait(os.path.join(self.dir, "bye_%d" % r))
            shutil.rmtree(self.dir, ignore_errors=True)


def carve_slab(engine, grid, n_ranks, rank, **carve_kwargs):
    """Restrict ``engine`` to rank's slab and carve it; returns the local survivor count."""
    z0, z1 = slab_range(grid[2], n_ranks, rank)
    if engine.slab != (z0, z1):
        engine.set_slab(z0, z1)
    return engine.carve(**carve_kwargs)

"""Multi-GPU carve: block-split of the grid along z and the survivor all-gather.

Every voxel is independent (reference voxel_reconstruction.py:105-122 loops over them),
so the grid shards with no exchange until the end: rank r of G owns the z-slab
iz in [floor(r*nz/G), floor((r+1)*nz/G)), a contiguous range of the linear voxel index
(i = iz*nx*ny + ...), and the rank-ordered concatenation of the per-rank ascending survivor
lists IS the reference's output order (assignment.py:121-133).  One process per GPU; the
one collective is a variable-length all-gather of 8-byte survivor records, done by RCCL
over xGMI inside libvoxcarve (vc_allgather).  The reference has no distributed code.

The product path is device-to-device: CarveEngine.comm_init / allgather (RCCL inside libvoxcarve).  One host-side
transport moves the same data when no communicator can be made:
  ShmTransport    /dev/shm files, bench.py --allow-host-fallback only
(The torch.distributed / gloo transport of the CPU tests and of the one-GPU rehearsal lives in tests/torch_transport.py:
nothing in this package imports a framework.)
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


def _launch_dir(ppid=None):
    """Node-local directory of THIS launch: keyed by the launcher's PID *and its start time* (identical for all ranks of one
    launch, different for any other launch even when the PID is reused), MASTER_PORT and the elastic run id."""
    import os
    ppid = os.getppid() if ppid is None else ppid
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]          # field 22: start time in clock ticks since boot
    except OSError:
        start = "0"
    return os.path.join("/tmp", "voxcarve_rdzv_%d_%s_%s_%s" % (ppid, start, os.environ.get("MASTER_PORT", "0"),
                                                              os.environ.get("TORCHELASTIC_RUN_ID", "none")))


def _pid_alive(pid):
    import os
    try:
        os.kill(pid, 0)
        return True
    except ProcessLookupError:
        return False
    except PermissionError:
        return True


def file_rendezvous(rank, payload=None, tag="uid", timeout=120.0):
    """One-shot broadcast from rank 0 through the node-local filesystem (one process per GPU on ONE
    node).  Used to hand the RCCL unique id to the other ranks without importing a framework: a
    process that uses vc_comm_* must not load a second ROCm runtime (torch wheels bundle their own
    libhsa / librccl, and RCCL then resolves HSA from the wrong copy).  The file starts with the writer's PID:
    a reader ignores a file whose writer is no longer alive (left behind by a launch that crashed)."""
    import os
    import struct
    import time
    d = _launch_dir()
    path = os.path.join(d, tag)
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        for name in os.listdir(d):                          # nothing of an earlier launch under the same key survives
            try:
                os.remove(os.path.join(d, name))
            except OSError:
                pass
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(struct.pack("<q", os.getpid()) + payload)
        os.replace(tmp, path)
        return payload
    t_end = time.time() + timeout
    while time.time() < t_end:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) > 8 and _pid_alive(struct.unpack("<q", data[:8])[0]):
                return data[8:]
        except FileNotFoundError:
            pass
        time.sleep(0.002)
    raise TimeoutError("rendezvous file %s did not appear" % path)


def file_all_flags(rank, world, tag, text, timeout=180.0):
    """Every rank posts a short text ("" = fine) under `tag`; returns the list of all ranks' texts, the same on every rank.
    What makes a decision collective without a communicator: bench.py's "did the RCCL set-up work everywhere?"."""
    import os
    import time
    d = _launch_dir()
    os.makedirs(d, exist_ok=True)
    tmp = os.path.join(d, "%s_%d.tmp" % (tag, rank))
    with open(tmp, "w") as f:
        f.write("%d\n%s" % (os.getpid(), text))
    os.replace(tmp, os.path.join(d, "%s_%d" % (tag, rank)))
    out = []
    t_end = time.time() + timeout
    for r in range(world):
        path = os.path.join(d, "%s_%d" % (tag, r))
        while True:
            try:
                with open(path) as f:
                    pid, _, body = f.read().partition("\n")
                if pid:                 # (no liveness check here: a rank may have posted and left; rank 0 emptied the
                    out.append(body)    # directory before it published the unique id, and flags are posted after reading that)
                    break
            except (FileNotFoundError, ValueError):
                pass
            if time.time() > t_end:
                out.append("rank %d never reported (timeout)" % r)
                break
            time.sleep(0.002)
    return out


def file_rendezvous_cleanup(rank, world=1, timeout=60.0):
    """Every rank says it is done with the launch directory; rank 0 removes it once all have (nobody else waits)."""
    import os
    import shutil
    import time
    d = _launch_dir()
    try:
        open(os.path.join(d, "bye_%d" % rank), "w").close()
    except OSError:
        return
    if rank == 0:
        t_end = time.time() + timeout
        while time.time() < t_end and not all(os.path.exists(os.path.join(d, "bye_%d" % r)) for r in range(world)):
            time.sleep(0.002)
        shutil.rmtree(d, ignore_errors=True)


class ShmTransport:
    """Last-resort exchange through /dev/shm files (one node): every rank writes its records, waits for
    the others' files, reads them in rank order.  Slow (device -> host -> shared memory -> host) but free
    of any library; bench.py falls back to it only if the RCCL communicator cannot be created, and says so."""

    def __init__(self, n_ranks, rank):
        import os
        self.n_ranks, self.rank = n_ranks, rank
        self.dir = os.path.join("/dev/shm", os.path.basename(_launch_dir()).replace("rdzv", "xchg"))
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

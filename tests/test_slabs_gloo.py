"""world_size-2/3 gloo runs of the z-slab split + variable-length all-gather on CPU.

The per-rank carve is stood in for by the oracle (this is a test of the partition and
exchange logic of voxcarve.slabs, which on GPUs runs over RCCL inside libvoxcarve)."""
import os

import numpy as np
import pytest

import fixtures_util as fx


def _launch_key():
    """A number that stands in for MASTER_PORT where it is only a KEY of the launch directory (no socket is opened on it)."""
    return 20000 + os.getpid() % 20000


def _init_gloo(rank, world, out_dir):
    """Process group without a pre-probed port: rank 0 listens on port 0 (the kernel picks a free one) and publishes the
    number through the test's directory; the others connect to it.  Probing a port, closing it and handing the number to
    someone who listens later leaves a window for EADDRINUSE (GPUTEST_r02)."""
    import time
    from datetime import timedelta
    import torch.distributed as dist
    path = os.path.join(out_dir, "store_port")
    if rank == 0:
        store = dist.TCPStore("127.0.0.1", 0, world, is_master=True, timeout=timedelta(seconds=120), wait_for_workers=False)
        with open(path + ".tmp", "w") as f:
            f.write(str(store.port))
        os.replace(path + ".tmp", path)
    else:
        t_end = time.time() + 120
        while not os.path.exists(path):
            if time.time() > t_end:
                raise TimeoutError("rank 0 never published its store port")
            time.sleep(0.005)
        store = dist.TCPStore("127.0.0.1", int(open(path).read()), world, is_master=False, timeout=timedelta(seconds=120))
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
    return dist


def _entries_from_indices(idx, i0):
    """Host restatement of vc_pack_entries: non-zero 64-voxel words of a slab starting at global index i0."""
    local = idx.astype(np.int64) - i0
    w = local >> 6
    words, inv = np.unique(w, return_inverse=True)
    bits = np.zeros(words.size, np.uint64)
    np.bitwise_or.at(bits, inv, np.uint64(1) << (local & 63).astype(np.uint64))
    return np.stack([bits, (words * 64 + i0).astype(np.uint64)], axis=1)


def _indices_from_entries(ent):
    out = []
    for bits, base in ent:
        b = int(bits)
        out.extend(int(base) + k for k in range(64) if (b >> k) & 1)
    return np.array(out, dtype=np.uint32)


def _worker(rank, world, grid, out_dir):
    dist = _init_gloo(rank, world, out_dir)
    try:
        from oracle import carve_c
        from voxcarve import slabs
        cams, masks = fx.golden_cameras(), fx.golden_masks()
        frames = fx.synthetic_frames(4, *masks[0].shape)
        i0, i1 = slabs.slab_index_range(grid, world, rank)
        res = carve_c.carve(*grid, fx.oracle_cams(cams), masks, frames, index_range=(i0, i1), threads=1)
        b = res["bgr"].astype(np.uint64)
        rec = res["idx"].astype(np.uint64) | (b[:, 2] << 32) | (b[:, 1] << 40) | (b[:, 0] << 48) | (1 << 56)
        from torch_transport import TorchTransport
        tr = TorchTransport()
        counts, total = tr.allgather_records(rec)
        assert int(counts[rank]) == rec.size
        np.save(os.path.join(out_dir, "gathered_%d.npy" % rank), tr.fetch())
        np.save(os.path.join(out_dir, "counts_%d.npy" % rank), counts)
        # the compact exchange form over the same transport (what bench.py --transport host moves)
        ent = tr.allgather_entries(_entries_from_indices(res["idx"], i0))
        np.save(os.path.join(out_dir, "entries_%d.npy" % rank), ent)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid", [(2, (32, 32, 32)), (3, (16, 64, 7))])
def test_slab_split_allgather_equals_single_rank(built, tmp_path, world, grid):
    import torch.multiprocessing as mp
    from oracle import carve_c
    from voxcarve.engine import unpack_records
    mp.spawn(_worker, args=(world, grid, str(tmp_path)), nprocs=world, join=True)
    cams, masks = fx.golden_cameras(), fx.golden_masks()
    frames = fx.synthetic_frames(4, *masks[0].shape)
    full = carve_c.carve(*grid, fx.oracle_cams(cams), masks, frames)
    first = np.load(tmp_path / "gathered_0.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("gathered_%d.npy" % r)), first)     # every rank holds the list
        assert int(np.load(tmp_path / ("counts_%d.npy" % r)).sum()) == full["count"]
    idx, rgb, seen = unpack_records(first)
    assert np.array_equal(idx, full["idx"]) and np.array_equal(rgb[:, ::-1], full["bgr"]) and seen.all()
    ent0 = np.load(tmp_path / "entries_0.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("entries_%d.npy" % r)), ent0)
    assert np.array_equal(_indices_from_entries(ent0), full["idx"])


def _shm_worker(rank, world, port, out_dir):
    os.environ["MASTER_PORT"] = str(port)
    from voxcarve import slabs
    tr = slabs.ShmTransport(world, rank)
    try:
        for rnd in range(3):
            local = (np.arange(5 + rank + rnd, dtype=np.uint64) + 1000 * rank + 1)
            counts, total = tr.allgather_records(local)
            assert counts.tolist() == [5 + r + rnd for r in range(world)] and total == int(counts.sum())
            got = tr.fetch()
            want = np.concatenate([np.arange(5 + r + rnd, dtype=np.uint64) + 1000 * r + 1 for r in range(world)])
            assert np.array_equal(got, want)
        ent = tr.allgather_entries(np.array([[3, 64 * (10 * rank + 1)], [1 << 63, 64 * (10 * rank + 2)]], np.uint64))
        assert ent.shape == (2 * world, 2) and ent[:, 1].tolist() == sorted(ent[:, 1].tolist())
        assert tr.max(float(rank)) == float(world - 1)
        tr.barrier()
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    finally:
        tr.close()


def test_shm_fallback_transport(tmp_path):
    """The /dev/shm exchange bench.py falls back to when no RCCL communicator can be made."""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_shm_worker, args=(world, _launch_key(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok_%d.npy" % r)).exists() for r in range(world))


def _rdzv_worker(rank, world, port, out_dir, fail_rank):
    os.environ["MASTER_PORT"] = str(port)
    from voxcarve import slabs
    uid = slabs.file_rendezvous(rank, b"U" * 128 if rank == 0 else None, timeout=30)
    flags = slabs.file_all_flags(rank, world, "comm", "boom on %d" % rank if rank == fail_rank else "", timeout=30)
    slabs.file_rendezvous_cleanup(rank, world)
    with open(os.path.join(out_dir, "r%d" % rank), "w") as f:
        f.write(repr((uid == b"U" * 128, flags)))


@pytest.mark.parametrize("fail_rank", [None, 2])
def test_rendezvous_decision_is_collective(tmp_path, fail_rank):
    """bench.py's "did the RCCL set-up work everywhere?" without a communicator: the unique id reaches every rank, every
    rank sees the same list of per-rank outcomes (so all of them fall back, or all exit), a file a dead process left under
    the same key is ignored, and the directory is gone afterwards."""
    import multiprocessing as mp
    import struct
    from voxcarve import slabs
    port = _launch_key()
    os.environ["MASTER_PORT"] = str(port)
    d = slabs._launch_dir(os.getpid())           # the workers are children of THIS process: that is their launch key
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "uid"), "wb") as f:                       # left behind by a launch that crashed: writer pid is dead
        f.write(struct.pack("<q", 2 ** 22 + 12345) + b"S" * 128)
    ctx = mp.get_context("fork")
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 3, port, str(tmp_path), fail_rank)) for r in (1, 2, 0)]   # rank 0 starts last
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = [eval(open(os.path.join(str(tmp_path), "r%d" % r)).read()) for r in range(3)]
    want = ["", "", ""] if fail_rank is None else ["", "", "boom on 2"]
    assert all(ok and flags == want for ok, flags in seen), seen
    assert not os.path.exists(d)

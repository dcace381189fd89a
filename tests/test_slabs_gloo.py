"""world_size-2/3 gloo runs of the z-slab split + variable-length all-gather on CPU.

The per-rank carve is stood in for by the oracle (this is a test of the partition and
exchange logic of voxcarve.slabs, which on GPUs runs over RCCL inside libvoxcarve)."""
import os
import socket

import numpy as np
import pytest

import fixtures_util as fx


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, grid, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import carve_c
        from voxcarve import slabs
        cams, masks = fx.golden_cameras(), fx.golden_masks()
        frames = fx.synthetic_frames(4, *masks[0].shape)
        i0, i1 = slabs.slab_index_range(grid, world, rank)
        res = carve_c.carve(*grid, fx.oracle_cams(cams), masks, frames, index_range=(i0, i1), threads=1)
        b = res["bgr"].astype(np.uint64)
        rec = res["idx"].astype(np.uint64) | (b[:, 2] << 32) | (b[:, 1] << 40) | (b[:, 0] << 48) | (1 << 56)
        tr = slabs.TorchTransport()
        counts, total = tr.allgather_records(rec)
        assert int(counts[rank]) == rec.size
        np.save(os.path.join(out_dir, "gathered_%d.npy" % rank), tr.fetch())
        np.save(os.path.join(out_dir, "counts_%d.npy" % rank), counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid", [(2, (32, 32, 32)), (3, (16, 64, 7))])
def test_slab_split_allgather_equals_single_rank(built, tmp_path, world, grid):
    import torch.multiprocessing as mp
    from oracle import carve_c
    from voxcarve.engine import unpack_records
    port = _free_port()
    mp.spawn(_worker, args=(world, port, grid, str(tmp_path)), nprocs=world, join=True)
    cams, masks = fx.golden_cameras(), fx.golden_masks()
    frames = fx.synthetic_frames(4, *masks[0].shape)
    full = carve_c.carve(*grid, fx.oracle_cams(cams), masks, frames)
    first = np.load(tmp_path / "gathered_0.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("gathered_%d.npy" % r)), first)     # every rank holds the list
        assert int(np.load(tmp_path / ("counts_%d.npy" % r)).sum()) == full["count"]
    idx, rgb, seen = unpack_records(first)
    assert np.array_equal(idx, full["idx"]) and np.array_equal(rgb[:, ::-1], full["bgr"]) and seen.all()


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
        assert tr.max(float(rank)) == float(world - 1)
        tr.barrier()
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    finally:
        tr.close()


def test_shm_fallback_transport(tmp_path):
    """The /dev/shm exchange bench.py falls back to when no RCCL communicator can be made."""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_shm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok_%d.npy" % r)).exists() for r in range(world))

#!/usr/bin/env python3
"""bench.py -- visual-hull carve throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (reference voxel_reconstruction.py:89-124 + assignment.py:116-133)
over one frame set whose byte masks + colour image are resident in HBM: per-frame preparation on
the device (bit-pack, foreground boxes, cropped block grids, camera order, BGRX image -- two kernels, no host
round trip) -> carve kernels -> scan -> ordered survivor records (+ RCCL all-gather when N > 1).  Cameras and
(LUT mode) the packed lookup table are built once, before the timed region.  Every timed step prepares its
frame set again (vc_touch_masks): nothing derived from the masks is carried over from an earlier step.

Workloads (config.name):
  real     BASELINE configs[2]/[3] (default): 1024^3 grid x the reference's 4 calibrated cameras, frame-0 MOG mask
           fixtures rolled by a few columns per frame set, synthetic colour frames.  N > 1: the grid is block-split
           along z (STRONG scaling, as BASELINE's ">= 6x at 8 GPUs" is stated).
  config5  BASELINE configs[4]: 512^3 x 16 synthetic ring cameras, 1080x1920 masks with 0.5 % salt noise, colour on.
  big2048  2048 x 2048 x 1023 (4.29 G voxels, the u32 index limit) x the 4 real cameras, 68.7 GB of lookup tables on
           one GPU (--mode fused: table-free): the largest case an index of 32 bits can address.

  python bench.py [--gpus N --steps K --warmup W] [--workload real|config5|big2048] [--grid 1024] [--mode lut|fused]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  No framework is imported: for N > 1 the RCCL unique id travels through a
node-local file and barriers / the max-over-ranks timing go through RCCL.  A rank that cannot join the RCCL
communicator makes EVERY rank exit non-zero (no silent change of transport) unless --allow-host-fallback is given.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6        # MI355X_MICROARCH.md: FP64 vector (an FMA counts as two; the projection is contraction-free, so its
                               # own ceiling is 39.3 T mul-or-add per second -- the fraction is quoted against the spec figure)
FLOPS_PER_PROJECTION = 52      # 52 f64 flop + 1 f64 divide per voxel-view
LUT_BYTES_PER_VV = 4           # one packed int32 per voxel-view
N_SLOTS = 8                    # resident frame sets (distinct byte masks); every timed step prepares its set again


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("real", "config5", "big2048"), default="real")
    ap.add_argument("--grid", type=int, default=1024, help="workload real: N of the N^3 grid")
    ap.add_argument("--mode", choices=("lut", "lut_stream", "fused"), default=None,
                    help="default: lut")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--prewarm-seconds", type=float, default=1.0,
                    help="untimed launches before the W warm-up steps so the clocks have ramped")
    ap.add_argument("--depth", type=int, choices=(1, 2, 3), default=3,
                    help="steps in flight: step i+1 (and i+2) are queued on the device before the host collects step i")
    ap.add_argument("--transport", choices=("rccl", "host"), default="rccl",
                    help="N>1 survivor exchange: rccl (device, default) or host (gloo; rehearsal on one GPU)")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="N>1: if the RCCL communicator cannot be created, continue on a /dev/shm host transport (and say so) "
                         "instead of exiting non-zero")
    ap.add_argument("--exchange", choices=("compact", "records"), default="compact",
                    help="N>1 over RCCL: exchange the non-zero occupancy words and expand on every rank (default), "
                         "or the 8-byte survivor records themselves")
    ap.add_argument("--split", choices=("balanced", "even"), default="balanced",
                    help="N>1: z-slab boundaries from the measured cost of every 16-layer chunk (default) or nz/N layers each")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal: take the N > 1 path (file rendezvous, RCCL communicator, all-gather per step) even with one rank")
    ap.add_argument("--resident-prep", action="store_true",
                    help="steady state of round 1: frame sets prepared once, outside the timed region (default: every timed "
                         "step prepares its frame set on the device)")
    ap.add_argument("--only-headline", action="store_true", help="skip the other modes and the side measurements")
    ap.add_argument("--e2e-steps", type=int, default=5, help="steps of the PCIe-inclusive leg (0: skip it)")
    return ap.parse_args()


class Group:
    """Rank bookkeeping.  N > 1: the RCCL unique id travels through a node-local file, and barriers /
    max-reductions go through RCCL itself (vc_comm_allreduce_max) once the communicator exists -- no
    framework import: torch wheels bundle their own ROCm runtime, and a process holding two of them
    breaks RCCL's HSA lookup (found on the GPU box).  `--transport host` (rehearsal on one GPU) is the
    only path that imports torch.distributed (gloo)."""

    def __init__(self, n_gpus, use_torch=False):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != n_gpus:
            if self.world == 1 and n_gpus > 1:
                raise SystemExit("--gpus %d needs one process per GPU: launch with "
                                 "python -m torch.distributed.run --nproc-per-node %d bench.py ..." % (n_gpus, n_gpus))
            raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (self.world, n_gpus))
        self.dist = None
        self.eng = None
        self.shm = None
        if self.world > 1 and use_torch:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def attach(self, eng):
        """From here on barriers and reductions run over the engine's RCCL communicator."""
        self.eng = eng

    def attach_fallback(self, transport):
        """RCCL unavailable and --allow-host-fallback: barriers and reductions through the /dev/shm exchange."""
        self.shm = transport

    def barrier(self):
        if self.dist:
            self.dist.barrier()
        elif self.shm is not None:
            self.shm.barrier()
        elif self.eng is not None:
            self.eng.comm_max(0.0)

    def max(self, x):
        if self.dist:
            import torch
            t = torch.tensor([x], dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        if self.shm is not None:
            return self.shm.max(x)
        if self.eng is not None:
            return self.eng.comm_max(x)
        return x

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
        if self.shm is not None:
            self.shm.close()


def make_workload(args):
    """(grid, cameras, masks, frames, colour camera, default mode, text)."""
    import fixtures_util as fx
    if args.workload == "config5":
        from voxcarve import synthetic
        H, W, C = 1080, 1920, 16
        cams = synthetic.ring_cameras(C, H, W)
        masks = synthetic.ellipsoid_masks(cams, H, W)          # 0.5 % salt noise, as SURVEY 8(d) specifies
        frames = synthetic.random_frames(C, H, W)
        return (512, 512, 512), cams, masks, frames, 1, "lut", \
            "BASELINE config 5: 512^3 x 16 synthetic ring cameras, ellipsoid silhouettes XOR 0.5 % salt noise, colour on"
    cams, masks = fx.golden_cameras(), fx.golden_masks()
    frames = fx.synthetic_frames(len(cams), *masks[0].shape)
    if args.workload == "big2048":
        return (2048, 2048, 1023), cams, masks, frames, 1, "lut", \
            "2048x2048x1023 grid (u32 index limit) x the 4 real cameras (lookup tables: 68.7 GB)"
    G = args.grid
    return (G, G, G), cams, masks, frames, 1, "lut", "%d^3 voxel grid x the reference's 4 calibrated cameras" % G


def run_mode(eng, grp, mode, steps, warmup, multi, host_transport=None, depth=3, exchange="compact", overlap=1, fresh=True,
             detail=False, cc=1):
    """W untimed + K timed steps of one mode; returns (seconds, survivors of this rank, total, timing dict).
    overlap: the scan + record expansion of a step on a second stream, beside the next step's carve (the product's
    default).  lut_stream exists to measure ONE kernel against the HBM roof, so it runs on one stream with its events on.
    detail: also record the events around preparation and carve kernels (they cost the stream ~10 us each)."""
    stream = mode == "lut_stream"
    table_free = mode == "fused_table_free"                  # not even the colour camera's table: every survivor is projected again
    eng.set_option("fused_color_table", 0 if table_free else 1)
    mode = "fused" if table_free else mode
    eng.set_option("lut_hier", 0 if stream else 1)
    eng.set_option("overlap", 0 if stream else overlap)
    eng.set_option("timing_detail", 1 if (detail or stream) else 0)
    mode = "lut" if stream else mode

    def finish():
        n = eng.carve_end()
        if multi and host_transport is not None and exchange == "compact":
            # host rehearsal / fallback: the compact form through host memory, expanded on the device
            return n, eng.expand_entries(host_transport.allgather_entries(eng.pack_entries()))
        if multi and host_transport is not None:
            _, total = host_transport.allgather_records(eng.fetch_records(pinned=True))
            return n, total
        if multi:
            _, total = eng.allgather()
            return n, total
        return n, n

    # a rank of a communicator never reads its own slab's list: the records come out of the all-gather
    keep = not (multi and exchange == "compact")
    eng.set_option("gather_compact", 1 if exchange == "compact" else 0)
    # the all-gather of step i is only queued (its count is known from the ranks' counts); the next step is
    # enqueued behind it without a host round trip; synchronize() closes the timed region
    eng.set_option("gather_sync", 0 if (not keep and depth > 1 and host_transport is None) else 1)

    def begin(i):
        # fresh: the step takes its frame set's byte masks + colour image (resident in HBM) as NEW input:
        # bit-pack, foreground boxes, cropped block grids, camera order and BGRX expansion run on the device, queued
        # in front of the carve, inside the timed region -- no host round trip
        slot = i % N_SLOTS
        if fresh:
            eng.touch_masks(slot)
        eng.carve_begin(slot=slot, mode=mode, records=keep, color_cam=cc)

    def run(first, count):
        """`count` steps, `depth` of them in flight: step i+1 (and i+2) are enqueued before step i is collected, so the
        device never idles between steps."""
        last = (0, 0)
        if depth <= 1:
            for i in range(count):
                begin(first + i)
                last = finish()
            return last
        pending = 0
        for i in range(count):
            begin(first + i)
            pending += 1
            if pending == depth:
                last = finish()
                pending -= 1
        while pending:
            last = finish()
            pending -= 1
        return last

    if warmup:
        run(0, warmup)
    eng.synchronize()
    grp.barrier()
    eng.timing(reset=True)
    t0 = time.perf_counter()
    last = run(warmup, steps)
    eng.synchronize()
    grp.barrier()
    dt = grp.max(time.perf_counter() - t0)
    tm = eng.timing()
    eng.set_option("timing_detail", 0)
    return dt, last[0], last[1], tm


def cpu_baseline(grid, cams, masks, frames, seconds, cc, device_digest=None):
    """The C/OpenMP oracle ("port") on this host's cores: the WHOLE grid of the same workload when that
    fits the time budget (a few seconds on a many-core host), else a centred z-slab sized to it.  When it is the
    whole grid, its record list is also the checker of the device's (sha256 of the packed records)."""
    import fixtures_util as fx
    from oracle import carve_c
    oc = fx.oracle_cams(cams)
    nx, ny, nz = grid
    threads = len(os.sched_getaffinity(0))
    layer = nx * ny
    carve_c.carve(nx, ny, nz, oc, masks, frames, index_range=(0, layer * 2), threads=threads, cap=1 << 22, color_cam=cc)   # warm the pool
    probe = max(1, min(nz, 16))
    z_mid = nz // 2
    t0 = time.perf_counter()
    carve_c.carve(nx, ny, nz, oc, masks, frames, index_range=(z_mid * layer, (z_mid + probe) * layer),
                  threads=threads, cap=1 << 22, color_cam=cc)
    per_layer = (time.perf_counter() - t0) / probe
    layers = int(max(1, min(nz, seconds / max(per_layer, 1e-9))))
    if per_layer * nz <= 3.0 * seconds:      # the whole grid is within reach: then the baseline also CHECKS the device's records
        layers = nz
    z0 = max(0, (nz - layers) // 2)
    t0 = time.perf_counter()
    res = carve_c.carve(nx, ny, nz, oc, masks, frames, index_range=(z0 * layer, (z0 + layers) * layer),
                        threads=threads, cap=1 << 27, color_cam=cc)
    dt = time.perf_counter() - t0
    vv = layers * layer * len(cams)
    whole = layers == nz
    what = "the whole %dx%dx%d grid" % grid if whole else "z-layers [%d,%d) of the %dx%dx%d grid" % ((z0, z0 + layers) + grid)
    out = {"value": round(vv / dt / 1e6, 2), "unit": "Mvoxel-views/s", "cores": threads, "kind": "port",
           "sample": "oracle/carve_ref.c (C/OpenMP, -O2 -ffp-contract=off), %s, %.3g voxel-views in %.2f s, "
                     "%d survivors" % (what, vv, dt, res["count"])}
    if whole and device_digest is not None:
        bgr = res["bgr"].astype(np.uint64)
        rec = res["idx"].astype(np.uint64) | (bgr[:, 2] << np.uint64(32)) | (bgr[:, 1] << np.uint64(40)) \
            | (bgr[:, 0] << np.uint64(48)) | (np.uint64(1) << np.uint64(56))
        out["records_sha256"] = hashlib.sha256(rec.tobytes()).hexdigest()
        out["device_records_match"] = out["records_sha256"] == device_digest
    return out


def main():
    args = parse()
    import voxcarve
    voxcarve._lib.load()                 # libvoxcarve (system HIP runtime) is loaded before any torch import
    grp = Group(args.gpus, use_torch=(args.transport == "host"))
    from voxcarve import slabs

    grid, cams, masks, frames, cc, default_mode, workload_text = make_workload(args)
    if args.mode is None:
        args.mode = default_mode
    H, W = masks[0].shape
    C = len(cams)
    nx, ny, nz = grid

    eng = voxcarve.CarveEngine(0 if args.single_device else grp.local_rank)
    eng.set_grid(*grid)
    z0, z1 = slabs.slab_range(nz, grp.world, grp.rank)
    eng.set_slab(z0, z1)
    eng.set_cameras(cams, H, W)
    for s in range(N_SLOTS):
        eng.upload_masks([np.roll(m, 3 * s, axis=1) for m in masks], slot=s)
        eng.upload_frame(cc, np.roll(frames[cc], 3 * s, axis=1), slot=s)
    multi = grp.world > 1 or args.force_comm
    host_transport = None
    rccl_ranks = 0
    transport_note = ("rccl, %s" % ("non-zero occupancy words, expanded on every rank" if args.exchange == "compact"
                                    else "8-byte survivor records")) if multi else "none (one rank)"
    if multi and args.transport == "host":
        from torch_transport import TorchTransport      # tests/: the product package imports no framework
        host_transport = TorchTransport()
        transport_note = "host (gloo) rehearsal"
    elif multi:
        # the decision "RCCL or not" is COLLECTIVE: every rank reports how its communicator set-up went through the
        # rendezvous directory and all act on the same list -- a rank that failed alone cannot wander off to another
        # transport while the others sit in a collective
        err = ""
        # RCCL prints a version banner on STDOUT when it starts (NCCL_DEBUG=VERSION in this image): rank 0's stdout must
        # carry the one JSON line and nothing else, so file descriptor 1 points at stderr while the communicator comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = slabs.file_rendezvous(grp.rank, voxcarve.CarveEngine.comm_unique_id() if grp.rank == 0 else None)
            eng.comm_init(grp.world, grp.rank, uid)
        except Exception as exc:
            err = str(exc)[:300] or type(exc).__name__
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        flags = slabs.file_all_flags(grp.rank, grp.world, "comm", err)
        if all(f == "" for f in flags):
            grp.attach(eng)
            grp.barrier()
            rccl_ranks = grp.world
        else:
            why = "; ".join("rank %d: %s" % (r, f) for r, f in enumerate(flags) if f)
            if not args.allow_host_fallback:
                sys.stderr.write("[bench rank %d] RCCL communicator unavailable (%s); no --allow-host-fallback: exiting\n" % (grp.rank, why))
                slabs.file_rendezvous_cleanup(grp.rank, grp.world)
                eng.close()
                raise SystemExit(3)
            transport_note = "shm-fallback (--allow-host-fallback): RCCL communicator unavailable (%s)" % why[:200]
            sys.stderr.write("[bench rank %d] %s\n" % (grp.rank, transport_note))
            if not err:
                eng.comm_destroy()
            host_transport = slabs.ShmTransport(grp.world, grp.rank)
            grp.attach_fallback(host_transport)
        slabs.file_rendezvous_cleanup(grp.rank, grp.world)
    split_note = "even"
    if grp.world > 1 and args.split == "balanced":
        # the hull is not spread evenly over z: give every rank the same share of measured kernel time
        chunk = 16
        weights = slabs.measure_chunk_cost(eng, nz, chunk, mode="fused" if args.mode == "fused" else "lut", reduce_max=grp.max,
                                           color_cam=cc)
        bounds = slabs.balanced_bounds(weights, chunk, nz, grp.world)
        z0, z1 = bounds[grp.rank], bounds[grp.rank + 1]
        split_note = "balanced by measured chunk cost: z bounds %s" % bounds
    eng.set_slab(z0, z1)
    have_lut = not (args.mode == "fused" and (args.only_headline or args.workload == "big2048"))
    if have_lut:
        eng.build_lut()
    eng.synchronize()
    lut_ms = eng.timing()["lut_ms"] if have_lut else 0.0
    h2d_ms = eng.timing()["h2d_ms"]           # the last frame set's byte masks over PCIe (asynchronous, on the upload stream)

    t_end = time.perf_counter() + args.prewarm_seconds
    while time.perf_counter() < t_end:
        eng.carve(slot=0, mode="fused", color_cam=cc)
    short = max(10, args.steps // 5)
    common = dict(multi=multi, host_transport=host_transport, depth=args.depth, exchange=args.exchange, cc=cc)
    results = {}
    order = [args.mode]
    if not args.only_headline and have_lut:
        order += [m for m in ("lut", "lut_stream", "fused", "fused_table_free") if m != args.mode]
    for mode in order:
        dt, n_local, n_total, tm = run_mode(eng, grp, mode, args.steps, args.warmup, fresh=not args.resident_prep, **common)
        results[mode] = {"seconds": dt, "survivors": int(n_total), "survivors_this_rank": int(n_local), "tm": tm}
    head = results[args.mode]

    # side measurements of the headline mode (short runs, never `value`)
    if not multi and not args.only_headline:
        plain = run_mode(eng, grp, args.mode, short, 2, overlap=0, fresh=not args.resident_prep, **common)
        alone = run_mode(eng, grp, args.mode, short, 2, overlap=0, fresh=not args.resident_prep, detail=True, **common)
        head["one_stream"] = {"ms_per_step": plain[0] / short * 1e3, "tm": alone[3], "tm_plain": plain[3]}
        det = run_mode(eng, grp, args.mode, short, 2, fresh=not args.resident_prep, detail=True, **common)
        head["detail"] = {"ms_per_step": det[0] / short * 1e3, "tm": det[3]}
        if not args.resident_prep:
            res = run_mode(eng, grp, args.mode, short, 2, fresh=False, **common)
            head["ms_per_step_resident_prep"] = res[0] / short * 1e3
        if "fused" in results and args.mode != "fused":
            fdet = run_mode(eng, grp, "fused", short, 2, fresh=not args.resident_prep, detail=True, **common)
            results["fused"]["detail"] = {"ms_per_step": fdet[0] / short * 1e3, "tm": fdet[3], "steps": short}
        head["detail"]["steps"] = short
    n_local_vox = eng.n_voxels
    total_vv = float(nx) * ny * nz * C
    ms_per_step = head["seconds"] / args.steps * 1e3
    value = total_vv * args.steps / head["seconds"] / 1e6

    # ---- the records of frame set 0, as a digest: all ranks must hold the same list, and rank 0 at N = 1 checks it against
    # the CPU oracle's (cpu_baseline) and against the committed digest of an oracle-checked run
    eng.set_option("overlap", 1)
    eng.set_option("lut_hier", 1)
    eng.set_option("gather_sync", 1)
    eng.set_option("fused_color_table", 1)
    counts_note = None
    keep = not (multi and args.exchange == "compact")
    eng.carve_begin(slot=0, mode="lut" if args.mode == "lut_stream" else args.mode, records=keep, color_cam=cc)
    n_mine = eng.carve_end()
    if multi and host_transport is None:
        counts, n_all = eng.allgather()
        rec = eng.fetch_gathered()
        counts_note = [int(c) for c in counts]
    elif multi and args.exchange == "compact":
        n_all = eng.expand_entries(host_transport.allgather_entries(eng.pack_entries()))
        rec = eng.fetch_gathered()
    elif multi:
        _, n_all = host_transport.allgather_records(eng.fetch_records(pinned=True))
        rec = host_transport.fetch()
    else:
        n_all, rec = n_mine, eng.fetch_records(pinned=True)
    digest = hashlib.sha256(np.ascontiguousarray(rec).tobytes()).hexdigest()
    d48 = float(int(digest[:12], 16))                       # 48 bits: exact in a double
    same = (grp.max(d48) == d48) and (-grp.max(-d48) == d48)
    ranks_agree = grp.max(0.0 if same else 1.0) == 0.0
    golden_file = os.path.join(ROOT, "tests", "golden", "bench_digests.json")
    golden = json.load(open(golden_file)) if os.path.exists(golden_file) else {}
    gkey = "%s_%dx%dx%d" % (args.workload, nx, ny, nz)
    golden_match = (golden[gkey]["records_sha256"] == digest and golden[gkey]["survivors"] == int(n_all)) if gkey in golden else None

    # ---- roofline of the DOMINANT kernel of the timed region = the kernel with the largest summed time.  The record expansion is
    # timed in the timed region itself (its begin / end events ride on its own launch, which costs the stream nothing); the other
    # kernels' times come from the short side run in which every launch carries such events (2-3 us per launch: never the run
    # `value` comes from) and the kernels count the work they actually do (vc_timing_t::work).
    tm = head["tm"]
    emit_ms = tm["emit_ms_sum"] / tm["emit_launches"] if tm["emit_launches"] else None

    def kernel_objects(mode, dtm, dsteps, survivors_rank, live_emit_ms, color_table=True):
        """Per kernel kind of one mode: {avg_launch_ms, bound, algorithmic bytes or flops per launch, achieved, frac}."""
        out = {}
        work = {k: v / max(1, dsteps) for k, v in dtm.get("work", {}).items()}
        for kind, kv in dtm.get("kernels", {}).items():
            ms = kv["ms_sum"] / kv["launches"]
            o = {"avg_launch_ms": round(ms, 4), "launches_timed": kv["launches"],
                 "timed_by": "begin / end events on every launch of a %d-step side run (three steps in flight, as the timed region)" % dsteps}
            fused = mode == "fused"
            if kind == "k_emit":
                if live_emit_ms:
                    ms = live_emit_ms
                    o.update({"avg_launch_ms": round(ms, 4), "launches_timed": tm["emit_launches"],
                              "timed_by": "begin / end events riding on the launch itself, every step of the timed region"})
                if fused and not color_table:
                    o.update({"bound": "valu_f64", "kernel": "k_emit_busy<re-projecting> (record expansion, one float64 projection per survivor)",
                              "alg": FLOPS_PER_PROJECTION * float(survivors_rank), "per_unit": "52 flop per survivor"})
                else:
                    o.update({"bound": "hbm", "kernel": "k_emit_busy<FROM_LUT> (record expansion: occupancy words -> ordered {idx, rgb, seen} records)",
                              "alg": 12.0 * survivors_rank, "per_unit": "12 B per survivor (8 B record written + 4 B table entry read)",
                              "units_per_launch": survivors_rank})
            elif kind == "k_voxel_words" and fused:
                o.update({"bound": "valu_f64", "kernel": "k_voxel_words<table-free> (exact float64 projection + mask bit of the undecided words' voxels)",
                          "alg": FLOPS_PER_PROJECTION * work.get("projections", 0.0), "per_unit": "52 flop per projection executed (lanes that asked)",
                          "units_per_launch": work.get("projections", 0.0)})
            elif kind == "k_voxel_words":
                o.update({"bound": "hbm", "kernel": "k_voxel_words<LUT> (table entry + mask bit of the undecided words' voxels)",
                          "alg": 4.0 * work.get("table_entries", 0.0), "per_unit": "4 B per table entry read (lanes that asked)",
                          "units_per_launch": work.get("table_entries", 0.0)})
            elif kind == "k_brick_words":
                o.update({"bound": "hbm", "kernel": "k_brick_words (word boxes of the listed bricks against the block grids in LDS)",
                          "alg": 8.0 * work.get("word_boxes", 0.0), "per_unit": "8 B per word box read (cameras asked)",
                          "units_per_launch": work.get("word_boxes", 0.0)})
            elif kind == "k_cull_bricks":
                o.update({"bound": "hbm", "kernel": "k_cull_bricks (brick boxes against the block grids in LDS)",
                          "alg": 8.0 * work.get("brick_boxes", 0.0), "per_unit": "8 B per brick box read (cameras asked)",
                          "units_per_launch": work.get("brick_boxes", 0.0)})
            else:
                o.update({"bound": "latency", "kernel": kind, "alg": None})
            if o.get("alg"):
                if o["bound"] == "hbm":
                    ach = o["alg"] / (ms * 1e-3) / 1e9
                    o.update({"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                              "algorithmic_bytes_per_launch": o["alg"]})
                else:
                    ach = o["alg"] / (ms * 1e-3) / 1e12
                    o.update({"achieved": round(ach, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP64_PEAK_TFLOPS, 4),
                              "algorithmic_flops_per_launch": o["alg"]})
            o.pop("alg", None)
            o["traffic"] = None
            out[kind] = o
        return out

    per_survivor = 12
    roof, kernel_table = None, None
    ctab = True                                              # (VC_MODE_FUSED colours from the colour camera's table by default)
    if "detail" in head:
        objs = kernel_objects(args.mode, head["detail"]["tm"], head["detail"]["steps"], head["survivors_this_rank"], emit_ms, ctab)
        kernel_table = {k: v["avg_launch_ms"] for k, v in objs.items()}
        ranked = sorted((k for k in objs if objs[k].get("frac") is not None), key=lambda k: -objs[k]["avg_launch_ms"])
        if ranked:
            roof = objs[ranked[0]]
            roof["dominant_of"] = kernel_table
    if roof is None and emit_ms:
        alg = per_survivor * float(head["survivors_this_rank"])
        ach = alg / (emit_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_emit_busy<FROM_LUT> (record expansion: occupancy words -> ordered {idx, rgb, seen} records)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes_per_launch": alg, "per_unit": "12 B per survivor (8 B record written + 4 B table entry read)",
                "units_per_launch": head["survivors_this_rank"], "avg_launch_ms": round(emit_ms, 4),
                "timed_by": "begin / end events riding on the launch itself, every step of the timed region"}
    elif roof is None and multi and tm["gathers"]:
        # a rank of a communicator expands the gathered words of ALL ranks (k_emit_lanes<INDIRECT>) inside vc_allgather
        g_ms = tm["gather_ms_sum"] / tm["gathers"]
        x_ms = tm["exchange_ms"]
        alg = per_survivor * float(head["survivors"])
        ach = alg / (max(g_ms - x_ms, 1e-6) * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_emit_lanes<INDIRECT> (expansion of all ranks' occupancy words on every rank)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(g_ms - x_ms, 4)}
    # Mode F: the table-free kernel against the FP64 vector peak, from the fused leg's side run
    roof_fused = None
    fd = head.get("detail") if args.mode == "fused" else results.get("fused", {}).get("detail")
    if fd:
        fobjs = kernel_objects("fused", fd["tm"], fd["steps"], head["survivors_this_rank"], None, ctab)
        cand = [o for o in fobjs.values() if o.get("bound") == "valu_f64"]
        if cand:
            roof_fused = max(cand, key=lambda o: o["avg_launch_ms"])
            roof_fused["step_kernel_times_ms"] = {k: v["avg_launch_ms"] for k, v in fobjs.items()}
    # the contract's own figure for the whole step (4 B per voxel-view + 8 B per survivor + the masks): how much
    # of the table the hierarchy never touches -- a skip factor, not a bandwidth
    contract_bytes = LUT_BYTES_PER_VV * total_vv + 8.0 * head["survivors"] + C * H * W
    skip = {"contract_bytes_per_step": contract_bytes, "rate_gbs": round(contract_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "skip_factor_vs_hbm_peak": round(contract_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
            "note": "SURVEY 8(d) bytes of the streaming formulation / step time / 8 TB/s; > 1 because culling decides most of the "
                    "grid from brick and word pixel boxes without reading its table entries"}

    def stream_roofline(r):
        # k_lut_first streams ONE camera's packed table over every voxel of the slab:
        # units per launch = n voxel-views, 4 B each + its mask bits + the alive words.
        first_ms = r["tm"]["first_ms_sum"] / max(1, r["tm"]["carve_launches"])
        alg = LUT_BYTES_PER_VV * float(n_local_vox) + H * W / 8.0 + n_local_vox / 8.0
        ach = alg / (first_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "k_lut_first (first-camera table stream, LDS-resident mask)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(first_ms, 4),
                "frac_of_measured_copy_ceiling_6290": round(ach / 6290.0, 4)}

    # (masks above 64 KB of bits per camera do not fit k_lut_first's LDS: that configuration takes the generic kernel)
    roof_stream = stream_roofline(results["lut_stream"]) if "lut_stream" in results and (H * W + 31) // 32 * 4 <= 65536 else None
    if args.mode == "lut_stream" and roof_stream:
        roof = roof_stream
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file) and roof is not None:
        tj = json.load(open(traffic_file))
        kname = roof["kernel"].split("<")[0].split(" ")[0]
        t = tj.get("%s_%s_%s_g%d" % (kname, args.mode, gkey, grp.world)) or \
            (tj.get("emit_%s_%s_g%d" % (args.mode, gkey, grp.world)) if kname == "k_emit_busy" else None)
        if t:
            t = dict(t)
            t["from"] = "profiles/ (rocprofv3 --pmc passes of an earlier session of this round), NOT measured in this run"
            roof["traffic"] = t
            gbs = t["hbm_bytes"] / (roof["avg_launch_ms"] * 1e-3) / 1e9
            roof["traffic_rate_gbs"] = round(gbs, 1)
            roof["traffic_frac_of_peak"] = round(gbs / HBM_PEAK_GBS, 4)
        if roof_stream and tj.get("lut_stream_%s_g%d" % (gkey, grp.world)):
            roof_stream["traffic"] = tj["lut_stream_%s_g%d" % (gkey, grp.world)]

    others = {}
    for m in results:
        if m != args.mode:
            o = results[m]
            others[m] = {"value": round(total_vv * args.steps / o["seconds"] / 1e6, 1),
                         "ms_per_step": round(o["seconds"] / args.steps * 1e3, 4), "survivors": o["survivors"]}
    phases = {"record_expansion": round(emit_ms, 4) if emit_ms else None,
              "gather": round(tm["gather_ms_sum"] / max(1, tm["gathers"]), 4), "gather_exchange_part": round(tm["exchange_ms"], 4),
              "lut_build_once": round(lut_ms, 3), "steps_that_prepared": int(tm["preps"]),
              "mask_upload_h2d_outside_timed_region": round(h2d_ms, 4)}
    if "detail" in head:
        d, o = head["detail"]["tm"], head["one_stream"]["tm"]
        phases.update({
            "frame_set_prep_on_device": round(d["prep_ms_sum"] / max(1, d["preps_timed"]), 4),
            "carve_kernels": round(d["carve_ms_sum"] / max(1, d["carve_launches"]), 4),
            "carve_kernels_one_stream": round(o["carve_ms_sum"] / max(1, o["carve_launches"]), 4),
            "record_expansion_one_stream": round(head["one_stream"]["tm_plain"]["emit_ms_sum"] / max(1, head["one_stream"]["tm_plain"]["emit_launches"]), 4),
            "ms_per_step_one_stream": round(head["one_stream"]["ms_per_step"], 4),
            "ms_per_step_with_kernel_events": round(head["detail"]["ms_per_step"], 4),
            "note": "prep / carve / per-kernel figures come from short extra runs with timing_detail on (events around the phases, "
                    "begin / end events on every launch, work counters): that costs a step 20-50 us, so the run `value` comes "
                    "from carries only the expansion's own two events"})
    if head.get("ms_per_step_resident_prep") is not None:
        phases["ms_per_step_with_frame_sets_prepared_once"] = round(head["ms_per_step_resident_prep"], 4)
    # the same kernel without the other stream's kernels beside it (short one-stream side run): how fast it is by itself
    if roof is not None and "one_stream" in head and "dominant_of" in roof:
        o = head["one_stream"]["tm"]
        kind = max(roof["dominant_of"], key=lambda k: roof["dominant_of"][k] if kernel_table and objs[k].get("frac") is not None else -1.0)
        kv = o.get("kernels", {}).get(kind)
        if kind == "k_emit":                                   # (its events ride on its launch in every run: the plain one-stream run has it undisturbed)
            op = head["one_stream"]["tm_plain"]
            kv = {"ms_sum": op["emit_ms_sum"], "launches": op["emit_launches"]}
        if kv and kv["launches"]:
            alone_ms = kv["ms_sum"] / kv["launches"]
            roof["avg_launch_ms_alone"] = round(alone_ms, 4)
            per_launch = roof.get("algorithmic_bytes_per_launch") or roof.get("algorithmic_flops_per_launch")
            scale = 1e9 * HBM_PEAK_GBS if roof["bound"] == "hbm" else 1e12 * FP64_PEAK_TFLOPS
            roof["frac_alone"] = round(per_launch / (alone_ms * 1e-3) / scale, 4)
        phases["kernels_one_stream"] = {k: round(v["ms_sum"] / v["launches"], 4) for k, v in o.get("kernels", {}).items()}
        phases["kernels_pipelined"] = kernel_table
    out = {
        "metric": "Mvoxel-views/s (grid N^3 x 4 cams)", "value": round(value, 1), "unit": "Mvoxel-views/s",
        "n_gpus": grp.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if args.mode == "fused" else "i32",
        "data": ("reference calibration (4x config.xml) + frame-0 MOG mask fixtures rolled per frame set; synthetic colour frames"
                 if args.workload != "config5" else
                 "synthetic ring cameras, ellipsoid silhouettes XOR 0.5 % salt noise rolled per frame set, random colour frames"),
        "config": {"workload": "%s (%dx%d masks), z-slab split over %d GPU(s), mode=%s, per-frame preparation on the device + "
                               "ordered survivor list + colour%s" % (workload_text, W, H, grp.world, args.mode,
                                                                     (" + RCCL all-gather" if host_transport is None else " + host-side gather") if multi else ""),
                   "name": args.workload, "grid": [nx, ny, nz], "cameras": C, "mode": args.mode, "survivors": head["survivors"],
                   "steps_in_flight": args.depth, "exchange": transport_note, "rccl_ranks": rccl_ranks, "split": split_note,
                   "survivors_per_rank": counts_note, "records_sha256_frame_set_0": digest, "survivors_frame_set_0": int(n_all),
                   "ranks_agree_on_records": bool(ranks_agree), "matches_committed_digest": golden_match},
        "scaling_model": None if not multi else {
            "note": "DESIGN.md section 5, an ESTIMATE made on one GPU (no N > 1 RCCL run exists from the builder's pool): every rank "
                    "expands ALL survivors itself, so the record expansion does not divide; expected speed-up at 8 GPUs ~1x for "
                    "workload real, ~2x for config5, ~1.2x for big2048 (BASELINE's >= 6x assumed a 1e9 voxel-views/s kernel)",
            "expected_speedup_at_8": {"real": 1.0, "config5": 2.0, "big2048": 1.2}.get(args.workload)},
        "roofline": roof,
        "roofline_fused": roof_fused,
        "contract_skip": skip,
        "roofline_stream": roof_stream,
        "other_modes": others,
        "phases_ms": phases,
    }
    if args.e2e_steps > 0 and grp.world == 1 and not args.force_comm:
        # PCIe-inclusive rate (never `value`): "writes a packed surviving-voxel list (+ sampled colour) back to host".
        # Per step: byte masks + colour frame up (page-locked staging, upload stream, beside the previous carve),
        # preparation + carve on the device, the records down into a page-locked buffer.
        rolled = [[np.roll(m, 3 * s, axis=1) for m in masks] for s in range(2)]
        K = args.e2e_steps
        e2e_mode = "lut" if args.mode == "lut_stream" else args.mode
        eng.carve(slot=0, mode=e2e_mode, color_cam=cc)
        eng.fetch_records(pinned=True)          # allocate the page-locked read-back buffer once
        eng.synchronize()
        t0 = time.perf_counter()
        eng.upload_masks(rolled[0], slot=0)
        eng.upload_frame(cc, frames[cc], slot=0)
        eng.carve_begin(slot=0, mode=e2e_mode, color_cam=cc)
        for i in range(1, K + 1):
            if i < K:                            # the next frame set goes up and is queued while this one is collected
                eng.upload_masks(rolled[i % 2], slot=i % 2)
                eng.upload_frame(cc, frames[cc], slot=i % 2)
                eng.carve_begin(slot=i % 2, mode=e2e_mode, color_cam=cc)
            eng.carve_end()
            rec = eng.fetch_records(pinned=True)
        dt = (time.perf_counter() - t0) / K
        out["pcie_inclusive"] = {"value": round(total_vv / dt / 1e6, 1), "unit": "Mvoxel-views/s", "steps": K,
                                 "ms_per_step": round(dt * 1e3, 3), "bytes_down_per_step": int(rec.nbytes),
                                 "bytes_up_per_step": int(C * H * W + H * W * 3),
                                 "note": "host byte masks + colour frame in, packed survivor records (8 B each) out, per step; "
                                         "the read-back of %.0f MB is the PCIe-bound part" % (rec.nbytes / 1e6)}
    rc = 0
    if grp.rank == 0 and grp.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, cams, masks, frames, args.cpu_seconds, cc, device_digest=digest)
        if out["cpu_baseline"].get("device_records_match") is False:
            sys.stderr.write("[bench] device records differ from the CPU oracle's\n")
            rc = 4
    elif grp.rank == 0:
        out["cpu_baseline"] = None
    if not ranks_agree or golden_match is False:
        sys.stderr.write("[bench rank %d] record digest check failed: ranks agree %s, committed digest %s\n" % (grp.rank, ranks_agree, golden_match))
        rc = 4
    eng.close()
    grp.close()
    if grp.rank == 0:
        print(json.dumps(out))
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()

"""bench.py itself on the GPU box, as the driver launches it (child processes; N > 1 through torch.distributed.run).

This file sorts AFTER test_gpu_parity.py on purpose: every comparison with the oracle runs before anything here starts a
launcher, so a launcher hiccup under `pytest -x` cannot hide a parity test (GPUTEST_r02: a port probe raced torchrun's
listen and the property test was never reached)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _run_bench(args, nproc=1, timeout=300, expect_rc=None):
    """bench.py in a child process (as the driver launches it); returns the parsed JSON line of rank 0.
    N > 1: the launcher picks its own port (--standalone: the agent's store listens on port 0 and exports MASTER_PORT to
    the workers; bench.py's file rendezvous keys on whatever arrives) -- nothing is probed here and handed on.  A launcher
    that still fails to listen is started once more, as a fresh child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(nproc)]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(nproc)] + args
    for attempt in range(2):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root)
        if nproc > 1 and attempt == 0 and out.returncode != 0 and ("EADDRINUSE" in out.stderr or "address already in use" in out.stderr):
            continue
        break
    if expect_rc is not None:
        assert out.returncode != 0, "bench.py was expected to fail"
        return out
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_single_gpu(built):
    """The JSON line the driver reads: keys, types, and the numbers that must hang together."""
    d = _run_bench(["--grid", "256", "--steps", "7", "--warmup", "2", "--cpu-seconds", "1"])   # last step lands on frame set 0
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Mvoxel-views/s" and "workload" in d["config"] and "model" not in d["config"]
    vv = 256 ** 3 * 4
    assert abs(d["value"] - vv / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu_f64") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["config"]["survivors"] == 461113                      # the 256^3 golden count: the bench ran the real path
    assert r["frac"] <= 1.0 and r["avg_launch_ms"] > 0 and r["kernel"].startswith("k_")
    # the roofline names the kernel with the largest time of the step, whichever it is at this size, out of the per-kernel table
    assert max(r["dominant_of"].values()) >= r["avg_launch_ms"] * 0.5 and "k_emit" in r["dominant_of"]
    # Mode F: the table-free projection kernel against the FP64 vector peak
    f = d["roofline_fused"]
    assert f["bound"] == "valu_f64" and f["unit"] == "TFLOP/s" and 0 < f["frac"] <= 1.0 and f["units_per_launch"] > 0
    assert abs(f["algorithmic_flops_per_launch"] - 52 * f["units_per_launch"]) < 1 and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-3
    assert set(d["other_modes"]) >= {"fused", "fused_table_free", "lut_stream"}
    assert d["contract_skip"]["skip_factor_vs_hbm_peak"] > 0
    # the device's record list of frame set 0 is the CPU oracle's, byte for byte (the baseline leg carved the whole grid)
    assert c["device_records_match"] is True and c["records_sha256"] == d["config"]["records_sha256_frame_set_0"]
    assert d["config"]["survivors_frame_set_0"] == 461113 and d["config"]["ranks_agree_on_records"] is True
    # every timed step prepared its frame set on the device, inside the timed region; the PCIe-inclusive figure is there
    ph = d["phases_ms"]
    # (the prep / carve figures come from a side run with extra events and from kernels that run beside other streams' kernels:
    # at 256^3 they are of the order of the step itself, so only their order of magnitude is checked)
    assert ph["steps_that_prepared"] == 7 and 0 < ph["frame_set_prep_on_device"] < 3 * d["ms_per_step"]
    assert 0 < ph["carve_kernels"] < 3 * d["ms_per_step"] and ph["record_expansion"] > 0
    assert d["pcie_inclusive"]["value"] > 0 and d["pcie_inclusive"]["value"] < d["value"]


def test_bench_two_ranks_host_transport(built):
    """The N > 1 flow of bench.py (work-balanced slab bounds, records-free steps, compact word exchange, expansion of
    all ranks' words on the device) with two processes sharing this GPU; the exchange itself goes through gloo
    because RCCL refuses two ranks on one device.  The gathered list must be the single-rank one."""
    d = _run_bench(["--grid", "256", "--steps", "8", "--warmup", "1", "--no-cpu-baseline", "--single-device",
                    "--transport", "host"], nproc=2, timeout=600)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["cpu_baseline"] is None
    assert "balanced" in d["config"]["split"] and "host" in d["config"]["exchange"]
    # frame sets are rolled per step: the last timed step (index warmup + steps - 1 = 8 -> slot 0) is the unrolled set
    assert d["config"]["survivors"] == 461113
    for m in d["other_modes"].values():
        assert m["survivors"] == 461113
    assert d["config"]["ranks_agree_on_records"] is True and d["config"]["survivors_frame_set_0"] == 461113
    assert d["config"]["rccl_ranks"] == 0


def test_bench_rccl_failure_is_collective_and_loud(built):
    """Two ranks on ONE device: RCCL refuses the duplicate device.  Without --allow-host-fallback every rank exits non-zero
    (no JSON line, no silently different transport); with it the run completes on the /dev/shm transport and says so."""
    out = _run_bench(["--grid", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--single-device", "--only-headline"],
                     nproc=2, timeout=600, expect_rc=True)
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "RCCL communicator unavailable" in out.stderr
    d = _run_bench(["--grid", "256", "--steps", "8", "--warmup", "1", "--no-cpu-baseline", "--single-device", "--only-headline",
                    "--allow-host-fallback"], nproc=2, timeout=600)
    assert "shm-fallback" in d["config"]["exchange"] and d["config"]["rccl_ranks"] == 0
    assert d["config"]["survivors"] == 461113 and d["config"]["ranks_agree_on_records"] is True


def test_bench_two_ranks_two_devices_over_rccl(built):
    """The real N > 1 path -- one rank per GPU, RCCL communicator over two devices, grouped per-root broadcasts, expansion of
    all ranks' words on every rank: needs a box with at least two GPUs (gpurun boxes have one: skipped there; the driver's
    8-GPU scaling run exercises the same code).  Every rank must hold the committed, oracle-checked record list."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    hip.hipGetDeviceCount(ctypes.byref(n))
    if n.value < 2:
        pytest.skip("one GPU on this box")
    d = _run_bench(["--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--only-headline"], nproc=2, timeout=900)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["rccl_ranks"] == 2 and c["exchange"].startswith("rccl")
    assert c["ranks_agree_on_records"] is True and c["matches_committed_digest"] is True
    assert c["survivors_frame_set_0"] == 29802555 and sum(c["survivors_per_rank"]) == 29802555

"""bench.py's multi-rank code path (id broadcast, sharded create, barrier + max-over-ranks timing, the
cross-rank bit-identity check, rank-0 JSON) exercised end to end on the 1-GPU box: two ranks on device 0
(mesh exchange over HIP IPC), launcher collectives over gloo.  On a real node the driver launches the same
script with one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,solver_name,fused", [(2, "pdlp", ""), (4, "pdlp", ""), (8, "pdlp", ""), (2, "hipdlp", ""), (4, "hipdlp", ""),
                                                     (2, "pdlp", "2")])  # "2": the multi-GPU form (an exchange per launch), forced
def test_bench_two_ranks_one_device(world, solver_name, fused):
    env = dict(os.environ, PDLP_BENCH_SINGLE_DEVICE="1", PDLP_BENCH_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PDLP_MI355X_MESH_TIMEOUT_MS="30000")
    if fused:
        env.update(PDLP_MI355X_DEV="1", PDLP_MI355X_MESH_FUSED_WAIT=fused)
    if world >= 8:  # (folded ranks + this process oversubscribe the device's hardware queues: see test_gpu_mesh._run_ranks)
        env.setdefault("GPU_MAX_HW_QUEUES", "1")
    port = 29700 + os.getpid() % 200 + world
    cmd = ["timeout", "400", sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
           "--config", "a", "--steps", "400", "--warmup", "80", "--solver", solver_name]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 400 and d["scaling"] == "strong"
    assert d["ranks_bit_identical"] is True
    assert "mesh" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0
    # which exchange ran, and the device-side time rank 0 spent waiting for its peers in each hot-loop exchange
    assert d["exchange"].startswith("direct xGMI mesh") and d["exchange_fallback"] is None
    w = d["exchange_waits"]
    assert w["waits"][0] > 0 and w["waits"][1] > 0 and w["X_allgather_x"] > 0 and w["P_reduce_scatter_aty"] > 0
    if solver_name == "pdlp":
        assert w["waits"][2] > 0 and w["S_scalars"] > 0
        # the checks of the sharded solve run on the device, collectives enqueued with them
        assert (d["trial_launches"], d["check_launches"]) == ((5, 14) if fused == "2" else (9, 26))


def test_bench_rccl_exchange_when_two_devices_are_visible():
    """The launcher path over RCCL (nccl backend) and the RCCL all-reduce exchange with N > 1: needs two physical
    GPUs (RCCL refuses two ranks on one device), so it only runs where the box has them."""
    import ctypes
    n = ctypes.c_int(0)
    ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n))
    if n.value < 2:
        pytest.skip("one GPU visible: RCCL needs one device per rank")
    for k, ex in enumerate(("rccl", "")):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if ex:
            env["PDLP_MI355X_EXCHANGE"] = ex
        port = 29900 + os.getpid() % 45 + 45 * k  # (a port of its own per launch)
        cmd = ["timeout", "400", sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
               "--config", "a", "--steps", "400", "--warmup", "80"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["ranks_bit_identical"] is True
        assert ("RCCL" in d["exchange"]) == (ex == "rccl") or d["exchange_fallback"]

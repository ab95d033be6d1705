"""GPU tests of the multi-GPU path's direct-exchange ("mesh") protocol, run on ONE device: WORLD processes,
one per rank, all on cuda:0, exchanging through HIP-IPC-mapped arenas exactly as they would over xGMI
(RCCL cannot put two ranks on one device; the mesh does not care).  Checks: every rank ends with
bit-identical results (same decisions everywhere), the sharded solve agrees with the single-GPU solve to
the tolerances of SURVEY §8c, and a fixed number of iterations gives the same iterate to rounding."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from highs_amd import solver
from highs_amd import lp as L

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _lp(name):
    return L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))


def _run_ranks(world, case, tmp_path, extra_env=None, must_finish=None):
    # the mesh exchange only hashes the 128-byte id into the name of its rendezvous segment; random bytes
    # keep librccl (whose first ncclGetUniqueId can take a minute on a box without network) out of these tests
    uid = (C.c_ubyte * 128).from_buffer_copy(os.urandom(128))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world >= 8:
        # Eight rank processes FOLDED onto one device plus this pytest process (which holds a used HIP context) need more
        # hardware queues than the device maps at once (three per process by default): the scheduler then time-slices the
        # processes, and a rank that spins on a peer which is switched out runs into the exchange's time-out —
        # 6 of 10 runs on the round-6 box (tools/r6_flake.sh; 0 of 10 without the ninth process, 0 of 10 with one queue per
        # rank).  A property of folding, not of the exchange: on a node every rank has a device of its own.
        env.setdefault("GPU_MAX_HW_QUEUES", "1")
    env.update(extra_env or {})
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen(["timeout", "240", sys.executable, os.path.join(HERE, "mesh_worker.py"), str(r),
                               str(world), bytes(uid).hex(), case, outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate()[0].decode(errors="replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-2000:]}"
    return [dict(np.load(o)) for r, o in enumerate(outs) if must_finish is None or r in must_finish]


# (8 processes time-share the one GPU's hardware queues: keep the 8-rank case small)
@pytest.mark.parametrize("name,world", [("afiro", 2), ("afiro", 4), ("afiro", 8), ("e226", 2), ("e226", 4)])
def test_mesh_sharded_solve(world, name, tmp_path):
    lp = _lp(name)
    base = solver.solveLpCupdlp(lp)
    res = _run_ranks(world, f"solve:{name}", tmp_path)
    for r in res:
        assert r["exchange"] in (2.0, 3.0), "the direct mesh exchange must be the one in use"
    # identical control flow and identical bits on every rank
    for r in res[1:]:
        for k in ("col_value", "col_dual", "row_value", "row_dual", "num_iter", "num_trials", "primal_obj", "dual_obj"):
            assert np.array_equal(r[k], res[0][k]), k
    r0 = res[0]
    assert int(r0["term"]) == 0
    obj = lp.objective_value(r0["col_value"])
    b = base.info["objective_function_value"]
    assert abs(obj - b) <= 1e-6 * (1 + abs(b))
    assert r0["primal_feas"] < 1e-7 * (1 + r0["norm_rhs"]) and r0["rel_gap"] < 1e-7
    assert 0.5 * base.pdlp_iteration_count <= int(r0["num_iter"]) <= 2 * base.pdlp_iteration_count
    # row activity really is A x (the gathered row-sharded vectors are in the right places)
    assert np.allclose(lp.row_activity(r0["col_value"]), r0["row_value"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,world", [("afiro", 2), ("e226", 2), ("e226", 4)])
def test_sharded_device_driven_checks_give_the_bits_of_host_driven_checks(name, world, tmp_path):
    """Round 5: the check iterations of the row-block sharded solve (two-all-gathers layout) run on the device — the
    statistics, the termination tests, the restart decision and the primal-weight update behind the trial batches, with
    the check's collectives (three all-gathers, two scalar all-reduces) enqueued between them instead of being driven from
    the host.  Same kernels on the same slices, same rank-ordered sums: whole solves equal the host-driven ones bit for
    bit, on every rank."""
    dev = _run_ranks(world, f"solve:{name}", tmp_path)
    (tmp_path / "host").mkdir()
    host = _run_ranks(world, f"solve:{name}", tmp_path / "host", extra_env={"PDLP_MI355X_DEVICE_CHECK": "0"})
    for a, b in zip(dev, host):
        for k in ("col_value", "col_dual", "row_value", "row_dual", "num_iter", "num_trials", "primal_obj", "dual_obj", "term"):
            assert np.array_equal(a[k], b[k]), k
    assert int(dev[0]["term"]) == 0 and int(dev[0]["num_iter"]) > 100


def test_sharded_hard_instance_device_driven_checks(tmp_path):
    """perold — the LP of check/instances that takes the method millions of iterations and hundreds of restarts — on two
    ranks (folded on this device), 4 000 iterations with the checks, restarts and primal-weight updates on the device: both
    ranks hold the same bits, and they are the bits of the host-driven checks (round 6)."""
    dev = _run_ranks(2, "solve:perold:0:4000", tmp_path)
    (tmp_path / "host").mkdir()
    host = _run_ranks(2, "solve:perold:0:4000", tmp_path / "host", extra_env={"PDLP_MI355X_DEVICE_CHECK": "0"})
    for k in ("col_value", "col_dual", "row_value", "row_dual", "num_iter", "num_trials", "primal_obj", "dual_obj", "term"):
        assert np.array_equal(dev[0][k], dev[1][k]), k
        assert np.array_equal(dev[0][k], host[0][k]), k
    assert int(dev[0]["num_iter"]) == 3999 and int(dev[0]["num_trials"]) > 3999


@pytest.mark.parametrize("case,world", [("iterate:25fv47:40", 4), ("iterate:synth:120", 2), ("iterate:synth:120", 8),
                                        ("iterate:synthbig:80", 4)])  # the bench workload (1M x 1M, slab layout, 2 MB slices)
def test_mesh_fixed_iterations_match_single_gpu(case, world, tmp_path):
    _, name, k = case.split(":")
    k = int(k)
    if name in ("synth", "synthbig"):
        sp_ = solver.SyntheticProblem(*((20000, 20000, 160000, 3) if name == "synth" else (1000000, 1000000, 8000000, 1)))
        S = solver.DeviceSolver(problem_struct=sp_.struct)
    else:
        S = solver.DeviceSolver(lp=_lp(name))
    S.iterate(k)
    x1 = S.get("x", S.n)
    S.close()
    res = _run_ranks(world, case, tmp_path)
    for r in res[1:]:
        assert np.array_equal(r["x"], res[0]["x"]) and np.array_equal(r["steps"], res[0]["steps"])
    assert int(res[0]["iters"]) == k
    # same iterate up to the different grouping of the reduction partials (a handful of ulps early on)
    err = np.linalg.norm(res[0]["x"] - x1) / (1e-300 + np.linalg.norm(x1))
    assert err < 1e-9, err


def test_mesh_features_off_and_rccl_switch(tmp_path):
    """Fixed step (power method through the generic collectives) + no restart, and the RCCL switch
    being honoured (single rank)."""
    lp = _lp("afiro")
    base = solver.solveLpCupdlp(lp, pdlp_features_off=6)
    res = _run_ranks(2, "solve:afiro:6", tmp_path)
    assert np.array_equal(res[0]["col_value"], res[1]["col_value"])
    obj = lp.objective_value(res[0]["col_value"])
    b = base.info["objective_function_value"]
    assert abs(obj - b) <= 1e-5 * (1 + abs(b))


@pytest.mark.parametrize("exchange", ["mesh", "rccl"])
def test_sharded_sequence_single_rank(exchange, monkeypatch):
    """Both exchanges forced onto one rank reproduce the single-GPU solve."""
    lp = _lp("e226")
    base = solver.solveLpCupdlp(lp)
    monkeypatch.setenv("PDLP_MI355X_FORCE_COMM", "1")
    monkeypatch.setenv("PDLP_MI355X_EXCHANGE", exchange)
    sh = solver.solveLpCupdlp(lp, time_limit=1000.0)
    assert sh.model_status == solver.kOptimal
    a, b = sh.info["objective_function_value"], base.info["objective_function_value"]
    assert abs(a - b) <= 1e-6 * (1 + abs(b))
    assert 0.5 * base.pdlp_iteration_count <= sh.pdlp_iteration_count <= 2 * base.pdlp_iteration_count


def test_vanished_peer_becomes_an_error_not_a_hang(tmp_path):
    """Every device-side wait is bounded: when a peer disappears the surviving rank's next exchange times
    out and the C ABI reports it (here after 2 s), instead of spinning on the GPU forever."""
    res = _run_ranks(2, "die:afiro", tmp_path, extra_env={"PDLP_MI355X_MESH_TIMEOUT_MS": "2000"}, must_finish={0})
    assert "timed out" in str(res[0]["msg"]), res[0]["msg"]
    assert float(res[0]["seconds"]) < 30.0


@pytest.mark.parametrize("name", sorted(L.special_lps()))
def test_mesh_special_lps_same_status_as_single_gpu(name, tmp_path):
    """The reference's unit-test LPs (optimal, infeasible, unbounded, ranged rows, maximisation) on two ranks:
    the certificates and objectives are assembled from row-sharded AND column-sliced statistics."""
    lp = L.special_lps()[name]
    base = solver.solveLpCupdlp(lp)
    res = _run_ranks(2, f"solve:{name}", tmp_path)
    assert np.array_equal(res[0]["col_value"], res[1]["col_value"]) and int(res[0]["term"]) == int(res[1]["term"])
    assert int(res[0]["term"]) == int(base.result.term_code)
    if base.model_status == solver.kOptimal:
        a, b = lp.objective_value(res[0]["col_value"]), base.info["objective_function_value"]
        assert abs(a - b) <= 1e-6 * (1 + abs(b))


@pytest.mark.parametrize("name,world", [("afiro", 2), ("afiro", 4), ("adlittle", 2), ("shell", 4), ("e226", 2)])
def test_mesh_sharded_hipdlp_solve(name, world, tmp_path):
    """The second reference path (solver="hipdlp") over row-block shards: per Halpern step one reduce-scatter of
    the partial A'y and one all-gather of the reflected x, no scalar exchange.  All ranks bit-identical; the
    result agrees with the single-GPU solve (only the rank-ordered partial sums group differently)."""
    lp = _lp(name)
    base = solver.solveLpHiPdlp(lp)
    res = _run_ranks(world, f"hsolve:{name}", tmp_path)
    for r in res:
        assert r["exchange"] == 2.0
    for r in res[1:]:
        for k in ("col_value", "col_dual", "row_value", "row_dual", "num_iter", "num_restarts", "primal_obj", "dual_obj"):
            assert np.array_equal(r[k], res[0][k]), k
    r0 = res[0]
    assert int(r0["term"]) == 0
    a, b = lp.objective_value(r0["col_value"]), lp.objective_value(base.solution.col_value)
    assert abs(a - b) <= 1e-6 * (1 + abs(b))
    assert abs(int(r0["num_iter"]) - base.pdlp_iteration_count) <= max(80, 0.1 * base.pdlp_iteration_count)
    assert r0["primal_feas"] < 1e-7 * (1 + r0["norm_rhs"]) and r0["dual_feas"] < 1e-7 * (1 + r0["norm_cost"])
    assert np.allclose(lp.row_activity(r0["col_value"]), r0["row_value"], rtol=1e-9, atol=1e-9)


def test_sharded_hipdlp_single_rank_reproduces_single_gpu(monkeypatch):
    """The sharded kernel sequence forced onto one rank: same arithmetic, so the same solve bit for bit."""
    lp = _lp("adlittle")
    base = solver.solveLpHiPdlp(lp)
    monkeypatch.setenv("PDLP_MI355X_FORCE_COMM", "1")
    sh = solver.solveLpHiPdlp(lp)
    assert sh.pdlp_iteration_count == base.pdlp_iteration_count
    assert np.array_equal(sh.solution.col_value, base.solution.col_value)
    assert np.array_equal(sh.solution.row_dual, base.solution.row_dual)


def test_mesh_with_release_acquire_fences_gives_the_same_bits(tmp_path):
    """PDLP_MI355X_MESH_FENCES=1 (the belt-and-braces form of the exchange that bench.py tries before falling
    back to RCCL) changes ordering instructions only: same iterates, bit for bit."""
    a = _run_ranks(2, "solve:e226", tmp_path)
    for level in ("1", "2"):  # (2: fences and a kernel per exchange step, bench.py's last stop in front of RCCL)
        b = _run_ranks(2, "solve:e226", tmp_path, extra_env={"PDLP_MI355X_MESH_FENCES": level})
        for k in ("col_value", "row_dual", "num_iter", "num_trials", "primal_obj"):
            assert np.array_equal(a[0][k], b[0][k]) and np.array_equal(b[0][k], b[1][k]), (level, k)
        assert b[0]["exchange"] in (2.0, 3.0)


@pytest.mark.parametrize("case,world", [("solve:e226", 2), ("iterate:synth:120", 2), ("iterate:synthbig:40", 2)])
def test_mesh_consumers_that_wait_themselves_give_the_same_bits(case, world, tmp_path):
    """With every rank on a GPU of its own an exchange of the hot loop is ONE kernel (MeshArgs::fusedWait == 2, round 6:
    push, epoch, wait, copy — five launches per trial; 1 = round 5's form, the consumers of the two all-gathers poll the
    peers' flags themselves); with ranks folded onto one device — this box — the wait is a single-block kernel of its own
    (0), so that a spinning grid cannot keep the producers off the CUs.  PDLP_MI355X_MESH_FUSED_WAIT=1|2 runs the multi-GPU
    forms here with two ranks (room for both): ordering of launches only, same iterates bit for bit."""
    a = _run_ranks(world, case, tmp_path, extra_env={"PDLP_MI355X_MESH_FUSED_WAIT": "0"})
    keys = [k for k in a[0] if k not in ("seconds",)]
    assert keys
    for level in ("1", "2"):
        b = _run_ranks(world, case, tmp_path, extra_env={"PDLP_MI355X_MESH_FUSED_WAIT": level})
        for r in range(world):
            for k in keys:
                assert np.array_equal(a[r][k], b[r][k]), (level, r, k)


@pytest.mark.parametrize("case", ["solve:e226", "solve:perold:0:4000"])
def test_sharded_checks_with_one_launch_per_exchange_give_the_same_bits(case, tmp_path):
    """Round 6: with every rank on a GPU of its own a device-driven check of the sharded solve is 14 launches instead of
    26 — an all-gather is one kernel (push, epoch, wait, copy, rendezvous by the block that finishes last), the statistics'
    fixed-order reduction carries its all-reduce (the block that finishes last runs the exchange), the two restart norms
    likewise.  Forced here on two folded ranks: whole solves with their checks, restarts and primal-weight updates equal
    the 26-launch form bit for bit on both ranks."""
    a = _run_ranks(2, case, tmp_path, extra_env={"PDLP_MI355X_MESH_FUSED_WAIT": "0"})
    (tmp_path / "one").mkdir()
    b = _run_ranks(2, case, tmp_path / "one", extra_env={"PDLP_MI355X_MESH_FUSED_WAIT": "2"})
    for r in range(2):
        for k in ("col_value", "col_dual", "row_value", "row_dual", "num_iter", "num_trials", "primal_obj", "dual_obj", "term"):
            assert np.array_equal(a[r][k], b[r][k]), (r, k)
            assert np.array_equal(b[0][k], b[r][k]), (r, k)
    assert int(b[0]["num_iter"]) > 100


@pytest.mark.parametrize("case,world", [("solve:e226", 2), ("iterate:synth:120", 4)])
def test_sharded_ranks_prepared_on_the_device_give_the_same_bits(case, world, tmp_path):
    """Sharded ranks prepare the whole problem on their device (formulate + scaling + both orientations, automatic
    from 200k nonzeros; forced here) and only cut their row block on the host: bit-identical to the host set-up."""
    a = _run_ranks(world, case, tmp_path, extra_env={"PDLP_MI355X_GPU_SETUP": "0"})
    b = _run_ranks(world, case, tmp_path, extra_env={"PDLP_MI355X_GPU_SETUP": "1"})
    keys = [k for k in a[0] if k not in ("seconds",)]
    assert keys
    for r in range(world):
        for k in keys:
            assert np.array_equal(a[r][k], b[r][k]), (r, k)

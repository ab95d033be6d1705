"""GPU parity tests of the HiPDLP path (solver="hipdlp"), through the C ABI, against the
oracle (oracle/hipdlp_oracle.c, itself pinned on the reference binary) and the reference goldens."""
import json
import os

import numpy as np
import pytest

import oraclelib as O
from highs_amd import solver
from highs_amd import lp as L

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_hipdlp.json")))


def _lp(name):
    return L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))


@pytest.fixture(params=["csr", "slab"])
def spmv_layout(request, monkeypatch):
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1" if request.param == "slab" else "0")
    return request.param


@pytest.mark.parametrize("name,steps", [("afiro", 1), ("afiro", 40), ("adlittle", 7), ("25fv47", 40), ("shell", 40),
                                        ("standgub", 23)])
def test_halpern_steps_bit_exact(name, steps, spmv_layout):
    """The fused step kernels (A'y + primal projection/reflection/blend, A x + dual ones) reproduce the
    oracle's x, y, x_next, y_next bit for bit over a block of Halpern steps with the same step sizes
    (element-wise arithmetic in the reference's order, majors summed left to right).  standgub has rows longer
    than the 512-entry SpMV block of small operands, 25fv47 rows longer than the slab layout's 256: the GPU sums
    those as segment tasks, and there the oracle runs in its device reduction order (oracle/gpu_order.h
    g_major_sum)."""
    lp = _lp(name)
    long_majors = name == "standgub" or (name == "25fv47" and spmv_layout == "slab")
    ref = O.hipdlp_probe(lp, steps, **(dict(device_reduction_order=True, device_layout=spmv_layout) if long_majors else {}))
    S = solver.DeviceSolver(lp=lp, solver="hipdlp")
    st = S.get("steps", 8)
    # the device power method differs from the CPU's only in the order of its dot products
    assert abs(st[4] - ref["lam"]) <= 1e-12 * ref["lam"]
    S.set("steps", [ref["tau"], ref["sigma"]])
    S.stage("steps", init=[steps])
    assert np.array_equal(S.get("x", S.n), ref["x_cur"])
    assert np.array_equal(S.get("y", S.m), ref["y_cur"])
    assert np.array_equal(S.get("x_next", S.n), ref["x_next"])
    assert np.array_equal(S.get("y_next", S.m), ref["y_next"])
    S.close()


def test_fixed_point_error_matches_oracle():
    lp = _lp("25fv47")
    ref = O.hipdlp_probe(lp, 40)
    S = solver.DeviceSolver(lp=lp, solver="hipdlp")
    S.set("steps", [ref["tau"], ref["sigma"]])
    fpe = S.stage("block")[0]
    assert abs(fpe - ref["fpe"]) <= 1e-10 * abs(ref["fpe"])
    S.close()


@pytest.mark.parametrize("name", ["afiro", "adlittle", "avgas", "blending", "chip", "shell"])
def test_solve_matches_oracle_and_reference(name):
    lp = _lp(name)
    ora = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn())
    gpu = solver.solveLpHiPdlp(lp)
    assert gpu.model_status == solver.kOptimal
    a, b = lp.objective_value(gpu.solution.col_value), lp.objective_value(ora.solution.col_value)
    assert abs(a - b) <= 1e-6 * (1 + abs(b))  # north-star tolerance on objectives
    assert abs(gpu.result.dual_obj - ora.result.dual_obj) <= 1e-6 * (1 + abs(ora.result.dual_obj))
    # the trajectories only differ by the summation order of the check-iteration reductions
    assert abs(gpu.pdlp_iteration_count - ora.pdlp_iteration_count) <= max(80, 0.1 * ora.pdlp_iteration_count)
    g = REF.get(name, {}).get("default", {})
    if g.get("objective"):
        assert abs(a - float(g["objective"])) <= 1e-6 * (1 + abs(a))
    r = gpu.result
    assert r.primal_feas < 1e-7 * (1 + r.norm_rhs) and r.dual_feas < 1e-7 * (1 + r.norm_cost) and r.rel_gap < 1e-7
    assert np.allclose(lp.row_activity(gpu.solution.col_value), gpu.solution.row_value, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("opts", [{"pdlp_step_size_strategy": 0}, {"pdlp_features_off": 1},
                                  {"pdlp_scaling_mode": 7, "pdlp_ruiz_iterations": 4}])
def test_option_variants_match_oracle(opts):
    lp = _lp("afiro")
    ora = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), kkt_tolerance=1e-5, **opts)
    gpu = solver.solveLpHiPdlp(lp, kkt_tolerance=1e-5, **opts)
    assert gpu.model_status == ora.model_status == solver.kOptimal
    a, b = lp.objective_value(gpu.solution.col_value), lp.objective_value(ora.solution.col_value)
    assert abs(a - b) <= 1e-4 * (1 + abs(b))
    assert abs(gpu.pdlp_iteration_count - ora.pdlp_iteration_count) <= max(80, 0.1 * ora.pdlp_iteration_count)


def test_iteration_limit_semantics():
    lp = _lp("adlittle")
    gpu = solver.solveLpHiPdlp(lp, pdlp_iteration_limit=200)
    assert gpu.model_status == solver.kIterationLimit and gpu.pdlp_iteration_count == 200
    assert not gpu.solution.col_value.any()  # the reference returns the zero start (pdhg.cc:866-877)


def test_large_synthetic_block_bit_exact_and_runs():
    """20k x 20k synthetic LP (slab layout off/on by size rule): first block vs the oracle, then 400 more
    iterations stay finite and inside the bounds."""
    sp_ = solver.SyntheticProblem(20000, 20000, 160000, 3)
    lp = sp_.to_lp()
    ref = O.hipdlp_probe(lp, 40)
    S = solver.DeviceSolver(problem_struct=sp_.struct, solver="hipdlp")
    S.set("steps", [ref["tau"], ref["sigma"]])
    S.stage("steps", init=[40])
    assert np.array_equal(S.get("x", S.n), ref["x_cur"]) and np.array_equal(S.get("y", S.m), ref["y_cur"])
    S.reset()
    st = S.iterate(400)
    assert st.iters == 400 and st.checks == 10
    x = S.get("x_next", S.n)
    lo, up = S.get("lower", S.n), S.get("upper", S.n)
    assert np.all(np.isfinite(x)) and np.all(x >= lo) and np.all(x <= up)
    S.close()


@pytest.mark.parametrize("seed", [0, 4, 14, 28, 3])
def test_random_lps_with_every_row_and_column_kind(seed):
    import lpgen
    lp = lpgen.random_lp(seed)  # free rows included here (the oracle covers them; MPS goldens cannot)
    limit = 20000 if seed % 2 == 0 else 400
    ora = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), kkt_tolerance=1e-6, pdlp_iteration_limit=limit)
    gpu = solver.solveLpHiPdlp(lp, kkt_tolerance=1e-6, pdlp_iteration_limit=limit)
    assert gpu.model_status == ora.model_status
    assert abs(gpu.pdlp_iteration_count - ora.pdlp_iteration_count) <= max(40, 0.1 * ora.pdlp_iteration_count)
    a, b = lp.objective_value(gpu.solution.col_value), lp.objective_value(ora.solution.col_value)
    assert abs(a - b) <= 1e-6 * (1 + abs(b))
    assert np.allclose(gpu.solution.row_dual, ora.solution.row_dual, rtol=1e-4, atol=1e-5 * (1 + np.abs(ora.solution.row_dual).max()))


@pytest.mark.parametrize("name,opts", [("afiro", {}), ("adlittle", {}), ("shell", {}), ("blending", {}), ("standgub", {}),
                                       ("e226", {}), ("afiro", {"pdlp_step_size_strategy": 0, "kkt_tolerance": 1e-5}),
                                       ("adlittle", {"pdlp_features_off": 1, "kkt_tolerance": 1e-4})])
def test_whole_solve_bit_exact_in_device_reduction_order(name, opts, monkeypatch):
    """With the oracle summing its reductions in the HIP kernels' order (oracle/gpu_order.h) a complete
    HiPDLP solve on the GPU — power method, every Halpern step, fixed-point errors, checks, restarts, PID
    primal weight, post-solve — is reproduced BIT FOR BIT.  (In its other mode the same oracle reproduces
    the reference binary's iteration counts and objectives: tests/test_hipdlp_oracle.py.)"""
    monkeypatch.setenv("PDLP_MI355X_SLAB", "0")  # the oracle's mode restates the CSR-stream SpMV (short majors)
    lp = _lp(name)
    ora = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), device_reduction_order=True, **opts)
    gpu = solver.solveLpHiPdlp(lp, **opts)
    assert gpu.model_status == ora.model_status == solver.kOptimal
    assert gpu.pdlp_iteration_count == ora.pdlp_iteration_count
    assert gpu.result.num_restarts == ora.result.num_restarts
    for k in ("col_value", "col_dual", "row_value", "row_dual"):
        assert np.array_equal(getattr(gpu.solution, k), getattr(ora.solution, k)), k
    for k in ("primal_obj", "dual_obj", "primal_feas", "dual_feas", "rel_gap"):
        assert getattr(gpu.result, k) == getattr(ora.result, k), k


@pytest.mark.parametrize("name", ["25fv47", "shell", "standgub", "random6", "synthetic"])
@pytest.mark.parametrize("opts", [{}, {"pdlp_features_off": 1}, {"pdlp_scaling_mode": 7, "pdlp_ruiz_iterations": 3}])
def test_device_side_setup_bit_identical_to_host_setup(name, opts, monkeypatch):
    """Preprocessing + scaling + both orientations built on the device (pdlp_setup.hip, HiPDLP form) vs the
    host path, which is bit-identical to the oracle: every prepared vector, and a block of Halpern steps
    through the device-built matrices."""
    sp_ = None
    if name == "synthetic":
        sp_ = solver.SyntheticProblem(30000, 25000, 240000, 7)
        kw = dict(problem_struct=sp_.struct)
    elif name.startswith("random"):
        import lpgen
        kw = dict(lp=lpgen.random_lp(int(name[6:])))  # has ranged AND free rows
    else:
        kw = dict(lp=_lp(name))
    out = []
    for dev in ("0", "1"):
        monkeypatch.setenv("PDLP_MI355X_GPU_SETUP", dev)
        S = solver.DeviceSolver(solver="hipdlp", **kw, **opts)
        vec = {k: S.get(k, S.n if k in ("cost", "lower", "upper", "col_scale") else S.m)
               for k in ("cost", "lower", "upper", "col_scale", "row_lower", "row_upper", "row_scale")}
        S.set("steps", [0.37, 0.21])
        S.stage("steps", init=[40])
        vec["x"], vec["y"], vec["xn"], vec["yn"] = S.get("x", S.n), S.get("y", S.m), S.get("x_next", S.n), S.get("y_next", S.m)
        out.append(((S.n, S.m, S.nnz, S.n_eqs), vec))
        S.close()
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k

"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oraclelib as O
from highs_amd import abi, solver
from highs_amd import lp as L

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_pdlp.json")))


def _lp(name):
    return L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))


def _spmv(beg, idx, val, x, m, long_limit=None):
    """Left-to-right row sums (the reference's order); with long_limit, majors longer than that in the product's
    segment-task order (oracle/gpu_order.h g_long_major_sum)."""
    out = np.zeros(m)
    p = lambda a, t: np.ascontiguousarray(a).ctypes.data_as(t)
    args = (m, p(beg, abi.c_i32p), p(idx, abi.c_i32p), p(val, abi.c_f64p),
            np.ascontiguousarray(x).ctypes.data_as(abi.c_f64p), out.ctypes.data_as(abi.c_f64p))
    if long_limit is None:
        O.oracle().pdlp_oracle_spmv_csr(*args)
    else:
        O.oracle().pdlp_oracle_spmv_csr_device_order(*args, long_limit)
    return out


def _problems():
    yield "25fv47", _lp("25fv47"), None
    yield "shell", _lp("shell"), None
    sp_ = solver.SyntheticProblem(30000, 25000, 240000, 7)
    yield "synthetic", None, sp_


@pytest.fixture(params=["csr", "slab"])
def spmv_layout(request, monkeypatch):
    """Run a test under both SpMV layouts: plain CSR stream and row-block x column-slab."""
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1" if request.param == "slab" else "0")
    return request.param


@pytest.mark.parametrize("which", ["25fv47", "shell", "synthetic"])
def test_spmv_bit_exact(which, spmv_layout):
    """A x (CSR) and A' y (CSC) — integer-exact placement and, because every major is summed left to
    right like AxCPU/ATyCPU (cupdlp_linalg.c:35-109), bit-identical values.  (Slab layout: majors longer than 256
    — 25fv47 has some — are segment tasks; those follow the modelled order of oracle/gpu_order.h.)"""
    for name, lp, sp_ in _problems():
        if name != which:
            continue
        kw = dict(problem_struct=sp_.struct) if sp_ else dict(lp=lp)
        P = solver.Prepared(**kw)
        S = solver.DeviceSolver(**kw)
        rng = np.random.default_rng(0)
        x = rng.standard_normal(P.n)
        y = rng.standard_normal(P.m)
        S.set("x", x)
        S.set("y", y)
        S.stage("ax")
        S.stage("aty")
        limit = 256 if spmv_layout == "slab" else None
        if limit is None:
            assert max(np.diff(P.csr_beg).max(), np.diff(P.csc_beg).max()) <= (512 if P.nnz < 2**18 else 2048)
        assert np.array_equal(S.get("ax", P.m), _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m, limit))
        assert np.array_equal(S.get("aty", P.n), _spmv(P.csc_beg, P.csc_idx, P.csc_val, y, P.n, limit))
        S.close()


def test_spmv_wide_matrix_many_slabs(spmv_layout):
    """A operand with 200k columns = 4 slabs of 65536: slab-ordered accumulation stays bit-exact."""
    sp_ = solver.SyntheticProblem(3000, 200000, 36000, 5)
    P = solver.Prepared(problem_struct=sp_.struct)
    S = solver.DeviceSolver(problem_struct=sp_.struct)
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(P.n), rng.standard_normal(P.m)
    S.set("x", x); S.set("y", y)
    S.stage("ax"); S.stage("aty")
    assert np.array_equal(S.get("ax", P.m), _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m))
    assert np.array_equal(S.get("aty", P.n), _spmv(P.csc_beg, P.csc_idx, P.csc_val, y, P.n))
    S.close()


def test_spmv_long_and_empty_majors(spmv_layout):
    """Rows longer than one LDS chunk (512 nnz for operands below 2^18 nonzeros, 2048 above: spmvChunkFor; 256 in
    the slab layout) are cut into segment tasks; empty rows/cols give 0."""
    rng = np.random.default_rng(1)
    n, m = 6000, 40
    rows, cols, vals = [], [], []
    for i in range(m):
        k = {0: 5000, 1: 0, 2: 2049, 3: 1}.get(i, int(rng.integers(0, 30)))
        c = np.sort(rng.choice(n, size=k, replace=False))
        rows += [i] * k
        cols += list(c)
        vals += list(rng.standard_normal(k))
    r_start = np.searchsorted(np.array(rows), np.arange(m + 1))
    inf = float("inf")
    lp = L.HighsLp.from_rowwise(n, m, r_start, cols, vals, col_cost=rng.standard_normal(n), col_lower=np.zeros(n),
                                col_upper=np.ones(n), row_lower=np.full(m, -inf), row_upper=np.ones(m))
    P = solver.Prepared(lp, pdlp_features_off=1)
    S = solver.DeviceSolver(lp, pdlp_features_off=1)
    x = rng.standard_normal(P.n)
    y = rng.standard_normal(P.m)
    S.set("x", x); S.set("y", y)
    S.stage("ax"); S.stage("aty")
    ax_o = _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m)
    ax_g = S.get("ax", P.m)
    lens = np.diff(P.csr_beg)
    limit = 256 if spmv_layout == "slab" else 512  # this operand is small: chunk 512
    short = lens <= limit
    assert (~short).sum() == 2
    assert np.array_equal(ax_g[short], ax_o[short])
    assert ax_g[lens == 0].tolist() == [0.0] * int((lens == 0).sum())
    scale = _spmv(P.csr_beg, P.csr_idx, np.abs(P.csr_val), np.abs(x), P.m)
    assert np.all(np.abs(ax_g - ax_o) <= 1e-14 * (scale + 1))  # long rows: segment sums
    assert np.array_equal(ax_g, _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m, limit))  # ... in exactly the modelled order
    assert np.array_equal(S.get("aty", P.n), _spmv(P.csc_beg, P.csc_idx, P.csc_val, y, P.n))
    S.close()


@pytest.mark.parametrize("which", ["25fv47", "synthetic"])
def test_trial_step_matches_oracle(which, spmv_layout):
    """One trial of cupdlp_step.c:241-257: x+, y+, A x+, A' y+ bit-identical; the three reductions
    (tree order on the GPU, left-to-right in the oracle) to 1e-12 relative."""
    for name, lp, sp_ in _problems():
        if name == which:
            _trial_step_check(lp, sp_)


def test_trial_step_matches_oracle_on_the_bench_workload(monkeypatch):
    """The same check on BASELINE config 4 itself (1M x 1M, 8M nnz; automatic layout = slab, device-side
    setup): the three trial sums of the GPU's reduction tree against the SERIAL oracle to 1e-12 — the guard
    against a mistake mirrored in both the kernels and the oracle's device-order model."""
    monkeypatch.delenv("PDLP_MI355X_SLAB", raising=False)
    _trial_step_check(None, solver.SyntheticProblem(1000000, 1000000, 8000000, 1))


def _trial_step_check(lp, sp_):
    if True:
        kw = dict(problem_struct=sp_.struct) if sp_ else dict(lp=lp)
        S = solver.DeviceSolver(**kw)
        if sp_:
            lp = sp_.to_lp()
        Fv = O.FormulatedView(lp)
        n, m = Fv.n, Fv.m
        rng = np.random.default_rng(5)
        x = np.clip(rng.standard_normal(n), Fv.lower, Fv.upper)
        y = rng.standard_normal(m)
        y[Fv.n_eqs:] = np.maximum(y[Fv.n_eqs:], 0.0)
        S.set("x", x); S.set("y", y)
        S.stage("ax"); S.stage("aty")
        ax, aty = S.get("ax", m), S.get("aty", n)
        tau, sigma, beta = 0.37, 0.21, 0.21 / 0.37
        S.set("steps", [tau, sigma, beta])
        out = S.stage("trial")
        accepted = out[3] == 1.0
        pre = "" if accepted else "_next"  # accepted -> parity flipped, the new iterate is "current"
        xg, yg = S.get("x" + pre if accepted else "x_next", n), S.get("y" + pre if accepted else "y_next", m)
        axg, atyg = S.get("ax" if accepted else "ax_next", m), S.get("aty" if accepted else "aty_next", n)
        # oracle
        P = abi.ProblemHandle(lp)
        F = O.Formulated()
        prm = abi.default_params()
        assert O.oracle().pdlp_oracle_formulate_scale(C.byref(P.struct), C.byref(prm), C.byref(F)) == 0
        xo, yo, axo, atyo, o3 = np.zeros(n), np.zeros(m), np.zeros(m), np.zeros(n), np.zeros(3)
        d = lambda a: a.ctypes.data_as(abi.c_f64p)
        O.oracle().pdlp_oracle_trial_step(C.byref(F), tau, sigma, d(x), d(y), d(ax), d(aty), d(xo), d(yo), d(axo),
                                          d(atyo), d(o3))
        O.oracle().pdlp_oracle_free_formulated(C.byref(F))
        slab = os.environ.get("PDLP_MI355X_SLAB") == "1" or (os.environ.get("PDLP_MI355X_SLAB") is None and max(n, m) >= 2**18)
        longest = max(np.diff(Fv.csr_beg).max(), np.bincount(Fv.csr_idx, minlength=n).max())
        if longest <= (256 if slab else 512 if Fv.nnz < 2**18 else 2048):
            assert np.array_equal(xg, xo) and np.array_equal(axg, axo)
            assert np.array_equal(yg, yo) and np.array_equal(atyg, atyo)
        else:  # long majors are summed as segment tasks (bit-exactness in THAT order: test_gpu_bitexact.py)
            assert np.array_equal(xg, xo) and np.allclose(axg, axo, rtol=1e-12, atol=1e-13)
            assert np.allclose(yg, yo, rtol=1e-12, atol=1e-13) and np.allclose(atyg, atyo, rtol=1e-11, atol=1e-12)
        assert np.allclose(out[:3], o3, rtol=1e-12, atol=0)
        # the decision itself (cupdlp_step.c:266-285)
        sb = np.sqrt(beta)
        movement = o3[0] * 0.5 * sb + o3[1] / (2 * sb)
        limit = movement / abs(o3[2])
        eta = np.sqrt(tau * sigma)
        assert accepted == (eta <= limit)
        S.close()


def test_residuals_match_numpy_restatement():
    lp = _lp("e226")
    S = solver.DeviceSolver(lp)
    S.iterate(120)
    Fv = O.FormulatedView(lp)
    n, m = Fv.n, Fv.m
    out = S.stage("residuals")
    x, y, ax, aty = S.get("x", n), S.get("y", m), S.get("ax", m), S.get("aty", n)
    r = ax - Fv.rhs
    r[Fv.n_eqs:] = np.minimum(r[Fv.n_eqs:], 0)
    pfeas = np.linalg.norm(r * Fv.row_scale)
    pobj = float(x @ Fv.cost) * lp.sense + lp.offset
    rc = Fv.cost - aty
    hasL, hasU = np.isfinite(Fv.lower), np.isfinite(Fv.upper)
    sp_, sn_ = np.maximum(rc, 0) * hasL, -np.minimum(rc, 0) * hasU
    lF, uF = np.where(hasL, Fv.lower, 0.0), np.where(hasU, Fv.upper, 0.0)
    dobj = (float(y @ Fv.rhs) + float(sp_ @ lF) - float(sn_ @ uF)) * lp.sense + lp.offset
    dfeas = np.linalg.norm((rc - sp_ + sn_) * Fv.col_scale)
    assert np.allclose(out[:4], [pobj, dobj, pfeas, dfeas], rtol=1e-11, atol=1e-13)
    assert np.array_equal(S.get("slack_pos", n), sp_) and np.array_equal(S.get("slack_neg", n), sn_)
    S.close()


SPECIAL_EXPECT = {"distillation": 31.2, "3d": 7.0, "boxed_row": -16.0, "blending": -2850.0}


@pytest.mark.parametrize("name", sorted(SPECIAL_EXPECT))
def test_special_lps_reference_unit_tests(name):
    """check/TestPdlp.cpp: objective within 1e-3 at kkt_tolerance 1e-4, status Optimal;
    and to 1e-6 relative at the default tolerance."""
    lp = L.special_lps()[name]
    out = solver.solveLpCupdlp(lp, kkt_tolerance=1e-4)
    assert out.status == solver.kOk and out.model_status == solver.kOptimal
    # TestPdlp.cpp uses an absolute 1e-3 on objectives of magnitude <= 31.2; blending (|obj| = 2850,
    # not in TestPdlp.cpp) gets the same bound relative to its magnitude
    assert abs(out.info["objective_function_value"] - SPECIAL_EXPECT[name]) < 1e-3 * max(1.0, abs(SPECIAL_EXPECT[name]) / 31.2)
    assert out.pdlp_iteration_count > 0
    out = solver.solveLpCupdlp(lp)
    assert out.model_status == solver.kOptimal
    assert abs(out.info["objective_function_value"] - SPECIAL_EXPECT[name]) <= 1e-6 * abs(SPECIAL_EXPECT[name])


def test_infeasible_and_unbounded_status():
    # check/TestPdlp.cpp:186-239 (kUnbounded comes from HiGHS' KKT check upgrading the same PDLP status)
    for name in ("infeasible", "unbounded"):
        out = solver.solveLpCupdlp(L.special_lps()[name], kkt_tolerance=1e-4)
        assert out.status == solver.kOk and out.model_status == solver.kUnboundedOrInfeasible


def test_iteration_limit_semantics():
    # check/TestPdlp.cpp:53-61: limit 80 -> 79 iterations, kIterationLimit, HighsStatus::kWarning
    out = solver.solveLpCupdlp(L.special_lps()["distillation"], kkt_tolerance=1e-4, pdlp_iteration_limit=80)
    assert out.model_status == solver.kIterationLimit and out.pdlp_iteration_count == 79
    assert out.status == solver.kWarning


def test_hot_start():
    lp = L.special_lps()["restart_lp"]
    a = solver.solveLpCupdlp(lp, kkt_tolerance=1e-4)
    start = {"col_value": a.solution.col_value, "row_value": a.solution.row_value, "row_dual": a.solution.row_dual}
    b = solver.solveLpCupdlp(lp, start=start, kkt_tolerance=1e-4)
    assert b.model_status == solver.kOptimal and b.pdlp_iteration_count < a.pdlp_iteration_count
    assert abs(b.info["objective_function_value"] - a.info["objective_function_value"]) < 1e-3


@pytest.mark.parametrize("features_off", [1, 2, 4, 7])
def test_feature_switches_converge_to_same_optimum(features_off):
    lp = _lp("afiro")
    out = solver.solveLpCupdlp(lp, kkt_tolerance=1e-6, pdlp_features_off=features_off, pdlp_iteration_limit=400000)
    assert out.model_status == solver.kOptimal
    assert abs(out.info["objective_function_value"] - (-464.7531428571)) <= 1e-4 * 464.75


@pytest.mark.parametrize("name", sorted(REF))
def test_instances_match_reference_cpu_pdlp(name):
    """BASELINE config 1 and the 13 ctest instances: same .mps input, default tolerance (1e-7);
    objective and KKT measures agree with the reference CPU pdlp to 1e-6 relative."""
    lp = _lp(name)
    g = REF[name]
    out = solver.solveLpCupdlp(lp)
    assert out.model_status == solver.kOptimal  # what PDLP itself reports (HiGHS may downgrade standata/standgub)
    ref_obj = g["cupdlp"]["objective_function_value"]
    obj = out.info["objective_function_value"]
    assert abs(obj - ref_obj) <= 1e-6 * (1.0 + abs(ref_obj)), (obj, ref_obj)
    R = out.result
    assert abs(R.primal_obj - g["cupdlp"]["primal_obj"]) <= 1e-6 * (1 + abs(ref_obj))
    assert abs(R.dual_obj - g["cupdlp"]["dual_obj"]) <= 1e-6 * (1 + abs(ref_obj))
    # KKT residuals in the reference's own normalisation (termination test, cupdlp_solver.c:813-816)
    assert R.primal_feas < 1e-7 * (1 + R.norm_rhs) and R.dual_feas < 1e-7 * (1 + R.norm_cost) and R.rel_gap < 1e-7
    assert R.norm_rhs == g["cupdlp"]["norm_rhs"] and R.norm_cost == g["cupdlp"]["norm_cost"]
    assert out.info["primal_dual_objective_error"] <= 2e-7
    # same ballpark of work as the CPU trajectory (not required to be equal: TestPdlp.cpp:98-113).  The count is
    # sensitive to the summation order: 80bau3b took 289 600 iterations with 2048-entry SpMV blocks and 93 360
    # with 512-entry ones (reference CPU: 374 520), hence the wide band
    assert 0.2 * g["cupdlp"]["num_iter"] <= out.pdlp_iteration_count <= 5 * g["cupdlp"]["num_iter"]
    assert out.pdlp_iteration_count == GPU_PINS[name]["pdlp_iteration_count"]  # exact: the device's sums have a fixed order


MORE = json.load(open(os.path.join(GOLD, "reference_pdlp_more.json")))
# The GPU path's own iteration counts (tools/make_gpu_iteration_pins.py, run on an MI355X): every sum on the device has
# a fixed order, so they are reproducible exactly — a different count means a summation order or a decision changed.
# Regenerate the file then, and say in the commit which change moved them.
GPU_PINS = json.load(open(os.path.join(GOLD, "gpu_iteration_counts.json")))


@pytest.mark.parametrize("name", sorted(MORE))
def test_more_instances_match_reference_cpu_pdlp(name):
    """The other LPs of the reference's check/instances that its CPU pdlp finishes in minutes (make_golden_more.py):
    five it solves to optimality — objectives and KKT norms to 1e-6 — and nine it reports as primal infeasible or
    unbounded — same verdict (cuPDLP termination code and HiGHS model status)."""
    lp = _lp(name)
    g = MORE[name]
    out = solver.solveLpCupdlp(lp)
    R = out.result
    assert R.term_code == g["cupdlp"]["term_code"]
    assert R.norm_rhs == g["cupdlp"]["norm_rhs"] and R.norm_cost == g["cupdlp"]["norm_cost"]
    if g["expect"] == "optimal":
        assert out.model_status == solver.kOptimal
        ref_obj = g["cupdlp"]["objective_function_value"]
        # 1e-6 relative — except etamacro, 2e-6: both runs stop on the same criterion (relative KKT 1e-7) with a primal
        # residual of 2e-4 against duals of order 10, which leaves the objective itself determined to ~1e-3 absolute
        # on 756 (measured difference 8.0e-4 = 1.05e-6 relative)
        tol = (2e-6 if name == "etamacro" else 1e-6) * (1.0 + abs(ref_obj))
        assert abs(out.info["objective_function_value"] - ref_obj) <= tol
        assert abs(R.primal_obj - g["cupdlp"]["primal_obj"]) <= tol
        assert abs(R.dual_obj - g["cupdlp"]["dual_obj"]) <= tol
        assert R.primal_feas < 1e-7 * (1 + R.norm_rhs) and R.dual_feas < 1e-7 * (1 + R.norm_cost) and R.rel_gap < 1e-7
    else:
        assert out.model_status == solver.kUnboundedOrInfeasible
        assert g["highs"]["model_status"] == "Primal infeasible or unbounded"
    assert 0.2 * g["cupdlp"]["num_iter"] <= out.pdlp_iteration_count <= max(5 * g["cupdlp"]["num_iter"], 40)
    assert out.pdlp_iteration_count == GPU_PINS[name]["pdlp_iteration_count"]


HARD = json.load(open(os.path.join(GOLD, "reference_hard.json"))) if os.path.exists(os.path.join(GOLD, "reference_hard.json")) else {}


@pytest.mark.parametrize("layout", ["csr", "slab"])
@pytest.mark.parametrize("name", ["perold", "greenbea", "gas11", "primal1"])
def test_hard_instances_first_iterations_bit_exact(name, layout, monkeypatch):
    """The LPs of check/instances on which a first-order method struggles (millions of iterations, many restarts;
    make_golden_hard.py): the first 4 000 iterations of the GPU solve, bit for bit against the oracle's device-order mode
    — iterates of the last iteration, step sizes, trial and restart counts.  (primal1 is a QP with a diagonal Hessian: the
    oracle's QP extension.)  Round 6: also in the SLAB layout — the work-balanced partition, the XCD-affine segment tasks
    (greenbea has majors beyond 256 entries) and the fused two-launch trial, where this round's kernel changes live."""
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1" if layout == "slab" else "0")
    lp = _lp(name)
    kw = dict(kkt_tolerance=1e-12, pdlp_iteration_limit=4000)
    cpu = O.oracle_solve(lp, device_reduction_order=True, device_layout=layout, **kw)
    gpu = solver.solveLpCupdlp(lp, **kw)
    R = gpu.result
    assert (R.term_code, R.num_iter, R.num_trials, R.num_restarts) == (cpu.term_code, cpu.num_iter, cpu.num_trials, cpu.num_restarts)
    assert R.num_iter == 3999 or R.term_code == 0  # (the reference stops at limit - 1; primal1 is solved to 1e-12 after 680)
    assert R.primal_obj == cpu.primal_obj and R.dual_obj == cpu.dual_obj and R.primal_feas == cpu.primal_feas and R.dual_feas == cpu.dual_feas
    assert np.array_equal(gpu.solution.col_value, cpu.col_value) and np.array_equal(gpu.solution.row_dual, cpu.row_dual)


@pytest.mark.slow
def test_perold_whole_solve_in_the_slab_layout(monkeypatch):
    """perold solved to the default tolerance in the SLAB layout (two launches per trial with the in-kernel grid barrier,
    work-balanced partition): millions of iterations on the code path of this round's kernel changes.  The layouts sum in
    different orders, so the trajectory is not the CSR pin's; every run stops on the same criterion (relative KKT 1e-7,
    with a primal residual of ~4e-4 against duals of order 10^3), which determines the objective itself to a few 1e-6
    relative on this LP: the real cuPDLP-C core ends 3.8e-7 from the reference simplex optimum, the CSR layout 4.1e-7, this
    layout 1.7e-6 (round 6, measured) — the bound here is 3e-6, as for etamacro above.  (~3 minutes on the device: slab
    launches are sized for operands a thousand times larger.)"""
    g = HARD.get("perold")
    pin = GPU_PINS.get("perold")
    if g is None or pin is None:
        pytest.skip("no golden for perold")
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1")
    lp = _lp("perold")
    out = solver.solveLpCupdlp(lp, time_limit=900.0)
    ref = g["simplex"]["objective_value"]
    assert out.model_status == solver.kOptimal
    assert abs(out.info["objective_function_value"] - ref) <= 3e-6 * (1.0 + abs(ref))
    R = out.result
    assert R.primal_feas < 1e-7 * (1 + R.norm_rhs) and R.dual_feas < 1e-7 * (1 + R.norm_cost) and R.rel_gap < 1e-7
    assert out.pdlp_iteration_count >= 1_000_000
    assert 0.25 * pin["pdlp_iteration_count"] <= out.pdlp_iteration_count <= 4 * pin["pdlp_iteration_count"]


@pytest.mark.slow
@pytest.mark.parametrize("name", ["perold", "greenbea", "gas11", "primal1"])
def test_hard_instances_reach_the_reference_simplex_result(name):
    """... and the converged result at the DEFAULT tolerance against the reference's own simplex (primal1: its QP solver):
    optimal objectives to 1e-6, gas11 (unbounded) gets the reference's verdict.  No CPU-pdlp convergence is needed for
    the golden; where the real cuPDLP-C core does converge within its time budget the record holds
    its iteration count for comparison.  Iteration counts on the device are reproducible exactly and pinned."""
    g = HARD.get(name)
    pin = GPU_PINS.get(name)
    if g is None or pin is None or pin.get("skip"):
        pytest.skip("no GPU pin recorded for %s: %s" % (name, (pin or {}).get("skip", "run tools/r5_hard.py on an MI355X")))
    lp = _lp(name)
    out = solver.solveLpCupdlp(lp, time_limit=600.0)
    ref = g["simplex"]
    if ref["model_status"] == "Optimal":
        assert out.model_status == solver.kOptimal, out.model_status
        obj = lp.objective_value(out.solution.col_value)
        assert abs(obj - ref["objective_value"]) <= 1e-6 * max(1.0, abs(ref["objective_value"])), (obj, ref["objective_value"])
    else:
        assert ref["model_status"] == "Unbounded" and out.model_status == solver.kUnboundedOrInfeasible
    assert out.pdlp_iteration_count == pin["pdlp_iteration_count"]


SYNTH = json.load(open(os.path.join(GOLD, "reference_synth.json"))) if os.path.exists(os.path.join(GOLD, "reference_synth.json")) else {}


def _converged_against_reference(key, m, n, nnz):
    """BASELINE configs 2 / 4: the synthetic LP solved to the reference's DEFAULT tolerance (1e-7) on the GPU,
    against the converged solution of the real cuPDLP-C core on the same LP (tests/golden/reference_synth.json,
    generated by tests/golden/make_golden_synth.py): objectives to 1e-6 relative — the north_star tolerance —
    and KKT residuals inside the termination test both sides were held to."""
    g = SYNTH[key]
    sp_ = solver.SyntheticProblem(m, n, nnz, 1)
    S = solver.DeviceSolver(problem_struct=sp_.struct)  # default tolerances: 1e-7
    assert (S.n, S.m, S.nnz) == (g["n"], g["m"], g["nnz"])
    R = S.run(n, m)
    assert R.term_code == abi.TERM_OPTIMAL == g["term_code"]
    lp = sp_.to_lp()
    ref = g["objective_function_value"]
    obj = lp.objective_value(R.col_value)
    scale = 1.0 + abs(ref)
    assert abs(obj - ref) <= 1e-6 * scale, (obj, ref)
    assert abs(R.primal_obj - g["primal_obj"]) <= 1e-6 * scale and abs(R.dual_obj - g["dual_obj"]) <= 1e-6 * scale
    assert R.norm_rhs == g["norm_rhs"] and R.norm_cost == g["norm_cost"]  # same formulated LP, bit for bit
    # the reference's own termination test (cupdlp_solver.c:813-816), which its golden satisfies as well
    assert R.primal_feas < 1e-7 * (1 + R.norm_rhs) and R.dual_feas < 1e-7 * (1 + R.norm_cost) and R.rel_gap < 1e-7
    assert g["primal_feas"] < 1e-7 * (1 + g["norm_rhs"]) and g["rel_gap"] < 1e-7
    k = L.kkt_measures(lp, R.col_value, R.col_dual, R.row_value, R.row_dual)
    kr = g["kkt"]
    assert k["max_primal_residual_error"] < 1e-9 and k["max_dual_residual_error"] < 1e-9
    assert k["max_primal_infeasibility"] <= 10 * max(kr["max_primal_infeasibility"], 1e-7)
    assert k["max_dual_infeasibility"] <= 10 * max(kr["max_dual_infeasibility"], 1e-7)
    assert k["primal_dual_objective_error"] <= 2e-7
    # same ballpark of work as the CPU trajectory (not required to be equal: TestPdlp.cpp:98-113)
    assert 0.25 * g["num_iter"] <= R.num_iter <= 4 * g["num_iter"]
    S.close()
    return R


@pytest.mark.skipif("a_tol1e-07" not in SYNTH, reason="golden for config 2 not generated")
def test_synthetic_100k_converged_matches_reference():
    _converged_against_reference("a_tol1e-07", 100000, 100000, 1000000)


@pytest.mark.skipif("c_small_tol1e-07" not in SYNTH, reason="golden for the structured LP not generated")
@pytest.mark.parametrize("key", ["c_small_tol1e-07", "c_tol1e-07"])
def test_structured_lp_converged_matches_reference(key):
    """BASELINE config 3 stand-in: block-angular network LP with dense linking rows (long majors -> CSR side
    kernel next to the slab kernel), ranged and free rows; converged objectives against the real cuPDLP-C core."""
    if key not in SYNTH:
        pytest.skip("golden not generated")
    from lpgen import structured_lp
    kw = dict(commodities=16, nodes=1024, arcs=8192, link_rows=64, link_nnz=2048, extra_rows=128) if "small" in key else {}
    lp = structured_lp(1, **kw)
    g = SYNTH[key]
    assert (lp.num_row, lp.num_col, lp.num_nz) == (g["m"], g["n"], g["nnz"])
    out = solver.solveLpCupdlp(lp)
    assert out.model_status == solver.kOptimal
    R = out.result
    ref = g["objective_function_value"]
    scale = 1.0 + abs(ref)
    assert abs(out.info["objective_function_value"] - ref) <= 1e-6 * scale
    assert abs(R.primal_obj - g["primal_obj"]) <= 1e-6 * scale and abs(R.dual_obj - g["dual_obj"]) <= 1e-6 * scale
    assert R.norm_rhs == g["norm_rhs"] and R.norm_cost == g["norm_cost"]
    assert out.info["max_primal_residual_error"] <= max(10 * g["kkt"]["max_primal_residual_error"], 1e-9)
    assert out.info["max_dual_residual_error"] <= max(10 * g["kkt"]["max_dual_residual_error"], 1e-9)
    assert 0.25 * g["num_iter"] <= R.num_iter <= 4 * g["num_iter"]


D_SMALL = dict(periods=64, rows_per=1024, cols_per=896, dense_cols=48, dense_nnz=4000, tail_rows=512, tail_max=2500)


@pytest.mark.parametrize("key", ["d_small_tol1e-07", "d_tol1e-07"])
def test_dense_column_lp_converged_matches_reference(key):
    """Second structured family (round 4, tests/lpgen.py::dense_column_lp): a staircase LP with DENSE COLUMNS (thousands
    of entries: segment tasks in the A'y launch — the transpose of config c's dense rows), power-law row lengths,
    <= / >= / ranged rows and a maximisation sense; converged objectives against the real cuPDLP-C core
    (tests/golden/make_golden_synth.py d_small / d), 1e-6 relative, and the trial loop must not fall off its fast path
    because of the long columns."""
    if key not in SYNTH:
        pytest.skip("golden not generated")
    from lpgen import dense_column_lp
    lp = dense_column_lp(1, **(D_SMALL if "small" in key else {}))
    g = SYNTH[key]
    assert (lp.num_row, lp.num_col, lp.num_nz) == (g["m"], g["n"], g["nnz"])
    out = solver.solveLpCupdlp(lp)
    assert out.model_status == solver.kOptimal
    R = out.result
    ref = g["objective_function_value"]
    scale = 1.0 + abs(ref)
    assert abs(out.info["objective_function_value"] - ref) <= 1e-6 * scale
    assert abs(R.primal_obj - g["primal_obj"]) <= 1e-6 * scale and abs(R.dual_obj - g["dual_obj"]) <= 1e-6 * scale
    assert R.norm_rhs == g["norm_rhs"] and R.norm_cost == g["norm_cost"]
    assert out.info["max_primal_residual_error"] <= max(10 * g["kkt"]["max_primal_residual_error"], 1e-9)
    assert out.info["max_dual_residual_error"] <= max(10 * g["kkt"]["max_dual_residual_error"], 1e-9)
    assert 0.25 * g["num_iter"] <= R.num_iter <= 4 * g["num_iter"]


HELD_OUT_SMALL = {"e_small": dict(n=20_000, m=150_000, window=1024, dense_rows=24, dense_nnz=3000),
                  "f_small": dict(n=90_000, m=80_000, band=2048, hubs=400)}


@pytest.mark.parametrize("key", ["e_small_tol1e-07", "f_small_tol1e-07"])
def test_held_out_families_converged_match_reference(key):
    """The held-out families of round 6 (tests/lpgen.py tall_lp / powerlaw_band_lp) at a size the real cuPDLP-C core
    converges on in minutes: objective and cuPDLP primal / dual objective to 1e-6 relative, KKT residuals no worse than ten
    times the reference's (tests/golden/make_golden_synth.py e_small / f_small)."""
    if key not in SYNTH:
        pytest.skip("golden not generated")
    from lpgen import powerlaw_band_lp, tall_lp
    lp = (tall_lp if key.startswith("e") else powerlaw_band_lp)(1, **HELD_OUT_SMALL[key[:7]])
    g = SYNTH[key]
    assert (lp.num_row, lp.num_col, lp.num_nz) == (g["m"], g["n"], g["nnz"])
    out = solver.solveLpCupdlp(lp)
    assert out.model_status == solver.kOptimal
    R = out.result
    ref = g["objective_function_value"]
    scale = 1.0 + abs(ref)
    assert abs(out.info["objective_function_value"] - ref) <= 1e-6 * scale
    assert abs(R.primal_obj - g["primal_obj"]) <= 1e-6 * scale and abs(R.dual_obj - g["dual_obj"]) <= 1e-6 * scale
    assert R.norm_rhs == g["norm_rhs"] and R.norm_cost == g["norm_cost"]
    assert out.info["max_primal_residual_error"] <= max(10 * g["kkt"]["max_primal_residual_error"], 1e-9)
    assert out.info["max_dual_residual_error"] <= max(10 * g["kkt"]["max_dual_residual_error"], 1e-9)
    assert 0.25 * g["num_iter"] <= R.num_iter <= 4 * g["num_iter"]


@pytest.mark.skipif("b_tol1e-07" not in SYNTH, reason="golden for config 4 not generated")
def test_synthetic_1m_converged_matches_reference():
    _converged_against_reference("b_tol1e-07", 1000000, 1000000, 8000000)


def test_full_size_properties_1m():
    """BASELINE config 4 size (1M x 1M, 8M nnz), size-independent properties:
    adjointness <A x, y> = <x, A' y>, linearity of both SpMVs, idempotent projection,
    and the per-iteration invariants ax == A x, aty == A' y after real iterations."""
    sp_ = solver.SyntheticProblem(1000000, 1000000, 8000000, 1)
    S = solver.DeviceSolver(problem_struct=sp_.struct, kkt_tolerance=1e-4)
    n, m = S.n, S.m
    assert (n, m, S.nnz) == (1000000, 1000000, 7999977)
    rng = np.random.default_rng(11)
    x1, x2, y1 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)

    def Ax(x):
        S.set("x", x); S.stage("ax"); return S.get("ax", m)

    def ATy(y):
        S.set("y", y); S.stage("aty"); return S.get("aty", n)

    a1, a2, a12 = Ax(x1), Ax(x2), Ax(x1 + x2)
    assert np.allclose(a12, a1 + a2, rtol=0, atol=1e-12 * np.abs(a12).max())
    t1 = ATy(y1)
    lhs, rhs = float(a1 @ y1), float(x1 @ t1)
    assert abs(lhs - rhs) <= 1e-11 * (np.linalg.norm(a1) * np.linalg.norm(y1))
    S.reset()
    st = S.iterate(60)
    assert st.iters == 60 and st.trials >= 60
    x, y = S.get("x", n), S.get("y", m)
    ax, aty = S.get("ax", m), S.get("aty", n)
    lo, up = S.get("lower", n), S.get("upper", n)
    assert np.all(x >= lo) and np.all(x <= up) and np.all(y[S.n_eqs:] >= 0)
    assert np.array_equal(Ax(x), ax) and np.array_equal(ATy(y), aty)
    S.close()


@pytest.mark.parametrize("name", ["25fv47", "shell", "boxed_row", "restart_lp", "synthetic"])
@pytest.mark.parametrize("features_off", [0, 1])
def test_gpu_setup_bit_identical_to_host_setup(name, features_off, monkeypatch):
    """Formulate + Ruiz/Pock-Chambolle scaling on the device (pdlp_setup.hip) vs the host path (which is
    bit-identical to the oracle / reference): every prepared vector must agree bit for bit."""
    monkeypatch.setenv("PDLP_MI355X_GPU_SETUP", "1")  # force it also for the small LPs
    sp_ = None
    if name == "synthetic":
        sp_ = solver.SyntheticProblem(30000, 25000, 240000, 7)
        kw = dict(problem_struct=sp_.struct)
    elif name in L.special_lps():
        kw = dict(lp=L.special_lps()[name])
    else:
        kw = dict(lp=_lp(name))
    P = solver.Prepared(pdlp_features_off=features_off, **kw)
    S = solver.DeviceSolver(pdlp_features_off=features_off, **kw)  # GPU-side setup is the default
    assert (S.n, S.m, S.nnz, S.n_eqs) == (P.n, P.m, P.nnz, P.n_eqs)
    for nm, ref in [("cost", P.cost), ("rhs", P.rhs), ("lower", P.lower), ("upper", P.upper),
                    ("col_scale", P.col_scale), ("row_scale", P.row_scale)]:
        assert np.array_equal(S.get(nm, len(ref)), ref), nm
    # the matrices: A x and A' y through the device-built layouts, bit-exact vs the host-built CSR/CSC
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal(P.n), rng.standard_normal(P.m)
    S.set("x", x); S.set("y", y); S.stage("ax"); S.stage("aty")
    assert np.array_equal(S.get("ax", P.m), _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m))
    assert np.array_equal(S.get("aty", P.n), _spmv(P.csc_beg, P.csc_idx, P.csc_val, y, P.n))
    S.close()


@pytest.mark.parametrize("layout", ["0", "1"])
def test_gpu_setup_gives_identical_solve(layout, monkeypatch):
    lp = _lp("e226")
    monkeypatch.setenv("PDLP_MI355X_SLAB", layout)
    monkeypatch.setenv("PDLP_MI355X_GPU_SETUP", "0")
    a = solver.solveLpCupdlp(lp)
    monkeypatch.setenv("PDLP_MI355X_GPU_SETUP", "1")
    b = solver.solveLpCupdlp(lp)
    assert a.pdlp_iteration_count == b.pdlp_iteration_count
    assert np.array_equal(a.solution.col_value, b.solution.col_value)
    assert np.array_equal(a.solution.row_dual, b.solution.row_dual)
    assert a.result.norm_rhs == b.result.norm_rhs and a.result.norm_cost == b.result.norm_cost


def test_gpu_setup_long_rows_slab_layout(monkeypatch):
    """Device-built slab layout with long majors (> 256 nnz) cut into segment tasks."""
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1")
    monkeypatch.setenv("PDLP_MI355X_GPU_SETUP", "1")
    rng = np.random.default_rng(1)
    n, m = 70000, 60
    rows, cols, vals = [], [], []
    for i in range(m):
        k = {0: 5000, 1: 0, 2: 2049, 3: 1, 4: 300}.get(i, int(rng.integers(0, 30)))
        c = np.sort(rng.choice(n, size=k, replace=False))
        rows += [i] * k; cols += list(c); vals += list(rng.standard_normal(k))
    r_start = np.searchsorted(np.array(rows), np.arange(m + 1))
    inf = float("inf")
    lp = L.HighsLp.from_rowwise(n, m, r_start, cols, vals, col_cost=rng.standard_normal(n), col_lower=np.zeros(n),
                                col_upper=np.ones(n), row_lower=np.full(m, -inf), row_upper=np.ones(m))
    P = solver.Prepared(lp)
    S = solver.DeviceSolver(lp)
    x, y = rng.standard_normal(P.n), rng.standard_normal(P.m)
    S.set("x", x); S.set("y", y); S.stage("ax"); S.stage("aty")
    ax_o = _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m)
    lens = np.diff(P.csr_beg)
    ax_g = S.get("ax", P.m)
    assert (lens > 256).sum() == 3
    assert np.array_equal(ax_g[lens <= 256], ax_o[lens <= 256])
    assert np.allclose(ax_g, ax_o, rtol=1e-13, atol=1e-13)
    assert np.array_equal(ax_g, _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m, 256))  # segment tasks, modelled order
    assert np.array_equal(S.get("aty", P.n), _spmv(P.csc_beg, P.csc_idx, P.csc_val, y, P.n))
    S.close()


def test_slab_width_does_not_change_the_bits(monkeypatch):
    """Structured operand (block-angular network LP in small): its row blocks touch few stretches of the gathered vector
    densely, so the slab layout is built with 2^14-column slabs instead of 2^17 (pdlp_kernels.hpp kSlabTileLog2).  Entry
    order inside a major, hence every sum, does not depend on the slab width: the plain SpMVs (also against the oracle's
    device order) and the iterates after 120 iterations with every fused epilogue in the loop agree bit for bit."""
    from lpgen import structured_lp
    lp = structured_lp(seed=2, commodities=12, nodes=1024, arcs=8192, link_rows=24, link_nnz=700, extra_rows=40)
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1")
    out = {}
    for width in ("", "17", "12"):
        if width:
            monkeypatch.setenv("PDLP_MI355X_SLAB_W", width)
        else:
            monkeypatch.delenv("PDLP_MI355X_SLAB_W", raising=False)
        P = solver.Prepared(lp)
        S = solver.DeviceSolver(lp)
        rng = np.random.default_rng(4)
        x, y = rng.standard_normal(P.n), rng.standard_normal(P.m)
        S.set("x", x); S.set("y", y); S.stage("ax"); S.stage("aty")
        ax, aty = S.get("ax", P.m), S.get("aty", P.n)
        assert np.array_equal(ax, _spmv(P.csr_beg, P.csr_idx, P.csr_val, x, P.m, 256))
        assert np.array_equal(aty, _spmv(P.csc_beg, P.csc_idx, P.csc_val, y, P.n, 256))
        S.close()
        S = solver.DeviceSolver(lp)
        S.iterate(120)
        out[width] = (ax, aty, S.get("x", P.n), S.get("y", P.m), S.get("steps", 8))
        S.close()
    for w in ("17", "12"):
        for a, b in zip(out[""], out[w]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("name,iters", [("afiro", 160), ("25fv47", 400), ("80bau3b", 400), ("synthetic", 240),
                                        # majors longer than a work block (segment tasks inside the persistent loop):
                                        ("standata", 400), ("standgub", 400), ("standmps", 400), ("cplex1", 400), ("staircase", 240)])
def test_trial_loop_variants_give_the_same_bits(name, iters, monkeypatch):
    """The trial loop exists as separate launches and as ONE persistent launch (pdlp_small.hip) — on one XCD, on all
    XCDs with every workgroup sweeping the arrival words, on all XCDs with the XCD-hierarchical barrier; mid-size LPs
    (2048-entry work blocks, hundreds of workgroups) only take the last.  Small work blocks without long rows run it
    with TWO barriers per trial (phase A recomputes x+ of the columns it gathers: no P phase), everything else — and
    PDLP_MI355X_PRIMAL_IN_A=0 — with three.  Work blocks, lanes and sums are the same in all of them: iterates, step
    sizes and trial counts after a few hundred iterations (several check iterations and restarts among them) must
    agree bit for bit."""
    sp_ = None
    if name == "synthetic":
        sp_ = solver.SyntheticProblem(40000, 35000, 400000, 9)  # above 2^18 nonzeros: ~200 work blocks of 2048 entries per operand
        kw = dict(problem_struct=sp_.struct)
    elif name == "staircase":  # dense columns (~2800 entries) and a power-law tail of rows, mid-size work blocks
        from lpgen import dense_column_lp
        kw = dict(lp=dense_column_lp(3, periods=48, rows_per=512, cols_per=448, dense_cols=24, dense_nnz=3000, tail_rows=256, tail_max=2600))
    else:
        kw = dict(lp=_lp(name))
    variants = {"launches": {"PDLP_MI355X_PERSISTENT": "0"},
                "persistent": {},
                "all-xcds-sweep": {"PDLP_MI355X_XCD_LOCAL": "0", "PDLP_MI355X_HIER_BARRIER": "0"},
                "all-xcds-hierarchical": {"PDLP_MI355X_XCD_LOCAL": "0", "PDLP_MI355X_HIER_BARRIER": "1"},
                "persistent-three-barriers": {"PDLP_MI355X_PRIMAL_IN_A": "0"},
                "all-xcds-sweep-three-barriers": {"PDLP_MI355X_XCD_LOCAL": "0", "PDLP_MI355X_HIER_BARRIER": "0", "PDLP_MI355X_PRIMAL_IN_A": "0"},
                "all-xcds-hierarchical-three-barriers": {"PDLP_MI355X_XCD_LOCAL": "0", "PDLP_MI355X_HIER_BARRIER": "1", "PDLP_MI355X_PRIMAL_IN_A": "0"}}
    if name == "staircase":  # 2048-entry blocks: only the hierarchical barrier
        del variants["all-xcds-sweep"], variants["all-xcds-sweep-three-barriers"]
    out = {}
    for vname, env in variants.items():
        for k in ("PDLP_MI355X_PERSISTENT", "PDLP_MI355X_XCD_LOCAL", "PDLP_MI355X_HIER_BARRIER", "PDLP_MI355X_PRIMAL_IN_A"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        S = solver.DeviceSolver(**kw)
        st = S.iterate(iters)
        launches = int(S.stage("trial_launches")[0])
        out[vname] = (S.get("x", S.n), S.get("y", S.m), S.get("steps", 8), int(st.trials), int(st.iters), launches, int(S.stage("trial_barriers")[0]))
        S.close()
    if sp_ is not None:
        sp_.close()
    assert out["launches"][5] in (2, 3) and out["persistent"][5] == 0 and out["all-xcds-hierarchical"][5] == 0
    # two barriers where the work blocks are small and no row is a segment task; three on request and everywhere else
    two = name in ("afiro", "25fv47", "80bau3b", "staircase")  # (cplex1 and the stand* LPs have rows longer than a work block)
    assert out["launches"][6] == 0 and out["persistent"][6] == (2 if two else 3) and out["persistent-three-barriers"][6] == 3, [o[6] for o in out.values()]
    ref = out["launches"]
    for vname, o in out.items():
        for a, b in zip(ref[:3], o[:3]):
            assert np.array_equal(a, b), vname
        assert ref[3:5] == o[3:5], vname


@pytest.mark.parametrize("name", ["afiro", "25fv47", "standata", "standmps", "80bau3b", "synthetic-stream", "synthetic-slab", "qp"])
def test_device_driven_checks_give_the_bits_of_host_driven_checks(name, monkeypatch):
    """Since round 4 the check iteration (residuals, termination, restart, primal-weight update) runs on the device
    behind the trial batch, several periods queued ahead of the host (pdlp_check.hip, Solver::doSolveDevice);
    PDLP_MI355X_DEVICE_CHECK=0 gives the host-driven loop back (what the sharded paths run).  Same kernels for the
    statistics, the same scalar arithmetic: complete solves must agree bit for bit — status, iteration / trial /
    restart counts, residuals, every solution vector."""
    sp_ = None
    if name == "synthetic-stream":
        sp_ = solver.SyntheticProblem(30000, 25000, 200000, 3)
        lp, kw = sp_.to_lp(), dict(kkt_tolerance=1e-5)
    elif name == "synthetic-slab":  # gathered vectors beyond 2^18 entries: slab layout, 2-launch fused trial, hipGraph batches
        sp_ = solver.SyntheticProblem(300000, 280000, 2000000, 5)
        lp, kw = sp_.to_lp(), dict(kkt_tolerance=1e-4, pdlp_iteration_limit=2500)
    elif name == "qp":
        lp, kw = L.HighsLp.from_npz(os.path.join(GOLD, "qp", "qp0.npz")), dict(kkt_tolerance=1e-8)
    else:
        lp, kw = _lp(name), {}
    # host-driven | device-driven as a sequence of ten launches | device-driven, the check as ONE launch where the LP runs
    # on the persistent loop with at most 64 workgroups (pdlp_check.hip k_check_small; elsewhere the same as the second)
    modes = {"host": {"PDLP_MI355X_DEVICE_CHECK": "0"}, "launches": {"PDLP_MI355X_CHECK_SMALL": "0"}, "default": {}}
    out = {}
    for mode, env in modes.items():
        for k in ("PDLP_MI355X_DEVICE_CHECK", "PDLP_MI355X_CHECK_SMALL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out[mode] = solver.solveLpCupdlp(lp, **kw)
    if sp_ is not None:
        sp_.close()
    a = out["host"].result
    for mode in ("launches", "default"):
        b = out[mode].result
        assert (a.term_code, a.term_iterate, a.num_iter, a.num_trials, a.num_restarts) == (b.term_code, b.term_iterate, b.num_iter, b.num_trials, b.num_restarts), mode
        assert (a.primal_obj, a.dual_obj, a.primal_feas, a.dual_feas, a.rel_gap) == (b.primal_obj, b.dual_obj, b.primal_feas, b.dual_feas, b.rel_gap), mode
        for v in ("col_value", "col_dual", "row_value", "row_dual"):
            assert np.array_equal(getattr(out["host"].solution, v), getattr(out[mode].solution, v)), (mode, v)
    assert a.num_iter > 0


def test_concurrent_solver_contexts_on_two_threads():
    """SURVEY §8b threading contract: several Highs instances may call the path concurrently from different
    threads, so a context holds no process-global mutable state.  Two threads solve different LPs (both
    reference paths) at the same time; every result must be bit-identical to the same solve done alone."""
    import threading
    lps = {"afiro": _lp("afiro"), "adlittle": _lp("adlittle"), "shell": _lp("shell"), "sctest": _lp("sctest")}
    fns = {"afiro": solver.solveLpCupdlp, "adlittle": solver.solveLpHiPdlp, "shell": solver.solveLpHiPdlp,
           "sctest": solver.solveLpCupdlp}
    alone = {k: fns[k](lp) for k, lp in lps.items()}
    out, errs = {}, []

    def work(name):
        try:
            for _ in range(3):
                out[name] = fns[name](lps[name])
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in lps]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k in lps:
        assert out[k].model_status == alone[k].model_status == solver.kOptimal
        assert out[k].pdlp_iteration_count == alone[k].pdlp_iteration_count
        assert np.array_equal(out[k].solution.col_value, alone[k].solution.col_value), k
        assert np.array_equal(out[k].solution.row_dual, alone[k].solution.row_dual), k


def test_fused_trial_keeps_long_columns(monkeypatch):
    """Dense columns in a slab-layout operand: the 2-launch fused trial used to be switched off by a single long column
    (its segment tasks ran in extra workgroups that took no part in the grid barrier).  Round 5: the 64-register variant
    of the fused kernel carries them as co-resident task workgroups that arrive at its barrier; where two blocks per CU do
    not fit (or with PDLP_MI355X_FUSED_COTASKS=0) the streaming blocks run the task passes themselves behind their stream
    (round 4's form).  Same lanes and sums either way: the bits of the 3-launch trial, and of the oracle."""
    from lpgen import dense_column_lp
    lp = dense_column_lp(5, periods=40, rows_per=512, cols_per=448, dense_cols=16, dense_nnz=5000, tail_rows=128, tail_max=700)
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1")
    monkeypatch.setenv("PDLP_MI355X_PERSISTENT", "0")
    out = {}
    for name, fused, cotasks in (("three", "0", "1"), ("inline", "1", "0"), ("cotasks", "1", "1")):
        monkeypatch.setenv("PDLP_MI355X_FUSED", fused)
        monkeypatch.setenv("PDLP_MI355X_FUSED_COTASKS", cotasks)
        out[name] = _iterate_state(dict(lp=lp), 240)
    assert out["three"][5] == 3 and out["inline"][5] == 2 and out["cotasks"][5] == 2, [o[5] for o in out.values()]
    for name in ("inline", "cotasks"):
        assert out[name][6] == 0, (name, "a barrier launch gave up")
        for a, b in zip(out["three"][:3], out[name][:3]):
            assert np.array_equal(a, b), name
        assert out["three"][3:5] == out[name][3:5], name


def _iterate_state(kw, iters):
    S = solver.DeviceSolver(**kw)
    st = S.iterate(iters)
    out = (S.get("x", S.n), S.get("y", S.m), S.get("steps", 8), int(st.trials), int(st.iters), int(S.stage("trial_launches")[0]),
           int(S.stage("barrier_fallbacks")[0]))
    S.close()
    return out


@pytest.mark.parametrize("which", ["persistent", "fused"])
def test_barrier_launch_that_cannot_be_resident_falls_back_to_plain_launches(which, monkeypatch):
    """Launches with in-kernel grid barriers (the persistent trial loop, the 2-launch fused trial) are plain launches: HIP
    promises no co-residency, and on a shared device a workgroup may wait for CUs another tenant holds.  The persistent
    launch therefore starts with a roll call and changes nothing if it fails; the fused trial's barrier fails for all
    blocks or for none and leaves the trial undecided (pdlp_devfn.hpp rollCall / gridBarrier).  PDLP_MI355X_FAULT makes
    exactly that happen once (one workgroup too many expected, 30 ms timeout): the solver must go on with plain launches
    — and, because a failed launch changes nothing, with the very same bits."""
    if which == "persistent":
        kw, iters, fault = dict(lp=_lp("25fv47")), 400, "1"
    else:
        sp_ = solver.SyntheticProblem(300000, 280000, 2000000, 5)
        kw, iters, fault = dict(problem_struct=sp_.struct), 160, "2"
    ref = _iterate_state(kw, iters)
    assert ref[5] == (0 if which == "persistent" else 2) and ref[6] == 0
    monkeypatch.setenv("PDLP_MI355X_FAULT", fault)
    monkeypatch.setenv("PDLP_MI355X_BARRIER_TIMEOUT_MS", "30")
    out = _iterate_state(kw, iters)
    assert out[5] == 3 and out[6] == 1, out[5:]
    for a, b in zip(ref[:3], out[:3]):
        assert np.array_equal(a, b)
    assert ref[3:5] == out[3:5]
    # the same through a whole solve (device-driven checks queued behind the failing launch must all stay shut)
    if which == "persistent":
        monkeypatch.delenv("PDLP_MI355X_FAULT")
        a = solver.solveLpCupdlp(kw["lp"])
        monkeypatch.setenv("PDLP_MI355X_FAULT", fault)
        b = solver.solveLpCupdlp(kw["lp"])
        assert a.pdlp_iteration_count == b.pdlp_iteration_count and np.array_equal(a.solution.col_value, b.solution.col_value)


def test_two_large_contexts_concurrently():
    """Two solver contexts with grid-barrier launches on ONE device at the same time (several Highs
    instances on several threads): two mid-size LPs on the persistent loop (hundreds of workgroups each, more than the
    device holds together) and two slab-layout LPs on the fused 2-launch trial, next to a third tenant that needs no
    barriers (a HiPDLP solve whose 1024-thread kernels keep taking and releasing every CU).  Barrier rounds of different
    contexts are ordered ON THE DEVICE (an event chain per device; the host-side gate is held only while a round is being
    enqueued): no deadlock, no barrier timeouts (the default timeout is 1 s: a stall would show in the wall clock), every
    result bit-identical to the same work done alone — and since a context's round is queued while another one's is
    still running, doing the work together costs no more than doing it one after the other (round 4 held the gate across
    the synchronisation and only promised 3x)."""
    import threading
    import time
    mids = [solver.SyntheticProblem(40000, 35000, 400000, 9), solver.SyntheticProblem(38000, 36000, 380000, 10)]
    bigs = [solver.SyntheticProblem(300000, 280000, 2000000, 5), solver.SyntheticProblem(290000, 300000, 2100000, 6)]
    jobs = {"mid0": (dict(problem_struct=mids[0].struct), 2000), "mid1": (dict(problem_struct=mids[1].struct), 2000),
            "big0": (dict(problem_struct=bigs[0].struct), 400), "big1": (dict(problem_struct=bigs[1].struct), 400)}
    t0 = time.time()
    alone = {k: _iterate_state(kw, it) for k, (kw, it) in jobs.items()}
    t_alone = time.time() - t0
    assert alone["mid0"][5] == 0 and alone["big0"][5] == 2
    hip_lp = bigs[1].to_lp()
    t0 = time.time()
    solver.solveLpHiPdlp(hip_lp, kkt_tolerance=1e-12, pdlp_iteration_limit=1200)
    t_alone += time.time() - t0
    out, errs = {}, []

    def work(name):
        try:
            out[name] = _iterate_state(*jobs[name])
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))

    def tenant():
        try:
            out["tenant"] = solver.solveLpHiPdlp(hip_lp, kkt_tolerance=1e-12, pdlp_iteration_limit=1200)
        except Exception as e:  # noqa: BLE001
            errs.append(("tenant", repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in jobs] + [threading.Thread(target=tenant)]
    t0 = time.time()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    t_together = time.time() - t0
    assert not errs, errs
    for k in jobs:
        assert out[k][6] == 0, (k, "a barrier launch gave up")
        for a, b in zip(alone[k][:3], out[k][:3]):
            assert np.array_equal(a, b), k
        assert alone[k][3:6] == out[k][3:6], k
    # the gate is correctness (bit identity, no barrier fall-back, above); the clock is informative on a device that may be
    # shared or throttled: a warning beyond 1.25x, a failure only when the contexts evidently ran one after the other
    if t_together >= 1.25 * t_alone + 1.0:
        import warnings
        warnings.warn("concurrent contexts took %.2f s against %.2f s one after the other" % (t_together, t_alone))
    assert t_together < 3.0 * t_alone + 3.0, (t_together, t_alone)
    for p in mids + bigs:
        p.close()


def test_time_limit_status_on_both_paths():
    """time_limit exceeded -> kTimeLimit (CupdlpWrapper.cpp:235: iters < limit-1; HiPdlpWrapper.cpp:120-123)."""
    sp_ = solver.SyntheticProblem(200000, 200000, 1600000, 5)
    lp = sp_.to_lp()
    a = solver.solveLpCupdlp(lp, kkt_tolerance=1e-12, time_limit=0.2)
    assert a.model_status == solver.kTimeLimit and a.pdlp_iteration_count > 0
    b = solver.solveLpHiPdlp(lp, kkt_tolerance=1e-12, time_limit=0.2)
    assert b.model_status == solver.kTimeLimit and b.pdlp_iteration_count > 0 and b.pdlp_iteration_count % 40 == 0


def test_bad_arguments_are_reported_not_thrown():
    import ctypes as C
    L_ = solver.lib()
    assert L_.pdlp_mi355x_solve(None, None, None) != 0
    assert b"null" in L_.pdlp_mi355x_last_error()
    prm = abi.default_params()
    prm.algorithm = 7
    P = abi.ProblemHandle(_lp("afiro"))
    R = abi.ResultHandle(32, 27)
    assert L_.pdlp_mi355x_solve(C.byref(P.struct), C.byref(prm), C.byref(R.struct)) != 0
    assert b"algorithm" in L_.pdlp_mi355x_last_error()


def test_lp_without_constraints_is_an_error_not_a_hang():
    inf = float("inf")
    no_rows = L.HighsLp(2, 0, np.array([1.0, -1.0]), np.zeros(2), np.array([1.0, 2.0]), np.zeros(0), np.zeros(0),
                        np.array([0, 0, 0], np.int32), np.zeros(0, np.int32), np.zeros(0), 1, 0.0, "norows").normalise()
    no_cols = L.HighsLp(0, 2, np.zeros(0), np.zeros(0), np.zeros(0), np.array([-inf, -1.0]), np.array([1.0, inf]),
                        np.array([0], np.int32), np.zeros(0, np.int32), np.zeros(0), 1, 0.0, "nocols").normalise()
    for lp in (no_rows, no_cols):
        for fn in (solver.solveLpCupdlp, solver.solveLpHiPdlp):
            out = fn(lp, pdlp_iteration_limit=1000)
            assert out.status == solver.kError and b"solveUnconstrainedLp" in solver.lib().pdlp_mi355x_last_error()


def test_nan_in_the_data_ends_in_an_error_not_an_endless_step_size_search():
    lp = _lp("afiro")
    lp.col_cost = lp.col_cost.copy()
    lp.col_cost[3] = float("nan")
    out = solver.solveLpCupdlp(lp, pdlp_iteration_limit=100000, time_limit=60.0)
    # the library reports the search that cannot end (RETCODE != OK -> HighsStatus::kError + kSolveError, CupdlpWrapper.cpp:225-251)
    assert out.status == solver.kError and out.model_status == solver.kSolveError

"""A family of random small LPs with every row/column kind: the oracle must equal the real cuPDLP-C
core bit for bit (CPU), and the GPU path must reach the same optimum and status (GPU)."""
import numpy as np
import pytest

import oraclelib as O
from highs_amd import abi, solver
from lpgen import random_lp

SEEDS = list(range(24))


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_equals_reference_core_on_random_lps(seed):
    lp = random_lp(seed)
    a = O.oracle_solve(lp, kkt_tolerance=1e-6, pdlp_iteration_limit=200000)
    b = O.ref_solve(lp, kkt_tolerance=1e-6, pdlp_iteration_limit=200000)
    assert (a.term_code, a.num_iter, a.num_trials) == (b.term_code, b.num_iter, b.num_trials)
    assert a.primal_obj == b.primal_obj and a.dual_obj == b.dual_obj
    assert np.array_equal(a.col_value, b.col_value) and np.array_equal(a.row_dual, b.row_dual)
    assert np.array_equal(a.row_value, b.row_value) and np.array_equal(a.col_dual, b.col_dual)


@pytest.mark.parametrize("seed", SEEDS)
def test_host_formulation_of_random_lps_bit_exact(seed):
    lp = random_lp(seed)
    P, F = solver.Prepared(lp), O.FormulatedView(lp)
    for k in ["csr_beg", "csr_idx", "csr_val", "cost", "rhs", "lower", "upper", "col_scale", "row_scale"]:
        assert np.array_equal(getattr(P, k), getattr(F, k)), k
    assert (P.n, P.m, P.n_eqs) == (F.n, F.m, F.n_eqs)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_matches_oracle_on_random_lps(seed, monkeypatch):
    lp = random_lp(seed)
    if seed % 4 == 0:
        monkeypatch.setenv("PDLP_MI355X_GPU_SETUP", "1")  # also exercise the device-side setup on odd shapes
    if seed % 4 == 1:
        monkeypatch.setenv("PDLP_MI355X_SLAB", "1")
    cpu = O.oracle_solve(lp, kkt_tolerance=1e-7, pdlp_iteration_limit=400000)
    gpu = solver.solveLpCupdlp(lp, kkt_tolerance=1e-7, pdlp_iteration_limit=400000)
    assert gpu.result.term_code == cpu.term_code
    if cpu.term_code == abi.TERM_OPTIMAL:
        ref = lp.objective_value(cpu.col_value)
        assert abs(gpu.info["objective_function_value"] - ref) <= 1e-5 * (1.0 + abs(ref))
        assert gpu.info["max_primal_infeasibility"] <= 1e-4 and gpu.info["max_primal_residual_error"] <= 1e-8


@pytest.mark.gpu
def test_limits():
    lp = random_lp(3)
    out = solver.solveLpCupdlp(lp, pdlp_iteration_limit=1)
    assert out.model_status == solver.kIterationLimit and out.pdlp_iteration_count == 0
    out = solver.solveLpCupdlp(lp, pdlp_iteration_limit=0)
    assert out.pdlp_iteration_count == 0 and out.status != solver.kError
    big = solver.SyntheticProblem(200000, 200000, 1600000, 2).to_lp()
    out = solver.solveLpCupdlp(big, time_limit=0.0)
    assert out.model_status == solver.kTimeLimit and out.status == solver.kWarning


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_reference_core_on_many_more_random_lps():
    """Seeds 100..299, two settings each (default features at 1e-6; a feature-switch combination chosen by the seed at
    1e-4): iteration and trial counts, termination code, objectives and the solution vectors, bit for bit."""
    for seed in range(100, 300):
        lp = random_lp(seed)
        for kw in (dict(kkt_tolerance=1e-6, pdlp_iteration_limit=40000),
                   dict(kkt_tolerance=1e-4, pdlp_iteration_limit=4000, pdlp_features_off=seed % 8)):
            a, b = O.oracle_solve(lp, **kw), O.ref_solve(lp, **kw)
            assert (a.term_code, a.num_iter, a.num_trials, a.primal_obj, a.dual_obj) == \
                   (b.term_code, b.num_iter, b.num_trials, b.primal_obj, b.dual_obj), (seed, kw)
            assert np.array_equal(a.col_value, b.col_value) and np.array_equal(a.row_dual, b.row_dual), (seed, kw)

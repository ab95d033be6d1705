"""CPU tests of the HiPDLP path (solver="hipdlp"):
 * the oracle restatement (oracle/hipdlp_oracle.c) is PINNED on outputs of the reference binary run in the
   build container (tests/golden/reference_hipdlp.json, made by tests/golden/make_golden_hipdlp.py):
   identical iteration counts, identical objective to the printed 16 digits, solutions to print precision;
 * the product's host-side preprocessing + scaling (pdlp_hipdlp_host.cpp, through the C ABI) is bit-identical
   to the oracle's."""
import json
import os

import numpy as np
import pytest

import oraclelib as O
from highs_amd import solver
from highs_amd import lp as L

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_hipdlp.json")))
CASES = [(n, k) for n, rec in sorted(REF.items()) for k, v in sorted(rec.items()) if "skipped" not in v]
KKT = {"default": {}, "kkt1e-4": {"kkt_tolerance": 1e-4}}


def _lp(name):
    return L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))


def test_golden_file_covers_the_reference_unit_test_case():
    # check/TestPdlpHi.cpp: afiro at kkt_tolerance 1e-4 must be Optimal
    assert REF["afiro"]["kkt1e-4"]["model_status"] == "Optimal"
    assert len(CASES) >= 10


@pytest.mark.parametrize("name,key", CASES)
def test_oracle_reproduces_reference_binary(name, key):
    g = REF[name][key]
    lp = _lp(name)
    out = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), **KKT[key])
    assert out.model_status == solver.kOptimal and g["model_status"] == "Optimal"
    assert out.pdlp_iteration_count == g["pdlp_iterations"]
    obj = lp.objective_value(out.solution.col_value)
    ref_obj = float(g["objective"])
    assert abs(obj - ref_obj) <= 1e-12 * max(1.0, abs(ref_obj)), (obj, ref_obj)  # the file prints 14-16 digits
    for k, a in (("col_value", out.solution.col_value), ("row_value", out.solution.row_value),
                 ("col_dual", out.solution.col_dual), ("row_dual", out.solution.row_dual)):
        b = np.asarray(g[k])
        # the solution file prints 15 significant digits (and flushes |v| < 1e-13 .. to fewer)
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9 * (1 + np.abs(b).max())), k


@pytest.mark.parametrize("name", ["afiro", "25fv47", "shell", "standgub", "e226"])
@pytest.mark.parametrize("opts", [{}, {"pdlp_features_off": 1}, {"pdlp_scaling_mode": 7, "pdlp_ruiz_iterations": 3}])
def test_product_host_preparation_bit_identical_to_oracle(name, opts):
    lp = _lp(name)
    P = solver.Prepared(lp=lp, solver="hipdlp", **opts)
    Q = O.hipdlp_prepared(lp, **opts)
    assert (P.n, P.m, P.n_eqs, P.nnz) == (Q["n"], Q["m"], Q["n_eqs"], Q["nnz"])
    assert np.array_equal(P.csc_beg, Q["beg"]) and np.array_equal(P.csc_idx, Q["idx"])
    assert np.array_equal(P.csc_val, Q["val"])
    for a, b in ((P.cost, Q["cost"]), (P.rhs, Q["row_lower"]), (P.lower, Q["lower"]), (P.upper, Q["upper"]),
                 (P.col_scale, Q["col_scale"]), (P.row_scale, Q["row_scale"])):
        assert np.array_equal(a, b)
    assert P.norm_cost == Q["norm_cost"] and P.norm_rhs == Q["norm_rhs"]


def test_iteration_limit_returns_the_zero_start_like_the_reference():
    """pdhg.cc:866-877: only a converged check writes the output vectors."""
    lp = _lp("adlittle")
    out = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), pdlp_iteration_limit=200)
    assert out.model_status == solver.kIterationLimit and out.pdlp_iteration_count == 200
    assert not out.solution.col_value.any() and not out.solution.row_dual.any()


def test_fixed_step_strategy_and_scaling_off_still_converge():
    lp = _lp("afiro")
    a = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), pdlp_step_size_strategy=0, kkt_tolerance=1e-4)
    b = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), pdlp_features_off=1, kkt_tolerance=1e-4)
    for o in (a, b):
        assert o.model_status == solver.kOptimal
        assert abs(lp.objective_value(o.solution.col_value) + 464.753) < 0.1


RAND = json.load(open(os.path.join(GOLD, "reference_hipdlp_random.json")))


@pytest.mark.parametrize("seed", sorted(int(k) for k in RAND))
def test_oracle_reproduces_reference_binary_on_random_lps(seed):
    """Every row kind (EQ / GEQ / LEQ / ranged) and column kind, empty rows and columns, an offset; odd
    seeds are maximisation LPs, which the reference MINIMISES on this path (pdhg.cc:171,481) — they hit the
    iteration limit and come back as the zero start."""
    import lpgen
    g = RAND[str(seed)]
    lp = lpgen.drop_free_rows(lpgen.random_lp(seed))
    out = solver.solveLpHiPdlp(lp, solve_fn=O.hipdlp_solve_fn(), kkt_tolerance=1e-6,
                               pdlp_iteration_limit=g["iteration_limit"])
    assert out.pdlp_iteration_count == g["pdlp_iterations"]
    assert (out.model_status == solver.kOptimal) == (g["model_status"] == "Optimal")
    if g["model_status"] != "Optimal":
        assert out.model_status == solver.kIterationLimit
    obj, ref_obj = lp.objective_value(out.solution.col_value), float(g["objective"])
    assert abs(obj - ref_obj) <= 1e-12 * max(1.0, abs(ref_obj))  # the file prints ~14 significant digits
    b = np.asarray(g["col_value"])
    assert np.allclose(out.solution.col_value, b, rtol=1e-9, atol=1e-9 * (1 + np.abs(b).max()))

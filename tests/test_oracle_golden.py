"""The oracle (CPU restatement) against the reference's own golden values.
Runs on CPU; this is what pins the oracle."""
import json
import os

import numpy as np
import pytest

import oraclelib as O
from highs_amd import lp as L

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_pdlp.json")))
SPECIAL = json.load(open(os.path.join(GOLD, "special_lps.json")))
FAST = ["afiro", "adlittle", "avgas", "blending", "chip", "sctest", "standata", "standgub", "e226", "shell"]
SLOW = ["25fv47", "scrs8", "stair", "80bau3b"]


def test_distillation_iteration_counts_pinned():
    # check/TestPdlp.cpp:29,44 (160) and :53-61 (79 under pdlp_iteration_limit=80)
    lp = L.special_lps()["distillation"]
    r = O.oracle_solve(lp, kkt_tolerance=1e-4)
    assert r.term_code == 0 and r.num_iter == 160
    assert abs(lp.objective_value(r.col_value) - 31.2) < 1e-3
    r = O.oracle_solve(lp, kkt_tolerance=1e-4, pdlp_iteration_limit=80)
    assert r.term_code == 4 and r.num_iter == 79


@pytest.mark.parametrize("name,obj,term", [("3d", 7.0, 0), ("boxed_row", -16.0, 0), ("infeasible", None, 3),
                                           ("unbounded", None, 3)])
def test_special_lps(name, obj, term):
    # check/TestPdlp.cpp:119-239
    lp = L.special_lps()[name]
    r = O.oracle_solve(lp, kkt_tolerance=1e-4)
    assert r.term_code == term
    if obj is not None:
        assert abs(lp.objective_value(r.col_value) - obj) < 1e-3
    g = SPECIAL[name]
    assert r.num_iter == g["num_iter"] and r.num_trials == g["num_trials"]
    assert r.primal_obj == g["primal_obj"] and r.dual_obj == g["dual_obj"]


@pytest.mark.parametrize("name", FAST + SLOW)
def test_instances_match_reference_binary_and_core(name):
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
    r = O.oracle_solve(lp)
    g = REF[name]
    # the real cuPDLP-C core (oracle/_ref): bit-for-bit
    assert r.num_iter == g["cupdlp"]["num_iter"]
    assert r.num_trials == g["cupdlp"]["num_trials"]
    assert r.primal_obj == g["cupdlp"]["primal_obj"]
    assert r.dual_obj == g["cupdlp"]["dual_obj"]
    assert r.primal_feas == g["cupdlp"]["primal_feas"]
    obj = lp.objective_value(r.col_value)
    assert obj == g["cupdlp"]["objective_function_value"]
    # the reference binary (highs --solver=pdlp --presolve=off): iterations and printed objective
    if g["highs"]:
        assert r.num_iter == g["highs"]["pdlp_iterations"]
        assert abs(obj - g["highs"]["objective_value"]) <= 1e-9 * max(1.0, abs(obj))
    # ctest's CPU objective prefix (check/CMakeLists.txt:321-335)
    if g["ctest_cpu_prefix"]:
        assert ("%.10e" % obj).replace("e+0", "e").startswith(g["ctest_cpu_prefix"][:6])


def test_hot_start_matches_reference_core():
    # PDHG_PreSolve semantics (check/TestPdlp.cpp:260-284 restart-lp): solve, then re-solve from the solution
    lp = L.special_lps()["restart_lp"]
    r1 = O.oracle_solve(lp, kkt_tolerance=1e-4)
    start = {"col_value": r1.col_value, "row_value": r1.row_value, "row_dual": r1.row_dual}
    r2 = O.oracle_solve(lp, start=start, kkt_tolerance=1e-4)
    assert r2.term_code == 0 and r2.num_iter < r1.num_iter
    if O.ref_available():
        q = O.ref_solve(lp, start=start, kkt_tolerance=1e-4)
        assert q.num_iter == r2.num_iter and q.primal_obj == r2.primal_obj


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("features_off", [0, 1, 2, 4, 7])
def test_oracle_equals_reference_core_feature_switches(features_off):
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", "afiro.npz"))
    a = O.oracle_solve(lp, kkt_tolerance=1e-5, pdlp_features_off=features_off, pdlp_iteration_limit=20000)
    b = O.ref_solve(lp, kkt_tolerance=1e-5, pdlp_features_off=features_off, pdlp_iteration_limit=20000)
    assert (a.term_code, a.num_iter, a.num_trials) == (b.term_code, b.num_iter, b.num_trials)
    assert a.primal_obj == b.primal_obj and a.dual_obj == b.dual_obj
    assert np.array_equal(a.col_value, b.col_value) and np.array_equal(a.row_dual, b.row_dual)


MORE = json.load(open(os.path.join(GOLD, "reference_pdlp_more.json")))


@pytest.mark.parametrize("name", sorted(MORE))
def test_more_instances_match_reference_binary_and_core(name):
    """The other LPs of the reference's check/instances that its CPU pdlp finishes in minutes (five optimal, nine
    primal infeasible or unbounded; make_golden_more.py): the restatement follows the real cuPDLP-C core bit for bit
    and gives the reference binary's iteration count."""
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
    r = O.oracle_solve(lp)
    g = MORE[name]
    for k in ("num_iter", "num_trials", "term_code", "primal_obj", "dual_obj", "primal_feas", "dual_feas"):
        assert getattr(r, k) == g["cupdlp"][k], k
    assert r.num_iter == g["highs"]["pdlp_iterations"]

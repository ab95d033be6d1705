"""The drop-in boundary on the driver's box: the REFERENCE's own unmodified CLI, Catch2 `[pdlp]` unit tests and
a plain C client of its C API, all running on a libhighs.so whose PDLP wrapper TUs are replaced by
integration/*Mi355x.cpp -> libpdlp_mi355x.so (built by `make -C integration` — a CMake-free recipe that compiles the
reference's TUs from where they lie; __graft_entry__.build() runs it wherever the reference tree exists — and
integration/_build travels with the repo snapshot).  A missing build FAILS where the reference tree is present."""
import json
import os
import re
import subprocess

import pytest

from highs_amd import abi, solver
from highs_amd import lp as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "integration", "_build")
GOLD = os.path.join(ROOT, "tests", "golden")
REF = json.load(open(os.path.join(GOLD, "reference_pdlp.json")))
needs_build = pytest.mark.usefixtures("dropin_build")  # tests/conftest.py: FAILS where the reference tree is present and the build is not


def _env(**extra):
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = BUILD + ":" + os.path.join(ROOT, "highs_amd", "lib") + ":" + e.get("LD_LIBRARY_PATH", "")
    e.update(extra)
    return e


def _cli(tmp_path, name, *flags, **env):
    mps = os.path.join(str(tmp_path), name + ".mps")
    L.write_mps(L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz")), mps)
    out = subprocess.run([os.path.join(BUILD, "highs_ref_cli"), *flags, mps], capture_output=True, text=True, timeout=300,
                         env=_env(**env), cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    g = lambda pat: (re.search(pat, txt) or [None, None])[1]
    return dict(rc=out.returncode, text=txt, status=(g(r"Model status\s*:\s*(.+)") or "").strip(),
                iters=int(g(r"PDLP\s+iterations:\s*(\d+)") or -1), objective=float(g(r"Objective value\s*:\s*(\S+)") or "nan"))


@needs_build
@pytest.mark.parametrize("name", ["afiro", "adlittle", "shell", "25fv47"])
def test_reference_cli_pdlp(tmp_path, name):
    r = _cli(tmp_path, name, "--solver=pdlp", "--presolve=off")
    assert r["rc"] == 0 and r["status"] == "Optimal", r["text"][-1500:]
    ref = REF[name]["highs"]["objective_value"]
    assert abs(r["objective"] - ref) <= 1e-6 * (1 + abs(ref)), (r["objective"], ref)
    assert r["iters"] > 0 and "MI355X" in r["text"]  # the library's banner arrives through HiGHS's log


@needs_build
@pytest.mark.parametrize("name", ["afiro", "adlittle"])
def test_reference_cli_hipdlp(tmp_path, name):
    r = _cli(tmp_path, name, "--solver=hipdlp", "--presolve=off")
    assert r["rc"] == 0 and r["status"] == "Optimal", r["text"][-1500:]
    ref = REF[name]["highs"]["objective_value"]
    assert abs(r["objective"] - ref) <= 1e-5 * (1 + abs(ref))


@needs_build
def test_reference_cli_presolve_on(tmp_path):
    r = _cli(tmp_path, "25fv47", "--solver=pdlp")  # HiGHS presolve -> PDLP on the reduced LP -> postsolve
    assert r["rc"] == 0 and r["status"] == "Optimal"
    assert abs(r["objective"] - 5.5018446801e+03) <= 1e-5 * 5.5e3


@needs_build
@pytest.mark.parametrize("case", ["pdlp-distillation-lp", "pdlp-3d-lp", "pdlp-boxed-row-lp", "pdlp-infeasible-lp",
                                  "pdlp-unbounded-lp", "pdlp-restart-lp", "pdlp-restart-add-row",
                                  "test-1966"])  # check/TestIpm.cpp:92: PDLP on a primal- and dual-infeasible LP
def test_reference_catch2_cases(tmp_path, case):
    """check/TestPdlp.cpp, unmodified binary: the asserts on iteration counts (160 / 79) are CPU-build-only
    in the reference (TestPdlp.cpp:98-113 drops them for its CUDA build); everything else must pass."""
    out = subprocess.run([os.path.join(BUILD, "unit_tests_ref"), case], capture_output=True, text=True, timeout=300,
                         env=_env(), cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    assert "All tests passed" in txt or re.search(r"test cases:\s+1\s+\|\s+1 passed", txt), txt[-2000:]


@needs_build
@pytest.mark.parametrize("which", ["pdlp", "hipdlp"])
def test_c_api_client(tmp_path, which):
    """integration/capi_check.c: Highs_create / Highs_passLp / Highs_setStringOptionValue("solver", ...) /
    Highs_run / Highs_getSolution (highs/interfaces/highs_c_api.h) on the distillation LP."""
    out = subprocess.run([os.path.join(BUILD, "capi_check"), which], capture_output=True, text=True, timeout=300,
                         env=_env(), cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-500:]
    assert "model_status=7" in out.stdout


REF_QP = dict(json.load(open(os.path.join(GOLD, "reference_qp.json"))), **json.load(open(os.path.join(GOLD, "reference_qp_sparse.json"))))


@needs_build
@pytest.mark.parametrize("name", ["qjh_mps", "qjh_qmatrix_mps", "qptestnw_lp", "qp3", "qp7", "sq5", "sq100"])
def test_qp_through_highs_run_with_the_gate_lifted(tmp_path, name):
    """SURVEY §8(f)-3 through the reference's own entry point: libhighs_qp.so.1 = the drop-in libhighs with the two gate
    TUs (lp_data/HighsOptions.cpp:1178-1181 solverValidForQp, lp_data/Highs.cpp:4139 callSolveQp) replaced by build-time
    copies carrying two edits (integration/qp_gate_patch.py).  A plain C client (capi_check.c: Highs_readModel,
    solver=pdlp, Highs_run) then solves the reference's own QP instances and random convex QPs on the MI355X path and
    must reach the optimum of the reference's QP solver (tests/golden/reference_qp*.json) — check/TestQpSolver.cpp's
    objective asserts, at 1e-6 relative."""
    assert os.path.exists(os.path.join(BUILD, "capi_check_qp")), "make -C integration (libhighs_qp.so.1, capi_check_qp)"
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "qp", name + ".npz"))
    mps = os.path.join(str(tmp_path), name + ".mps")
    L.write_mps(lp, mps)
    out = subprocess.run([os.path.join(BUILD, "capi_check_qp"), "pdlp", mps, "1e-8"], capture_output=True, text=True, timeout=300,
                         env=_env(), cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    m = re.search(r"capi_check: .*model_status=(\d+) objective=(\S+) pdlp_iteration_count=(-?\d+) qp_iteration_count=(-?\d+) hessian_nz=(\d+)", txt)
    assert m, txt[-2000:]
    status, obj, it, qit, hnz = int(m[1]), float(m[2]), int(m[3]), int(m[4]), int(m[5])
    ref = REF_QP[name]["objective_value"]
    assert "MI355X" in txt and "Quadratic objective" in txt  # the library's banner: the QP went down the PDLP path ...
    assert it > 0 and qit <= 0 and hnz > 0, txt[-3000:]       # ... not to the active-set solver
    assert status == 7, txt[-1500:]                           # kHighsModelStatusOptimal
    assert abs(obj - ref) <= 1e-6 * (1 + abs(ref)), (obj, ref)


@needs_build
def test_unpatched_libhighs_keeps_the_reference_gate(tmp_path):
    """The plain drop-in (libhighs.so.1) leaves the gate where the reference has it: solver=pdlp on a QP warns and runs the
    reference's own QP solver (lp_data/Highs.cpp:1367)."""
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "qp", "qjh_mps.npz"))
    mps = os.path.join(str(tmp_path), "qjh.mps")
    L.write_mps(lp, mps)
    out = subprocess.run([os.path.join(BUILD, "capi_check"), "pdlp", mps], capture_output=True, text=True, timeout=300, env=_env(),
                         cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    assert "not available for QP" in txt and re.search(r"qp_iteration_count=[1-9]", txt), txt[-1500:]
    assert abs(float(re.search(r"objective=(\S+)", txt)[1]) + 5.25) < 1e-6


@needs_build
def test_reference_cli_sharded_over_two_ranks(tmp_path):
    """Multi-GPU BEHIND the boundary: PDLP_MI355X_DEVICES=2 makes pdlp_mi355x_solve (what Highs::run reaches)
    shard the LP over two device ranks of the one process (host thread per rank, peer-access mesh).  On a
    1-GPU box both ranks fold onto device 0."""
    one = _cli(tmp_path, "adlittle", "--solver=pdlp", "--presolve=off")
    two = _cli(tmp_path, "adlittle", "--solver=pdlp", "--presolve=off", PDLP_MI355X_DEVICES="2",
               PDLP_MI355X_FOLD_DEVICES="1", PDLP_MI355X_VERIFY_RANKS="1")
    assert two["rc"] == 0 and two["status"] == "Optimal", two["text"][-1500:]
    assert "Row-block sharded over 2 GPUs" in two["text"]
    assert abs(two["objective"] - one["objective"]) <= 1e-6 * (1 + abs(one["objective"]))


_SHARD_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from highs_amd import solver, lp as L
lp = L.HighsLp.from_npz(os.path.join({root!r}, "tests", "golden", "instances", {inst!r}))
fn = solver.solveLpCupdlp if {which!r} == "pdlp" else solver.solveLpHiPdlp
one = fn(lp, kkt_tolerance=1e-6)
many = fn(lp, kkt_tolerance=1e-6, num_devices={ranks})
assert many.model_status == solver.kOptimal == one.model_status, (many.model_status, solver.lib().pdlp_mi355x_last_error())
a, b = lp.objective_value(one.solution.col_value), lp.objective_value(many.solution.col_value)
assert abs(a - b) <= 1e-5 * (1 + abs(a)), (a, b)  # two trajectories, each converged to a 1e-6 relative gap
print("sharded ok", a, b)
"""


@pytest.mark.parametrize("ranks", [2])
@pytest.mark.parametrize("which", ["pdlp", "hipdlp"])
def test_one_call_boundary_shards_in_process(ranks, which):
    """pdlp_mi355x_solve with num_devices = G: bit-identical results on every rank (checked inside the call,
    PDLP_MI355X_VERIFY_RANKS) and the single-device optimum.  On this 1-GPU box the G ranks are folded onto
    device 0; their G streams must then not share a hardware queue (a kernel spinning on a peer's flag would
    block the peer's own kernel behind it), hence GPU_MAX_HW_QUEUES and only two ranks here — with one device
    per rank, the real case, every rank has its device's queues to itself; 4 and 8 ranks on one device are
    covered with one PROCESS per rank in tests/test_gpu_mesh.py."""
    import sys
    env = dict(os.environ, PDLP_MI355X_FOLD_DEVICES="1", PDLP_MI355X_VERIFY_RANKS="1", GPU_MAX_HW_QUEUES="16")
    code = _SHARD_SCRIPT.format(root=ROOT, inst="e226.npz" if which == "pdlp" else "adlittle.npz", which=which, ranks=ranks)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "sharded ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_num_devices_beyond_the_visible_devices_is_an_error(monkeypatch):
    monkeypatch.delenv("PDLP_MI355X_FOLD_DEVICES", raising=False)
    lp = L.special_lps()["distillation"]
    P, R = abi.ProblemHandle(lp), abi.ResultHandle(lp.num_col, lp.num_row)
    import ctypes as C
    rc = solver.lib().pdlp_mi355x_solve(C.byref(P.struct), C.byref(abi.default_params(num_devices=64)), C.byref(R.struct))
    assert rc != 0 and b"devices" in solver.lib().pdlp_mi355x_last_error()


@pytest.mark.parametrize("name", ["afiro", "adlittle"])
def test_model_file_to_solution_without_highs(tmp_path, name):
    """MPS file -> pdlp_mi355x_read_mps -> pdlp_mi355x_solve (solver.run_model_file): the reference objective."""
    mps = os.path.join(str(tmp_path), name + ".mps")
    L.write_mps(L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz")), mps)
    out, lp, info = solver.run_model_file(mps, threads=3)
    ref = REF[name]["highs"]["objective_value"]
    assert out.model_status == solver.kOptimal and info["threads"] == 3
    assert abs(out.info["objective_function_value"] - ref) <= 1e-6 * (1 + abs(ref))

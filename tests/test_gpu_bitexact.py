"""Whole-solve bit-exactness.  The oracle has two summation modes: the reference's left-to-right
loops (pinned bit for bit to the real cuPDLP-C core) and "device reduction order", which restates
the lane/wave/block order of the HIP kernels' reductions and nothing else.  In that mode a complete
GPU solve — every iterate, every accept/reject decision, every restart — must be reproduced BIT FOR
BIT: the only arithmetic difference between this library and the reference is the order in which
five kinds of sums are added up."""
import os

import numpy as np
import pytest

import oraclelib as O
from highs_amd import abi, solver
from highs_amd import lp as L
from lpgen import random_lp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check(lp, layout="csr", **kw):
    cpu = O.oracle_solve(lp, device_reduction_order=True, device_layout=layout, **kw)
    gpu = solver.solveLpCupdlp(lp, **kw)
    R = gpu.result
    assert (R.term_code, R.term_iterate, R.num_iter, R.num_trials, R.num_restarts) == \
           (cpu.term_code, cpu.term_iterate, cpu.num_iter, cpu.num_trials, cpu.num_restarts)
    assert R.primal_obj == cpu.primal_obj and R.dual_obj == cpu.dual_obj
    assert R.primal_feas == cpu.primal_feas and R.dual_feas == cpu.dual_feas and R.rel_gap == cpu.rel_gap
    assert np.array_equal(gpu.solution.col_value, cpu.col_value)
    assert np.array_equal(gpu.solution.row_dual, cpu.row_dual)
    assert np.array_equal(gpu.solution.col_dual, cpu.col_dual)
    assert np.array_equal(gpu.solution.row_value, cpu.row_value)
    return R.num_iter


@pytest.fixture(autouse=True)
def _csr_layout(monkeypatch):
    monkeypatch.setenv("PDLP_MI355X_SLAB", "0")  # default for this module: the CSR-stream work plan


@pytest.mark.parametrize("name", ["afiro", "adlittle", "sctest", "e226", "shell", "25fv47",
                                  # LPs outside the ctest list: optimal ones and primal infeasible / unbounded ones
                                  "qap04", "standmps", "israel", "woodinfe", "box1", "bgetam", "galenet",
                                  # majors longer than a 512-entry work block: segment tasks inside the persistent loop
                                  "standata", "standgub", "cplex1"])
def test_instances_bit_exact(name):
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
    assert _check(lp) > 0


@pytest.mark.parametrize("name", sorted(L.special_lps()))
def test_special_lps_bit_exact(name):
    _check(L.special_lps()[name], kkt_tolerance=1e-4)


@pytest.mark.parametrize("seed", range(8))
def test_random_lps_bit_exact(seed):
    _check(random_lp(seed), kkt_tolerance=1e-6, pdlp_iteration_limit=200000)


@pytest.mark.parametrize("features_off", [1, 2, 4, 7])
def test_feature_switches_bit_exact(features_off):
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", "afiro.npz"))
    _check(lp, kkt_tolerance=1e-5, pdlp_features_off=features_off, pdlp_iteration_limit=50000)


def test_hot_start_bit_exact():
    lp = L.special_lps()["restart_lp"]
    a = solver.solveLpCupdlp(lp, kkt_tolerance=1e-4)
    start = {"col_value": a.solution.col_value, "row_value": a.solution.row_value, "row_dual": a.solution.row_dual}
    cpu = O.oracle_solve(lp, start=start, device_reduction_order=True, kkt_tolerance=1e-4)
    gpu = solver.solveLpCupdlp(lp, start=start, kkt_tolerance=1e-4)
    assert gpu.result.num_iter == cpu.num_iter and np.array_equal(gpu.solution.col_value, cpu.col_value)


def test_synthetic_20k_bit_exact():
    sp_ = solver.SyntheticProblem(20000, 20000, 160000, 1)
    _check(sp_.to_lp(), kkt_tolerance=1e-4)


@pytest.mark.parametrize("name", ["afiro", "e226", "25fv47"])
def test_slab_layout_bit_exact(name, monkeypatch):
    """Same check with the row-block x column-slab SpMV layout forced on both sides (25fv47 has majors
    longer than 256 entries, which go through the CSR side kernel)."""
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1")
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
    _check(lp, layout="slab")


@pytest.mark.parametrize("layout", ["csr", "slab"])
def test_dense_column_lp_bit_exact(layout, monkeypatch):
    """tests/lpgen.py::dense_column_lp in small: dense columns (segment tasks in A'y — inside the persistent loop in the
    stream layout, inside the streaming blocks of the fused trial in the slab layout), long rows, every row kind."""
    from lpgen import dense_column_lp
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1" if layout == "slab" else "0")
    lp = dense_column_lp(2, periods=12, rows_per=256, cols_per=224, dense_cols=6, dense_nnz=1500, tail_rows=64, tail_max=900)
    assert _check(lp, layout=layout, kkt_tolerance=1e-5, pdlp_iteration_limit=30000) > 0


@pytest.mark.parametrize("family", ["tall", "plband"])
@pytest.mark.parametrize("layout", ["csr", "slab"])
def test_held_out_families_bit_exact(family, layout, monkeypatch):
    """The two HELD-OUT structured families of round 6 in small (tests/lpgen.py tall_lp: m >> n with dense coupling rows;
    powerlaw_band_lp: power-law row and column lengths, hub columns that are segment tasks of A'y): whole solves bit for
    bit in both layouts — the XCD-affine task deal, the work-balanced partition and the co-resident task workgroups see
    shapes they were not developed on."""
    from lpgen import powerlaw_band_lp, tall_lp
    monkeypatch.setenv("PDLP_MI355X_SLAB", "1" if layout == "slab" else "0")
    lp = tall_lp(2, n=3000, m=20000, window=256, dense_rows=6, dense_nnz=1500) if family == "tall" else \
        powerlaw_band_lp(2, n=12000, m=10000, band=512, hubs=60)
    assert _check(lp, layout=layout, kkt_tolerance=1e-5, pdlp_iteration_limit=4000) > 0


@pytest.mark.parametrize("family", ["tall", "plband"])
def test_held_out_families_at_bench_size_first_iterations(family, monkeypatch):
    """... and at the size bench.py --config e / f runs them (device-side set-up, automatic layouts: tall = CSR stream for
    A x with 96 long rows + slab for A'y; plband = slab for both with hub columns up to 93k entries): the first 60
    iterations, checks included, bit for bit against the oracle's device-order mode."""
    from lpgen import powerlaw_band_lp, tall_lp
    monkeypatch.delenv("PDLP_MI355X_SLAB", raising=False)
    lp = tall_lp(1) if family == "tall" else powerlaw_band_lp(1)
    assert _check(lp, layout="auto", kkt_tolerance=1e-4, pdlp_iteration_limit=60) == 59


def test_bench_workload_bit_exact_first_iterations(monkeypatch):
    """The bench workload itself (1M x 1M, 8M nnz; automatic layout = slab, device-side setup): the first
    60 iterations — checks, restarts and rejected trials included — reproduced bit for bit."""
    monkeypatch.delenv("PDLP_MI355X_SLAB", raising=False)
    sp_ = solver.SyntheticProblem(1000000, 1000000, 8000000, 1)
    lp = sp_.to_lp()
    n_iter = _check(lp, layout="auto", kkt_tolerance=1e-4, pdlp_iteration_limit=60)
    assert n_iter == 59

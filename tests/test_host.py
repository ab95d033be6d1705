"""CPU tests of the PRODUCT's host logic: the C-ABI library loads, exports every
declared symbol, and its formulate/scale/transpose agree bit for bit with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oraclelib as O
from highs_amd import abi, solver
from highs_amd import lp as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["25fv47", "adlittle", "afiro", "avgas", "blending", "chip", "e226", "scrs8", "sctest", "shell", "stair",
         "standata", "standgub"]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pdlp_mi355x.h")).read()
    declared = set(re.findall(r"\b(pdlp_mi355x_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = solver.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/pdlp_mi355x.h but not exported"
    assert set(solver.EXPORTS) == declared
    assert lib.pdlp_mi355x_abi_version() == int(re.search(r"#define PDLP_MI355X_ABI_VERSION (\d+)", hdr).group(1)) == 6


def test_development_switches_need_the_master_switch():
    """A development variable in the environment changes nothing unless PDLP_MI355X_DEV=1 (highs_amd/csrc/pdlp_env.hpp): the
    library says once that it ignores it.  Checked on a switch that needs no GPU (the per-phase timing of the MPS reader)."""
    import subprocess
    import sys
    mps = os.path.join(ROOT, "tests", "golden", "mps_cases", "ranges.mps")
    if not os.path.exists(mps):
        mps = sorted(__import__("glob").glob(os.path.join(ROOT, "tests", "golden", "mps_cases", "*.mps")))[0]
    code = ("import sys; sys.path.insert(0, %r); from highs_amd import solver; solver.read_mps(%r)" % (ROOT, mps))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PDLP_MI355X_")}
    env["PDLP_MI355X_MPS_TIMING"] = "1"
    off = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert off.returncode == 0, off.stderr
    assert "PDLP_MI355X_MPS_TIMING is a development switch and is ignored" in off.stderr
    on = subprocess.run([sys.executable, "-c", code], env=dict(env, PDLP_MI355X_DEV="1"), capture_output=True, text=True, timeout=120)
    assert on.returncode == 0, on.stderr
    assert "development switch" not in on.stderr and on.stderr.strip() != ""  # the timing lines instead


def test_struct_sizes_match_ctypes_mirror():
    lib = solver.lib()
    for which, ty in enumerate([abi.PdlpProblem, abi.PdlpParams, abi.PdlpResult, abi.PdlpIterStats, abi.PdlpPrepared,
                              abi.PdlpSlabLayout, abi.PdlpMpsModel, abi.PdlpTaskPlan]):
        assert lib.pdlp_mi355x_sizeof(which) == C.sizeof(ty), ty.__name__


def test_default_params_match_highs_defaults():
    p = abi.PdlpParams()
    solver.lib().pdlp_mi355x_default_params(C.byref(p))
    q = abi.default_params()
    for k in ("primal_tol", "dual_tol", "gap_tol", "time_limit", "iter_limit", "features_off", "restart_method",
              "log_level"):
        assert getattr(p, k) == getattr(q, k), k


def _same(P, F):
    for k in ["csr_beg", "csr_idx", "csr_val", "cost", "rhs", "lower", "upper", "col_scale", "row_scale"]:
        assert np.array_equal(getattr(P, k), getattr(F, k)), k
    assert (P.n, P.m, P.n_eqs, P.nnz) == (F.n, F.m, F.n_eqs, F.nnz)
    assert P.norm_cost == F.norm_cost and P.norm_rhs == F.norm_rhs and P.mat_norm_inf == F.mat_norm_inf
    assert np.array_equal(P.row_new_idx, F.row_new_idx) and np.array_equal(P.row_kind, F.row_type)


@pytest.mark.parametrize("name", NAMES)
def test_formulate_scale_bit_exact_vs_oracle(name):
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
    _same(solver.Prepared(lp), O.FormulatedView(lp))


@pytest.mark.parametrize("name", sorted(L.special_lps()))
@pytest.mark.parametrize("features_off", [0, 1])
def test_formulate_special_lps(name, features_off):
    lp = L.special_lps()[name]
    _same(solver.Prepared(lp, pdlp_features_off=features_off), O.FormulatedView(lp, pdlp_features_off=features_off))


def test_boxed_and_free_rows_get_slack_columns():
    # ranged AND free rows -> a'x - z = 0 (CupdlpWrapper.cpp:327-343)
    inf = float("inf")
    lp = L.HighsLp(2, 3, np.array([1.0, 1.0]), np.zeros(2), np.full(2, inf), np.array([-inf, 1.0, -2.0]),
                   np.array([inf, 4.0, inf]), np.array([0, 3, 6], np.int32), np.array([0, 1, 2, 0, 1, 2], np.int32),
                   np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])).normalise()
    P = solver.Prepared(lp, pdlp_features_off=1)
    assert (P.n, P.m, P.n_eqs, P.nnz) == (4, 3, 2, 8)
    assert list(P.row_kind) == [3, 3, 2] and list(P.row_new_idx) == [0, 1, 2]
    assert P.lower[2] == -inf and P.upper[2] == inf and P.lower[3] == 1.0 and P.upper[3] == 4.0
    _same(P, O.FormulatedView(lp, pdlp_features_off=1))


def test_csc_is_transpose_of_csr_with_sorted_rows():
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", "e226.npz"))
    P = solver.Prepared(lp)
    import scipy.sparse as sp
    A = sp.csr_matrix((P.csr_val, P.csr_idx, P.csr_beg), shape=(P.m, P.n))
    At = sp.csr_matrix((P.csc_val, P.csc_idx, P.csc_beg), shape=(P.n, P.m))
    assert (A.T != At).nnz == 0
    for j in range(P.n):
        seg = P.csc_idx[P.csc_beg[j]:P.csc_beg[j + 1]]
        assert np.all(np.diff(seg) > 0)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_row_partition_covers_and_balances(world):
    sp_ = solver.SyntheticProblem(20000, 20000, 160000, 3)
    P = solver.Prepared(problem_struct=sp_.struct)
    off = P.row_partition(world)
    assert off[0] == 0 and off[-1] == P.m and np.all(np.diff(off) >= 0)
    nnz_per = np.diff(P.csr_beg[off])
    assert nnz_per.max() <= 1.1 * P.nnz / world + 64


def test_synthetic_generator_matches_survey_counts():
    # 100k x 100k, 1M draws -> 999 945 nonzeros after merging duplicates (seed 1)
    sp_ = solver.SyntheticProblem(100000, 100000, 1000000, 1)
    assert sp_.struct.num_nz == 999945
    lp = sp_.to_lp()
    assert np.all(lp.col_lower == 0) and np.all(lp.col_upper == 1)
    eq = lp.row_lower == lp.row_upper
    assert eq[0::2].all() and not eq[1::2].any() and np.isneginf(lp.row_lower[1::2]).all()


def test_bad_input_is_reported_not_fatal():
    lp = L.special_lps()["distillation"]
    h = abi.ProblemHandle(lp)
    h.a_index[0] = 99  # row index out of range
    F = abi.PdlpPrepared()
    p = abi.default_params()
    rc = solver.lib().pdlp_mi355x_host_prepare(C.byref(h.struct), C.byref(p), C.byref(F))
    assert rc != 0 and b"out of range" in solver.lib().pdlp_mi355x_last_error()


@pytest.mark.skipif(not os.path.isdir("/root/reference/check/instances"), reason="reference tree not present")
def test_mps_reader_reproduces_golden_npz():
    for name in ["afiro", "25fv47", "scrs8"]:
        a = L.read_mps(f"/root/reference/check/instances/{name}.mps")
        b = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
        for k in ["col_cost", "col_lower", "col_upper", "row_lower", "row_upper", "a_start", "a_index", "a_value"]:
            assert np.array_equal(getattr(a, k), getattr(b, k)), (name, k)


def test_solver_mirror_runs_against_oracle():
    # same marshalling as the GPU path, executed against the oracle: exercises solveLpCupdlp's status map
    lp = L.special_lps()["distillation"]
    out = solver.solveLpCupdlp(lp, solve_fn=O.oracle().pdlp_oracle_solve, kkt_tolerance=1e-4, pdlp_iteration_limit=80)
    assert out.model_status == solver.kIterationLimit and out.pdlp_iteration_count == 79 and out.status == solver.kWarning
    out = solver.solveLpCupdlp(L.special_lps()["infeasible"], solve_fn=O.oracle().pdlp_oracle_solve, kkt_tolerance=1e-4)
    assert out.model_status == solver.kUnboundedOrInfeasible and out.status == solver.kOk


def _slab_to_coo(sl, n_major):
    """Rebuild (major, minor, value) triplets from the slab layout, in storage order."""
    mb = sl["minor_bits"]
    wp, wb = sl["wave_ptr"], sl["wave_beg"]
    ent, val = sl["ent"].astype(np.int64), sl["val"]
    wave = np.repeat(np.arange(len(wp) - 1), np.diff(wp))
    majors = wb[wave] + (ent >> mb)
    minors = ent & ((1 << mb) - 1)
    return majors, minors, val, wave


def _slab_work(lens, long_limit, major_cost=2, cold=0):
    """pdlp_host.cpp slabMajorWork: entries + 3 x cold entries (they count four times) + the run-accumulation term + major_cost (2 for the operand by
    rows, 10 for the transposed one); a long major: major_cost alone."""
    lens = np.asarray(lens, dtype=np.int64)
    return np.where(lens > long_limit, major_cost, lens + 3 * cold + (lens * np.minimum(lens, 64)) // 32 + major_cost)


def _cold_counts(beg, idx, n_minor, long_limit):
    """pdlp_host.cpp slabColdCounts: entries 2^17 or more minors away from the major's middle entry, in a minor that at
    most 64 majors touch."""
    beg = np.asarray(beg, dtype=np.int64)
    idx = np.asarray(idx, dtype=np.int64)
    lens = np.diff(beg)
    count = np.bincount(idx, minlength=max(n_minor, 1))
    rows = np.repeat(np.arange(len(lens)), lens)
    mid = idx[np.minimum(beg[:-1] + lens // 2, max(len(idx) - 1, 0))] if len(idx) else np.zeros(len(lens), np.int64)
    is_cold = (np.abs(idx - mid[rows]) >= (1 << 17)) & (count[idx] <= 64) & ((lens >= 2) & (lens <= long_limit))[rows]
    return np.bincount(rows[is_cold], minlength=len(lens)).astype(np.int64)


def _oracle_blocks(beg, idx, n_major, n_minor, which, long_limit=256):
    """The slab block boundaries as the oracle's device-order mode models them (oracle/gpu_order.h, in C)."""
    ob = np.zeros(256 + n_major // 16 + 3, dtype=np.int32)
    b32 = np.ascontiguousarray(beg, dtype=np.int32)
    i32 = np.ascontiguousarray(idx if len(idx) else [0], dtype=np.int32)
    nb = O.oracle().pdlp_oracle_slab_blocks(n_major, n_minor, b32.ctypes.data_as(abi.c_i32p), i32.ctypes.data_as(abi.c_i32p), long_limit,
                                            which, ob.ctypes.data_as(abi.c_i32p))
    return ob[:nb + 1].astype(np.int64)


def _slab_partition_restated(beg, n_major, n_minor, long_limit, major_cost=2, cold=0):
    """pdlp_host.cpp slabPartition, restated: blocks, then the 16 waves of every block, filled one after the other by
    work = _slab_work."""
    lens = np.diff(beg)
    cost = _slab_work(lens, long_limit, major_cost, cold)
    mb = max(int(np.ceil(np.log2(max(n_minor, 1)))), 4)
    wave_cap = min(1 << (32 - mb), 16384)
    block_cap = min(16384, wave_cap * 16)
    nb = max(min(-(-n_major // 256), 256), -(-n_major // block_cap))

    def fill(r0, r1, units, cap, non_empty):
        out, r, rem = [r0], r0, int(cost[r0:r1].sum())
        for u in range(units):
            left = units - u
            target = -(-rem // left)
            rows = r1 - r
            min_rows = max(1 if non_empty and rows > 0 else 0, rows - (left - 1) * cap)
            max_rows = min(cap, max(rows - (left - 1), 1) if non_empty else rows)
            acc = cnt = 0
            while cnt < rows and cnt < max_rows:
                if cnt >= min_rows and 2 * acc + int(cost[r]) > 2 * target:
                    break
                acc += int(cost[r]); r += 1; cnt += 1
            rem -= acc
            out.append(r)
        return out
    bb = fill(0, n_major, nb, block_cap, True)
    wb = [0]
    for b in range(nb):
        wb += fill(bb[b], bb[b + 1], 16, wave_cap, False)[1:]
    return nb, mb, np.array(wb, dtype=np.int64)


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("long_limit", [256, 6])
def test_slab_layout_is_a_permutation_of_the_csr(which, long_limit):
    """Slab layout + long-major side list hold exactly the CSR's nonzeros; every major's entries stay
    in ascending minor order when read in storage order (=> same summation order as the reference)."""
    sp_ = solver.SyntheticProblem(3000, 200000, 24000, 5)  # wide: 4 slabs of 65536 columns for A
    P = solver.Prepared(problem_struct=sp_.struct, slab_long_limit=long_limit)
    sl = P.slab_layout(which)
    beg, idx, val = (P.csr_beg, P.csr_idx, P.csr_val) if which == 0 else (P.csc_beg, P.csc_idx, P.csc_val)
    n_major = P.m if which == 0 else P.n
    lens = np.diff(beg)
    long_rows = np.nonzero(lens > long_limit)[0]
    assert np.array_equal(sl["long_map"], long_rows)
    mask_rows = [r for r in range(n_major) if (sl["long_mask"][r >> 5] >> (r & 31)) & 1]
    assert mask_rows == list(long_rows)
    maj, mnr, v, wave = _slab_to_coo(sl, n_major)
    short = np.ones(n_major, bool)
    short[long_rows] = False
    rows_csr = np.repeat(np.arange(n_major), lens)
    keep = short[rows_csr]
    # same multiset of triplets
    a = np.lexsort((mnr, maj))
    assert np.array_equal(maj[a], rows_csr[keep]) and np.array_equal(mnr[a], idx[keep]) and np.array_equal(v[a], val[keep])
    # storage order: within a wave sorted by (slab, major, minor); so per major minors ascend
    cand = np.nonzero(short & (lens > 1))[0]
    for r in (np.random.default_rng(0).choice(cand, size=min(50, len(cand)), replace=False) if len(cand) else []):
        pos = np.nonzero(maj == r)[0]
        assert np.all(np.diff(pos) > 0) and np.all(np.diff(mnr[pos]) > 0)
    W, wb = sl["slab_width_log2"], sl["wave_beg"]
    assert (1 << sl["minor_bits"]) >= (P.n if which == 0 else P.m)
    assert wb[0] == 0 and wb[-1] == n_major and np.all(np.diff(wb) >= 0) and np.all(np.diff(wb) <= 1 << (32 - sl["minor_bits"]))
    blk = wb[::16]
    assert np.all(np.diff(blk) >= 1) and np.max(np.diff(blk)) == sl["rows_per_block"] <= 16384
    # the partition is the restated rule, and it balances work: no block above the mean by more than one major's worth
    cold = _cold_counts(beg, idx, P.n if which == 0 else P.m, long_limit)
    nb, mb, wb2 = _slab_partition_restated(beg, n_major, P.n if which == 0 else P.m, long_limit, 10 if which else 2, cold)
    assert nb == sl["n_blocks"] and mb == sl["minor_bits"] and np.array_equal(wb2, wb)
    assert np.array_equal(_oracle_blocks(beg, idx, n_major, P.n if which == 0 else P.m, which, long_limit), blk)  # the oracle's C restatement too
    cost = _slab_work(lens, long_limit, 10 if which else 2, cold)
    work = np.add.reduceat(cost, blk[:-1])
    assert work.max() <= work.mean() + cost.max()
    # a wave's entries are those of its majors
    assert np.all((maj >= wb[wave]) & (maj < wb[wave + 1]))
    assert np.all(np.diff(wave) >= 0)
    # inside a wave the key (slab, local major, minor) ascends
    key = (wave << 52) | ((mnr >> W) << 40) | ((maj - wb[wave]) << 28) | (mnr & ((1 << W) - 1))
    assert np.all(np.diff(key) > 0)


def test_slab_partition_counts_cold_entries_four_times():
    """Rows of random columns at the end of a banded matrix (the tail of bench.py --config c in small): their entries are
    2^17 or more columns away from the row's middle entry, in columns almost nobody else touches — cold gathers — and count
    four times; entries just as far away but in columns that many rows touch (a dense column) do not.  The product's partition is
    the restated rule's, on both operands."""
    rng = np.random.default_rng(5)
    n, m_band, m_tail = 600000, 6000, 300
    rows, cols = [], []
    for i in range(m_band):
        c0 = int(i * (n - 64) / m_band)
        cc = sorted(rng.choice(np.arange(c0 + 1, c0 + 64), size=6, replace=False))
        if i % 10 == 0:
            cc = sorted(set(cc) | {0 if c0 > n // 2 else n - 1})  # a far entry in one of two DENSE columns: hot
        cols += cc; rows += [i] * len(cc)
    for i in range(m_band, m_band + m_tail):
        cols += sorted(rng.choice(n, size=10, replace=False)); rows += [i] * 10
    m = m_band + m_tail
    r_start = np.searchsorted(np.array(rows), np.arange(m + 1))
    inf = float("inf")
    lp = L.HighsLp.from_rowwise(n, m, r_start, cols, rng.standard_normal(len(cols)), col_cost=np.ones(n), col_lower=np.zeros(n),
                                col_upper=np.ones(n), row_lower=np.full(m, -inf), row_upper=np.ones(m))
    P = solver.Prepared(lp, pdlp_features_off=1)
    for which in (0, 1):
        beg, idx = (P.csr_beg, P.csr_idx) if which == 0 else (P.csc_beg, P.csc_idx)
        n_major, n_minor = (P.m, P.n) if which == 0 else (P.n, P.m)
        sl = P.slab_layout(which)
        cold = _cold_counts(beg, idx, n_minor, 256)
        if which == 0:
            assert cold[:m_band].sum() == 0 and 1000 < cold[m_band:].sum() <= 3000  # the tail's entries, not the dense columns'
        nb, mb, wb = _slab_partition_restated(beg, n_major, n_minor, 256, 10 if which else 2, cold)
        assert nb == sl["n_blocks"] and np.array_equal(wb, sl["wave_beg"])
        assert np.array_equal(_oracle_blocks(beg, idx, n_major, n_minor, which), wb[::16])
        if which == 0:
            assert not np.array_equal(_slab_partition_restated(beg, n_major, n_minor, 256, 2, 0)[2], wb)  # the rule moved boundaries


@pytest.mark.parametrize("shape", [(40, 300, 900, 1), (300, 40, 900, 2), (5000, 70000, 30000, 3), (70000, 5000, 140000, 4),
                                   (1200, 1200, 200000, 5)])
def test_slab_partition_edge_shapes(shape):
    """Few majors (one or two blocks, empty waves), far more majors than entries (empty majors, blocks at their minimum
    size), dense-ish operands (every major long enough for the run term to matter): the product's partition is the restated
    rule's on both operands, covers every major exactly once, and no wave exceeds the local-major field of an entry."""
    m, n, nnz, seed = shape
    sp_ = solver.SyntheticProblem(m, n, nnz, seed)
    P = solver.Prepared(problem_struct=sp_.struct)
    for which in (0, 1):
        beg, idx = (P.csr_beg, P.csr_idx) if which == 0 else (P.csc_beg, P.csc_idx)
        n_major, n_minor = (P.m, P.n) if which == 0 else (P.n, P.m)
        sl = P.slab_layout(which)
        wb = sl["wave_beg"]
        cold = _cold_counts(beg, idx, n_minor, 256)
        nb, mb, wb2 = _slab_partition_restated(beg, n_major, n_minor, 256, 10 if which else 2, cold)
        assert nb == sl["n_blocks"] == max(1, min(256, -(-n_major // 256))) and mb == sl["minor_bits"]
        assert np.array_equal(wb, wb2)
        assert wb[0] == 0 and wb[-1] == n_major and np.all(np.diff(wb) >= 0) and np.all(np.diff(wb[::16]) >= 1)
        assert np.max(np.diff(wb)) <= 1 << (32 - mb) and np.max(np.diff(wb[::16])) == sl["rows_per_block"]
        # ... and the ORACLE's restatement in C (oracle/gpu_order.h, what its device-order mode sums by) gives the same blocks
        assert np.array_equal(_oracle_blocks(beg, idx, n_major, n_minor, which), wb[::16])
        # the entries of every wave's list are those of its majors (long majors left out)
        lens = np.diff(beg)
        short = lens <= 256
        per_wave = np.array([lens[wb[w]:wb[w + 1]][short[wb[w]:wb[w + 1]]].sum() for w in range(len(wb) - 1)])
        assert np.array_equal(per_wave, np.diff(sl["wave_ptr"]))
    sp_.close()


def test_slab_partition_balances_skewed_majors():
    """Power-law major lengths (the staircase LP of bench.py --config d): blocks of equal major COUNT would differ by 2x
    in entries; blocks cut by work do not."""
    rng = np.random.default_rng(3)
    n_major = 120000
    lens = np.minimum((rng.pareto(1.2, n_major) * 3 + 1).astype(np.int64), 600)
    lens[:20000] = 1  # a stretch of very short majors: these blocks hit no cap, they just own more majors
    beg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nb, mb, wb = _slab_partition_restated(beg, n_major, 300000, 256)
    cost = _slab_work(lens, 256)
    blk = wb[::16]
    work = np.add.reduceat(cost, blk[:-1])
    assert nb == 256 and work.max() <= 1.02 * work.mean() + cost.max()
    equal_count = np.add.reduceat(cost, np.arange(0, n_major, -(-n_major // 256)))
    assert equal_count.max() > 1.3 * equal_count.mean()  # what the old partition did on this operand
    for b in range(0, nb, 37):  # waves of a block carry equal work too
        w = wb[16 * b:16 * b + 17]
        ww = np.array([cost[w[k]:w[k + 1]].sum() for k in range(16)])
        assert ww.max() <= ww.mean() + cost.max()


def test_segment_tasks_are_dealt_to_the_xcd_their_entries_live_in():
    """The segment tasks of a slab operand's long majors (block-angular LP with dense linking rows — bench.py --config c in
    small): every segment of every long major exactly once, segment sums in slots of their own, whole task workgroups,
    and — the point of the deal — a task sits in a workgroup of the XCD whose streaming blocks gather from the stretch of
    the vector its entries lie in, no XCD carrying more than its share of the workgroups."""
    from lpgen import structured_lp
    lp = structured_lp(3, commodities=32, nodes=512, arcs=4096, link_rows=48, link_nnz=4096, extra_rows=64)  # 8 segments per linking row
    P = solver.Prepared(lp)
    for balance in (1, 0):
        T = P.task_plan(0, balance)
        tk, g, nb = T["tasks"], T["task_group"], T["n_blocks"]
        lb, li = T["long_beg"], T["long_idx"]
        assert T["n_long"] == 48 and T["n_tasks"] % g == 0 and 1 <= g <= 16
        real = tk[tk[:, 2] >= 0]
        assert len(real) > T["n_tasks"] - g  # one workgroup is not full at most
        slots = real[:, 3] + real[:, 7]
        assert len(set(slots.tolist())) == len(slots) and slots.max() < T["n_seg_slots"] == len(real)
        for c in range(T["n_long"]):
            mine = real[real[:, 2] == c]
            mine = mine[np.argsort(mine[:, 7])]
            n_seg = -(-(lb[c + 1] - lb[c]) // 512)
            assert len(mine) == n_seg and np.array_equal(mine[:, 7], np.arange(n_seg)) and np.all(mine[:, 4] == n_seg)
            assert mine[0, 0] == lb[c] and mine[-1, 1] == lb[c + 1] and np.array_equal(mine[1:, 0], mine[:-1, 1])
            assert np.all(mine[:, 6] == (1 if n_seg == 1 else 0)) and len(set(mine[:, 3].tolist())) == 1
        # XCD of a task's workgroup vs the owner of the tile its middle entry lies in
        grp = np.nonzero(tk[:, 2] >= 0)[0] // g
        xcd = (nb + grp) % 8
        mid = li[(real[:, 0] + real[:, 1]) // 2]
        home = T["tile_owner"][np.minimum(mid >> T["tile_log2"], len(T["tile_owner"]) - 1)]
        assert (home == xcd).mean() > 0.9
        per_xcd = np.bincount(xcd, minlength=8)
        assert per_xcd.max() <= g * -(-(T["n_tasks"] // g) // 8)
    # the tile owners follow the contiguous block -> XCD map: network block k's columns belong to the XCD that runs its rows
    own = P.task_plan(0)["tile_owner"]
    assert np.all(np.diff(own[:-1]) >= 0) and set(own.tolist()) == set(range(8))


@pytest.mark.parametrize("corrupt", ["start0", "decreasing", "row_index", "overrun"])
def test_malformed_csc_is_rejected_not_read_out_of_bounds(corrupt):
    """formulate()/formulateHipdlp()/the device-side setup validate the caller's CSC arrays first."""
    import ctypes as C
    lp = L.special_lps()["distillation"]
    H = abi.ProblemHandle(lp)
    start = np.array(lp.a_start, dtype=np.int32).copy()
    index = np.array(lp.a_index, dtype=np.int32).copy()
    if corrupt == "start0":
        start[0] = 1
    elif corrupt == "decreasing":
        start[1] = start[2] + 1
    elif corrupt == "row_index":
        index[0] = lp.num_row + 5
    else:
        start[-1] = len(index) + 7
    H.struct.a_start = start.ctypes.data_as(abi.c_i32p)
    H.struct.a_index = index.ctypes.data_as(abi.c_i32p)
    for alg in ("pdlp", "hipdlp"):
        F = abi.PdlpPrepared()
        rc = solver.lib().pdlp_mi355x_host_prepare(C.byref(H.struct), C.byref(abi.default_params(solver=alg)), C.byref(F))
        assert rc != 0 and solver.lib().pdlp_mi355x_last_error()


def test_lp_without_constraints_is_refused_by_the_oracle_not_looped_on():
    """HiGHS answers LPs without rows / nonzeros itself (solveUnconstrainedLp, HighsSolve.cpp:61-66); a direct
    caller of the ABI must get an error, not an endless PDHG loop (1/max|a_ij| is the initial step size)."""
    lp = L.HighsLp(2, 0, np.array([1.0, -1.0]), np.zeros(2), np.array([1.0, 2.0]), np.zeros(0), np.zeros(0),
                   np.array([0, 0, 0], np.int32), np.zeros(0, np.int32), np.zeros(0), 1, 0.0, "norows").normalise()
    with pytest.raises(RuntimeError):
        O.oracle_solve(lp, kkt_tolerance=1e-6, pdlp_iteration_limit=1000)


def test_plain_arithmetic_exp_log_within_one_ulp_of_libm():
    """The primal-weight update of a device-driven restart: exp and log in plain IEEE arithmetic, the same bits on host and
    device.  The product's functions (csrc/pdlp_detmath.h, through pdlp_mi355x_det_exp_log) and the oracle's separately
    written restatement (oracle/det_math.h) against numpy's long-double libm, error in units of the last place — and against
    each other, bit for bit."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-10, 10, 200000), rng.uniform(-1, 1, 200000),
                        rng.uniform(-1e-3, 1e-3, 100000), [0.0, 1.0, -1.0, 709.0, -745.0]])
    y = np.concatenate([np.exp(rng.uniform(-700, 700, 300000)), np.exp(rng.uniform(-2, 2, 300000)), [1.0, 2.0, 0.5, 5e-324, 1e308]])
    e, l = np.zeros(len(x)), np.zeros(len(x))
    O.oracle().pdlp_oracle_det_exp_log(len(x), x.ctypes.data_as(abi.c_f64p), e.ctypes.data_as(abi.c_f64p), l.ctypes.data_as(abi.c_f64p))
    ref = np.exp(x.astype(np.longdouble))
    assert np.max(np.abs(e - ref) / np.spacing(np.abs(ref.astype(np.float64)))) < 1.0
    e2, l2 = np.zeros(len(y)), np.zeros(len(y))
    O.oracle().pdlp_oracle_det_exp_log(len(y), y.ctypes.data_as(abi.c_f64p), e2.ctypes.data_as(abi.c_f64p), l2.ctypes.data_as(abi.c_f64p))
    refl = np.log(y.astype(np.longdouble))
    err = np.abs(l2 - refl) / np.maximum(np.spacing(np.abs(refl.astype(np.float64))), 5e-324)
    assert np.max(err[refl != 0]) < 1.0 and l2[len(y) - 5] == 0.0
    # the product's own functions: within one ulp of libm as well, and the same bits as the oracle's restatement
    for arg, oe, ol in ((x, e, l), (y, e2, l2)):
        pe, pl = np.zeros(len(arg)), np.zeros(len(arg))
        solver.lib().pdlp_mi355x_det_exp_log(len(arg), arg.ctypes.data_as(abi.c_f64p), pe.ctypes.data_as(abi.c_f64p), pl.ctypes.data_as(abi.c_f64p))
        assert np.array_equal(pe.view(np.uint64), oe.view(np.uint64))
        assert np.array_equal(pl.view(np.uint64), ol.view(np.uint64))
    pe = np.zeros(len(x)); pl = np.zeros(len(x))
    solver.lib().pdlp_mi355x_det_exp_log(len(x), x.ctypes.data_as(abi.c_f64p), pe.ctypes.data_as(abi.c_f64p), pl.ctypes.data_as(abi.c_f64p))
    assert np.max(np.abs(pe - ref) / np.spacing(np.abs(ref.astype(np.float64)))) < 1.0

"""The multi-threaded MPS reader (pdlp_mi355x_read_mps, csrc/pdlp_mps.cpp) against the reference's
own reader.  CPU only.

tests/golden/reference_mps.json holds digests of what Highs_readModel of the reference builds for every MPS file
of its check/instances and for the edge cases of tests/golden/mps_cases/ (make_golden_mps.py).  The reference's
instance files exist in the build container only; the edge cases and everything derived from committed fixtures
run anywhere."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_mps as MG  # noqa: E402  (hessian_canonical, shared with the golden generator)
from highs_amd import lp as L
from highs_amd import solver

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF = json.load(open(os.path.join(GOLD, "reference_mps.json")))
REF_INSTANCES = "/root/reference/check/instances"


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def normalised(lp, info):
    """What Highs::passModel makes of the parser's model: tiny matrix values dropped (small_matrix_value 1e-9,
    assessMatrix), bounds beyond 1e20 infinite (assessBounds / infinite_bound), Hessian lower triangle."""
    keep = np.abs(lp.a_value) > 1e-9
    col_of = np.repeat(np.arange(lp.num_col), np.diff(lp.a_start))
    cnt = np.bincount(col_of[keep], minlength=lp.num_col)
    a_start = np.r_[0, np.cumsum(cnt)].astype(np.int32)
    inf = np.inf
    big = lambda v: np.where(v >= 1e20, inf, np.where(v <= -1e20, -inf, v))
    integ = info["integrality"] if info["integrality"] is not None else np.zeros(lp.num_col, np.uint8)
    rec = dict(num_col=lp.num_col, num_row=lp.num_row, num_nz=int(keep.sum()), sense=int(lp.sense), offset=float(lp.offset),
               col_cost=digest(lp.col_cost), col_lower=digest(big(lp.col_lower)), col_upper=digest(big(lp.col_upper)),
               row_lower=digest(big(lp.row_lower)), row_upper=digest(big(lp.row_upper)),
               a_start=digest(a_start[:-1]), a_index=digest(lp.a_index[keep]), a_value=digest(lp.a_value[keep]),
               integrality=digest(integ.astype(np.uint8)))
    names = None
    if info["col_names"] is not None and info["row_names"] is not None:
        names = hashlib.sha256("\n".join(info["col_names"] + ["--"] + info["row_names"]).encode()).hexdigest()[:24]
    rec["names"] = names
    return rec


LP_KEYS = ("num_col", "num_row", "num_nz", "sense", "offset", "col_cost", "col_lower", "col_upper", "row_lower", "row_upper",
           "a_start", "a_index", "a_value", "integrality")


def check_against_reference(path, key, threads):
    ref = REF[key]
    if ref["status"] == -1:
        # Highs::readModel refuses the file: either the parser does (then so must this reader) or Highs::passModel
        # refuses what the parser built (a NaN among the bounds / costs: assessBounds, assessCosts)
        try:
            lp, info = solver.read_mps(path, threads)
        except (RuntimeError, solver.MpsFixedFormat):
            return None
        assert any(np.isnan(v).any() for v in (lp.col_cost, lp.col_lower, lp.col_upper, lp.row_lower, lp.row_upper)), key
        return None
    try:
        lp, info = solver.read_mps(path, threads)
    except solver.MpsFixedFormat:
        return "fixed"  # names with spaces: the reference's fixed-column reader took over, out of this reader's scope
    got = normalised(lp, info)
    for k in LP_KEYS:
        assert got[k] == ref[k], (key, k, got[k], ref[k])
    if ref["names"] is not None and got["names"] is not None:
        assert got["names"] == ref["names"], key
    # Hessian: the reference holds the lower triangle after Highs::passModel (normaliseHessian)
    assert (lp.hessian is not None) == (ref["hessian_num_nz"] > 0), key
    if lp.hessian is not None:
        assert MG.hessian_canonical(*lp.hessian) == ref["hessian_canonical"], key
    # return status: kWarning iff the parser's flag is set or Highs::passModel warns (tiny values dropped, Hessian
    # diagonal completed, inconsistent bounds ...)
    if info["warning_issued"]:
        assert ref["status"] == 1, (key, info["warnings"])
    elif ref["status"] == 1:
        pass_model_warns = got["num_nz"] < lp.num_nz or ref["hessian_num_nz"] or (lp.col_lower > lp.col_upper).any()
        assert pass_model_warns, key
    return lp, info


REF_KEYS = sorted(k for k in REF if k.startswith("ref/"))
CASE_KEYS = sorted(k for k in REF if k.startswith("case/"))


@pytest.mark.skipif(not os.path.isdir(REF_INSTANCES), reason="the reference's instance files exist in the build container only")
@pytest.mark.parametrize("key", REF_KEYS)
def test_reference_instances_read_as_the_reference_reads_them(key):
    path = os.path.join(REF_INSTANCES, key[4:])
    for threads in (1, 3, 8):
        check_against_reference(path, key, threads)


@pytest.mark.parametrize("key", CASE_KEYS)
def test_edge_cases_read_as_the_reference_reads_them(key):
    path = os.path.join(GOLD, "mps_cases", key[5:])
    for threads in (1, 2, 5, 16, 64):
        check_against_reference(path, key, threads)


def _same_model(a, b):
    assert (a.num_col, a.num_row, a.sense, a.offset) == (b.num_col, b.num_row, b.sense, b.offset)
    for k in ("a_start", "a_index", "a_value", "col_cost", "col_lower", "col_upper", "row_lower", "row_upper"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


@pytest.mark.parametrize("name", sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "instances", "*.npz"))))
def test_committed_instances_round_trip(name, tmp_path):
    """write_mps -> the library's reader gives back what the python restatement of the reader gives, for every
    number of pieces (column runs continue across piece boundaries: the files are cut every few hundred bytes)."""
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
    path = str(tmp_path / (name + ".mps"))
    L.write_mps(lp, path)
    ref = L.read_mps(path)
    for threads in (1, 2, 7, 64):
        got, info = solver.read_mps(path, threads)
        _same_model(got, ref)
        assert info["threads"] == threads and not info["warning_issued"]
        assert info["col_names"][0] == "C0" and info["row_names"][-1].startswith("R") and info["objective_name"] == "COST"


def test_structured_lp_round_trip_many_pieces(tmp_path):
    """A 330k-nonzero block LP with dense linking rows, ranged rows, fixed and boxed columns and an offset
    (tests/lpgen.py::structured_lp, the small version of BASELINE config 3's stand-in)."""
    from lpgen import structured_lp
    lp = structured_lp(1, commodities=16, nodes=1024, arcs=8192, link_rows=64, link_nnz=2048, extra_rows=128)
    path = str(tmp_path / "structured.mps")
    L.write_mps(lp, path)
    ref = L.read_mps(path)
    first = None
    for threads in (0, 1, 5, 32):
        got, info = solver.read_mps(path, threads)
        _same_model(got, ref)
        first = first or info
        assert info["col_names"] == first["col_names"] and info["row_names"] == first["row_names"]


def test_qp_round_trip(tmp_path):
    from lpgen import random_diag_qp
    lp = random_diag_qp(3, 40, 60)
    path = str(tmp_path / "qp.mps")
    L.write_mps(lp, path)
    got, info = solver.read_mps(path, 3)
    _same_model(got, L.read_mps(path))
    assert np.array_equal(got.hessian_diagonal(), lp.hessian_diagonal())
    hs = info["hessian_square"]
    assert len(hs[1]) == np.count_nonzero(lp.hessian_diagonal())  # diagonal entries are not mirrored


def test_reader_errors(tmp_path):
    p = tmp_path / "x.mps"
    with pytest.raises(FileNotFoundError):
        solver.read_mps(str(tmp_path / "missing.mps"))
    p.write_text("NAME t\nROWS\n N obj\n L r\nCOLUMNS\n x obj 1 r 1\n")  # truncated: no ENDATA (parse() ends in kFail)
    with pytest.raises(RuntimeError, match="ENDATA"):
        solver.read_mps(str(p))
    p.write_text("NAME t\nROWS\n N obj\n X r\nCOLUMNS\nENDATA\n")
    with pytest.raises(RuntimeError, match="unidentified"):
        solver.read_mps(str(p))
    p.write_text("NAME t\nROWS\n N obj\n L r\nCOLUMNS\n x r\nENDATA\n")
    with pytest.raises(RuntimeError, match="No coefficient"):
        solver.read_mps(str(p))
    p.write_text("NAME t\nROWS\n N obj\n L r\nCOLUMNS\n x obj 1 r 1\nBOUNDS\n XX b x 1\nENDATA\n")
    with pytest.raises(RuntimeError, match="BOUNDS"):
        solver.read_mps(str(p))
    p.write_text("NAME t\nROWS\n N obj\n L r\nCOLUMNS\n M 'MARKER' 'INTEND'\n x obj 1 r 1\nENDATA\n")
    with pytest.raises(RuntimeError, match="marker"):
        solver.read_mps(str(p))
    p.write_text("NAME t\nROWS\n N obj\n L r\nCOLUMNS\n x obj 1 r 1\nINDICATORS\n IF r x 1\nENDATA\n")
    with pytest.raises(RuntimeError, match="cannot parse"):
        solver.read_mps(str(p))
    p.write_text("NAME t\nROWS\n N obj\n L my row\nCOLUMNS\nENDATA\n")  # names with spaces: the fixed-column reader's job
    with pytest.raises(solver.MpsFixedFormat):
        solver.read_mps(str(p))
    p.write_text("")
    with pytest.raises(RuntimeError):
        solver.read_mps(str(p))


def test_reader_time_limit(tmp_path):
    """HMpsFF::time_limit_ (io/FilereaderMps.cpp:30-31, HMpsFF::timeout :218-220): a limit that has already passed
    when the first phase ends gives return code 5 (FilereaderRetcode::kTimeout in the drop-in's reader TU), a
    generous one and 'none' (<= 0, infinite) read the file."""
    mps = str(tmp_path / "a.mps")
    lp = L.HighsLp.from_npz(os.path.join(GOLD, "instances", "25fv47.npz"))
    L.write_mps(lp, mps)
    with pytest.raises(solver.MpsTimeout):
        solver.read_mps(mps, time_limit=1e-9)
    for limit in (0.0, -1.0, float("inf"), 3600.0):
        got, _ = solver.read_mps(mps, time_limit=limit)
        assert got.num_col == lp.num_col and got.num_row == lp.num_row


# ---- the reader behind Highs::readModel (integration/FilereaderMpsMi355x.cpp in the drop-in libhighs) ------------------
ROOT = os.path.dirname(HERE)
BUILD = os.path.join(ROOT, "integration", "_build")
needs_build = pytest.mark.usefixtures("dropin_build")  # tests/conftest.py: FAILS where the reference tree is present and the build is not


def _dropin_env():
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = BUILD + ":" + os.path.join(ROOT, "highs_amd", "lib") + ":" + e.get("LD_LIBRARY_PATH", "")
    return e


@needs_build
@pytest.mark.parametrize("name", ["afiro", "adlittle", "25fv47", "standgub"])
def test_reference_cli_solves_what_the_reader_read(name, tmp_path):
    """The reference's unmodified CLI on the drop-in libhighs: Highs::readModel goes through pdlp_mi355x_read_mps,
    then the reference's own simplex (no GPU involved) must find the reference's optimal objective."""
    import re
    import subprocess
    ref = json.load(open(os.path.join(GOLD, "reference_pdlp.json")))[name]["highs"]["objective_value"]
    mps = str(tmp_path / (name + ".mps"))
    L.write_mps(L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz")), mps)
    out = subprocess.run([os.path.join(BUILD, "highs_ref_cli"), "--solver=simplex", mps], capture_output=True, text=True,
                         timeout=300, env=_dropin_env(), cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    assert out.returncode == 0 and "Optimal" in txt, txt[-1500:]
    obj = float(re.search(r"Objective value\s*:\s*(\S+)", txt)[1])
    assert abs(obj - ref) <= 1e-6 * (1 + abs(ref)), (obj, ref)  # (the golden is the reference's PDLP objective at 1e-7)


@needs_build
@pytest.mark.skipif(not os.path.isdir(REF_INSTANCES), reason="the reference's unit tests read its instance files (build container only)")
def test_reference_filereader_unit_tests_pass_on_the_dropin_reader():
    """The reference's own Catch2 cases tagged [highs_filereader] (check/TestFilereader.cpp: free and fixed format,
    comments, D exponents, integrality markers, RANGES / BOUNDS rules, duplicate names, fixed-format fallback ...) run
    on the drop-in libhighs whose MPS reader TU is integration/FilereaderMpsMi355x.cpp."""
    import subprocess
    out = subprocess.run([os.path.join(BUILD, "unit_tests_ref"), "[highs_filereader]"], capture_output=True, text=True,
                         timeout=900, env=_dropin_env())
    assert out.returncode == 0 and "All tests passed" in out.stdout, out.stdout[-3000:]


def test_gzip_streams_are_inflated_by_the_reader(tmp_path):
    """gzip files (what the Mittelmann LPs ship as) give the same model as the plain text; concatenated members and
    a truncated stream are handled (the reference reads gzip through zstr when built with zlib, HMpsFF.cpp:253-261)."""
    import gzip
    src = os.path.join(GOLD, "mps_cases", "long_columns_duplicates.mps")
    text = open(src, "rb").read()
    plain, _ = solver.read_mps(src, 3)
    gz = tmp_path / "m.mps.gz"
    gz.write_bytes(gzip.compress(text))
    got, info = solver.read_mps(str(gz), 3)
    _same_model(got, plain)
    assert info["file_bytes"] == len(text)
    half = len(text) // 2
    while text[half - 1:half] != b"\n":
        half += 1
    gz.write_bytes(gzip.compress(text[:half]) + gzip.compress(text[half:]))  # two members
    got, _ = solver.read_mps(str(gz), 2)
    _same_model(got, plain)
    gz.write_bytes(gzip.compress(text)[:-40])
    with pytest.raises(RuntimeError, match="gzip"):
        solver.read_mps(str(gz))


@pytest.mark.parametrize("seed", range(12))
def test_random_lps_round_trip(seed, tmp_path):
    """tests/lpgen.py::random_lp (every row / column kind, maximisation, offsets) written as MPS and read back by the
    library's reader with a random number of pieces: the model the python restatement reads, and — for the LPs without
    free rows (the writer turns those into N rows, which every reader drops) — the LP that was written."""
    from lpgen import random_lp
    lp = random_lp(seed)
    path = str(tmp_path / "r.mps")
    L.write_mps(lp, path)
    got, info = solver.read_mps(path, 1 + seed % 7)
    _same_model(got, L.read_mps(path))
    free = np.isinf(lp.row_lower) & np.isinf(lp.row_upper)
    if not free.any():
        assert got.num_row == lp.num_row and np.array_equal(got.col_cost, lp.col_cost) and got.sense == lp.sense
        assert np.array_equal(got.col_lower, lp.col_lower) and np.array_equal(got.col_upper, lp.col_upper)
        assert np.array_equal(got.row_lower, lp.row_lower) and np.array_equal(got.row_upper, lp.row_upper)


@needs_build
def test_reference_cli_reads_a_gzip_file_through_the_reader(tmp_path):
    """afiro.mps.gz through the reference's unmodified CLI on the drop-in (Filereader::getFilereader strips ".gz" and
    dispatches to the MPS reader TU, which hands the gzip stream to pdlp_mi355x_read_mps)."""
    import gzip
    import re
    import subprocess
    ref = json.load(open(os.path.join(GOLD, "reference_pdlp.json")))["afiro"]["highs"]["objective_value"]
    mps = str(tmp_path / "afiro.mps")
    L.write_mps(L.HighsLp.from_npz(os.path.join(GOLD, "instances", "afiro.npz")), mps)
    gz = mps + ".gz"
    open(gz, "wb").write(gzip.compress(open(mps, "rb").read()))
    out = subprocess.run([os.path.join(BUILD, "highs_ref_cli"), "--solver=simplex", gz], capture_output=True, text=True,
                         timeout=300, env=_dropin_env(), cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    assert out.returncode == 0 and "Optimal" in txt, txt[-1500:]
    assert abs(float(re.search(r"Objective value\s*:\s*(\S+)", txt)[1]) - ref) <= 1e-6 * (1 + abs(ref))


@needs_build
def test_reference_cli_names_with_spaces_go_to_the_fixed_format_reader_once(tmp_path):
    """A fixed-column file whose names contain spaces: the drop-in's reader TU sees return code 3 and calls the
    reference's fixed-format reader directly (one warning, as the reference prints it: io/FilereaderMps.cpp:45-49);
    a time limit that has passed is reported the reference's way."""
    import subprocess
    mps = tmp_path / "sp.mps"
    # fields at the fixed columns 2-3, 5-12, 15-22, 25-36 (io/HMPSIO.cpp)
    mps.write_text("NAME          spaces\nROWS\n N  obj\n G  my row\nCOLUMNS\n"
                   "    col one   obj                1.0   my row             1.0\n"
                   "    col two   obj                2.0   my row             1.0\n"
                   "RHS\n    rhs       my row             1.0\nENDATA\n")
    out = subprocess.run([os.path.join(BUILD, "highs_ref_cli"), "--solver=simplex", str(mps)], capture_output=True, text=True,
                         timeout=300, env=_dropin_env(), cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    assert out.returncode == 0 and "Optimal" in txt, txt[-1500:]
    assert txt.count("switching to fixed format parser") == 1, txt[-1500:]
    import re
    assert abs(float(re.search(r"Objective value\s*:\s*(\S+)", txt)[1]) - 1.0) < 1e-9

"""Host-side LP container mirroring the fields of HiGHS' HighsLp
(highs/lp_data/HighsLp.h) that the PDLP path reads
(CupdlpWrapper.cpp:280-308), plus the small amount of data-format code the
tests and bench need on a box without the reference tree: an MPS reader, the
reference's in-code test LPs (check/SpecialLps.h, check/TestPdlp.cpp) and the
KKT measures HiGHS reports after a solve (lp_data/HighsSolution.cpp:1043+).
"""
from dataclasses import dataclass, field

import numpy as np

kHighsInf = float("inf")


@dataclass
class HighsLp:
    num_col: int = 0
    num_row: int = 0
    col_cost: np.ndarray = field(default_factory=lambda: np.zeros(0))
    col_lower: np.ndarray = field(default_factory=lambda: np.zeros(0))
    col_upper: np.ndarray = field(default_factory=lambda: np.zeros(0))
    row_lower: np.ndarray = field(default_factory=lambda: np.zeros(0))
    row_upper: np.ndarray = field(default_factory=lambda: np.zeros(0))
    # a_matrix_ column-wise (MatrixFormat::kColwise)
    a_start: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    a_index: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    a_value: np.ndarray = field(default_factory=lambda: np.zeros(0))
    sense: int = 1  # ObjSense::kMinimize = 1, kMaximize = -1
    offset: float = 0.0
    model_name: str = ""
    # HighsModel::hessian_ (model/HighsHessian.h): (start, index, value), lower-triangular column-wise; None = LP
    hessian: tuple = None

    def set_diagonal_hessian(self, q_diag):
        """+ 1/2 sum_j q_j x_j^2 as a HighsHessian with one entry per column."""
        q = np.ascontiguousarray(q_diag, dtype=np.float64)
        assert len(q) == self.num_col
        self.hessian = (np.arange(self.num_col + 1, dtype=np.int32), np.arange(self.num_col, dtype=np.int32), q)
        return self

    def hessian_diagonal(self):
        """Diagonal of Q as a dense vector (zeros for an LP); raises if Q has off-diagonal entries."""
        q = np.zeros(self.num_col)
        if self.hessian is None:
            return q
        st, idx, val = self.hessian
        cols = np.repeat(np.arange(len(st) - 1), np.diff(st))
        if np.any((idx != cols) & (val != 0)):
            raise ValueError("off-diagonal Hessian entries")
        np.add.at(q, cols[idx == cols], val[idx == cols])
        return q

    def hessian_times(self, x):
        """Q x for the symmetric Q whose lower triangle is stored (zeros for an LP)."""
        q = np.zeros(self.num_col)
        if self.hessian is None:
            return q
        x = np.asarray(x, dtype=np.float64)
        st, idx, val = self.hessian
        cols = np.repeat(np.arange(len(st) - 1), np.diff(st))
        np.add.at(q, idx, val * x[cols])
        off = idx != cols
        np.add.at(q, cols[off], val[off] * x[idx[off]])
        return q

    def set_hessian_from_dense(self, Q):
        """Lower triangle (column-wise) of a dense symmetric matrix."""
        Q = np.asarray(Q, dtype=np.float64)
        st, idx, val = [0], [], []
        for j in range(self.num_col):
            rows = np.nonzero(Q[j:, j])[0] + j
            idx += list(rows); val += list(Q[rows, j]); st.append(len(idx))
        self.hessian = (np.array(st, np.int32), np.array(idx, np.int32), np.array(val, np.float64))
        return self

    def normalise(self):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.col_cost, self.col_lower, self.col_upper = f(self.col_cost), f(self.col_lower), f(self.col_upper)
        self.row_lower, self.row_upper, self.a_value = f(self.row_lower), f(self.row_upper), f(self.a_value)
        self.a_start = np.ascontiguousarray(self.a_start, dtype=np.int32)
        self.a_index = np.ascontiguousarray(self.a_index, dtype=np.int32)
        return self

    @property
    def num_nz(self):
        return int(self.a_start[self.num_col])

    def objective_value(self, col_value):
        """HighsModel::objectiveValue: offset + c'x + 1/2 x'Qx (no sense factor)."""
        x = np.asarray(col_value)
        v = float(self.offset + np.dot(self.col_cost, x))
        if self.hessian is not None:
            v += 0.5 * float(np.dot(self.hessian_times(x), x))
        return v

    def row_activity(self, col_value):
        out = np.zeros(self.num_row)
        cols = np.repeat(np.arange(self.num_col), np.diff(self.a_start))
        np.add.at(out, self.a_index, self.a_value * np.asarray(col_value)[cols])
        return out

    def to_npz(self, path):
        np.savez_compressed(path, num_col=self.num_col, num_row=self.num_row, col_cost=self.col_cost,
                            col_lower=self.col_lower, col_upper=self.col_upper, row_lower=self.row_lower,
                            row_upper=self.row_upper, a_start=self.a_start, a_index=self.a_index,
                            a_value=self.a_value, sense=self.sense, offset=self.offset,
                            model_name=np.array(self.model_name),
                            **({} if self.hessian is None else {"q_start": self.hessian[0], "q_index": self.hessian[1],
                                                                 "q_value": self.hessian[2]}))

    @staticmethod
    def from_npz(path):
        z = np.load(path, allow_pickle=False)
        lp = HighsLp(int(z["num_col"]), int(z["num_row"]), z["col_cost"], z["col_lower"], z["col_upper"],
                     z["row_lower"], z["row_upper"], z["a_start"], z["a_index"], z["a_value"],
                     int(z["sense"]), float(z["offset"]), str(z["model_name"])).normalise()
        if "q_start" in z.files:
            lp.hessian = (z["q_start"].astype(np.int32), z["q_index"].astype(np.int32), z["q_value"].astype(np.float64))
        return lp

    @staticmethod
    def from_rowwise(num_col, num_row, r_start, r_index, r_value, **kw):
        """Row-wise (CSR) input -> column-wise storage, entries of a column in
        ascending row order (what HighsSparseMatrix::ensureColwise produces)."""
        r_start = np.asarray(r_start, dtype=np.int64)
        r_index = np.asarray(r_index, dtype=np.int64)
        r_value = np.asarray(r_value, dtype=np.float64)
        rows = np.repeat(np.arange(num_row, dtype=np.int64), np.diff(r_start))
        order = np.argsort(r_index, kind="stable")
        a_index = rows[order].astype(np.int32)
        a_value = r_value[order]
        counts = np.bincount(r_index, minlength=num_col)
        a_start = np.zeros(num_col + 1, dtype=np.int32)
        a_start[1:] = np.cumsum(counts)
        return HighsLp(num_col=num_col, num_row=num_row, a_start=a_start, a_index=a_index, a_value=a_value, **kw).normalise()


def _dense_lp(name, cost, cl, cu, rl, ru, start, index, value, sense=1, offset=0.0):
    return HighsLp(len(cost), len(rl), np.array(cost, float), np.array(cl, float), np.array(cu, float),
                   np.array(rl, float), np.array(ru, float), np.array(start, np.int32), np.array(index, np.int32),
                   np.array(value, float), sense, offset, name).normalise()


def special_lps():
    """The in-code LPs of the reference's PDLP unit tests."""
    inf = kHighsInf
    return {
        # check/SpecialLps.h:278-296, objective 31.2
        "distillation": _dense_lp("distillation", [8, 10], [0, 0], [inf, inf], [7, 12, 6], [inf, inf, inf],
                                  [0, 3, 6], [0, 1, 2, 0, 1, 2], [2, 3, 2, 2, 4, 1]),
        # check/SpecialLps.h:335-353, maximise, objective 7
        "3d": _dense_lp("3-d LP", [1, 2, 3], [0, 0, 0], [inf, inf, inf], [-inf, -inf], [3, 2],
                        [0, 1, 2, 4], [0, 1, 0, 1], [1, 1, 2, 2], sense=-1),
        # check/TestPdlp.cpp:150-184, boxed rows, objective -16
        "boxed_row": _dense_lp("boxed-row", [-1, -2], [0, 0], [inf, 6], [3, -4], [10, 2],
                               [0, 2, 4], [0, 1, 0, 1], [1, 1, 1, -1]),
        # check/TestPdlp.cpp:186-209 -> kUnboundedOrInfeasible
        "infeasible": _dense_lp("infeasible", [-1, -2], [0, 0], [inf, inf], [-inf], [-1],
                                [0, 1, 2], [0, 0], [1, 1]),
        # check/TestPdlp.cpp:211-239 -> kUnbounded (after HiGHS' KKT check)
        "unbounded": _dense_lp("unbounded", [-1, -2], [0, 0], [inf, inf], [1], [inf],
                               [0, 1, 2], [0, 0], [1, 1]),
        # check/TestPdlp.cpp:260-284: LB, EQ, BX, UB rows, maximise (hot-start test)
        "restart_lp": _dense_lp("restart-lp", [1, 3, 5], [0, 0, 0], [inf, inf, inf], [1, 3, 2, -inf],
                                [inf, 3, 10, 5], [0, 4, 8, 12], [0, 1, 2, 3] * 3,
                                [1, 1, 1, 1, 2, 1, 2, 2, 4, 3, 2, 3], sense=-1),
        # check/SpecialLps.h blendingLp, objective -2850
        "blending": _dense_lp("blending", [-8, -10], [0, 0], [inf, inf], [-inf, -inf], [120, 210],
                              [0, 2, 4], [0, 1, 0, 1], [0.3, 0.7, 0.5, 0.5]),
    }


def read_mps(path):
    """Minimal MPS reader (fixed or free format, whitespace-separated fields):
    ROWS / COLUMNS / RHS / RANGES / BOUNDS / OBJSENSE.  Semantics follow
    HiGHS' reader (io/HMpsFF.cpp): first N row is the objective, RHS on the
    objective row is minus the offset.  (The free-format reader has no special rule for a negative UP bound:
    the column keeps its lower bound 0, HMpsFF.cpp:1525-1533.)  Test helper; the library's own reader is
    csrc/pdlp_mps.cpp (solver.read_mps)."""
    import gzip

    op = gzip.open if str(path).endswith(".gz") else open
    rows, row_type, row_idx = [], [], {}
    obj_name = None
    cols, col_idx = [], {}
    entries = []  # (row, col, val)
    seen = set()
    cost = {}
    rhs, ranges = {}, {}
    bounds = []
    sense, offset, name = 1, 0.0, ""
    section = None
    with op(path, "rt") as f:
        for line in f:
            if not line.strip() or line[0] == "*":
                continue
            if line[0] not in " \t":
                t = line.split()
                section = t[0].upper()
                if section == "NAME":
                    name = t[1] if len(t) > 1 else ""
                elif section in ("OBJSENSE",) and len(t) > 1:
                    sense = -1 if t[1].upper().startswith("MAX") else 1
                elif section == "ENDATA":
                    break
                continue
            t = line.split()
            if section == "OBJSENSE":
                sense = -1 if t[0].upper().startswith("MAX") else 1
            elif section == "ROWS":
                ty, rn = t[0].upper(), t[1]
                if ty == "N":
                    if obj_name is None:
                        obj_name = rn
                    continue
                row_idx[rn] = len(rows)
                rows.append(rn)
                row_type.append(ty)
            elif section == "COLUMNS":
                if len(t) >= 3 and t[1] == "'MARKER'":
                    continue
                cn = t[0]
                if cn not in col_idx:
                    col_idx[cn] = len(cols)
                    cols.append(cn)
                j = col_idx[cn]
                for k in range(1, len(t) - 1, 2):
                    rn, v = t[k], float(t[k + 1])
                    if v == 0.0:
                        continue  # zero coefficients are dropped (HMpsFF.cpp:897)
                    if rn == obj_name:
                        cost.setdefault(j, v)  # the first value wins (:913-927)
                    elif rn in row_idx and (row_idx[rn], j) not in seen:
                        seen.add((row_idx[rn], j))
                        entries.append((row_idx[rn], j, v))
            elif section in ("RHS", "RANGES"):
                tt = t[1:] if len(t) % 2 == 1 else t
                for k in range(0, len(tt) - 1, 2):
                    rn, v = tt[k], float(tt[k + 1])
                    if section == "RHS":
                        if rn == obj_name:
                            offset = -v
                        elif rn in row_idx:
                            rhs[row_idx[rn]] = v
                    elif rn in row_idx:
                        ranges[row_idx[rn]] = v
            elif section == "BOUNDS":
                ty = t[0].upper()
                if ty in ("FR", "MI", "PL", "BV"):
                    cn = t[2] if len(t) >= 3 else t[1]
                    bounds.append((ty, col_idx[cn], 0.0))
                else:
                    cn, v = (t[2], float(t[3])) if len(t) >= 4 else (t[1], float(t[2]))
                    bounds.append((ty, col_idx[cn], v))
    n, m = len(cols), len(rows)
    inf = kHighsInf
    rl, ru = np.zeros(m), np.zeros(m)
    for i, ty in enumerate(row_type):
        b = rhs.get(i, 0.0)
        if ty == "E":
            rl[i] = ru[i] = b
        elif ty == "L":
            rl[i], ru[i] = -inf, b
        elif ty == "G":
            rl[i], ru[i] = b, inf
    for i, r in ranges.items():
        ty = row_type[i]
        if ty == "L":
            rl[i] = ru[i] - abs(r)
        elif ty == "G":
            ru[i] = rl[i] + abs(r)
        elif ty == "E":
            if r >= 0:
                ru[i] = rl[i] + r
            else:
                rl[i] = ru[i] + r
    cl, cu = np.zeros(n), np.full(n, inf)
    for ty, j, v in bounds:
        if ty == "UP":
            cu[j] = v
        elif ty == "LO":
            cl[j] = v
        elif ty == "FX":
            cl[j] = cu[j] = v
        elif ty == "FR":
            cl[j], cu[j] = -inf, inf
        elif ty == "MI":
            cl[j] = -inf
        elif ty == "PL":
            cu[j] = inf
        elif ty == "BV":
            cl[j], cu[j] = 0.0, 1.0
    c = np.zeros(n)
    for j, v in cost.items():
        c[j] = v
    # build CSC with each column's entries in file order (HiGHS keeps file order)
    ent = np.array(entries, dtype=np.float64).reshape(-1, 3)
    order = np.argsort(ent[:, 1], kind="stable")
    a_index = ent[order, 0].astype(np.int32)
    a_value = ent[order, 2]
    a_start = np.zeros(n + 1, dtype=np.int32)
    a_start[1:] = np.cumsum(np.bincount(ent[:, 1].astype(np.int64), minlength=n))
    return HighsLp(n, m, c, cl, cu, rl, ru, a_start, a_index, a_value, sense, offset, name).normalise()


def kkt_measures(lp, col_value, col_dual, row_value, row_dual, primal_feasibility_tolerance=1e-7):
    """The HighsInfo quantities HiGHS derives from a returned LP solution
    (getKktFailures lp_data/HighsSolution.cpp:73-565, getVariableKktFailures
    :567-660, computeDualObjectiveValue :1345-1392): objective, max primal /
    dual infeasibility, max primal / dual residual error and the relative
    primal-dual objective error.  Used for the parity columns."""
    x = np.asarray(col_value, dtype=np.float64)
    rv = np.asarray(row_value, dtype=np.float64)
    cd = np.asarray(col_dual, dtype=np.float64)
    rd = np.asarray(row_dual, dtype=np.float64)
    ax = lp.row_activity(x)
    obj = lp.objective_value(x)
    cols = np.repeat(np.arange(lp.num_col), np.diff(lp.a_start))
    aty = np.zeros(lp.num_col)
    np.add.at(aty, cols, lp.a_value * rd[lp.a_index])
    lower = np.concatenate([lp.col_lower, lp.row_lower])
    upper = np.concatenate([lp.col_upper, lp.row_upper])
    value = np.concatenate([x, rv])
    dual = lp.sense * np.concatenate([cd, rd])
    pinf = np.maximum(np.maximum(lower - value, value - upper), 0.0)
    free = np.isneginf(lower) & np.isposinf(upper)
    with np.errstate(invalid="ignore"):
        length = upper - lower
        middle = (lower + upper) * 0.5
    meaningful = (lower < upper) & ~free & (length * length > primal_feasibility_tolerance)
    below = value < middle
    dinf = np.zeros_like(value)
    dinf = np.where(free, np.abs(dual), dinf)
    dinf = np.where(meaningful & below, np.maximum(-dual, 0.0), dinf)
    dinf = np.where(meaningful & ~below, np.maximum(dual, 0.0), dinf)
    # residuals: |Ax - row_value| and |A'y + col_dual - c| (HighsSolution.cpp:196-199,263-268,400+)
    pres = np.abs(ax - rv)
    qx = lp.hessian_times(x)  # zeros for an LP
    dres = np.abs(aty + cd - lp.col_cost - qx)
    # dual objective: offset + sum bound * dual, bound = lower if primal < mid else upper; free -> 1
    ndual = np.concatenate([cd, rd])
    bound = np.where(free, 1.0, np.where(below, lower, upper))
    with np.errstate(invalid="ignore"):
        terms = np.where(ndual == 0.0, 0.0, bound * ndual)
    dobj = float(lp.offset + np.sum(terms) - 0.5 * np.dot(qx, x))
    return {
        "objective_function_value": obj,
        "dual_objective_value": dobj,
        "max_primal_infeasibility": float(pinf.max(initial=0.0)),
        "max_dual_infeasibility": float(dinf.max(initial=0.0)),
        "max_primal_residual_error": float(pres.max(initial=0.0)),
        "max_dual_residual_error": float(dres.max(initial=0.0)),
        "primal_dual_objective_error": abs(obj - dobj) / (1.0 + abs(obj) + abs(dobj)),
    }


def write_mps(lp, path):
    """Minimal free-format MPS writer (the inverse of read_mps) — lets the tests hand golden LPs to
    the reference's own CLI on a box without the reference tree."""
    inf = kHighsInf
    with open(path, "w") as f:
        f.write(f"NAME {lp.model_name or 'LP'}\n")
        if lp.sense < 0:
            f.write("OBJSENSE\n    MAX\n")
        f.write("ROWS\n N COST\n")
        rt = []
        for i in range(lp.num_row):
            lo, up = lp.row_lower[i], lp.row_upper[i]
            t = "E" if lo == up else ("G" if up >= inf else ("L" if lo <= -inf else "R"))
            if lo <= -inf and up >= inf:
                t = "N"
            rt.append(t)
            f.write(f" {'G' if t == 'R' else t} R{i}\n")
        f.write("COLUMNS\n")
        for j in range(lp.num_col):
            if lp.col_cost[j] != 0.0:
                f.write(f" C{j} COST {float(lp.col_cost[j])!r}\n")
            for p in range(lp.a_start[j], lp.a_start[j + 1]):
                f.write(f" C{j} R{lp.a_index[p]} {float(lp.a_value[p])!r}\n")
            if lp.col_cost[j] == 0.0 and lp.a_start[j] == lp.a_start[j + 1]:
                f.write(f" C{j} COST 0\n")
        f.write("RHS\n")
        if lp.offset != 0.0:
            f.write(f" RHS COST {float(-lp.offset)!r}\n")
        for i in range(lp.num_row):
            v = {"E": lp.row_lower[i], "G": lp.row_lower[i], "R": lp.row_lower[i], "L": lp.row_upper[i], "N": 0.0}[rt[i]]
            if v != 0.0:
                f.write(f" RHS R{i} {float(v)!r}\n")
        if "R" in rt:
            f.write("RANGES\n")
            for i in range(lp.num_row):
                if rt[i] == "R":
                    f.write(f" RNG R{i} {float((lp.row_upper[i] - lp.row_lower[i]))!r}\n")
        f.write("BOUNDS\n")
        for j in range(lp.num_col):
            lo, up = lp.col_lower[j], lp.col_upper[j]
            if lo == 0.0 and up >= inf:
                continue
            if lo <= -inf and up >= inf:
                f.write(f" FR BND C{j}\n")
            elif lo == up:
                f.write(f" FX BND C{j} {float(lo)!r}\n")
            else:
                if lo <= -inf:
                    f.write(f" MI BND C{j}\n")
                elif lo != 0.0:
                    f.write(f" LO BND C{j} {float(lo)!r}\n")
                if up < inf:
                    f.write(f" UP BND C{j} {float(up)!r}\n")
        if lp.hessian is not None:  # lower triangle of Q: QUADOBJ (what HMpsFF.cpp reads into HighsHessian)
            st, idx, val = lp.hessian
            f.write("QUADOBJ\n")
            for j in range(len(st) - 1):
                for p in range(st[j], st[j + 1]):
                    if val[p] != 0.0:
                        f.write(f" C{j} C{idx[p]} {float(val[p])!r}\n")
        f.write("ENDATA\n")

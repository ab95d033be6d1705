"""Random small feasible, bounded LPs covering every row kind (EQ / GEQ / LEQ / ranged / free) and
column bound kind (free / lower / upper / boxed / fixed) the formulation step distinguishes
(CupdlpWrapper.cpp:315-378), plus empty rows and columns, maximisation and an objective offset."""
import numpy as np

from highs_amd import lp as L


def random_lp(seed, m=None, n=None):
    rng = np.random.default_rng(seed)
    m = m or int(rng.integers(3, 30))
    n = n or int(rng.integers(3, 30))
    inf = float("inf")
    A = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.35)
    if seed % 3 == 0:
        A[rng.integers(0, m)] = 0.0  # an empty row
        A[:, rng.integers(0, n)] = 0.0  # an empty column
    xs = rng.uniform(-1, 2, n)
    cl, cu = np.full(n, -inf), np.full(n, inf)
    for j in range(n):
        k = rng.integers(0, 5)
        if k == 1: cl[j] = xs[j] - rng.random()
        elif k == 2: cu[j] = xs[j] + rng.random()
        elif k == 3: cl[j], cu[j] = xs[j] - rng.random(), xs[j] + rng.random()
        elif k == 4: cl[j] = cu[j] = xs[j]
    ax = A @ xs
    rl, ru = np.full(m, -inf), np.full(m, inf)
    for i in range(m):
        k = rng.integers(0, 5)
        if k == 0: rl[i] = ru[i] = ax[i]
        elif k == 1: rl[i] = ax[i] - rng.random()
        elif k == 2: ru[i] = ax[i] + rng.random()
        elif k == 3: rl[i], ru[i] = ax[i] - rng.random(), ax[i] + rng.random()
        # k == 4: free row
    # a cost that keeps the LP bounded: c = A'y0 + reduced costs compatible with the bounds
    y0 = rng.standard_normal(m)
    y0 = np.where(np.isinf(rl) & np.isinf(ru), 0.0, y0)
    y0 = np.where(np.isinf(ru) & ~np.isinf(rl), np.abs(y0), y0)   # >= rows: y >= 0
    y0 = np.where(np.isinf(rl) & ~np.isinf(ru), -np.abs(y0), y0)  # <= rows: y <= 0
    z = rng.standard_normal(n)
    z = np.where(np.isinf(cl) & np.isinf(cu), 0.0, z)
    z = np.where(np.isinf(cu) & ~np.isinf(cl), np.abs(z), z)
    z = np.where(np.isinf(cl) & ~np.isinf(cu), -np.abs(z), z)
    c = A.T @ y0 + z
    sense = -1 if seed % 2 else 1
    rows, cols = np.nonzero(A.T)  # column-major order
    a_start = np.zeros(n + 1, np.int32)
    a_start[1:] = np.cumsum(np.bincount(rows, minlength=n))
    return L.HighsLp(n, m, sense * c, cl, cu, rl, ru, a_start, cols.astype(np.int32), A.T[rows, cols], sense,
                     float(seed % 5) - 2.0, f"rand{seed}").normalise()


def drop_free_rows(lp):
    """The same LP without its free rows (an MPS file cannot carry them to the reference binary:
    HiGHS' reader discards extra N rows)."""
    keep = ~(np.isinf(lp.row_lower) & np.isinf(lp.row_upper))
    if keep.all():
        return lp
    newidx = np.cumsum(keep) - 1
    cols = np.repeat(np.arange(lp.num_col), np.diff(lp.a_start))
    sel = keep[lp.a_index]
    a_start = np.zeros(lp.num_col + 1, np.int32)
    a_start[1:] = np.cumsum(np.bincount(cols[sel], minlength=lp.num_col))
    return L.HighsLp(lp.num_col, int(keep.sum()), lp.col_cost, lp.col_lower, lp.col_upper, lp.row_lower[keep],
                     lp.row_upper[keep], a_start, newidx[lp.a_index[sel]].astype(np.int32), lp.a_value[sel],
                     lp.sense, lp.offset).normalise()


def random_diag_qp(seed, m=None, n=None):
    """random_lp(seed) made a convex QP: + 1/2 sum q_j x_j^2 with q_j = 0 for about a third of the columns and
    U(0.1, 3) for the rest (times the objective sense, so that the maximisation instances stay concave)."""
    lp = drop_free_rows(random_lp(seed, m, n))
    rng = np.random.default_rng(1000 + seed)
    q = np.where(rng.random(lp.num_col) < 0.35, 0.0, rng.uniform(0.1, 3.0, lp.num_col))
    return lp.set_diagonal_hessian(lp.sense * q)

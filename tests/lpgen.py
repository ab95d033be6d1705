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


def random_sparse_qp(seed, m=None, n=None, density=None):
    """random_lp(seed) made a convex QP with a SPARSE NON-DIAGONAL Hessian: Q = G'G + diag(d) for a sparse random G
    (so Q is PSD by construction, with off-diagonal entries where two columns share a row of G) and d = 0 for about
    a third of the columns; times the objective sense, so that the maximisation instances stay concave."""
    lp = drop_free_rows(random_lp(seed, m, n))
    rng = np.random.default_rng(2000 + seed)
    nc = lp.num_col
    k = max(2, nc // 2)
    dens = density if density is not None else min(0.5, 3.0 / nc)
    G = rng.standard_normal((k, nc)) * (rng.random((k, nc)) < dens)
    d = np.where(rng.random(nc) < 0.35, 0.0, rng.uniform(0.1, 2.0, nc))
    Q = G.T @ G + np.diag(d)
    return lp.set_hessian_from_dense(lp.sense * Q)


def structured_lp(seed=1, commodities=64, nodes=4096, arcs=32768, link_rows=256, link_nnz=4096, extra_rows=512):
    """A block-structured LP of the kind BASELINE config 3 stands for (pds-class multi-commodity / staircase
    models; pds-100 itself is not in the reference tree): `commodities` network blocks — node-balance EQUALITY
    rows over that block's arc columns, 2 nonzeros (+1/-1) per column, so a row block only touches its own
    column range — tied together by `link_rows` DENSE rows of `link_nnz` nonzeros each across all blocks
    (bundle capacities; longer than the slab kernel's 256-nonzero limit, so they exercise the long-major side
    kernel), plus `extra_rows` RANGED and FREE rows, boxed and fixed columns, an objective offset.  Feasible and
    bounded by construction (a primal point inside all bounds, costs = A'y0 + sign-compatible reduced costs).
    Defaults: 2.1M columns, 263k rows, 5.3M nonzeros.  Deterministic for a given seed (numpy default_rng)."""
    rng = np.random.default_rng(seed)
    inf = float("inf")
    K, N, Aa = commodities, nodes, arcs
    n = K * Aa
    # --- network blocks: arc (tail -> head) of block k is column k*Aa + a, rows k*N + node
    tail = rng.integers(0, N, size=(K, Aa))
    head = (tail + 1 + rng.integers(0, N - 1, size=(K, Aa))) % N
    col = np.arange(n, dtype=np.int64)
    blk = np.repeat(np.arange(K, dtype=np.int64), Aa)
    r_net = np.concatenate([blk * N + tail.ravel(), blk * N + head.ravel()])
    c_net = np.concatenate([col, col])
    v_net = np.concatenate([np.ones(n), -np.ones(n)])
    m_net = K * N
    # --- dense linking rows
    c_link = rng.integers(0, n, size=(link_rows, link_nnz))
    c_link.sort(axis=1)
    keep = np.ones_like(c_link, dtype=bool)
    keep[:, 1:] = c_link[:, 1:] != c_link[:, :-1]
    r_link = (m_net + np.repeat(np.arange(link_rows, dtype=np.int64), link_nnz))[keep.ravel()]
    v_link = rng.uniform(0.5, 2.0, size=c_link.size)[keep.ravel()]
    c_link = c_link.ravel()[keep.ravel()]
    # --- extra rows: short random rows, alternately ranged and free
    per = 12
    c_ex = rng.integers(0, n, size=(extra_rows, per))
    c_ex.sort(axis=1)
    keep = np.ones_like(c_ex, dtype=bool)
    keep[:, 1:] = c_ex[:, 1:] != c_ex[:, :-1]
    r_ex = (m_net + link_rows + np.repeat(np.arange(extra_rows, dtype=np.int64), per))[keep.ravel()]
    v_ex = rng.standard_normal(c_ex.size)[keep.ravel()]
    c_ex = c_ex.ravel()[keep.ravel()]
    m = m_net + link_rows + extra_rows
    rows = np.concatenate([r_net, r_link, r_ex])
    cols = np.concatenate([c_net, c_link, c_ex])
    vals = np.concatenate([v_net, v_link, v_ex])
    order = np.lexsort((rows, cols))  # column-major, rows ascending within a column
    rows, cols, vals = rows[order], cols[order], vals[order]
    a_start = np.zeros(n + 1, np.int64)
    a_start[1:] = np.cumsum(np.bincount(cols, minlength=n))
    # --- primal point, bounds
    xs = rng.uniform(0.0, 1.0, n)
    cl, cu = np.zeros(n), xs + rng.uniform(0.1, 1.0, n)
    fixed = rng.random(n) < 0.01
    cl[fixed] = cu[fixed] = xs[fixed]
    unb = rng.random(n) < 0.2
    cu[unb & ~fixed] = inf
    ax = np.zeros(m)
    np.add.at(ax, rows, vals * xs[cols])
    rl, ru = ax.copy(), ax.copy()  # network rows: equalities
    ru[m_net:m_net + link_rows] = ax[m_net:m_net + link_rows] + rng.uniform(0.0, 1.0, link_rows)  # capacities: <=
    rl[m_net:m_net + link_rows] = -inf
    ex = np.arange(m_net + link_rows, m)
    ranged = ex[::2]
    rl[ranged], ru[ranged] = ax[ranged] - rng.uniform(0.1, 1.0, ranged.size), ax[ranged] + rng.uniform(0.1, 1.0, ranged.size)
    free = ex[1::2]
    rl[free], ru[free] = -inf, inf
    # --- a bounded objective: c = A'y0 + z with multipliers / reduced costs of admissible sign
    y0 = rng.standard_normal(m)
    y0[m_net:m_net + link_rows] = -np.abs(y0[m_net:m_net + link_rows])  # <= rows
    y0[free] = 0.0
    z = rng.standard_normal(n)
    z = np.where(np.isinf(cu), np.abs(z), z)  # only a lower bound: z >= 0
    c = z.copy()
    np.add.at(c, cols, vals * y0[rows])
    return L.HighsLp(n, m, c, cl, cu, rl, ru, a_start.astype(np.int32), rows.astype(np.int32), vals, 1, 3.0,
                     f"structured{seed}").normalise()


def dense_column_lp(seed=1, periods=512, rows_per=1024, cols_per=896, dense_cols=192, dense_nnz=6000, tail_rows=2048,
                    tail_max=3000):
    """A second structured family next to structured_lp (BASELINE config 3 stands for real-world LPs; pds-100 itself is
    not in the tree): a STAIRCASE of `periods` time periods — period t owns `cols_per` columns and `rows_per` rows whose
    entries sit in the columns of periods t-1 and t (multi-period production / inventory models: a banded matrix, every
    row block gathers from two neighbouring column blocks) — plus what such models carry in practice and what the
    block-angular LP does not have:
      * DENSE COLUMNS: `dense_cols` linking columns (capacity-expansion variables) with ~`dense_nnz` nonzeros each
        spread over ALL periods — longer than the slab layout's 256-entry limit and the stream plan's 2048-entry blocks,
        so A'y runs them as segment tasks (the transpose of structured_lp's dense rows);
      * POWER-LAW ROW LENGTHS: `tail_rows` extra rows whose lengths follow a Pareto law (most below 20 entries, the
        longest ~`tail_max`), entries anywhere in the staircase;
      * <=, >=, ranged and equality rows, boxed / fixed / free / one-sided columns, a maximisation sense and an offset.
    Feasible and bounded by construction (a point strictly inside the bounds; costs = A'y0 + reduced costs of the sign
    each bound admits).  Defaults: 459k columns, 526k rows, ~4.4M nonzeros.  Deterministic for a given seed."""
    rng = np.random.default_rng(seed)
    inf = float("inf")
    T, R, Cp = periods, rows_per, cols_per
    n_st = T * Cp
    n = n_st + dense_cols
    m_st = T * R
    # --- staircase rows: 2..6 entries in the own period's columns, 1..3 in the previous period's
    k_own = rng.integers(2, 7, size=m_st)
    k_prev = rng.integers(1, 4, size=m_st)
    k_prev[:R] = 0
    per = np.repeat(np.arange(T, dtype=np.int64), R)
    r_own = np.repeat(np.arange(m_st, dtype=np.int64), k_own)
    c_own = np.repeat(per, k_own) * Cp + rng.integers(0, Cp, size=r_own.size)
    r_prev = np.repeat(np.arange(m_st, dtype=np.int64), k_prev)
    c_prev = (np.repeat(per, k_prev) - 1) * Cp + rng.integers(0, Cp, size=r_prev.size)
    # --- power-law rows
    lens = np.minimum(tail_max, (4.0 * (1.0 + rng.pareto(1.1, size=tail_rows))).astype(np.int64))
    r_tail = m_st + np.repeat(np.arange(tail_rows, dtype=np.int64), lens)
    c_tail = rng.integers(0, n_st, size=r_tail.size)
    m = m_st + tail_rows
    # --- dense columns: ~dense_nnz rows each, anywhere
    r_dense = rng.integers(0, m, size=(dense_cols, dense_nnz)).ravel()
    c_dense = n_st + np.repeat(np.arange(dense_cols, dtype=np.int64), dense_nnz)
    rows = np.concatenate([r_own, r_prev, r_tail, r_dense])
    cols = np.concatenate([c_own, c_prev, c_tail, c_dense])
    vals = rng.uniform(0.2, 2.0, size=rows.size) * np.where(rng.random(rows.size) < 0.4, -1.0, 1.0)
    # one entry per (column, row): sort column-major and drop repeats
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    first = np.ones(rows.size, dtype=bool)
    first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    rows, cols, vals = rows[first], cols[first], vals[first]
    a_start = np.zeros(n + 1, np.int64)
    a_start[1:] = np.cumsum(np.bincount(cols, minlength=n))
    # --- a primal point and the column bounds around it
    xs = rng.uniform(-1.0, 1.0, n)
    cl = xs - rng.uniform(0.1, 1.0, n)
    cu = xs + rng.uniform(0.1, 1.0, n)
    kind = rng.random(n)
    cl[kind < 0.15] = -inf                      # only an upper bound
    cu[(kind >= 0.15) & (kind < 0.35)] = inf    # only a lower bound
    free = (kind >= 0.35) & (kind < 0.38)
    cl[free], cu[free] = -inf, inf
    fixed = (kind >= 0.38) & (kind < 0.39)
    cl[fixed] = cu[fixed] = xs[fixed]
    ax = np.zeros(m)
    np.add.at(ax, rows, vals * xs[cols])
    rk = rng.random(m)
    rl, ru = ax.copy(), ax.copy()                                   # rk < 0.3: equality
    le = (rk >= 0.3) & (rk < 0.55)
    rl[le], ru[le] = -inf, ax[le] + rng.uniform(0.0, 1.0, int(le.sum()))
    ge = (rk >= 0.55) & (rk < 0.8)
    rl[ge], ru[ge] = ax[ge] - rng.uniform(0.0, 1.0, int(ge.sum())), inf
    rg = rk >= 0.8
    rl[rg], ru[rg] = ax[rg] - rng.uniform(0.1, 1.0, int(rg.sum())), ax[rg] + rng.uniform(0.1, 1.0, int(rg.sum()))
    # --- a bounded objective for the MINIMISATION form, then handed over as a maximisation: c = A'y0 + z
    y0 = rng.standard_normal(m)
    y0[le] = -np.abs(y0[le])
    y0[ge] = np.abs(y0[ge])
    z = rng.standard_normal(n)
    z = np.where(np.isinf(cl) & ~np.isinf(cu), -np.abs(z), z)   # only an upper bound: z <= 0
    z = np.where(np.isinf(cu) & ~np.isinf(cl), np.abs(z), z)    # only a lower bound: z >= 0
    z[free] = 0.0
    c = z.copy()
    np.add.at(c, cols, vals * y0[rows])
    return L.HighsLp(n, m, -c, cl, cu, rl, ru, a_start.astype(np.int32), rows.astype(np.int32), vals, -1, -7.5,
                     f"staircase{seed}").normalise()


def bench_qp_at_scale(n, banded, seed=1):
    """bench.py's QP configurations (`--config qp`: diagonal Q ~ U(0,1); `--config qpn`: tridiagonal, diagonally dominant
    PSD Q) built by the SAME generators at a size the reference's active-set QP solver can still take (at n = 400 it needs
    181 000 iterations and misses the optimum in the fourth digit; n = 1000 does not finish in 50 minutes): the library's seeded synthetic LP with 8 nonzeros per row plus the
    Hessian of bench.build_workload."""
    from highs_amd import solver
    sp_ = solver.SyntheticProblem(n, n, 8 * n, seed)
    lp = sp_.to_lp()
    sp_.close()
    rng = np.random.default_rng(1)
    if banded:
        off = rng.uniform(-0.5, 0.5, n - 1)
        diag = np.abs(np.concatenate([off, [0.0]])) + np.abs(np.concatenate([[0.0], off])) + rng.uniform(0.0, 1.0, n)
        st = np.zeros(n + 1, np.int32)
        st[1:] = np.cumsum(np.concatenate([np.full(n - 1, 2), [1]]))
        qi = np.empty(2 * n - 1, np.int32)
        qv = np.empty(2 * n - 1)
        qi[0::2], qv[0::2] = np.arange(n), diag
        qi[1::2], qv[1::2] = np.arange(1, n), off
        lp.hessian = (st, qi, qv)
    else:
        lp.hessian = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), rng.uniform(0.0, 1.0, n))
    return lp


def _finish_lp(rng, rows, cols, vals, m, n, name, sense=1, offset=0.0):
    """Triplets -> a feasible, bounded HighsLp: one entry per (row, column), column-major storage, a primal point strictly
    inside mixed column bounds, rows of every kind around its activity, costs = A'y0 + reduced costs of admissible sign."""
    inf = float("inf")
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    first = np.ones(rows.size, dtype=bool)
    first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    rows, cols, vals = rows[first], cols[first], vals[first]
    a_start = np.zeros(n + 1, np.int64)
    a_start[1:] = np.cumsum(np.bincount(cols, minlength=n))
    xs = rng.uniform(-1.0, 1.0, n)
    cl = xs - rng.uniform(0.1, 1.0, n)
    cu = xs + rng.uniform(0.1, 1.0, n)
    kind = rng.random(n)
    cl[kind < 0.1] = -inf
    cu[(kind >= 0.1) & (kind < 0.3)] = inf
    fixed = (kind >= 0.3) & (kind < 0.31)
    cl[fixed] = cu[fixed] = xs[fixed]
    ax = np.zeros(m)
    np.add.at(ax, rows, vals * xs[cols])
    rk = rng.random(m)
    rl, ru = ax.copy(), ax.copy()
    le = (rk >= 0.35) & (rk < 0.6)
    rl[le], ru[le] = -inf, ax[le] + rng.uniform(0.0, 1.0, int(le.sum()))
    ge = (rk >= 0.6) & (rk < 0.85)
    rl[ge], ru[ge] = ax[ge] - rng.uniform(0.0, 1.0, int(ge.sum())), inf
    rg = rk >= 0.85
    rl[rg], ru[rg] = ax[rg] - rng.uniform(0.1, 1.0, int(rg.sum())), ax[rg] + rng.uniform(0.1, 1.0, int(rg.sum()))
    y0 = rng.standard_normal(m)
    y0[le] = -np.abs(y0[le])
    y0[ge] = np.abs(y0[ge])
    z = rng.standard_normal(n)
    z = np.where(np.isinf(cl) & ~np.isinf(cu), -np.abs(z), z)
    z = np.where(np.isinf(cu) & ~np.isinf(cl), np.abs(z), z)
    c = z.copy()
    np.add.at(c, cols, vals * y0[rows])
    return L.HighsLp(n, m, sense * c, cl, cu, rl, ru, a_start.astype(np.int32), rows.astype(np.int32), vals, sense, offset,
                     name).normalise()


def tall_lp(seed=1, n=160_000, m=1_200_000, row_nnz=4, window=2048, dense_rows=96, dense_nnz=8000):
    """HELD-OUT family 1 (round 6: the slab partition's constants were chosen on structured_lp and dense_column_lp; this
    one and powerlaw_band_lp were written afterwards and the constants were NOT re-tuned on them): a TALL LP, m >> n —
    scenario / sample-average models: every row has `row_nnz` entries inside a window of `window` columns that slides
    with the row index, so every COLUMN is touched by ~m * row_nnz / n = 30 rows spread over a stretch of rows; plus
    `dense_rows` coupling rows of ~`dense_nnz` entries anywhere (segment tasks of A x; seen from the columns they are
    the rows that EVERY block of the transposed operand gathers from).  The gathered vector of A x (n doubles = 1.3 MB)
    fits an XCD's L2: that operand runs the CSR stream layout, the transposed one (y: 9.6 MB) the slab layout — a mix
    the two fitted families do not have.  Defaults: 1.2 M rows, 160 k columns, ~5.5 M nonzeros."""
    rng = np.random.default_rng(seed)
    m_reg = m - dense_rows
    r = np.repeat(np.arange(m_reg, dtype=np.int64), row_nnz)
    base = (np.arange(m_reg, dtype=np.int64) * (n - window)) // max(m_reg - 1, 1)
    c = np.repeat(base, row_nnz) + rng.integers(0, window, size=r.size)
    r_d = m_reg + np.repeat(np.arange(dense_rows, dtype=np.int64), dense_nnz)
    c_d = rng.integers(0, n, size=r_d.size)
    rows, cols = np.concatenate([r, r_d]), np.concatenate([c, c_d])
    vals = rng.uniform(0.2, 2.0, size=rows.size) * np.where(rng.random(rows.size) < 0.4, -1.0, 1.0)
    return _finish_lp(rng, rows, cols, vals, m, n, f"tall{seed}", 1, 1.5)


def powerlaw_band_lp(seed=1, n=700_000, m=650_000, band=4096, hubs=3000, hub_share=0.12, max_len=240):
    """HELD-OUT family 2 (see tall_lp): POWER-LAW lengths in BOTH orientations with banded locality.  Row lengths follow
    a Pareto law (most below 10 entries, capped at `max_len`, so no row is a segment task); a row's entries lie in a band
    of `band` columns around its own position — except a `hub_share` of them, which go to one of `hubs` hub columns
    chosen by a Zipf law, so that column lengths are power-law too: the heaviest hubs have thousands of entries (segment
    tasks of A'y), hundreds of them lie between the slab layout's 256-entry limit and a segment.  Both gathered vectors
    exceed an L2 (5.6 / 5.2 MB): both operands run the slab layout.  Defaults: ~5.4 M nonzeros."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(max_len, (3.0 * (1.0 + rng.pareto(1.3, size=m))).astype(np.int64))
    r = np.repeat(np.arange(m, dtype=np.int64), lens)
    centre = (np.repeat(np.arange(m, dtype=np.int64), lens) * (n - 1)) // max(m - 1, 1)
    c = np.clip(centre + rng.integers(-band // 2, band // 2, size=r.size), 0, n - 1)
    to_hub = rng.random(r.size) < hub_share
    hub_cols = rng.choice(n, size=hubs, replace=False)
    zipf = 1.0 / np.arange(1, hubs + 1) ** 1.1
    pick = rng.choice(hubs, size=int(to_hub.sum()), p=zipf / zipf.sum())
    c[to_hub] = hub_cols[pick]
    vals = rng.uniform(0.2, 2.0, size=r.size) * np.where(rng.random(r.size) < 0.4, -1.0, 1.0)
    return _finish_lp(rng, r, c, vals, m, n, f"plband{seed}", -1, -2.0)

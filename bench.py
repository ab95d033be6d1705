#!/usr/bin/env python3
"""bench.py — PDHG iterations/s of the MI355X PDLP path on the BASELINE.json workload.

  python bench.py --gpus 1 --steps K --warmup W            (default: N=1, a few seconds)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one accepted PDHG iteration (two SpMVs + the fused level-1 work +
the device-side step-size decision), including the reference's check/restart
schedule (every 40th iteration) — i.e. exactly what `pdlp_iteration_count`
counts.  The LP (synthetic 1M x 1M, 8M nnz, SURVEY §8d generator, seed 1) is
formulated, scaled and uploaded before the timed region; the K timed steps run
with everything resident in HBM.  With N > 1 the SAME LP is row-block sharded
over the N GPUs (strong scaling) and the A'y partials are all-reduced with RCCL.
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# dmabuf-only hosts: HIP IPC (the direct xGMI exchange at N > 1, RCCL's own buffers) needs this before the first HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[3] — the configuration the metric is quoted on; fits one GPU (0.42 GB)
    "b": dict(m=1_000_000, n=1_000_000, nnz=8_000_000, name="synthetic random sparse LP 1Mx1M, 8M nnz (seed 1)"),
    # BASELINE.json configs[1]
    "a": dict(m=100_000, n=100_000, nnz=1_000_000, name="synthetic random sparse LP 100kx100k, 1M nnz (seed 1)"),
    # BASELINE.json configs[4]: QP prox path — the same generator at 500k x 500k (8 nnz/row) plus a random PSD
    # diagonal Q, q_j ~ U(0,1) (numpy default_rng(1)).  No PDLP-QP exists in the reference: parity unpinned.
    # BASELINE.json configs[2] stand-in (pds-100 is not in the reference tree, no network): the block-angular
    # multi-commodity network LP of tests/lpgen.py::structured_lp — 64 network blocks, 256 dense linking rows of
    # 4096 nonzeros (segment tasks), ranged and free rows: 2.1M columns, 263k rows, 5.3M nonzeros
    "c": dict(structured=True, name="structured block-angular network LP, 263k x 2.1M, 5.3M nnz, 256 dense linking rows (seed 1)"),
    # second structured family (round 4): tests/lpgen.py::dense_column_lp — staircase LP with 192 DENSE COLUMNS of ~6000
    # nonzeros (segment tasks in the A'y launch), power-law row lengths: 459k columns, 526k rows, 4.4M nonzeros
    "d": dict(staircase=True, name="structured staircase LP with dense columns, 526k x 459k, 4.4M nnz, 192 dense columns (seed 1)"),
    # HELD-OUT structured families (round 6): written after the slab partition's constants were chosen on c and d, never tuned
    # on.  tests/lpgen.py::tall_lp — 1.2M rows x 160k columns, 5.5M nnz, 96 dense coupling rows (A x: CSR stream + segment
    # tasks, A'y: slab layout); powerlaw_band_lp — 650k x 700k, 6.3M nnz, power-law row AND column lengths, hub columns
    "e": dict(family="tall", name="held-out tall LP, 1.2M x 160k, 5.5M nnz, 96 dense rows (seed 1)"),
    "f": dict(family="plband", name="held-out power-law banded LP, 650k x 700k, 6.3M nnz, hub columns up to 93k entries (seed 1)"),
    "qp": dict(m=500_000, n=500_000, nnz=4_000_000, qp=True,
               name="synthetic random sparse QP 500kx500k, 4M nnz, diagonal Q ~ U(0,1) (seed 1)"),
    # the same with a NON-diagonal Hessian (the Q x SpMV of the general-Q path, a fourth launch per trial): tridiagonal,
    # diagonally dominant (PSD), off-diagonal entries ~ U(-0.5, 0.5), diagonal = their absolute row sums + U(0,1)
    "qpn": dict(m=500_000, n=500_000, nnz=4_000_000, qp=True, banded=True,
                name="synthetic random sparse QP 500kx500k, 4M nnz, tridiagonal PSD Q (seed 1)"),
}
PRE_ROLL = 40  # iterations of start-up excluded from every timed window
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy ceiling


def algorithmic_bytes(n, m, nnz):
    """B_iter = 2*nnz*(8+4) + 4(m+1) + 4(n+1) + 8*(15n + 12m), and the share of the
    dominant kernel (A x+ SpMV fused with the dual step): 12 B/nnz + row pointers + read x (n) +
    write ax (m) + dual step R4W1 (5m)."""
    b_iter = 2 * nnz * 12 + 4 * (m + 1) + 4 * (n + 1) + 8 * (15 * n + 12 * m)
    b_spmv_ax = 12 * nnz + 4 * (m + 1) + 8 * n + 8 * m + 8 * 5 * m
    b_spmv_aty = 12 * nnz + 4 * (n + 1) + 8 * m + 8 * n + 8 * 4 * n
    return b_iter, b_spmv_ax, b_spmv_aty


def algorithmic_bytes_hipdlp(n, m, nnz):
    """One Halpern step = two fused kernels.  A'y side: matrix 12 B/nnz + column pointers + read y (m)
    + read x, c, anchor, l, u (5n) + write x, reflected x (2n).  A x side: matrix + row pointers + read reflected
    x (n) + read y, anchor, row bounds (4m) + write y (m).  (Major steps, 2 in 40, also write x_next/slack/y_next.)"""
    b_aty = 12 * nnz + 4 * (n + 1) + 8 * m + 8 * 7 * n
    b_ax = 12 * nnz + 4 * (m + 1) + 8 * n + 8 * 5 * m
    return b_ax + b_aty, b_ax, b_aty


def build_workload(config):
    """The LP / QP of one bench configuration as a problem handle (`.struct` = pdlp_problem_t) plus the arrays that must
    stay alive next to it.  Shared with tools/kbench.py (the command the rocprofv3 passes run)."""
    from highs_amd import abi, solver
    cfg = CONFIGS[config]
    if cfg.get("structured") or cfg.get("staircase") or cfg.get("family"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from lpgen import dense_column_lp, powerlaw_band_lp, structured_lp, tall_lp
        gen = tall_lp if cfg.get("family") == "tall" else powerlaw_band_lp if cfg.get("family") == "plband" else \
            dense_column_lp if cfg.get("staircase") else structured_lp
        sp_ = abi.ProblemHandle(gen(1))  # same attribute (.struct) as the library-generated problem
    else:
        sp_ = solver.SyntheticProblem(cfg["m"], cfg["n"], cfg["nnz"], 1)
    qkeep = None
    if cfg.get("qp"):
        import numpy as np
        ncol = sp_.struct.num_col
        if cfg.get("banded"):
            rng = np.random.default_rng(1)
            off = rng.uniform(-0.5, 0.5, ncol - 1)
            diag = np.abs(np.concatenate([off, [0.0]])) + np.abs(np.concatenate([[0.0], off])) + rng.uniform(0.0, 1.0, ncol)
            st = np.zeros(ncol + 1, np.int32)
            st[1:] = np.cumsum(np.concatenate([np.full(ncol - 1, 2), [1]]))
            qi = np.empty(2 * ncol - 1, np.int32)
            qv = np.empty(2 * ncol - 1)
            qi[0::2], qv[0::2] = np.arange(ncol), diag
            qi[1::2], qv[1::2] = np.arange(1, ncol), off
            qkeep = (st, qi, qv)
        else:
            qkeep = (np.arange(ncol + 1, dtype=np.int32), np.arange(ncol, dtype=np.int32),
                     np.random.default_rng(1).uniform(0.0, 1.0, ncol))
        sp_.struct.q_dim = ncol
        sp_.struct.q_start = qkeep[0].ctypes.data_as(abi.c_i32p)
        sp_.struct.q_index = qkeep[1].ctypes.data_as(abi.c_i32p)
        sp_.struct.q_value = qkeep[2].ctypes.data_as(abi.c_f64p)
    return sp_, qkeep


def needed_bytes(n, m, nnz, fused, qp=False):
    """Bytes the two launches of a trial MUST move, counted from what the kernels read and write —
    next to SURVEY section 8(d)'s per-operation formula (algorithmic_bytes), which credits the fused launches with the
    re-reads of x, x+ and A'y that fusing the primal step into the A'y launch removed (it is the formula for the
    reference's separate level-1 calls).  A x+ launch: matrix 12 B/nnz, row pointers, gathered x+ (n), y / b / A x in,
    A x+ / y+ out, ySum in and out (7m).  A'y+ launch: matrix, column pointers, gathered y+ (m), x / x+ / A'y in, A'y+ out
    (4n); fused: + c / l / u / xSum in, x++ / xSum out (6n; QP: + the diagonal of Q)."""
    ax = 12 * nnz + 4 * (m + 1) + 8 * n + 8 * 7 * m
    aty = 12 * nnz + 4 * (n + 1) + 8 * m + 8 * 4 * n + (8 * 6 * n + (8 * n if qp else 0) if fused else 0)
    return ax, aty


def scaling_model(n, m, nnz, G, ax_us_1gpu, aty_us_1gpu, vec_us_1gpu):
    """Predicted microseconds per phase of one trial on G GPUs with the two-all-gathers layout, from
    the single-GPU kernel times and the node's link figures: xGMI 76.8 GB/s per direction and link (153.6 GB/s
    bidirectional, 7 links per GPU: MI355X_MICROARCH.md / the task statement), every rank sends its slice to the G-1
    peers over G-1 separate links, a flag hop ~3 us, a kernel ~3 us at least, a kernel boundary ~1.7 us; FIVE launches
    per trial with every rank on a GPU of its own (round 6, MeshArgs::fusedWait == 2: an exchange is one kernel — push,
    epoch, wait, copy), i.e. four boundaries (round 5: seven).  The measured `exchange_waits` of an N > 1 run are
    printed next to it."""
    link_gbs, hop_us, floor_us, boundary_us = 76.8, 3.0, 3.0, 1.7
    x_us = 8.0 * n / G / (link_gbs * 1e3) + hop_us   # the own slice to each peer, one link per peer
    y_us = 8.0 * m / G / (link_gbs * 1e3) + hop_us
    phases = {"primal_step_own_columns": max(floor_us, vec_us_1gpu / G), "X_allgather_x": x_us,
              "spmv_ax_dual_own_rows": max(floor_us, ax_us_1gpu / G), "Y_allgather_y": y_us,
              "spmv_aty_interact_own_columns": max(floor_us, aty_us_1gpu / G), "S_scalars_and_decision": hop_us + 1.0,
              "kernel_boundaries": 4 * boundary_us}
    total = sum(phases.values())
    return {"ranks": G, "us_per_phase": phases, "us_per_trial": total,
            "bytes_sent_per_rank_and_trial": 8 * (n + m) * (G - 1) // G,
            "assumptions": "xGMI %.1f GB/s per direction and link, flag hop %.0f us, kernel floor %.0f us, boundary %.1f us"
                           % (link_gbs, hop_us, floor_us, boundary_us)}


def _reference_highs_fn(lib_path, solver_name):
    """A solve function with the signature of the oracle's (problem, params, result) that runs the reference itself:
    highs_c_api.h of libhighs_reference.so.1 (interfaces/highs_c_api.h: Highs_passLp, Highs_setStringOptionValue, Highs_run)."""
    H = C.CDLL(lib_path)
    H.Highs_create.restype = C.c_void_p
    vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    H.Highs_passLp.argtypes = [vp, i32, i32, i32, i32, i32, f64, pd, pd, pd, pd, pd, pi, pi, pd]
    H.Highs_setStringOptionValue.argtypes = [vp, C.c_char_p, C.c_char_p]
    H.Highs_setBoolOptionValue.argtypes = [vp, C.c_char_p, i32]
    H.Highs_setIntOptionValue.argtypes = [vp, C.c_char_p, i32]
    H.Highs_setDoubleOptionValue.argtypes = [vp, C.c_char_p, f64]
    H.Highs_getIntInfoValue.argtypes = [vp, C.c_char_p, pi]
    H.Highs_run.argtypes = [vp]
    H.Highs_destroy.argtypes = [vp]

    def fn(p_ref, o_ref, r_ref):
        P, o, R = p_ref._obj, o_ref._obj, r_ref._obj
        h = H.Highs_create()
        H.Highs_setBoolOptionValue(h, b"output_flag", 0)
        H.Highs_setStringOptionValue(h, b"solver", solver_name.encode())
        H.Highs_setStringOptionValue(h, b"presolve", b"off")
        H.Highs_setDoubleOptionValue(h, b"kkt_tolerance", o.gap_tol)
        H.Highs_setIntOptionValue(h, b"pdlp_iteration_limit", int(o.iter_limit))
        cast = lambda q, t: C.cast(q, t)
        rc = H.Highs_passLp(h, P.num_col, P.num_row, int(P.a_start[P.num_col]), 1, int(P.sense), float(P.offset), cast(P.col_cost, pd),
                            cast(P.col_lower, pd), cast(P.col_upper, pd), cast(P.row_lower, pd), cast(P.row_upper, pd),
                            cast(P.a_start, pi), cast(P.a_index, pi), cast(P.a_value, pd))
        if rc not in (0, 1):
            H.Highs_destroy(h)
            return 1
        t0 = time.perf_counter()
        H.Highs_run(h)
        R.solve_seconds = time.perf_counter() - t0
        it = C.c_int32(0)
        H.Highs_getIntInfoValue(h, b"pdlp_iteration_count", C.byref(it))
        R.num_iter = it.value
        H.Highs_destroy(h)
        return 0
    return fn


def cpu_baseline(sp_struct, cfg, limits, solver_name="pdlp"):
    """Reference CPU pdlp (single thread) on a bounded sample of the same LP, as SURVEY §8(d) prescribes: the same
    solve at TWO iteration limits, iterations/s = the slope between them — set-up and the start-up phase (a check at
    every one of the first 10 iterations) cancel, what is left is the steady loop with its one check in 40."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    from highs_amd import abi
    ref_lib = os.path.join(ROOT, "integration", "_build", "libhighs_reference.so.1")
    if solver_name == "hipdlp" and os.path.exists(ref_lib):
        # the UNMODIFIED reference library (integration/Makefile `reference`: every TU compiled from the reference tree, travels
        # with the snapshot) through its own C API: Highs_passLp + solver=hipdlp + Highs_run
        kind, fn = "reference", _reference_highs_fn(ref_lib, "hipdlp")
    elif solver_name == "hipdlp":  # no reference build on this box: the oracle restatement
        kind, fn = "port", O.hipdlp_solve_fn()
    elif sp_struct.q_dim > 0:  # no PDLP-QP in the reference: this repository's own CPU restatement
        kind, fn = "port", O.oracle().pdlp_oracle_solve
    elif O.ref_available():
        kind, fn = "reference", O.ref().pdlp_ref_solve
    else:
        kind, fn = "port", O.oracle().pdlp_oracle_solve
    runs = []
    for lim in limits:
        params = abi.default_params(kkt_tolerance=1e-4, pdlp_iteration_limit=lim, solver=solver_name)
        R = abi.ResultHandle(sp_struct.num_col, sp_struct.num_row)
        t0 = time.perf_counter()
        rc = fn(C.byref(sp_struct), C.byref(params), C.byref(R.struct))
        wall = time.perf_counter() - t0
        if rc != 0 or R.num_iter <= 0:
            return None
        runs.append({"iteration_limit": lim, "iters": int(R.num_iter), "wall_s": wall, "loop_s": R.solve_seconds})
    (a, b) = runs
    if b["iters"] <= a["iters"] or b["wall_s"] <= a["wall_s"]:
        return None  # (the LP converged before the first limit: no slope)
    return {"value": (b["iters"] - a["iters"]) / (b["wall_s"] - a["wall_s"]), "unit": "it/s", "cores": 1, "kind": kind,
            "host_cpus": os.cpu_count(), "runs": runs,
            "single_run_value": b["iters"] / b["loop_s"],
            "sample": "slope between %d and %d PDHG iterations of the same LP (kkt 1e-4), %.1f s and %.1f s of wall clock, "
                      "1 thread" % (a["iters"], b["iters"], a["wall_s"], b["wall_s"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="b")
    ap.add_argument("--cpu-iters", type=int, default=None, help="CPU baseline sample size (0 disables)")
    ap.add_argument("--kernels", action="store_true", help="also print per-kernel timings to stderr")
    ap.add_argument("--solver", choices=["pdlp", "hipdlp"], default="pdlp",
                    help="pdlp = the headline path (cuPDLP-C semantics); hipdlp = the reference's Halpern PDHG path")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with `python -m torch.distributed.run --nproc-per-node %d "
                         "... bench.py --gpus %d`" % (args.gpus, world, args.gpus, args.gpus))
    cfg = CONFIGS[args.config]

    import torch
    from highs_amd import abi, solver

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PDLP path has no CPU fallback")
    # Test hooks (the 1-GPU build box): PDLP_BENCH_SINGLE_DEVICE=1 puts every rank on device 0 (the mesh
    # exchange only needs HIP IPC) and PDLP_BENCH_DIST_BACKEND=gloo replaces RCCL for the launcher-side
    # collectives, which RCCL cannot do with two ranks on one device.  The driver uses neither.
    if os.environ.get("PDLP_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("PDLP_BENCH_DIST_BACKEND", "nccl")
    tdev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def fresh_unique_id():
        """128-byte communicator id of one solver generation, made on rank 0 and broadcast by the launcher."""
        if world == 1:
            return None
        idbuf = (C.c_uint8 * 128)()
        if rank == 0:
            assert solver.lib().pdlp_mi355x_comm_unique_id(idbuf) == 0, solver.lib().pdlp_mi355x_last_error()
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=tdev)
        dist.broadcast(t, src=0)
        return (C.c_uint8 * 128)(*t.cpu().tolist())

    if cfg.get("qp") and args.solver != "pdlp":
        raise SystemExit("--config qp runs on the pdlp path only")
    sp_, qkeep = build_workload(args.config)
    params = abi.default_params(kkt_tolerance=1e-4, device=local_rank, solver=args.solver)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Start-up phase, reported separately: the reference checks convergence at EVERY one of the first 10
    # iterations (cupdlp_solver.c:953-962) and restarts for the first time; from iteration 40 on the
    # schedule is one check in 40.  The timed region therefore never starts before iteration 40, whatever
    # --warmup says (the hipGraph of a trial batch is captured in create(), i.e. in setup_seconds).
    # With N > 1 the direct xGMI exchange guards itself (known-answer test at creation, checksum of the replicated
    # iterate at every check): if it reports an inconsistency on this machine, every rank starts over with the
    # RCCL all-reduce exchange and the line says so.
    exchange_fallback = None
    chain = [None, "fences", "fences, a kernel per step"]
    for attempt in (chain + ["rccl"] if world > 1 and args.solver == "pdlp" else chain if world > 1 else [None]):
        if attempt == "fences":  # the direct exchange with system-scope release / acquire fences around the flags
            os.environ["PDLP_MI355X_MESH_FENCES"] = "1"
        elif attempt == chain[2]:  # ... and every step of an exchange a kernel of its own (round 5's folded form)
            os.environ["PDLP_MI355X_MESH_FENCES"] = "2"
        elif attempt:
            os.environ["PDLP_MI355X_EXCHANGE"] = attempt
        err, S = None, None
        try:
            uid = fresh_unique_id()
            t_setup = time.time()
            S = solver.DeviceSolver(problem_struct=sp_.struct, params=params, rank=rank, world=world, unique_id=uid)
            t_setup = time.time() - t_setup
            n, m, nnz = S.n, S.m, S.nnz
            sync()
            t0 = time.perf_counter()
            S.iterate(PRE_ROLL)
            sync()
            startup_ms = (time.perf_counter() - t0) * 1e3
        except RuntimeError as e:
            err = str(e)
        if dist is not None:
            flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = flag.item() > 0
        else:
            failed = err is not None
        if not failed:
            break
        if attempt == "rccl" or world == 1 or (attempt == chain[2] and args.solver != "pdlp"):
            raise SystemExit("bench.py: the solver failed: %s" % err)
        exchange_fallback = ("%sdirect xGMI exchange%s rejected on this machine (%s); next: %s"
                             % (exchange_fallback + " | " if exchange_fallback else "", " with " + attempt if attempt else "",
                                err or "error on another rank",
                                "the same with release/acquire fences" if attempt is None else
                                "the same with a kernel per exchange step" if attempt == "fences" else "RCCL all-reduce"))
        if S is not None:
            try:
                S.close()
            except Exception:
                pass
    S.iterate(args.warmup)

    def timed(iters):
        sync()
        t0 = time.perf_counter()
        r = S.iterate(iters)
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return r, el

    # The K steps the flags ask for, timed exactly as the contract says.  `value` must carry the reference's check
    # schedule (one check iteration in 40): when the K-step window is not a whole number of check periods starting
    # on one (the driver's --steps 20 --warmup 5 covers iterations 45..65: no check at all), `value` comes from a
    # second window of >= 400 iterations that is, and the K-step window is reported next to it.
    win_start = PRE_ROLL + args.warmup
    kst, k_elapsed = timed(args.steps)
    k_window = {"value": kst.iters / k_elapsed, "unit": "it/s", "iters": int(kst.iters), "checks": int(kst.checks),
                "ms_per_step": k_elapsed * 1e3 / kst.iters, "window": "iterations %d..%d" % (win_start, win_start + int(kst.iters))}
    if win_start % 40 == 0 and args.steps % 40 == 0 and args.steps >= 400:
        st, elapsed, val_start = kst, k_elapsed, win_start
    else:
        pos = win_start + int(kst.iters)
        if pos % 40:
            S.iterate(40 - pos % 40)
            pos += 40 - pos % 40
        st, elapsed = timed(max(400, (args.steps + 39) // 40 * 40))
        val_start = pos

    exchange = {0.0: "none", 1.0: "RCCL all-reduce of A'y", 2.0: "direct xGMI mesh (all-gather x+, reduce-scatter A'y+)",
                3.0: "direct xGMI mesh, two all-gathers (x+ column slices, y+ row blocks; A'y from the rank's column block)"}[
        float(S.stage("exchange")[0])]
    exchange_waits = None
    if world > 1 and exchange.startswith("direct"):
        # device-side wall-clock time a rank spent waiting for its peers' flags, per hot-loop exchange (rank 0's view)
        ph = S.stage("mesh_phases")
        exchange_waits = {"unit": "us per wait (rank 0, 100 MHz device clock)", "X_allgather_x": ph[0],
                          "P_reduce_scatter_aty": ph[1],  # (two-all-gathers layout: the wait for the y+ row blocks)
                          "S_scalars": ph[2], "waits": [int(ph[3]), int(ph[4]), int(ph[5])]}
    rank_consistent = None
    if dist is not None:
        # every rank must hold bit-identical iterates (same decisions everywhere): compare a checksum of x
        import numpy as np
        # (cuPDLP path: x is replicated; HiPDLP path: the reflected x is the vector every rank holds in full)
        rep = "x" if args.solver == "pdlp" else "x_reflected"
        chk = float(np.frombuffer(S.get(rep, n).tobytes(), dtype=np.uint64).astype(np.float64).sum())
        lo = torch.tensor([chk], dtype=torch.float64, device=tdev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rank_consistent = bool(lo.item() == hi.item())
    b_iter, b_ax, b_aty = (algorithmic_bytes if args.solver == "pdlp" else algorithmic_bytes_hipdlp)(n, m, nnz)
    if cfg.get("qp"):
        b_iter += 8 * n  # the primal step also reads the diagonal of Q
    if cfg.get("banded"):  # N x+ SpMV: 12 B per off-diagonal entry (both triangles) + row pointers + x, x+, N x in, N x+ out; + N x in the primal step
        b_iter += 12 * 2 * (n - 1) + 4 * (n + 1) + 8 * 5 * n
    ms_step = elapsed * 1e3 / st.iters
    # dominant kernel, timed live with HIP events on the solver's own stream, IN the loop (same
    # kernel sequence and cache state as the timed region; what rocprofv3 --kernel-trace reports)
    iso_ax = S.time_kernel("spmv_ax", 50)
    iso_aty = S.time_kernel("spmv_aty", 50)
    k_ax, k_aty, prof_launches = iso_ax, iso_aty, 0
    if world == 1:
        S.stage("profile_on")
        ps = S.iterate(max(200, min(args.steps, 1000)))
        S.stage("profile_off")
        if ps.reserved[0] > 0:
            k_ax, k_aty, prof_launches = ps.spmv_ax_ms, ps.spmv_aty_ms, int(ps.reserved[0])
    fused = args.solver == "pdlp" and world == 1 and int(S.stage("trial_launches")[0]) == 2
    if fused:
        # the A'y launch of the 2-launch trial also holds the grid barrier, the decision and the NEXT primal step: its
        # algorithmic bytes are SURVEY §8(d)'s terms for that work — SpMV A'y + movement/interaction (b_aty), the
        # primal step R5W1 (6n words) and the x half of the running average R2W1 (3n words)
        b_aty = b_aty + 8 * 9 * n + (8 * n if cfg.get("qp") else 0)
    dom_name, dom_ms, dom_bytes = ("spmv_ax_dual", k_ax, b_ax) if k_ax >= k_aty else (
        "spmv_aty_interact_decide_primal" if fused else "spmv_aty_interact", k_aty, b_aty)
    persistent = args.solver == "pdlp" and world == 1 and int(S.stage("trial_launches")[0]) == 0
    if persistent:
        # the whole trial batch is ONE launch (pdlp_small.hip): there is no per-SpMV launch to time in the loop; the
        # launch's algorithmic bytes per trial are the iteration's, its duration per trial the loop's (check iterations
        # included, so this understates the kernel a little).  k_ax / k_aty below are the stand-alone SpMV kernels
        # re-launched in isolation — what the 3-launch loop would run.
        dom_name, dom_ms, dom_bytes = "trials_persistent(per trial)", ms_step * st.iters / max(int(st.trials), 1), b_iter
    # what the two launches really have to move (needed_bytes) next to SURVEY's per-operation formula
    nb_ax, nb_aty = needed_bytes(n, m, nnz, fused, bool(cfg.get("qp")))
    if fused:  # bounds that all columns of a block share are not loaded per column (round 6): 8 bytes less for each of them
        ub = S.stage("uniform_bound_columns")
        nb_aty -= 8 * int(ub[0] + ub[1])
    if args.solver != "pdlp":  # (the Halpern launches: their formula already counts what the fused kernels move)
        nb_ax, nb_aty = b_ax, b_aty
    dom_needed = b_iter if persistent else nb_ax if dom_name == "spmv_ax_dual" else nb_aty
    if world > 1:  # each rank streams 1/world of the matrix
        dom_bytes = dom_bytes / world
        dom_needed = dom_needed / world
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    copy_gbs = None
    if args.solver == "pdlp" and world == 1:
        # measured device-copy ceiling: 2 x 512 MiB hipMemcpyDtoD, beyond the Infinity Cache
        copy_gbs = 2 * 512 * 2**20 / (S.time_kernel("copy", 10) * 1e-3) / 1e9
    traffic = traffic_miss = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if args.solver == "hipdlp":  # same SpMV kernels with the Halpern epilogues
        dom_name = {"spmv_ax_dual": "spmv_ax_halpern_dual", "spmv_aty_interact": "spmv_aty_halpern_primal"}.get(dom_name, dom_name)
    traffic_key = "trials_persistent_per_trial" if persistent else dom_name
    if os.path.exists(tpath):
        try:
            # (counters exist per configuration; the HiPDLP path was profiled on the headline LP only — another config: none)
            cfg_key = args.config if (args.solver == "pdlp" or args.config == "b") else None
            tj = json.load(open(tpath))
            traffic = tj.get(cfg_key, {}).get(traffic_key) if cfg_key else None
            # the cross-check of the guide's x2 on FETCH_SIZE: L2 misses of the same launches x 128 B lines
            raw = tj.get("raw_per_launch", {}).get(cfg_key if args.solver == "pdlp" else "hipdlp_b", {}).get(traffic_key, {})
            traffic_miss = raw.get("TCC_MISS_sum", {}).get("mean_working")
            traffic_miss = traffic_miss * 128.0 if traffic_miss is not None and not persistent else None
        except Exception:
            traffic = traffic_miss = None
    out = {
        "metric": "PDHG iterations/sec" if args.solver == "pdlp" else "PDHG iterations/sec (HiPDLP path)",
        "value": st.iters / elapsed, "unit": "it/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["name"], "m": m, "n": n, "nnz": nnz,
                   "parallelism": "single GPU" if world == 1 else "row-block x%d, %s" % (world, exchange),
                   "options": "presolve=off, kkt_tolerance=1e-4, adaptive step + restarts (reference defaults)"},
        "trial_steps": int(st.trials), "rejected_trials": int(st.trials - st.iters), "checks": int(st.checks),
        "trial_launches": int(S.stage("trial_launches")[0]) if args.solver == "pdlp" else 2,
        # launches of one check iteration queued behind the trial batches (0: the host drives the checks and waits for them)
        "check_launches": int(S.stage("check_launches")[0]) if args.solver == "pdlp" else None,
        "restarts": int(st.restarts), "setup_seconds": t_setup, "ranks_bit_identical": rank_consistent,
        "exchange_fallback": exchange_fallback, "exchange": exchange if world > 1 else None, "exchange_waits": exchange_waits,
        "startup_ms_first_40": startup_ms,
        # `value` / `ms_per_step` / `checks`: this window — whole check periods of the reference's schedule
        "timed_window": "iterations %d..%d" % (val_start, val_start + int(st.iters)), "timed_steps": int(st.iters),
        "steps_note": "`steps` / `warmup` echo the flags; `value` and `ms_per_step` are over `timed_steps` iterations (whole check "
                      "periods of the reference's schedule); the flags' own K-step window is `window_of_the_K_steps`",
        "window_of_the_K_steps": k_window,
        "iter_algorithmic_bytes": b_iter,
        "iter_hbm_gbs": b_iter / (ms_step * 1e-3) / 1e9,
        "iter_hbm_frac_of_peak": b_iter / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world,
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_tcc_miss_x_128B": traffic_miss,
                     "traffic_source": "profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of tools/make_profiles.sh, not "
                                       "measured in this run)" if traffic is not None else None,
                     "algorithmic_bytes_per_launch": dom_bytes,
                     # the bytes this launch must move, counted from what the kernel reads and writes (needed_bytes): the
                     # SURVEY formula above credits a fused launch with the re-reads that the fusion removed
                     "needed_bytes_per_launch": dom_needed,
                     "frac_needed": dom_needed / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic_over_needed": (traffic / dom_needed) if traffic else None,
                     "measured_copy_ceiling_gbs": copy_gbs,
                     "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                     "avg_launch_ms": dom_ms, "timed_launches_in_loop": prof_launches,
                     "other_kernels_ms": {"spmv_ax_dual": k_ax, "spmv_aty_interact": k_aty},
                     "per_kernel": {"spmv_ax_dual": {"ms": k_ax, "algorithmic_bytes": b_ax / world, "frac": b_ax / world / (k_ax * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                    ("spmv_aty_interact_decide_primal" if fused else "spmv_aty_interact"):
                                        {"ms": k_aty, "algorithmic_bytes": b_aty / world, "frac": b_aty / world / (k_aty * 1e-3) / 1e9 / HBM_PEAK_GBS}},
                     "isolated_relaunch_ms": {"spmv_ax_dual": iso_ax, "spmv_aty_interact": iso_aty}},
    }
    if args.solver == "pdlp" and not cfg.get("qp"):
        # the model of DESIGN.md §6 for this LP, fed with THIS run's kernel times (us): the two SpMV launches re-launched in
        # isolation on one GPU (what a rank runs on its 1/G share; the fused A'y launch would carry the primal step, which the
        # sharded sequence runs as its own kernel) and the stand-alone primal-step kernel
        if world == 1:
            base = (iso_ax * 1e3, iso_aty * 1e3, S.time_kernel("decide_primal", 30) * 1e3)
        else:  # (a rank of a sharded run measures its own share of the two SpMVs: scaled back to one GPU; the primal step is
            # timed full length on this rank's device — the stand-alone kernel a single GPU would run)
            base = (iso_ax * 1e3 * world, iso_aty * 1e3 * world, S.time_kernel("primal_step", 30) * 1e3)
        out["scaling_model"] = {("G=%d" % G): scaling_model(n, m, nnz, G, *base) for G in ((world,) if world > 1 else (2, 4, 8))}
        out["scaling_model"]["inputs_us_one_gpu"] = {"spmv_ax_dual": base[0], "spmv_aty_interact": base[1], "primal_step": base[2],
                                                     "source": "measured in this run (isolated re-launches)"}
        if world == 1:
            one = ms_step * 1e3 * st.iters / max(int(st.trials), 1)
            for G in (2, 4, 8):
                out["scaling_model"]["G=%d" % G]["speedup_vs_this_run"] = one / out["scaling_model"]["G=%d" % G]["us_per_trial"]
        if world > 1:
            out["scaling_model"]["measured_us_per_trial"] = ms_step * 1e3 * st.iters / max(int(st.trials), 1)
    if cfg.get("qp"):
        out["metric"] = "PDHG iterations/sec (QP prox path)"
        out["parity"] = ("unpinned: the reference has no PDLP for QPs (HighsOptions.cpp:1178-1181); optimal objectives are "
                         "pinned on the reference's QP solver for small instances (tests/golden/reference_qp.json)")
    if args.solver == "hipdlp":
        out["config"]["options"] = "presolve=off, kkt_tolerance=1e-4, Halpern restarts + PID primal weight (reference defaults)"
    if (cfg.get("structured") or cfg.get("staircase")) and world == 1 and args.solver == "pdlp":
        # the dense linking rows (> 256 nonzeros) are cut into segment tasks that run as extra workgroups of the SAME
        # launch: what they add to the plain A x (they hold 20 % of the nonzeros of this LP)
        full, nolong = S.time_kernel("spmv_ax_plain", 50), S.time_kernel("spmv_ax_plain_nolong", 50)
        out["long_majors"] = {"spmv_ax_plain_ms": full, "without_the_long_majors_ms": nolong, "share": (full - nolong) / full,
                              "share_of_nonzeros": (256 * 4096 / nnz) if cfg.get("structured") else None}
    if args.kernels and rank == 0 and world == 1 and args.solver == "pdlp":
        ks = {k: S.time_kernel(k, 50) for k in ("decide_primal", "primal_step", "spmv_ax", "spmv_aty", "decide", "trial",
                                                "spmv_ax_plain", "spmv_aty_plain", "copy")}
        ks["copy_GBs"] = 2 * 512 * 2**20 / (ks["copy"] * 1e-3) / 1e9
        print(json.dumps({"kernels_ms": ks}), file=sys.stderr)
        out["kernels_ms"] = ks
    if rank == 0 and world == 1:
        # two iteration limits (whole check periods + 1: the reference stops at limit - 1), about 10-30 s of one core
        budget = args.cpu_iters if args.cpu_iters is not None else {"b": 201, "qp": 361, "c": 361, "d": 361, "e": 361, "f": 361, "qpn": 361}.get(args.config, 3001)
        if budget > 0:
            lo = max(41, (budget - 1) * 2 // 5 // 40 * 40 + 1)
            hi = max(budget, lo + 40)
            if args.solver == "hipdlp":  # (whole blocks of 40 Halpern steps)
                lo, hi = lo - 1, hi - 1
            cb = cpu_baseline(sp_.struct, cfg, (lo, hi), args.solver)
            out["cpu_baseline"] = cb
            if cb:
                out["speedup_vs_cpu_pdlp"] = out["value"] / cb["value"]
    S.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

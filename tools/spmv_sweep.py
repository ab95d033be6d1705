#!/usr/bin/env python3
"""SpMV layout / pipeline sweep on the bench workload (development tool; one process, one solver per variant).

For every variant (environment switches read by the library at solver creation / launch) it reports the
PDHG iteration time, the in-loop HIP-event times of the two fused SpMVs, their isolated re-launch times
and whether the iterate after 40 iterations is bit-identical to the first variant's.
    python tools/spmv_sweep.py [--variants "slab=1;slab=1,w=16;slab=1,xcd=0;slab=0"]
"""
import argparse
import hashlib
import json
import os
os.environ.setdefault("PDLP_MI355X_DEV", "1")  # development switches below
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highs_amd import abi, solver  # noqa: E402

ENV = {"slab": "PDLP_MI355X_SLAB", "w": "PDLP_MI355X_SLAB_W", "xcd": "PDLP_MI355X_XCD_MAP", "graph": "PDLP_MI355X_GRAPH",
       "gpusetup": "PDLP_MI355X_GPU_SETUP", "pipe": "PDLP_MI355X_SLAB_PIPE"}
DEFAULT = "slab=1;slab=1,w=15;slab=1,w=16;slab=1,w=18;slab=1,xcd=0;slab=1,xcd=1;slab=0"

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1_000_000)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nnz", type=int, default=8_000_000)
ap.add_argument("--iters", type=int, default=600)
ap.add_argument("--solver", choices=["pdlp", "hipdlp"], default="pdlp")
ap.add_argument("--variants", default=DEFAULT)
ap.add_argument("--structured", action="store_true", help="the block-angular LP of bench.py --config c instead")
args = ap.parse_args()

if args.structured:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from lpgen import structured_lp
    sp_ = abi.ProblemHandle(structured_lp(1))
else:
    sp_ = solver.SyntheticProblem(args.m, args.n, args.nnz, 1)
ref_hash = None
for spec in args.variants.split(";"):
    for k in ENV.values():
        os.environ.pop(k, None)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        os.environ[ENV[k]] = v
    t0 = time.time()
    S = solver.DeviceSolver(problem_struct=sp_.struct, params=abi.default_params(kkt_tolerance=1e-4, solver=args.solver))
    setup = time.time() - t0
    S.iterate(40)
    h = hashlib.sha256(S.get("x" if args.solver == "pdlp" else "x_reflected", S.n).tobytes()).hexdigest()[:12]
    if ref_hash is None:
        ref_hash = h
    S.iterate(260)
    st = S.iterate(args.iters)
    rec = {"variant": spec, "us_per_iter": 1e3 * st.gpu_ms / st.iters, "trials_per_iter": st.trials / st.iters,
           "bit_identical": h == ref_hash, "setup_s": round(setup, 2)}
    if args.solver == "pdlp":
        S.stage("profile_on")
        ps = S.iterate(200)
        S.stage("profile_off")
        rec.update(loop_ax_us=1e3 * ps.spmv_ax_ms, loop_aty_us=1e3 * ps.spmv_aty_ms)
        rec.update(iso_ax_us=1e3 * S.time_kernel("spmv_ax", 30), iso_aty_us=1e3 * S.time_kernel("spmv_aty", 30),
                   iso_ax_plain_us=1e3 * S.time_kernel("spmv_ax_plain", 30),
                   iso_aty_plain_us=1e3 * S.time_kernel("spmv_aty_plain", 30))
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
    S.close()

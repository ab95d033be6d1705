#!/usr/bin/env python3
"""Kernel micro-benchmark on the bench workload: times individual kernels with HIP events
(pdlp_mi355x_time_kernel) — also the command run under rocprofv3 for profiles/."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highs_amd import abi, solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1_000_000)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nnz", type=int, default=8_000_000)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--iters", type=int, default=0, help="also run this many PDHG iterations first")
ap.add_argument("--solver", choices=["pdlp", "hipdlp"], default="pdlp")
ap.add_argument("--structured", action="store_true", help="the block-angular LP of bench.py --config c instead")
ap.add_argument("--config", default=None, help="a bench.py configuration (b, a, c, d, qp, qpn) instead of --m/--n/--nnz")
ap.add_argument("--kernels", default="primal_step,spmv_ax,spmv_aty,decide,trial,spmv_ax_plain,spmv_aty_plain")
args = ap.parse_args()
keep = None
if args.config:
    import bench
    sp_, keep = bench.build_workload(args.config)
elif args.structured:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from lpgen import structured_lp
    sp_ = abi.ProblemHandle(structured_lp(1))
else:
    sp_ = solver.SyntheticProblem(args.m, args.n, args.nnz, 1)
S = solver.DeviceSolver(problem_struct=sp_.struct, params=abi.default_params(kkt_tolerance=1e-4, solver=args.solver))
out = {}
if args.iters:
    st = S.iterate(args.iters)
    out["iterate_ms_per_iter"] = st.gpu_ms / st.iters
    out["iters"], out["trials"], out["trial_launches"] = int(st.iters), int(st.trials), int(S.stage("trial_launches")[0])
for k in args.kernels.split(","):
    out[k] = S.time_kernel(k, args.reps)
print(json.dumps(out))

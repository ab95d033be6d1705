#!/usr/bin/env python3
"""tools/bringup_multi_gpu.py STAGE ... — one stage of the staged multi-GPU bring-up (driven by tools/bringup_multi_gpu.sh).
Every stage prints ONE JSON line {"stage": ..., "ok": true|false, ...} and exits 0 / 1, so that a failed first run on an
8-GPU node says WHICH layer failed: peer access, HIP IPC + the mesh's known-answer test, the flag hop, RCCL, the solver.

  peers                 (one process)   device count, hipDeviceCanAccessPeer matrix, HSA_ENABLE_IPC_MODE_LEGACY
  mesh RANK WORLD IDHEX (one per rank)  arenas exported / mapped (HIP IPC across processes), known-answer self-test of the
                                        three exchanges, then 200 scalar all-reduces: microseconds per flag round trip
  rccl                  (under torchrun) ncclAllReduce of n + 1 doubles with N ranks through torch.distributed (backend nccl = RCCL)
"""
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def out(stage, ok, **kw):
    print(json.dumps(dict(stage=stage, ok=bool(ok), **kw)), flush=True)
    sys.exit(0 if ok else 1)


def peers():
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    rc = hip.hipGetDeviceCount(C.byref(n))
    if rc != 0 or n.value <= 0:
        out("peers", False, error="hipGetDeviceCount rc=%d count=%d" % (rc, n.value))
    mat = []
    for a in range(n.value):
        row = []
        for b in range(n.value):
            can = C.c_int(0)
            if a != b:
                hip.hipDeviceCanAccessPeer(C.byref(can), a, b)
            row.append(int(can.value) if a != b else 1)
        mat.append(row)
    full = all(all(r) for r in mat)
    out("peers", full or n.value == 1, devices=n.value, can_access_peer=mat,
        HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        note=None if full else "some device pairs report no peer access: the mesh exchange maps arenas through HIP IPC across "
                               "processes and needs peer access inside one process; RCCL is the fall-back")


def mesh(rank, world, idhex):
    from highs_amd import abi, solver
    uid = (C.c_ubyte * 128).from_buffer_copy(bytes.fromhex(idhex))
    hip = C.CDLL("libamdhip64.so")
    nd = C.c_int(0)
    hip.hipGetDeviceCount(C.byref(nd))
    dev = rank % max(nd.value, 1)  # (fewer devices than ranks: folded, as on the one-GPU test box)
    sp_ = solver.SyntheticProblem(20000, 20000, 160000, 3)
    t0 = time.time()
    try:
        S = solver.DeviceSolver(problem_struct=sp_.struct, params=abi.default_params(kkt_tolerance=1e-4, device=dev), rank=rank,
                                world=world, unique_id=uid)
    except RuntimeError as e:
        out("mesh", False, rank=rank, error=str(e))
    ex = int(S.stage("exchange")[0])  # 3 = mesh (two all-gathers), 2 = mesh (partials), 1 = RCCL: the self-test voted the mesh out
    hop = float(S.stage("mesh_hop_us")[0]) if ex >= 2 else None
    st = S.iterate(200)
    S.close()
    out("mesh", ex >= 2, rank=rank, world=world, device=dev, folded=nd.value < world, exchange={1: "rccl", 2: "mesh-partials", 3: "mesh-two-allgathers"}.get(ex, ex),
        create_seconds=round(time.time() - t0, 2), scalar_allreduce_us=hop, iterations=int(st.iters), trials=int(st.trials),
        note="exchange = rccl means the known-answer test of the direct exchange failed on some rank (see stderr)" if ex < 2 else None)


def rccl():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dist.init_process_group("nccl")
    n = 1_000_001
    x = torch.full((n,), float(rank + 1), dtype=torch.float64, device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    good = bool((x == world * (world + 1) / 2).all().item())
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        out("rccl", good, world=world, doubles=n, allreduce_us=round(us, 1))
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    st = sys.argv[1]
    if st == "peers":
        peers()
    elif st == "mesh":
        mesh(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    elif st == "rccl":
        rccl()
    else:
        raise SystemExit(__doc__)

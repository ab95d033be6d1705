#!/usr/bin/env python3
"""tools/slab_blocks.py RAW [RAW...] — the per-block phase table that PDLP_MI355X_SLAB_PROF=<path> leaves behind (two launches
x 1024 logical blocks x {launches, ticks to the end of the stream, of the epilogue, of the grid barrier, of the kernel};
100 MHz wall clock) as JSON: per launch the per-block mean microseconds of every phase, indexed like the slab partition
(logical block), so that a slow block can be looked up in the host-built layout."""
import json
import sys

import numpy as np

for path in sys.argv[1:]:
    t = np.fromfile(path, dtype=np.uint64).reshape(2, 1024, 8)
    out = {}
    for half, name in enumerate(["ax_dual", "aty_fused"]):
        n = t[half, :, 0].astype(np.float64)
        tn = n[512:]
        if (tn > 0).any():  # rows 512..: when the task workgroups of the launch were done
            tu = t[half, 512:, 1][tn > 0].astype(np.float64) * 0.01 / tn[tn > 0]
            out[name + "_tasks_done"] = tu.round(2).tolist()
            print(path, name, "task workgroups: %d, done at mean %.2f us, first %.2f, last %.2f" % (len(tu), tu.mean(), tu.min(), tu.max()))
        n = n.copy(); n[512:] = 0
        used = n > 0
        if not used.any():
            continue
        us = t[half][used][:, 1:5].astype(np.float64) * 0.01 / n[used, None]
        out[name] = {"blocks": int(used.sum()), "stream": us[:, 0].round(2).tolist(), "epilogue": us[:, 1].round(2).tolist(),
                     "barrier": us[:, 2].round(2).tolist(), "kernel": us[:, 3].round(2).tolist()}
    json.dump(out, open(path + ".json", "w"))
    for k, v in out.items():
        s = np.array(v["stream"])
        print(path, k, "stream mean %.2f max %.2f; slowest blocks:" % (s.mean(), s.max()), np.argsort(-s)[:12].tolist())

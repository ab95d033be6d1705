"""bench.py under the DRIVER's flags (--steps 20 --warmup 5): the short window must measure the loop, not
start-up — the value has to agree with the long-run rate of the same process and with the sum of the in-loop
kernel times (what rocprofv3 --kernel-trace reports for the three launches of a trial)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_flags_measure_the_steady_loop():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--cpu-iters", "0", "--kernels"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["steps"] == 20 and rec["warmup"] == 5 and rec["n_gpus"] == 1
    assert rec["timed_window"] == "iterations 45..65" and rec["checks"] == 0
    ss = rec["steady_state"]
    assert ss["iters"] >= 400 and ss["checks"] == ss["iters"] // 40
    # no check iteration falls into the 20-step window, so it may be a little faster than the long-run rate
    # (one check per 40 iterations), never slower than 10 % below it
    assert 0.90 * ss["value"] <= rec["value"] <= 1.20 * ss["value"], (rec["value"], ss["value"])
    # the three launches of a trial, timed in the loop / in isolation, add up to the step time within 15 %
    k = rec["roofline"]["other_kernels_ms"]
    trial = k["spmv_ax_dual"] + k["spmv_aty_interact"] + rec["kernels_ms"]["decide_primal"]
    assert abs(rec["ms_per_step"] - trial) <= 0.15 * rec["ms_per_step"], (rec["ms_per_step"], trial)
    r = rec["roofline"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert rec["startup_ms_first_40"] > 0 and rec["setup_seconds"] > 0

"""bench.py under the DRIVER's flags (--steps 20 --warmup 5): the short window must measure the loop, not
start-up, and `value` must include the reference's check schedule — it comes from a window of whole check periods,
the 20-step window is reported next to it, and both agree with the sum of the in-loop kernel times (what rocprofv3
--kernel-trace reports for the launches of a trial)."""
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
    # `value` carries the reference's check schedule: its window is whole 40-iteration check periods
    assert rec["timed_window"] == "iterations 80..480" and rec["timed_steps"] == 400
    assert rec["checks"] == rec["timed_steps"] // 40
    assert abs(rec["value"] - 1e3 / rec["ms_per_step"]) < 1e-6 * rec["value"]
    # the K = 20 steps the flags ask for are timed too (iterations 45..65: no check falls into them), and may only
    # be a little faster than `value`, never slower than 10 % below it
    kw = rec["window_of_the_K_steps"]
    assert kw["window"] == "iterations 45..65" and kw["iters"] == 20 and kw["checks"] == 0
    assert 0.90 * rec["value"] <= kw["value"] <= 1.20 * rec["value"], (kw["value"], rec["value"])
    # the launches of a trial, timed in the loop, add up to the step time within 15 % (2 launches when the trial is
    # fused — the A' y kernel then holds the decision and the next primal step —, else 3)
    k = rec["roofline"]["other_kernels_ms"]
    trial = k["spmv_ax_dual"] + k["spmv_aty_interact"] + (0.0 if rec["trial_launches"] == 2 else rec["kernels_ms"]["decide_primal"])
    assert abs(rec["ms_per_step"] - trial) <= 0.15 * rec["ms_per_step"], (rec["ms_per_step"], trial)
    r = rec["roofline"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert rec["startup_ms_first_40"] > 0 and rec["setup_seconds"] > 0

"""bench.py prints ONE JSON line with the fields the driver and the judge read (headline metric, roofline object with its
live-measured launch duration, CPU baseline on a bounded sample)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("crystal structures/sec") and d["unit"] == "structures/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 256 * 4 / (1000 * d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]   # B*K / (T * elapsed)
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["launches"] == 4 * 2 * 6 * rf["concurrent_streams"] and rf["avg_launch_ms"] > 0
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "structures/s" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    assert d["config"]["final_state_finite"] is True
    # what the timed region leaves out is said in the line, and priced (round 6): recording off in the window, a record=True figure beside it
    assert d["config"]["record"] is False and "log-probabilities" in d["config"]["record_note"] and d["config"]["gc_disabled_in_window"] is True
    rc = d["extra"]["recording_chain"]
    assert "error" not in rc, rc
    assert rc["record"] is True and rc["value"] > 0 and rc["states_kept"] >= 4 and rc["final_state_bit_identical_to_timed_chain"] is True
    assert {"log_prob_l", "log_prob_t", "log_prob_x"} <= set(rc["fields_per_state"])
    # BASELINE configs[2] rides along with a roofline object over the WHOLE micro-step
    ft = d["extra"]["fine_tune"]
    assert "error" not in ft, ft
    assert ft["roofline"]["scope"].startswith("whole micro-step") and 0 < ft["roofline"]["frac"] < 1

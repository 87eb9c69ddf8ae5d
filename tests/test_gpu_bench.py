"""The driver's own command, `python bench.py --steps 20 --warmup 5`, end to end in a child
process: one JSON line with the contract's fields, the roofline tied to the library that ran, the
direct-form figure and the CPU baseline beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_at_the_drivers_arguments():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MHX_SPLIT", "MHX_TSPLIT"):
        env.pop(k, None)
    env["MHX_BENCH_CPU_THREADS"] = "4"  # (a short CPU leg: the figure itself is not asserted)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f64"
    assert d["unit"] == "chain-steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["chains_per_gpu"] == 4096 and d["config"]["n_points"] == 100000
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / 4096 - 1.0) < 1e-9
    r = d["roofline"]
    assert r["bound"] == "fp64_valu_issue" and r["unit"] == "G wave-instr/s" and 0.3 < r["frac"] <= 1.0 and r["instr_source_stale"] is False
    assert r["kernel_ms_per_launch"] * 1e-3 <= d["ms_per_step"] * 1e-3 * 20 * 1.001
    assert d["build"]["id"].startswith("csrc:")
    assert d["value_direct_form"] > 0 and d["direct_form"]["ratio_to_value"] > 1.0
    er = d["early_reject"]   # (a walk's iterations 6-25 accept nothing: every sweep is left early)
    assert "+early-reject" in er["kernel"] and d["value_early_reject"] > 1.5 * d["value"], er
    sb = d["small_batches"]
    for k in ("walkers_1", "walkers_64"):   # (one persistent launch per portion: well under the 16-20 us of two launches)
        assert "persistent tsplit" in sb[k]["kernel"] and 0 < sb[k]["us_per_iteration"] < 14 and not sb[k]["trapped"], sb
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["single_core"]["value"] > 0
    assert d["value"] > 1e3 * c["value"]

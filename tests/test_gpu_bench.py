"""bench.py's contract on a real GPU (small shape, seconds): ONE JSON line on stdout with the driver's keys, the roofline and CPU-baseline
objects, and the sequences-in-flight section."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_single_gpu_bench_prints_one_json_line_with_the_contract_keys(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--steps", "12", "--warmup", "3", "--prompt-len", "16",
                        "--cpu-steps", "2", "--concurrent", "3", "--profile-iters", "4", "--mode", mode],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True and d["value"] > 0
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) / d["value"] < 0.02          # tokens/s of EXACTLY `steps` timed steps
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert isinstance(rf["traffic_measured_in_run"], bool)
    if rf["traffic_measured_in_run"]:                       # the rocprofv3 FETCH_SIZE probe of the dominant kernel ran
        assert rf["traffic"] > 0 and "rocprofv3" in rf["traffic_source"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "tokens/s"
    sf = d["sequences_in_flight"]
    assert sf["n"] == 3 and sf["tokens_per_s"] > 0
    assert sf["sequence0_tokens_vs_single_run"]["identical_prefix"] == sf["sequence0_tokens_vs_single_run"]["compared"] > 0

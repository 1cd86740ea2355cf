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
    mm = rf["measured_model"]                               # hardware-only bound next to the current kernels taken apart by this run's cycle stamps
    assert mm["hardware_bound"]["tokens_per_s"] > 0 and 0 < mm["hardware_bound"]["achieved_frac_of_bound"] <= 1.0
    assert set(mm["per_kernel"]) >= {"wo+residual GEMV", "w2+residual GEMV", "norm+output GEMV", "attention"}
    assert all("hardware_bound_us" in v for k, v in mm["per_kernel"].items() if k != "attention")
    # the roofline object is the kernel SYMBOL with the largest share of the token's GPU time; the heaviest launch class sits next to it
    assert 0 < rf["share_of_gpu_time"] <= 1 and rf["kernel"] in rf["symbols"] and rf["dominant_class"]["frac"] > 0
    assert sf["schedule"] == "throughput" and sf["latency_forms_same_run"]["tokens_per_s"] > 0 and sf["n2"]["n"] == 2
    rp = d["config"]["timed_region_repeats_ms_per_step"]   # three repeats of the same K steps, the median one reported
    assert len(rp) == 3 and sorted(rp)[1] == pytest.approx(d["ms_per_step"], rel=1e-3)
    if mode == "exact":                                    # batched exact decode: n prompts per pass over the weights, sequence 0 = the single run's prompt
        nw = d["config"]["norm_item_walk"]                 # rows of the fused RMSNorm that needed the record walk: a minority
        assert nw["rows"] > 0 and 0 <= nw["fallback_rows"] <= nw["rows"] // 2
        sb = d["sequences_in_flight_batched"]
        assert [r["n"] for r in sb["runs"]] == [2, 4, 8, 16, 32, 64, 128] and sb["weights_second_copy_bytes"] > 0
        for r in sb["runs"]:
            assert r["tokens_per_s"] > 0 and r["sequence0_tokens_vs_single_run"]["identical_prefix"] == r["sequence0_tokens_vs_single_run"]["compared"] > 0
        assert set(sb["runs"][-1]["kernels_us"]) >= {"norm+wqkv+rope", "attention", "w2+residual", "norm+output"}
    else:
        assert "sequences_in_flight_batched" not in d


def test_bench_gpus_2_as_a_plain_command_on_one_gpu_with_gloo():
    """the driver's N-GPU command line -- `python bench.py --gpus N --steps K --warmup W`, no launcher -- on a one-GPU box: the two ranks
    bench.py spawns share GPU 0 and exchange through gloo (LNB_PIPELINE_BACKEND=gloo; RCCL wants one GPU per rank)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--steps", "6", "--warmup", "2", "--prompt-len", "20", "--cpu-steps", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, LNB_PIPELINE_BACKEND="gloo", LNB_PIPELINE_PROBE="0"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and d["config"]["parallelism"] == "pp2" and d["config"]["sequences_in_flight"] == 4


def test_forced_pipeline_line_carries_the_single_stream_figure_and_the_transport_rank_count():
    """LNB_FORCE_PIPELINE=1: the N-GPU code path (native ticks) on one GPU -- the JSON line of an N > 1 run has the same extra fields"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "tiny", "--steps", "8", "--warmup", "2", "--prompt-len", "20", "--cpu-steps", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, LNB_FORCE_PIPELINE="1", MASTER_PORT="29581"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    c = d["config"]
    assert c["rccl_comm_count_per_rank"] == [1] and c["single_stream"]["tokens_per_s"] > 0 and c["single_stream"]["tokens_equal_sequence0_of_the_batch"] is True
    assert d["cpu_baseline"]["value"] > 0
    # the value is the UNBATCHED figure (comparable with the N = 1 line); single-stream and batched sit next to it, each workload with its
    # one-GPU anchor measured in the same run, the efficiency derived from it, and the predicted stage times next to the measured tick
    assert abs(d["value"] - d["steps"] * c["sequences_in_flight"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-3
    assert c["value_single_stream"] == c["single_stream"]["tokens_per_s"] and c["value_batched"] == c["batched"]["tokens_per_s"] > 0
    an = c["one_gpu_anchor"]
    assert an["unbatched_tokens_per_s"] > 0 and an["unbatched_sequences_in_flight"] == c["sequences_in_flight"] and an["batched_tokens_per_s"] > 0
    assert c["efficiency_vs_one_gpu"]["unbatched"] > 0 and c["efficiency_vs_one_gpu"]["batched"] > 0
    assert c["stage_time_us"]["measured_tick"] > 0 and len(c["stage_time_us"]["predicted_per_rank_from_the_probe"]) == 1
    assert "whole blocks" in c["batched"]["cut"]

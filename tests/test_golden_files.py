"""The committed oracle goldens are present and well-formed (CPU suite): the GPU suite replays them device-only and SKIPS a missing file, so
a missing golden must fail here.  Each file records its generator (tests/golden/make_*.py), seeds and hashes."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("n_layers", [2, 8, 32])
def test_configs2_golden_of_every_depth_is_committed(n_layers):
    path = os.path.join(GOLD, "configs2_%dlayer_tokens.json" % n_layers)
    assert os.path.exists(path), "run tests/golden/make_configs2_cut_tokens.py %d on a host with cores to spare" % n_layers
    g = json.load(open(path))
    assert g["n_layers"] == n_layers and g["prompt_len"] == 4096 and g["weights_seed"] == 1234 and g["prompt_seed"] == 99
    toks = np.array(g["tokens"], dtype="<i4")
    assert len(toks) >= 65 and ((0 <= toks) & (toks < orc.LLAMA_8B["vocab_size"])).all()        # configs[2]: the first token + 64 decode steps at least
    assert hashlib.sha256(toks.tobytes()).hexdigest() == g["tokens_sha256"]
    prompt = orc.synth_tokens(g["prompt_seed"], g["prompt_len"], orc.LLAMA_8B["vocab_size"])
    assert hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest() == g["prompt_sha256"]    # the oracle's prompt generator still makes the prompt the file was made from


def test_bench_knows_the_full_depth_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.GOLDENS[("llama8b", 4096)] == "configs2_32layer_tokens.json" and b.CFG2_P == 4096 and b.CFG2_W + b.CFG2_K + 1 <= 65 + 4


def test_configs4_ten_layer_golden_is_committed():
    """one stage of the 8-GPU pipeline of configs[4] (10 of the 80 layers of the 70B-like shape): tests/test_gpu_round6.py replays it on the device"""
    path = os.path.join(GOLD, "configs4_10layer_tokens.json")
    assert os.path.exists(path), "run tests/golden/make_configs4_cut_tokens.py 10"
    g = json.load(open(path))
    assert g["n_layers"] == 10 and g["model"]["dim"] == 8192 and g["model"]["n_heads"] == 64 and g["model"]["n_kv_heads"] == 8 and g["model"]["multiple_of"] == 4096
    assert g["prompt_len"] == 16 and g["weights_seed"] == 1234 and g["prompt_seed"] == 99
    toks = np.array(g["tokens"], dtype="<i4")
    assert len(toks) >= 17 and ((0 <= toks) & (toks < g["model"]["vocab_size"])).all()
    assert hashlib.sha256(toks.tobytes()).hexdigest() == g["tokens_sha256"]
    prompt = orc.synth_tokens(g["prompt_seed"], g["prompt_len"], g["model"]["vocab_size"])
    assert hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest() == g["prompt_sha256"]

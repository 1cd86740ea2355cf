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
    assert b.GOLDENS[("llama70b-like", 16)] == "configs4_80layer_tokens.json" and b.CFG4_P == 16 and b.CFG4["n_layers"] == 80


@pytest.mark.parametrize("n_layers,n_min", [(10, 17), (80, 19)])
def test_configs4_goldens_are_committed(n_layers, n_min):
    """configs[4] (the 70B-like shape): one stage of its 8-GPU pipeline (10 of the 80 layers) and the FULL 80-layer model (made on the GPU box's host: 141 GB of
    synthetic weights in the oracle's memory, 64 threads); tests/test_gpu_round6.py replays both on the device, bench.py checks its configs4_one_gpu run against the second"""
    path = os.path.join(GOLD, "configs4_%dlayer_tokens.json" % n_layers)
    assert os.path.exists(path), "run tests/golden/make_configs4_cut_tokens.py %d" % n_layers
    g = json.load(open(path))
    assert g["n_layers"] == n_layers and g["model"]["dim"] == 8192 and g["model"]["n_heads"] == 64 and g["model"]["n_kv_heads"] == 8 and g["model"]["multiple_of"] == 4096
    assert g["prompt_len"] == 16 and g["weights_seed"] == 1234 and g["prompt_seed"] == 99
    toks = np.array(g["tokens"], dtype="<i4")
    assert len(toks) >= n_min and ((0 <= toks) & (toks < g["model"]["vocab_size"])).all()        # (80 layers: bench.py's configs4_one_gpu run is 1 + 2 + 16 tokens)
    assert hashlib.sha256(toks.tobytes()).hexdigest() == g["tokens_sha256"]
    prompt = orc.synth_tokens(g["prompt_seed"], g["prompt_len"], g["model"]["vocab_size"])
    assert hashlib.sha256(prompt.astype("<i4").tobytes()).hexdigest() == g["prompt_sha256"]


@pytest.mark.parametrize("P,n_min", [(128, 53), (512, 9)])
def test_multi_prompt_goldens_are_committed(P, n_min):
    """the oracle's continuations of SEVERAL prompts on the full 8B shape (sequence s: synth_tokens(99 + s, P); sparse: the file lists the sequences it holds): what sequences of a
    batch, of the sequences in flight and of configs[3]'s literal shape are compared with (bench.py check_multi_golden, pipeline.py _multi_golden_check, tests/test_gpu_round6.py)"""
    path = os.path.join(GOLD, "configs1_multi_P%d_tokens.json" % P)
    assert os.path.exists(path), "run tests/golden/make_multi_prompt_tokens.py %d <n_seq> %d on a host with cores to spare" % (P, n_min + 3)
    g = json.load(open(path))
    ids = g["sequences"]
    assert g["prompt_len"] == P and len(ids) >= 4 and ids == sorted(ids) and ids[0] == 0 and g["weights_seed"] == 1234 and g["prompt_seed_base"] == 99
    toks = np.concatenate([np.array(g["tokens"][str(k)], dtype="<i4") for k in ids])       # (ragged: later runs add sequences with fewer tokens)
    assert ((0 <= toks) & (toks < orc.LLAMA_8B["vocab_size"])).all() and all(len(g["tokens"][str(k)]) >= (n_min if k % (32 if P == 128 else 4) == 0 else 9) for k in ids)
    assert hashlib.sha256(toks.tobytes()).hexdigest() == g["tokens_sha256"]
    prompts = np.stack([orc.synth_tokens(99 + k, P, orc.LLAMA_8B["vocab_size"]) for k in ids]).astype("<i4")
    assert hashlib.sha256(prompts.tobytes()).hexdigest() == g["prompts_sha256"]
    if P == 128:                                             # sequence 0 is configs[1]'s prompt: the two files agree
        one = json.load(open(os.path.join(GOLD, "configs1_tokens.json")))["tokens"]
        assert g["tokens"]["0"] == one[:len(g["tokens"]["0"])]


@pytest.mark.parametrize("which,P,layers", [("configs4", 16, 80), ("configs1", 128, 32)])
def test_logits_hash_goldens_are_committed(which, P, layers):
    """every logit of every row at full depth (tests/golden/make_logits_hashes.py; replayed by tests/test_gpu_round6.py)"""
    path = os.path.join(GOLD, "%s_logits.json" % which)
    assert os.path.exists(path), "run tests/golden/make_logits_hashes.py %s" % which
    g = json.load(open(path))
    assert g["prompt_len"] == P and g["model"]["n_layers"] == layers and g["weights_seed"] == 1234 and g["prompt_seed"] == 99
    assert len(g["prompt_rows_logits_sha256"]) == P and len(g["steps"]) >= 4 and all(len(h) == 64 for h in g["prompt_rows_logits_sha256"])
    assert g["steps"][0]["input_token"] == g["first_token"] and all(a["argmax"] == b["input_token"] for a, b in zip(g["steps"], g["steps"][1:]))
    tok_file = "configs4_80layer_tokens.json" if which == "configs4" else "configs1_tokens.json"
    toks = json.load(open(os.path.join(GOLD, tok_file)))["tokens"]                       # the token goldens tell the same story
    assert [g["first_token"]] + [st["argmax"] for st in g["steps"]] == toks[:1 + len(g["steps"])]


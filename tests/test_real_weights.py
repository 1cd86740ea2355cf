"""Replay of the reference's real-checkpoint test (src/model/llamatransformer_simulated_test.go, TestSimulatedOnlyFirstLayer) on the
MI355X path.  It needs Meta's files -- consolidated.00.pth, params.json, tokenizer.model of Meta-Llama-3.1-8B-Instruct -- which are
not redistributable and not in any build image, so the tests SKIP when the directory is absent, exactly like the reference does
(:1344-1348: `Model directory "..." is not found, passing this test`).  Looked for in $LNB_MODEL_DIR, then
models-original/Meta-Llama-3.1-8B-Instruct under the repository root and its two parents (the reference's relative path).

With the files present this pins, against values the reference repository holds:
  * the tokenizer + chat template: "What is your name?" from the user -> the 15 prompt ids of :1369  (tests/golden/reference_kat.json);
  * the checkpoint reader + NewLlamaTransformer binding + Forward: the ONLY-FIRST-LAYER model (block 0 + norm + output) gives next token
    114545 (:1434), the greedy continuation {114545, 80657, 20508, 21053, 71434} (:1466) and the printed logits corners (:1400-1424)
    within the reference's own tolerance 30*THRESHOLD_BF16;
  * and the device path against the oracle on real weights, bit for bit.
"""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))["simulated_token_ids"]
REL = os.path.join("models-original", "Meta-Llama-3.1-8B-Instruct")
THRESHOLD_BF16 = 1e-2           # src/common/utils.go:16


def model_dir():
    cands = [os.environ.get("LNB_MODEL_DIR", "")] + [os.path.join(ROOT, up, REL) for up in ("", "..", os.path.join("..", ".."))]
    for d in cands:
        if d and os.path.isfile(os.path.join(d, "consolidated.00.pth")) and os.path.isfile(os.path.join(d, "params.json")):
            return os.path.abspath(d)
    return None


MODEL_DIR = model_dir()
needs_model = pytest.mark.skipif(MODEL_DIR is None, reason='Model directory "%s" is not found, passing this test' % os.path.join("..", "..", REL))


@needs_model
def test_tokenizer_reproduces_the_reference_prompt_ids():
    import lnb
    lnb.build()
    tk = lnb.Tokenizer(os.path.join(MODEL_DIR, "tokenizer.model"))
    assert tk.vocab_size == 128256
    assert tk.encode_chat([(KAT["prompt_header"], KAT["prompt_text"])]) == KAT["prompt"]        # llamatransformer_simulated_test.go:1362-1369
    tk.close()


@needs_model
@pytest.mark.gpu
def test_first_layer_only_model_reproduces_the_reference_tokens_and_logits():
    import lnb
    from oracle import oracle as orc
    lnb.build()
    args = lnb.model_args_from_json(os.path.join(MODEL_DIR, "params.json"))
    ck = lnb.Checkpoint(os.path.join(MODEL_DIR, "consolidated.00.pth"))
    i = ck.find("tok_embeddings.weight")
    assert i >= 0
    _, dt, shape, _ = ck.tensor(i)
    assert dt == "bf16"
    args.update(vocab_size=int(shape[0]), n_layers=1)                                          # vocab_size: "defined later by tokenizer" (modelargs.go:17)
    gm = lnb.LlamaTransformer(**args).load_checkpoint(ck).finalize()
    prompt = np.array(KAT["prompt"], dtype=np.int32)
    seq_len = KAT["sequence_length"]
    gc = lnb.InferenceContext(gm, seq_len)
    logits, first = gc.Forward(prompt, 0)
    assert first == KAT["expected_first_layer_only_next_token"]                                 # :1434
    for row, (head, tail) in KAT["expected_logits_first_layer_only"]["rows"].items():          # :1400-1424, the reference's tolerance
        assert np.abs(logits[int(row), :3] - np.array(head, dtype=np.float32)).max() <= 30 * THRESHOLD_BF16
        assert np.abs(logits[int(row), -3:] - np.array(tail, dtype=np.float32)).max() <= 30 * THRESHOLD_BF16
    more, _ = gc.decode_greedy(first, len(prompt), seq_len - len(prompt) - 1)
    assert [first] + [int(t) for t in more] == KAT["expected_first_layer_only_tokens"]         # :1466
    # the oracle on the same real tensors: bit-identical logits
    om = orc.Model(**{k: args[k] for k in ("dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "multiple_of", "ffn_dim_multiplier",
                                            "norm_eps", "use_scaled_rope", "rope_theta", "max_seq_len")})
    for name, _ in gm.tensor_infos():
        _, _, _, arr = ck.tensor(ck.find(name))
        om.set_tensor(name, np.ascontiguousarray(arr).ravel())
    om.finalize()
    lo, ao = orc.Context(om, seq_len).forward(prompt, 0)
    assert ao == first and (lo.view(np.uint32) == logits.view(np.uint32)).all()
    gc.close(); gm.close(); om.close(); ck.close()

"""Weight ingestion (SURVEY.md 8f "next" #2): the C-ABI checkpoint reader against files written by torch.save itself.

Oracle = PyTorch: the test writes real zip checkpoints (Meta's key names, bf16 tensors, protocol-2 pickle, 64-byte aligned
STORED entries), reads them back through lnb_checkpoint_* (mmap + zip central directory + pickle VM, no PyTorch involved)
and compares names, shapes and raw bytes with the tensors that were saved.  Error texts follow the reference
(src/torch/torchmodelreader.go, src/pickle/pickledispatch.go, src/model/loader.go:183-192).  No GPU needed."""
import json
import os
import pickle
import zipfile

import numpy as np
import pytest

import lnb

torch = pytest.importorskip("torch")

TINY = dict(dim=64, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=96, multiple_of=32, ffn_dim_multiplier=1.3)


def tiny_state(seed=0):
    g = torch.Generator().manual_seed(seed)
    hd, d = TINY["dim"] // TINY["n_heads"], TINY["dim"]
    ffn = 192          # 4*64=256 -> 170 -> int(1.3*170)=221 -> round up to 32 -> 224?  (shape only matters to the model test)
    def t(*shape):
        return (torch.randn(*shape, generator=g) * 0.05).to(torch.bfloat16)
    sd = {"tok_embeddings.weight": t(TINY["vocab_size"], d)}
    for l in range(TINY["n_layers"]):
        p = "layers.%d." % l
        sd[p + "attention.wq.weight"] = t(d, d)
        sd[p + "attention.wk.weight"] = t(TINY["n_kv_heads"] * hd, d)
        sd[p + "attention.wv.weight"] = t(TINY["n_kv_heads"] * hd, d)
        sd[p + "attention.wo.weight"] = t(d, d)
        sd[p + "feed_forward.w1.weight"] = t(ffn, d)
        sd[p + "feed_forward.w2.weight"] = t(d, ffn)
        sd[p + "feed_forward.w3.weight"] = t(ffn, d)
        sd[p + "attention_norm.weight"] = t(d) + 1
        sd[p + "ffn_norm.weight"] = t(d) + 1
    sd["norm.weight"] = t(d) + 1
    sd["output.weight"] = t(TINY["vocab_size"], d)
    return sd


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def test_reads_what_torch_save_wrote(tmp_path):
    sd = tiny_state()
    path = str(tmp_path / "consolidated.00.pth")
    torch.save(sd, path)                                       # plain dict, like Meta's consolidated.00.pth
    ck = lnb.Checkpoint(path)
    assert len(ck) == len(sd)
    names = []
    for i in range(len(ck)):
        name, dt, shape, arr = ck.tensor(i)
        names.append(name)
        assert dt == "bf16" and shape == tuple(sd[name].shape)
        assert (arr == bits(sd[name])).all()
        assert ck.find(name) == i
    assert names == list(sd.keys())                            # pickle (insertion) order, like the reference's PickleDict
    assert ck.find("no.such.tensor") == -1
    ck.close()


def test_views_shared_storage_other_dtypes_and_ordered_dict(tmp_path):
    from collections import OrderedDict
    base = (torch.arange(64, dtype=torch.float32) / 7).to(torch.bfloat16)
    sd = OrderedDict()                                         # torch state_dicts are OrderedDicts
    sd["whole"] = base.view(8, 8)
    sd["tail_view"] = base[16:48].view(4, 8)                   # same storage, storage_offset 16 (the reference ignores it; we do not)
    sd["f32"] = torch.linspace(-1, 1, 12).view(3, 4)
    sd["f16"] = torch.linspace(-2, 2, 6).to(torch.float16)
    sd["transposed"] = base.view(8, 8).t()                     # non-contiguous: must be reported, not mis-read
    sd["scalar_meta"] = 7                                      # non-tensor entries are not weights
    path = str(tmp_path / "views.pth")
    torch.save(sd, path)
    ck = lnb.Checkpoint(path)
    got = {}
    for i in range(len(ck)):
        try:
            name, dt, shape, arr = ck.tensor(i)
            got[name] = (dt, shape, arr.copy())
        except lnb.LnbError as e:
            got["error"] = str(e)
    assert (got["whole"][2] == bits(sd["whole"])).all()
    assert got["tail_view"][1] == (4, 8) and (got["tail_view"][2] == bits(sd["tail_view"])).all()
    assert got["f32"][0] == "f32" and np.array_equal(got["f32"][2], sd["f32"].numpy())
    assert got["f16"][0] == "f16" and (got["f16"][2] == sd["f16"].view(torch.int16).numpy().view(np.uint16)).all()
    assert "transposed" in got["error"] and "not contiguous" in got["error"]
    assert "scalar_meta" not in got
    ck.close()


def test_big_archive_uses_zip64_records(tmp_path):
    """torch writes ZIP64 end records when asked to / when large; force them with a hand-made archive of the same layout."""
    sd = {"a.weight": (torch.randn(5, 8) * 0.1).to(torch.bfloat16), "b.weight": (torch.randn(3) * 0.1).to(torch.bfloat16)}
    ref = str(tmp_path / "ref.pth")
    torch.save(sd, ref)
    out = str(tmp_path / "z64.pth")
    with zipfile.ZipFile(ref) as zin, zipfile.ZipFile(out, "w", zipfile.ZIP_STORED, allowZip64=True) as zout:
        for info in zin.infolist():
            with zout.open(info.filename, "w", force_zip64=True) as f:   # zip64 extra fields + zip64 EOCD
                f.write(zin.read(info.filename))
    ck = lnb.Checkpoint(out)
    assert len(ck) == 2
    for i in range(2):
        name, dt, shape, arr = ck.tensor(i)
        assert (arr == bits(sd[name])).all()
    ck.close()


def test_error_behaviour(tmp_path):
    with pytest.raises(lnb.LnbError, match="open .*missing.pth"):
        lnb.Checkpoint(str(tmp_path / "missing.pth"))
    junk = tmp_path / "junk.pth"; junk.write_bytes(b"this is not a zip archive at all, just bytes" * 3)
    with pytest.raises(lnb.LnbError, match="not a zip archive"):
        lnb.Checkpoint(str(junk))
    nopkl = str(tmp_path / "nopkl.pth")
    with zipfile.ZipFile(nopkl, "w") as z:
        z.writestr("archive/version", "3\n")
    with pytest.raises(lnb.LnbError, match="no .pkl file found in Torch model file"):          # torchmodelreader.go:48-50
        lnb.Checkpoint(nopkl)
    comp = str(tmp_path / "compressed.pth")
    with zipfile.ZipFile(comp, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("archive/data.pkl", pickle.dumps({"x": 1}, protocol=2) * 50)
    with pytest.raises(lnb.LnbError, match="is compressed"):
        lnb.Checkpoint(comp)
    bad = str(tmp_path / "badclass.pth")
    with zipfile.ZipFile(bad, "w") as z:
        z.writestr("archive/data.pkl", pickle.dumps({"x": np.float64}, protocol=2))            # GLOBAL numpy.float64
    with pytest.raises(lnb.LnbError, match=r'unknown class "numpy\.float64" not found'):         # torchmodelreader.go:99-108
        lnb.Checkpoint(bad)
    proto4 = str(tmp_path / "proto4.pth")
    with zipfile.ZipFile(proto4, "w") as z:
        z.writestr("archive/data.pkl", pickle.dumps({"x": 1}, protocol=4))                     # FRAME / MEMOIZE opcodes
    with pytest.raises(lnb.LnbError, match="unsupported Pickle op code"):                       # pickledispatch.go:97
        lnb.Checkpoint(proto4)


def test_params_json_defaults_and_values(tmp_path):
    p = tmp_path / "params.json"
    p.write_text(json.dumps({"dim": 4096, "n_layers": 32, "n_heads": 32, "n_kv_heads": 8, "vocab_size": 128256,
                             "ffn_dim_multiplier": 1.3, "multiple_of": 1024, "norm_eps": 1e-05, "rope_theta": 500000.0,
                             "use_scaled_rope": True}))
    a = lnb.model_args_from_json(str(p))
    assert (a["dim"], a["n_layers"], a["n_heads"], a["n_kv_heads"], a["vocab_size"], a["multiple_of"]) == (4096, 32, 32, 8, 128256, 1024)
    assert a["ffn_dim_multiplier"] == 1.3 and a["use_scaled_rope"] == 1 and a["rope_theta"] == 500000.0 and a["max_seq_len"] == 2048
    assert abs(a["norm_eps"] - 1e-5) < 1e-12
    p.write_text("{}")                                          # NewModelArgs defaults, modelargs.go:29-44
    a = lnb.model_args_from_json(str(p))
    assert (a["dim"], a["n_layers"], a["n_heads"], a["n_kv_heads"], a["vocab_size"], a["multiple_of"]) == (4096, 32, 32, -1, -1, 256)
    assert a["ffn_dim_multiplier"] == -1 and a["use_scaled_rope"] == 0 and a["rope_theta"] == 500000.0


def test_truncated_and_corrupted_files_fail_cleanly(tmp_path):
    """The reader parses an untrusted file straight out of an mmap: every prefix of a valid checkpoint and a few hundred random
    byte flips must end in an error message or a successful open -- never in a crash or an out-of-bounds read."""
    sd = {"a.weight": (torch.randn(6, 8) * 0.1).to(torch.bfloat16), "b.weight": (torch.randn(5) * 0.1).to(torch.bfloat16)}
    ref = str(tmp_path / "ref.pth")
    torch.save(sd, ref)
    blob = open(ref, "rb").read()
    rng = np.random.default_rng(0)
    victim = str(tmp_path / "victim.pth")
    outcomes = {"ok": 0, "error": 0}
    cuts = sorted(set([0, 1, 21, 22, 23, len(blob) - 1] + list(rng.integers(0, len(blob), 60))))
    for cut in cuts:
        open(victim, "wb").write(blob[:cut])
        try:
            ck = lnb.Checkpoint(victim)
            for i in range(len(ck)):
                ck.tensor(i)
            ck.close(); outcomes["ok"] += 1
        except lnb.LnbError:
            outcomes["error"] += 1
    for trial in range(300):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        open(victim, "wb").write(bytes(b))
        try:
            ck = lnb.Checkpoint(victim)
            for i in range(len(ck)):
                name, dt, shape, arr = ck.tensor(i)
                _ = arr.sum()                                   # touch every byte the reader says belongs to the tensor
            ck.close(); outcomes["ok"] += 1
        except lnb.LnbError:
            outcomes["error"] += 1
    assert outcomes["error"] > 30 and outcomes["ok"] > 30, outcomes

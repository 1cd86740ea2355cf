"""Prompt tokenisation (SURVEY.md 8f "next" #3) against a Python restatement of the reference's algorithm.

The real tokenizer.model is not available offline, so the vocabulary is synthetic but well formed: all 256 single bytes plus
byte-pair merges learned from a small corpus (ranked like a tiktoken file), written in the tiktoken format.  The oracle is the
`regex` module running the reference's split pattern (src/model/vocabulary.go:32, with RE2's ASCII \\s spelled out) followed by
a restatement of bytePairMerge (src/inference/tokenize.go:109-176).  Parity with the real Llama-3 vocabulary is UNPINNED."""
import base64
import collections

import numpy as np
import pytest

import lnb

regex = pytest.importorskip("regex")

WS = r"\t\n\f\r "
PATTERN = regex.compile(r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^" + WS + r"\p{L}\p{N}]+[\r\n]*|[" + WS + r"]*[\r\n]+|[" + WS + r"]+")

CORPUS = ("The quick brown fox jumps over the lazy dog. It's 2024, isn't it? We're here; they've gone, I'm fine, he'll come, she'd stay.\n"
          "Llama nuts and bolts: RMSNorm, RoPE, attention, SwiGLU!  naïve café Ünïcödé Straße 東京 こんにちは 12345 3.14159 🙂👍🏽\r\n\r\n"
          "def forward(x):\n    return x @ w.T  # matmul\n\n\tindented\twith\ttabs   and   spaces\n") * 3


def learn_bpe(corpus, n_merges):
    """classic BPE on the pattern's pieces; returns {bytes: rank} with the 256 bytes first (tiktoken layout)"""
    ranks = {bytes([b]): b for b in range(256)}
    words = collections.Counter(tuple(bytes([b]) for b in m.group(0).encode()) for m in PATTERN.finditer(corpus))
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _cnt = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        ranks.setdefault(a + b, len(ranks))
        merged = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(w[i]); i += 1
            merged[tuple(out)] += c
        words = merged
    return ranks


def oracle_bpe(piece, ranks):
    """src/inference/tokenize.go:109-176 restated"""
    NONE = 2 ** 31 - 1
    n = len(piece)
    parts = [[ranks.get(piece[i:i + 2], NONE) if i + 1 < n else NONE, i] for i in range(n + 1)]
    if n >= 1:
        parts[n - 1] = [NONE, n - 1]
    parts[n] = [NONE, n]

    def rank_at(i):
        return ranks.get(piece[parts[i][1]:parts[i + 3][1]], NONE) if i + 3 < len(parts) else NONE
    while True:
        best, bi = NONE, None
        for i in range(len(parts) - 1):
            if parts[i][0] < best:
                best, bi = parts[i][0], i
        if bi is None:
            break
        if bi > 0:
            parts[bi - 1][0] = rank_at(bi - 1)
        parts[bi][0] = rank_at(bi)
        del parts[bi + 1]
    return [ranks.get(piece[parts[i][1]:parts[i + 1][1]], 0) for i in range(len(parts) - 1)]


def oracle_encode(text, ranks):
    out = []
    for m in PATTERN.finditer(text):
        piece = m.group(0).encode()
        if piece in ranks:
            out.append(ranks[piece])
        else:
            out.extend(oracle_bpe(piece, ranks))
    return out


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    ranks = learn_bpe(CORPUS, 400)
    path = tmp_path_factory.mktemp("tok") / "tokenizer.model"
    with open(path, "w") as f:
        for piece, rank in sorted(ranks.items(), key=lambda kv: kv[1]):
            f.write("%s %d\n" % (base64.b64encode(piece).decode(), rank))
    t = lnb.Tokenizer(str(path))
    yield t, ranks
    t.close()


def test_vocabulary_layout_and_special_tokens(tok):
    t, ranks = tok
    n = len(ranks)
    assert t.vocab_size == n + 256
    assert (t.bos, t.eos) == (n, n + 1) and (t.eom, t.eot) == (n + 8, n + 9)       # tiktokenreader.go:49-61 order
    assert t.piece(t.bos) == b"<|begin_of_text|>" and t.piece(n + 11) == b"<|reserved_special_token_2|>"
    assert t.piece(n + 255) == b"<|reserved_special_token_246|>"
    for piece, rank in list(ranks.items())[::37]:
        assert t.token_id(piece) == rank and t.piece(rank) == piece
    assert t.token_id(b"\xff\xfe not a token") == -1


def test_encode_matches_the_oracle_on_fixed_strings(tok):
    t, ranks = tok
    cases = ["", "a", "Hello world", "  two  spaces  ", "It's HERE'S we'RE can'T I'M she'LL he'D 'twas", "x\n\ny\r\n\r\n  \n z", "tabs\t\tand\fformfeed",
             "12345678 3.14 1,000,000 ٣٤٥ ⅓", "naïve café Straße Ünïcödé", "東京タワー こんにちは 你好世界", "emoji 🙂👍🏽 done", "  \n", "\n\n\n", " \t \n \t ",
             "a'sſ 'ſ", "snake_case CamelCase kebab-case ###!!!", "def f(x):\n    return x**2  # comment\n", CORPUS[:400]]
    for s in cases:
        assert t.encode(s) == oracle_encode(s, ranks), repr(s)


def test_encode_matches_the_oracle_on_random_strings(tok):
    t, ranks = tok
    rng = np.random.default_rng(5)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789") + list(" \t\n\r\f'\".,;:!?-_()[]{}@#$%^&*+=/\\|<>~`") \
        + list("äöüßéèêñçøåÆŁžščě") + list("αβγδεζηθ") + list("абвгдеж") + list("東京日本語中文") + list("٠١٢٣٤") + list("🙂👍🏽🚀") + ["'s", "'T", "'re", "'LL", " ", " ", "\n", " ", " ", "́"]
    for trial in range(1500):
        s = "".join(rng.choice(alphabet) for _ in range(int(rng.integers(0, 40))))
        assert t.encode(s) == oracle_encode(s, ranks), repr(s)


def test_chat_template(tok):
    t, ranks = tok
    parts = [("system", "You are helpful."), ("user", ""), ("user", "Hi there!\nHow are you?")]
    n = len(ranks)
    bos, sh, eh, eot = n, n + 6, n + 7, n + 9
    exp = [bos]
    for header, content in parts + [("assistant", None)]:
        if content == "":
            continue                                           # empty parts are skipped (tokenize.go:43-45)
        exp += [sh] + oracle_encode(header, ranks) + [eh] + oracle_encode("\n\n", ranks)
        if content is not None:
            exp += oracle_encode(content, ranks) + [eot]
    assert t.encode_chat(parts) == exp


def test_error_behaviour(tmp_path):
    with pytest.raises(lnb.LnbError, match="open .*nope"):
        lnb.Tokenizer(str(tmp_path / "nope.model"))
    bad = tmp_path / "bad.model"; bad.write_text("!!!notbase64!!! 0\n")
    with pytest.raises(lnb.LnbError, match="illegal base64 data"):          # base64.StdEncoding.DecodeString's error text
        lnb.Tokenizer(str(bad))
    bad.write_text("YQ== x\n")
    with pytest.raises(lnb.LnbError, match="invalid rank"):
        lnb.Tokenizer(str(bad))

"""ctypes binding of the CPU oracle (oracle/liblnb_oracle.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblnb_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("lnb_oracle.c", "lnb_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Args(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int), ("n_kv_heads", C.c_int),
                ("vocab_size", C.c_int), ("multiple_of", C.c_int), ("ffn_dim_multiplier", C.c_double),
                ("norm_eps", C.c_float), ("use_scaled_rope", C.c_int), ("rope_theta", C.c_double),
                ("max_seq_len", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u16p, f32p, i32p, vp = C.POINTER(C.c_uint16), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    sig = {
        "orc_f32_to_bf16": (C.c_uint16, [C.c_float]),
        "orc_bf16_to_f32": (C.c_float, [C.c_uint16]),
        "orc_synth_bf16": (C.c_uint16, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_float]),
        "orc_synth_token": (C.c_int32, [C.c_uint64, C.c_uint64, C.c_int32]),
        "orc_synth_fill": (None, [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_float]),
        "orc_linear_bf16": (None, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_linear_f32": (None, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
        "orc_matmul_bf16": (None, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_arange_bf16": (C.c_int, [C.c_int, C.c_int, C.c_int, vp]),
        "orc_arange_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, vp]),
        "orc_outer_bf16": (None, [vp, C.c_int, vp, C.c_int, vp]),
        "orc_polar_f32": (None, [vp, vp, vp, C.c_int]),
        "orc_polar_bf16": (None, [vp, vp, vp, C.c_int]),
        "orc_triu_bf16": (None, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "orc_pow_bf16": (None, [vp, vp, C.c_int, C.c_double]),
        "orc_mean_f32": (None, [vp, vp, C.c_int, C.c_int]),
        "orc_softmax_f32": (None, [vp, vp, C.c_int, C.c_int]),
        "orc_argmax_f32": (C.c_int32, [vp, C.c_int]),
        "orc_silu_table": (f32p, []),
        "orc_rmsnorm_bf16": (None, [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
        "orc_rope_freqs": (None, [C.c_int, C.c_double, C.c_int, vp]),
        "orc_rope_table": (None, [C.c_int, C.c_int, C.c_double, C.c_int, vp, vp]),
        "orc_rope_apply": (None, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "orc_ffn_hidden_dim": (C.c_int, [C.POINTER(Args)]),
        "orc_model_create": (vp, [C.POINTER(Args)]),
        "orc_model_destroy": (None, [vp]),
        "orc_model_set_tensor": (C.c_int, [vp, C.c_char_p, vp, C.c_int64]),
        "orc_model_get_tensor": (u16p, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
        "orc_model_fill_synthetic": (None, [vp, C.c_uint64, C.c_int]),
        "orc_model_finalize": (C.c_int, [vp]),
        "orc_model_rope_table": (f32p, [vp, C.POINTER(C.c_int)]),
        "orc_ctx_create": (vp, [vp, C.c_int]),
        "orc_ctx_destroy": (None, [vp]),
        "orc_ctx_set_threads": (None, [vp, C.c_int]),
        "orc_set_exp_impl": (None, [C.c_int]),
        "orc_exp_f64": (C.c_double, [C.c_double]),
        "orc_ctx_cache": (u16p, [vp, C.c_int, C.c_int]),
        "orc_ctx_set_dump": (None, [vp, vp, vp]),
        "orc_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, i32p]),
        "orc_generate": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp]),
        "orc_last_error": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32_to_bf16(a):
    """numpy f32 -> uint16 by truncation (src/dtype/bfloat16.go:31-33)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return (a.view(np.uint32) >> 16).astype(np.uint16)


def bf16_to_f32(a):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    return (a.astype(np.uint32) << 16).view(np.float32)


LLAMA_8B = dict(dim=4096, n_layers=32, n_heads=32, n_kv_heads=8, vocab_size=128256, multiple_of=1024,
                ffn_dim_multiplier=1.3, norm_eps=1e-5, use_scaled_rope=1, rope_theta=500000.0, max_seq_len=2048)
TINY = dict(dim=256, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=1024, multiple_of=64,
            ffn_dim_multiplier=1.3, norm_eps=1e-5, use_scaled_rope=1, rope_theta=500000.0, max_seq_len=2048)


def make_args(**kw):
    d = dict(LLAMA_8B)
    d.update(kw)
    return Args(**d)


DUMPFN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int)


class Model:
    def __init__(self, **kw):
        self.args = make_args(**kw)
        self.L = lib()
        self.h = self.L.orc_model_create(C.byref(self.args))
        self.ffn_hidden = self.L.orc_ffn_hidden_dim(C.byref(self.args))

    def fill_synthetic(self, seed=1234, nthreads=0):
        self.L.orc_model_fill_synthetic(self.h, seed, nthreads or default_threads())
        return self

    def set_tensor(self, name, arr_u16):
        a = np.ascontiguousarray(arr_u16, dtype=np.uint16)
        if self.L.orc_model_set_tensor(self.h, name.encode(), _p(a), a.size) != 0:
            raise ValueError(self.L.orc_last_error().decode())

    def get_tensor(self, name):
        n = C.c_int64()
        p = self.L.orc_model_get_tensor(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def tensor_names(self):
        a = self.args
        names = ["tok_embeddings.weight"]
        for l in range(a.n_layers):
            for s in ("attention_norm", "attention.wq", "attention.wk", "attention.wv", "attention.wo",
                      "ffn_norm", "feed_forward.w1", "feed_forward.w2", "feed_forward.w3"):
                names.append("layers.%d.%s.weight" % (l, s))
        return names + ["norm.weight", "output.weight"]

    def finalize(self):
        self.L.orc_model_finalize(self.h)
        return self

    def rope_table(self):
        rows = C.c_int()
        p = self.L.orc_model_rope_table(self.h, C.byref(rows))
        hd = self.args.dim // self.args.n_heads
        return np.ctypeslib.as_array(p, shape=(rows.value, hd // 2, 2)).copy()

    def close(self):
        if self.h:
            self.L.orc_model_destroy(self.h)
            self.h = None


def default_threads():
    """Output-parallel worker count.  Capped at 64: on the 2 x 64-core / 256-thread GPU hosts the OpenMP team collapses
    beyond the physical cores of one socket (measured: 64 threads 114 GMAC/s, 256 threads 0.5 GMAC/s)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


class Context:
    def __init__(self, model, seq_len, nthreads=None):
        self.m, self.L = model, model.L
        self.seq_len = seq_len
        self.h = self.L.orc_ctx_create(model.h, seq_len)
        self.nthreads = nthreads or default_threads()
        self.L.orc_ctx_set_threads(self.h, self.nthreads)
        self._cb = None

    def forward(self, tokens, start_pos, want_logits=True):
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        S = tok.size
        V = self.m.args.vocab_size
        logits = np.empty((S, V), dtype=np.float32) if want_logits else None
        am = C.c_int32(-2)
        rc = self.L.orc_forward(self.h, _p(tok), S, start_pos, _p(logits) if want_logits else None, C.byref(am))
        if rc != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return logits, am.value

    def generate(self, prompt, n_out):
        pr = np.ascontiguousarray(prompt, dtype=np.int32)
        out = np.empty(n_out, dtype=np.int32)
        secs = np.zeros(n_out, dtype=np.float64)
        n = self.L.orc_generate(self.h, _p(pr), pr.size, _p(out), n_out, _p(secs))
        if n < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return out[:n], secs[:n]

    def cache(self, layer, which):
        a = self.m.args
        hd = a.dim // a.n_heads
        p = self.L.orc_ctx_cache(self.h, layer, which)
        return np.ctypeslib.as_array(p, shape=(self.seq_len, a.n_kv_heads, hd))

    def capture(self):
        """Install a dump hook; returns dict filled as {(stage, layer): ndarray} during forward."""
        store = {}

        def cb(user, stage, layer, dtype, data, shape, rank):
            shp = tuple(int(shape[i]) for i in range(rank))
            n = int(np.prod(shp))
            ct = C.c_uint16 if dtype == 0 else C.c_float
            arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(ct)), shape=(n,)).reshape(shp).copy()
            store[(stage.decode(), layer)] = arr

        self._cb = DUMPFN(cb)
        self.L.orc_ctx_set_dump(self.h, C.cast(self._cb, C.c_void_p), None)
        return store

    def close(self):
        if self.h:
            self.L.orc_ctx_destroy(self.h)
            self.h = None


def synth_tokens(seed, n, vocab):
    L = lib()
    return np.array([L.orc_synth_token(seed, i, vocab) for i in range(n)], dtype=np.int32)

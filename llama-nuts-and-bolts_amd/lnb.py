"""ctypes plumbing over liblnb_hip.so (the C ABI of include/lnb.h) for tests and bench.py.

This is NOT a compute path: every call goes straight into the HIP library.  There is no CPU
fallback -- if the shared library is missing or no MI355X is visible, the calls raise.
Names mirror the reference's Go API (src/model: LlamaTransformer / InferenceContext,
src/inference: InferenceEngine) so the parity tests read like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("LNB_SO") or os.path.join(_HERE, "liblnb_hip.so")     # LNB_SO: measurement aid (A/B builds of the same ABI)
_CSRC = os.path.join(_HERE, "csrc")


MODE_EXACT, MODE_FAST = 0, 1


class LnbError(RuntimeError):
    pass


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the library, in-tree (the .so travels with the repo snapshot)."""
    # every source and header under csrc/ plus the ABI header: a stale git-ignored .so must never be what the tests validate
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".cpp", ".h", ".hpp")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "lnb.h"))
    if os.environ.get("LNB_SO"):
        return _SO
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _CSRC, "-s"])
    return _SO


class ModelArgs(C.Structure):
    """model.ModelArgs (src/model/modelargs.go:12-27)"""
    _fields_ = [("dim", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("vocab_size", C.c_int32), ("multiple_of", C.c_int32), ("ffn_dim_multiplier", C.c_double),
                ("norm_eps", C.c_float), ("use_scaled_rope", C.c_int32), ("rope_theta", C.c_double),
                ("max_seq_len", C.c_int32)]


LLAMA_8B = dict(dim=4096, n_layers=32, n_heads=32, n_kv_heads=8, vocab_size=128256, multiple_of=1024,
                ffn_dim_multiplier=1.3, norm_eps=1e-5, use_scaled_rope=1, rope_theta=500000.0, max_seq_len=2048)

_lib = None
EXPORTS = [
    "lnb_last_error", "lnb_device_count", "lnb_device_info", "lnb_device_can_access_peer", "lnb_device_pci_bus_id", "lnb_model_create", "lnb_model_create_parts", "lnb_model_destroy", "lnb_model_ffn_hidden_dim",
    "lnb_model_set_tensor", "lnb_model_get_tensor", "lnb_model_fill_synthetic", "lnb_model_finalize",
    "lnb_model_rope_table", "lnb_model_weight_bytes", "lnb_ctx_create", "lnb_ctx_destroy", "lnb_ctx_reset",
    "lnb_ctx_read_kv", "lnb_ctx_set_layer_callback", "lnb_forward", "lnb_decode_greedy", "lnb_ctx_hidden_ptr",
    "lnb_forward_stage", "lnb_forward_stage_begin", "lnb_forward_stage_end", "lnb_ctx_synchronize", "lnb_ctx_stream", "lnb_op_linear", "lnb_op_rmsnorm_linear", "lnb_op_argmax", "lnb_op_exp_table", "lnb_op_linear_mode", "lnb_ctx_set_mode", "lnb_ctx_get_mode", "lnb_ctx_set_schedule", "lnb_ctx_get_schedule", "lnb_ctx_set_attention", "lnb_ctx_zseq_count", "lnb_ctx_norm_fallbacks", "lnb_ctx_prefill_attention_form",
    "lnb_profile_kernel", "lnb_profile_kernel_stamps", "lnb_model_num_tensors", "lnb_model_tensor_info",
    "lnb_checkpoint_open", "lnb_checkpoint_close", "lnb_checkpoint_num_tensors", "lnb_checkpoint_find", "lnb_checkpoint_tensor",
    "lnb_model_load_checkpoint", "lnb_model_args_from_json",
    "lnb_tokenizer_load", "lnb_tokenizer_free", "lnb_tokenizer_vocab_size", "lnb_tokenizer_special", "lnb_tokenizer_token_id",
    "lnb_tokenizer_piece", "lnb_tokenizer_encode", "lnb_tokenizer_encode_chat",
    "lnb_tokenizer_stream_create", "lnb_tokenizer_stream_free", "lnb_tokenizer_decode_stream", "lnb_tokenizer_stream_pending",
    "lnb_pipeline_unique_id", "lnb_pipeline_init", "lnb_pipeline_init_loopback", "lnb_pipeline_destroy", "lnb_pipeline_tick", "lnb_pipeline_sync", "lnb_pipeline_read_tokens", "lnb_pipeline_selftest", "lnb_pipeline_comm_count",
    "lnb_model_enable_batch", "lnb_model_batch_bytes", "lnb_batch_create", "lnb_batch_destroy", "lnb_batch_decode", "lnb_batch_decode_until", "lnb_ctx_set_stop_ids", "lnb_decode_greedy_until", "lnb_batch_profile_kernel", "lnb_batch_set_state", "lnb_batch_check_error", "lnb_pipeline_tick_batch",
    "lnb_pipeline_init_host", "lnb_batch_boundary_ptr",
    "lnb_abi_version", "lnb_runtime_info", "lnb_profile_ffn_pair",
]
ABI_VERSION = 6          # LNB_ABI_VERSION of include/lnb.h this binding was written against (tests/test_cabi.py compares it with the header's)


class RuntimeInfo(C.Structure):
    """lnb_runtime_info_t (include/lnb.h)"""
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("n_cus", C.c_int32),
                ("shader_clock_khz", C.c_int32), ("memory_clock_khz", C.c_int32), ("wall_clock_khz", C.c_int32),
                ("hw_queues_env", C.c_int32), ("hw_queues_set_by_library", C.c_int32), ("hip_initialised_before_load", C.c_int32),
                ("hw_queues_expected", C.c_int32), ("hw_queues_measured", C.c_int32), ("probe_ms", C.c_float),
                ("device_name", C.c_char * 64), ("arch", C.c_char * 32)]



def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise LnbError("liblnb_hip.so is not built (run __graft_entry__.build()); there is no fallback path")
    L = C.CDLL(_SO)
    vp, i32p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    # a stale library behind a newer binding (or the reverse) must fail HERE, not write through a mistyped pointer later
    if not hasattr(L, "lnb_abi_version") or L.lnb_abi_version() != ABI_VERSION:
        raise LnbError("liblnb_hip.so reports ABI version %s, this binding is written against %d: rebuild the library (make -C csrc)"
                       % (L.lnb_abi_version() if hasattr(L, "lnb_abi_version") else "none", ABI_VERSION))
    L.lnb_runtime_info.argtypes = [C.c_int, C.c_int, C.POINTER(RuntimeInfo)]
    L.lnb_profile_ffn_pair.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.lnb_last_error.restype = C.c_char_p
    L.lnb_device_count.argtypes = [C.POINTER(C.c_int)]
    L.lnb_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.lnb_device_can_access_peer.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.lnb_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.lnb_model_create.argtypes = [C.POINTER(ModelArgs), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.lnb_model_create_parts.argtypes = [C.POINTER(ModelArgs), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.lnb_model_destroy.argtypes = [vp]
    L.lnb_model_ffn_hidden_dim.argtypes = [C.POINTER(ModelArgs)]
    L.lnb_model_set_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int]
    L.lnb_model_get_tensor.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    L.lnb_model_fill_synthetic.argtypes = [vp, C.c_uint64]
    L.lnb_model_finalize.argtypes = [vp, C.c_int]
    L.lnb_model_rope_table.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int)]
    L.lnb_model_weight_bytes.argtypes = [vp]
    L.lnb_model_weight_bytes.restype = C.c_int64
    L.lnb_ctx_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.lnb_ctx_destroy.argtypes = [vp]
    L.lnb_ctx_reset.argtypes = [vp]
    L.lnb_ctx_read_kv.argtypes = [vp, C.c_int, C.c_int, vp]
    L.lnb_ctx_set_layer_callback.argtypes = [vp, vp, vp]
    L.lnb_forward.argtypes = [vp, vp, C.c_int, C.c_int, vp, i32p]
    L.lnb_decode_greedy.argtypes = [vp, C.c_int32, C.c_int, C.c_int, vp, f32p]
    L.lnb_ctx_hidden_ptr.argtypes = [vp, C.c_int]
    L.lnb_ctx_hidden_ptr.restype = vp
    L.lnb_forward_stage.argtypes = [vp, vp, C.c_int, C.c_int, vp, i32p]
    L.lnb_forward_stage_begin.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.lnb_forward_stage_end.argtypes = [vp, i32p]
    L.lnb_ctx_synchronize.argtypes = [vp]
    L.lnb_ctx_stream.argtypes = [vp]
    L.lnb_ctx_stream.restype = vp
    L.lnb_profile_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_int, f32p]
    L.lnb_profile_kernel_stamps.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int)]
    L.lnb_op_linear.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lnb_op_rmsnorm_linear.argtypes = [C.c_int, vp, vp, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lnb_ctx_set_mode.argtypes = [vp, C.c_int]
    L.lnb_ctx_get_mode.argtypes = [vp]
    L.lnb_ctx_set_schedule.argtypes = [vp, C.c_int]
    L.lnb_ctx_get_schedule.argtypes = [vp]
    L.lnb_ctx_set_attention.argtypes = [vp, C.c_int, C.c_int]
    L.lnb_ctx_zseq_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.lnb_ctx_norm_fallbacks.argtypes = [vp, C.POINTER(C.c_int)]
    L.lnb_ctx_prefill_attention_form.argtypes = [vp, C.POINTER(C.c_int)]
    L.lnb_pipeline_unique_id.argtypes = [vp]
    L.lnb_pipeline_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.lnb_pipeline_init_loopback.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]
    L.lnb_pipeline_init_host.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.lnb_batch_boundary_ptr.argtypes = [vp, C.c_int]
    L.lnb_batch_boundary_ptr.restype = C.c_void_p
    L.lnb_pipeline_selftest.argtypes = [C.c_int, C.c_int]
    L.lnb_pipeline_destroy.argtypes = [vp]
    L.lnb_pipeline_tick.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.lnb_pipeline_sync.argtypes = [vp]
    L.lnb_pipeline_comm_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.lnb_model_enable_batch.argtypes = [vp]
    L.lnb_model_batch_bytes.argtypes = [vp]
    L.lnb_model_batch_bytes.restype = C.c_int64
    L.lnb_batch_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
    L.lnb_batch_destroy.argtypes = [vp]
    L.lnb_batch_decode.argtypes = [vp, i32p, i32p, C.c_int, vp, f32p]
    L.lnb_batch_decode_until.argtypes = [vp, i32p, i32p, C.c_int, vp, vp, vp, f32p]
    L.lnb_ctx_set_stop_ids.argtypes = [vp, vp, C.c_int]
    L.lnb_decode_greedy_until.argtypes = [vp, C.c_int32, C.c_int, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), f32p]
    L.lnb_batch_profile_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_int, f32p]
    L.lnb_batch_set_state.argtypes = [vp, i32p, i32p]
    L.lnb_batch_check_error.argtypes = [vp]
    L.lnb_pipeline_tick_batch.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int)]
    L.lnb_pipeline_read_tokens.argtypes = [vp, C.c_int, C.c_int, vp]
    L.lnb_op_linear_mode.argtypes = [C.c_int, vp, vp, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lnb_op_argmax.argtypes = [C.c_int, vp, C.c_int, i32p]
    L.lnb_op_exp_table.argtypes = [C.c_int, C.c_float, vp]
    L.lnb_model_num_tensors.argtypes = [vp]
    L.lnb_model_tensor_info.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.lnb_checkpoint_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.lnb_checkpoint_close.argtypes = [vp]
    L.lnb_checkpoint_close.restype = None
    L.lnb_checkpoint_num_tensors.argtypes = [vp]
    L.lnb_checkpoint_find.argtypes = [vp, C.c_char_p]
    L.lnb_checkpoint_tensor.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                        C.POINTER(vp), C.POINTER(C.c_int64)]
    L.lnb_model_load_checkpoint.argtypes = [vp, vp]
    L.lnb_model_args_from_json.argtypes = [C.c_char_p, C.POINTER(ModelArgs)]
    L.lnb_tokenizer_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.lnb_tokenizer_free.argtypes = [vp]
    L.lnb_tokenizer_free.restype = None
    L.lnb_tokenizer_vocab_size.argtypes = [vp]
    L.lnb_tokenizer_special.argtypes = [vp, i32p, i32p, i32p, i32p]
    L.lnb_tokenizer_token_id.argtypes = [vp, C.c_char_p, C.c_int]
    L.lnb_tokenizer_piece.argtypes = [vp, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.lnb_tokenizer_encode.argtypes = [vp, C.c_char_p, C.c_int, i32p, C.c_int]
    L.lnb_tokenizer_encode_chat.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, i32p, C.c_int]
    L.lnb_tokenizer_stream_create.argtypes = [vp, C.POINTER(vp)]
    L.lnb_tokenizer_stream_free.argtypes = [vp]
    L.lnb_tokenizer_stream_free.restype = None
    L.lnb_tokenizer_decode_stream.argtypes = [vp, C.c_int32, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.lnb_tokenizer_stream_pending.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise LnbError(lib().lnb_last_error().decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    _chk(lib().lnb_device_count(C.byref(n)))
    return n.value


def device_info(device=0):
    """{name, arch, hbm_bytes, n_cus} of a device index (lnb_device_info)"""
    name, arch = C.create_string_buffer(256), C.create_string_buffer(256)
    hbm, cus = C.c_int64(0), C.c_int(0)
    _chk(lib().lnb_device_info(device, name, 256, C.byref(hbm), C.byref(cus), arch, 256))
    return {"name": name.value.decode(), "arch": arch.value.decode(), "hbm_bytes": hbm.value, "n_cus": cus.value}


def runtime_info(device=0, probe_queues=False):
    """lnb_runtime_info as a dict: hardware queues the HIP runtime was told to use / really uses, clocks, device"""
    ri = RuntimeInfo()
    _chk(lib().lnb_runtime_info(device, 1 if probe_queues else 0, C.byref(ri)))
    d = {k: getattr(ri, k) for k, _ in RuntimeInfo._fields_}
    d["device_name"] = ri.device_name.decode("utf-8", "replace"); d["arch"] = ri.arch.decode("utf-8", "replace")
    d["probe_ms"] = round(float(ri.probe_ms), 3)
    return d


def queue_warning(contexts_in_flight, info):
    """the sentence a host prints when it keeps more contexts in flight than the runtime has hardware queues (streams on one queue serialise)"""
    q = info.get("hw_queues_measured") or info.get("hw_queues_expected") or 4
    if contexts_in_flight <= q:
        return None
    why = (" -- HIP had been initialised before liblnb_hip.so was loaded, so its GPU_MAX_HW_QUEUES=16 default came too late: export the variable "
           "before the process starts" if info.get("hip_initialised_before_load") and info.get("hw_queues_set_by_library") else "")
    return "%d contexts in flight on %d hardware queues: streams that share a queue run one after the other%s" % (contexts_in_flight, q, why)


def can_access_peer(device, peer):
    out = C.c_int(0)
    _chk(lib().lnb_device_can_access_peer(device, peer, C.byref(out)))
    return bool(out.value)


def pci_bus_id(device):
    buf = C.create_string_buffer(32)
    _chk(lib().lnb_device_pci_bus_id(device, buf, 32))
    return buf.value.decode()


def rccl_selftest(device=0, n_bytes=1 << 20):
    """one-rank RCCL communicator, the pipeline's grouped send + recv to itself, bytes compared (lnb_pipeline_selftest)"""
    _chk(lib().lnb_pipeline_selftest(device, n_bytes))


DTYPES = {0: ("bf16", np.uint16), 1: ("f16", np.uint16), 2: ("f32", np.float32)}


class Checkpoint:
    """Read-only mmap of a PyTorch zip checkpoint (torch.TorchModelReader, src/torch/torchmodelreader.go:39-145)."""

    def __init__(self, path):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.lnb_checkpoint_open(os.fsencode(path), C.byref(self.h)))

    def __len__(self):
        return self.L.lnb_checkpoint_num_tensors(self.h)

    def tensor(self, i):
        """(name, dtype string, shape, numpy view of the mmap'ed bytes -- valid until close())"""
        name, dt, shape, rank, data, nb = C.c_char_p(), C.c_int(), (C.c_int64 * 4)(), C.c_int(), C.c_void_p(), C.c_int64()
        _chk(self.L.lnb_checkpoint_tensor(self.h, i, C.byref(name), C.byref(dt), shape, C.byref(rank), C.byref(data), C.byref(nb)))
        dname, npdt = DTYPES[dt.value]
        shp = tuple(shape[:rank.value])
        n = nb.value // np.dtype(npdt).itemsize
        arr = np.ctypeslib.as_array(C.cast(data.value, C.POINTER(C.c_uint16 if npdt is np.uint16 else C.c_float)), shape=(n,)) if n else np.empty(0, npdt)
        return name.value.decode("utf-8", "replace"), dname, shp, arr.reshape(shp)

    def find(self, name):
        return self.L.lnb_checkpoint_find(self.h, name.encode())

    def close(self):
        if self.h:
            self.L.lnb_checkpoint_close(self.h)
            self.h = C.c_void_p()


class Tokenizer:
    """tiktoken vocabulary + the reference's TokenizeString / Tokenize (src/inference/tokenize.go)."""

    def __init__(self, path):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.lnb_tokenizer_load(os.fsencode(path), C.byref(self.h)))
        self.vocab_size = self.L.lnb_tokenizer_vocab_size(self.h)
        b, e, t, m = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _chk(self.L.lnb_tokenizer_special(self.h, C.byref(b), C.byref(e), C.byref(t), C.byref(m)))
        self.bos, self.eos, self.eot, self.eom = b.value, e.value, t.value, m.value

    def encode(self, text):
        raw = text if isinstance(text, bytes) else text.encode("utf-8")
        cap = 4 * len(raw) + 16
        out = (C.c_int32 * cap)()
        n = self.L.lnb_tokenizer_encode(self.h, raw, len(raw), out, cap)
        if n < 0:
            raise LnbError(self.L.lnb_last_error().decode("utf-8", "replace"))
        return list(out[:n])

    def encode_chat(self, parts):
        """parts: [(header, content), ...] -> ids with the chat template of Tokenize(promptParts)"""
        hs = (C.c_char_p * len(parts))(*[h.encode("utf-8") for h, _ in parts])
        cs = (C.c_char_p * len(parts))(*[c.encode("utf-8") for _, c in parts])
        cap = 4 * sum(len(h) + len(c) for h, c in parts) + 64 * (len(parts) + 2)
        out = (C.c_int32 * cap)()
        n = self.L.lnb_tokenizer_encode_chat(self.h, hs, cs, len(parts), out, cap)
        if n < 0:
            raise LnbError(self.L.lnb_last_error().decode("utf-8", "replace"))
        return list(out[:n])

    def piece(self, tid):
        p, n = C.c_void_p(), C.c_int()
        _chk(self.L.lnb_tokenizer_piece(self.h, tid, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value) if n.value else b""

    def token_id(self, piece_bytes):
        return self.L.lnb_tokenizer_token_id(self.h, piece_bytes, len(piece_bytes))

    def stream(self):
        """a TokenToString decoding context (one per generation): .feed(token_id) -> (bytes released, added_to_waiting)"""
        return DecodeStream(self)

    def close(self):
        if self.h:
            self.L.lnb_tokenizer_free(self.h)
            self.h = C.c_void_p()


class DecodeStream:
    """InferenceEngine.TokenToString + generationDecodingContext.waitingBytes (src/inference/tokenize.go:197-239)"""

    def __init__(self, tok):
        self.L, self.h = tok.L, C.c_void_p()
        _chk(self.L.lnb_tokenizer_stream_create(tok.h, C.byref(self.h)))
        self.buf = C.create_string_buffer(1 << 16)

    def feed(self, tid):
        w = C.c_int(0)
        n = self.L.lnb_tokenizer_decode_stream(self.h, int(tid), self.buf, len(self.buf), C.byref(w))
        if n < 0:
            raise LnbError(self.L.lnb_last_error().decode("utf-8", "replace"))
        return self.buf.raw[:n], bool(w.value)

    def pending(self):
        p, n = C.c_void_p(), C.c_int()
        _chk(self.L.lnb_tokenizer_stream_pending(self.h, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value) if n.value else b""

    def close(self):
        if self.h:
            self.L.lnb_tokenizer_stream_free(self.h)
            self.h = C.c_void_p()


def model_args_from_json(path):
    """params.json -> dict of ModelArgs fields with the reference's defaults (src/model/modelargs.go:29-65)"""
    a = ModelArgs()
    _chk(lib().lnb_model_args_from_json(os.fsencode(path), C.byref(a)))
    return {f: getattr(a, f) for f, _ in ModelArgs._fields_}


class LlamaTransformer:
    """Device-resident model (one pipeline stage): model.NewLlamaTransformer, llamatransformer.go:64-113."""

    def __init__(self, device=0, layer_begin=0, layer_end=None, part_begin=None, part_end=None, **kw):
        """[layer_begin, layer_end) in blocks, or [part_begin, part_end) in thirds of a block (3l = attention part of block l,
        3l+1 = gate/up part, 3l+2 = down part: lnb_model_create_parts)"""
        d = dict(LLAMA_8B)
        d.update(kw)
        self.args = ModelArgs(**d)
        self.L = lib()
        self.h = C.c_void_p()
        if part_begin is None and part_end is None:
            part_begin, part_end = 3 * layer_begin, 3 * (self.args.n_layers if layer_end is None else layer_end)
        self.part_begin, self.part_end = int(part_begin), int(part_end)
        self.layer_begin, self.layer_end = self.part_begin // 3, (self.part_end + 2) // 3
        _chk(self.L.lnb_model_create_parts(C.byref(self.args), device, self.part_begin, self.part_end, C.byref(self.h)))
        self.ffn_hidden = self.L.lnb_model_ffn_hidden_dim(C.byref(self.args))
        self.head_dim = self.args.dim // self.args.n_heads

    def set_tensor(self, name, arr_u16, shape=None):
        a = np.ascontiguousarray(arr_u16, dtype=np.uint16)
        shp = tuple(shape) if shape is not None else a.shape
        s = (C.c_int64 * len(shp))(*shp)
        _chk(self.L.lnb_model_set_tensor(self.h, name.encode(), _p(a), s, len(shp)))

    def get_tensor(self, name, nelem):
        out = np.empty(nelem, dtype=np.uint16)
        _chk(self.L.lnb_model_get_tensor(self.h, name.encode(), _p(out), nelem))
        return out

    def fill_synthetic(self, seed=1234):
        _chk(self.L.lnb_model_fill_synthetic(self.h, seed))
        return self

    def load_checkpoint(self, ckpt):
        """bind every tensor of this stage from an open Checkpoint (torch.TorchModelReader + getTensor, loader.go:183-192)"""
        _chk(self.L.lnb_model_load_checkpoint(self.h, ckpt.h))
        return self

    def tensor_infos(self):
        out = []
        for k in range(self.L.lnb_model_num_tensors(self.h)):
            name, shape, rank = C.c_char_p(), (C.c_int64 * 2)(), C.c_int()
            _chk(self.L.lnb_model_tensor_info(self.h, k, C.byref(name), shape, C.byref(rank)))
            out.append((name.value.decode(), tuple(shape[:rank.value])))
        return out

    def finalize(self, rope_rows=0):
        _chk(self.L.lnb_model_finalize(self.h, rope_rows))
        return self

    @property
    def PrecomputedFreqsCis(self):
        rows = C.c_int(0)
        _chk(self.L.lnb_model_rope_table(self.h, None, 0, C.byref(rows)))
        out = np.empty((rows.value, self.head_dim // 2, 2), dtype=np.float32)
        _chk(self.L.lnb_model_rope_table(self.h, _p(out), out.size, C.byref(rows)))
        return out

    def weight_bytes(self):
        return int(self.L.lnb_model_weight_bytes(self.h))

    def enable_batch(self):
        """build the matrix-core friendly copy of the weights that batched decode streams (lnb_model_enable_batch); after finalize()"""
        _chk(self.L.lnb_model_enable_batch(self.h))
        return self

    def batch_bytes(self):
        return int(self.L.lnb_model_batch_bytes(self.h))

    def close(self):
        if self.h:
            self.L.lnb_model_destroy(self.h)
            self.h = C.c_void_p()


class InferenceContext:
    """model.NewInferenceContext (inferencecontext.go:17-46): device KV cache for one generation."""

    def __init__(self, transformer, seq_len):
        self.t, self.L = transformer, transformer.L
        self.SequenceLength = seq_len
        self.h = C.c_void_p()
        _chk(self.L.lnb_ctx_create(transformer.h, seq_len, C.byref(self.h)))

    def set_mode(self, mode):
        """MODE_EXACT (default, bit-identical to the reference) or MODE_FAST (split-K / bf16-MFMA tolerance mode)"""
        _chk(self.L.lnb_ctx_set_mode(self.h, {"exact": 0, "fast": 1}.get(mode, mode)))
        return self

    def set_schedule(self, sched):
        """'latency' (default: one generation owns the chip) or 'throughput' (several contexts in flight on one GPU share the CUs); same bits"""
        _chk(self.L.lnb_ctx_set_schedule(self.h, {"latency": 0, "throughput": 1}.get(sched, sched)))
        return self

    def set_attention(self, long_threshold=-1, force_zseq=0):
        """contexts above long_threshold use the long-context decode attention; force_zseq: always walk the serial f64 sum"""
        _chk(self.L.lnb_ctx_set_attention(self.h, long_threshold, force_zseq))
        return self

    def zseq_count(self):
        n = C.c_int(0)
        _chk(self.L.lnb_ctx_zseq_count(self.h, C.byref(n)))
        return n.value

    def norm_fallbacks(self):
        """fused-RMSNorm rows (decode) that left the branch-free item walk for the record walk since the context was created"""
        n = C.c_int(0)
        _chk(self.L.lnb_ctx_norm_fallbacks(self.h, C.byref(n)))
        return n.value

    def prefill_attention_form(self):
        """3: the last multi-row call ran attn_mfma3_kernel (scores once), 1: attn_mfma_kernel (scores twice: scratch refused), 0: neither"""
        n = C.c_int(0)
        _chk(self.L.lnb_ctx_prefill_attention_form(self.h, C.byref(n)))
        return n.value

    def Forward(self, tokens, start_pos, want_logits=True):
        """(*LlamaTransformer).Forward (llamatransformer.go:145-180) -> (logits f32 [S,V] | None, argmax of last row)."""
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        S, V = tok.size, self.t.args.vocab_size
        logits = np.empty((S, V), dtype=np.float32) if want_logits else None
        am = C.c_int32(-2)
        _chk(self.L.lnb_forward(self.h, _p(tok), S, start_pos, _p(logits) if want_logits else None, C.byref(am)))
        return logits, am.value

    def decode_greedy(self, token, start_pos, n_steps):
        out = np.empty(n_steps, dtype=np.int32)
        ms = C.c_float(0)
        _chk(self.L.lnb_decode_greedy(self.h, int(token), start_pos, n_steps, _p(out), C.byref(ms)))
        return out, ms.value

    def profile_kernel(self, which, pos, iters):
        ms = C.c_float(0)
        _chk(self.L.lnb_profile_kernel(self.h, which, pos, iters, C.byref(ms)))
        return ms.value

    def profile_ffn_pair(self, pos, iters, w2_delay_us=0, w2_lds_pad=0):
        """gate|up and down kernels of a block on two streams, w2 launched w2_delay_us behind w1|w3 (lnb_profile_ffn_pair): ms per pair"""
        ms = C.c_float(0)
        _chk(self.L.lnb_profile_ffn_pair(self.h, pos, iters, w2_delay_us, w2_lds_pad, C.byref(ms)))
        return ms.value

    def profile_kernel_stamps(self, which, pos):
        """in-kernel cycle stamps of one launch of a GEMV class -> (float64 [8 waves, 16], wall clock kHz); layout: include/lnb.h"""
        out = np.zeros((8, 16), dtype=np.float64)
        khz = C.c_int(0)
        _chk(self.L.lnb_profile_kernel_stamps(self.h, which, pos, _p(out), C.byref(khz)))
        return out, khz.value

    def set_stop_ids(self, ids):
        """model.StopTokenIds on the device (inference.go:233-252)"""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        _chk(self.L.lnb_ctx_set_stop_ids(self.h, _p(a) if a.size else None, int(a.size)))
        return self

    def decode_greedy_until(self, token, start_pos, max_steps):
        """-> (tokens generated: max_steps of them unless a stop id ended the run, finished flag, device ms)"""
        out = np.empty(max_steps, dtype=np.int32)
        ms, n, fin = C.c_float(0), C.c_int(0), C.c_int(0)
        _chk(self.L.lnb_decode_greedy_until(self.h, int(token), start_pos, max_steps, _p(out), C.byref(n), C.byref(fin), C.byref(ms)))
        return out[:n.value].copy(), bool(fin.value), ms.value

    def CacheK(self, layer):
        return self._kv(layer, 0)

    def CacheV(self, layer):
        return self._kv(layer, 1)

    def _kv(self, layer, which):
        a = self.t.args
        out = np.empty((self.SequenceLength, a.n_kv_heads, self.t.head_dim), dtype=np.uint16)
        _chk(self.L.lnb_ctx_read_kv(self.h, layer, which, _p(out)))
        return out

    def reset(self):
        _chk(self.L.lnb_ctx_reset(self.h))

    def close(self):
        if self.h:
            _chk(self.L.lnb_ctx_destroy(self.h))             # refused while a live Batch holds the context: the handle (and the device memory) stays
            self.h = C.c_void_p()


class Batch:
    """1..128 InferenceContexts of one transformer decoded together: one pass over the weights per step for all of them, every sequence
    bit-identical to its single-sequence run (lnb_batch_*).  The contexts keep their own caches and positions."""

    def __init__(self, contexts):
        self.ctxs = list(contexts)
        self.L = self.ctxs[0].L
        arr = (C.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs])
        self.h = C.c_void_p()
        _chk(self.L.lnb_batch_create(arr, len(self.ctxs), C.byref(self.h)))

    def decode(self, tokens, start_pos, n_steps):
        """sequence s continues from tokens[s] at position start_pos[s]; -> (int32 [n, n_steps], device ms of the n_steps)"""
        n = len(self.ctxs)
        tok = np.ascontiguousarray(tokens, dtype=np.int32); pos = np.ascontiguousarray(start_pos, dtype=np.int32)
        assert tok.size == n and pos.size == n
        out = np.empty((n, n_steps), dtype=np.int32)
        ms = C.c_float(0)
        _chk(self.L.lnb_batch_decode(self.h, tok.ctypes.data_as(C.POINTER(C.c_int32)), pos.ctypes.data_as(C.POINTER(C.c_int32)), n_steps, _p(out), C.byref(ms)))
        return out, ms.value

    def decode_until(self, tokens, start_pos, max_steps):
        """decode with the member contexts' stop ids: -> (list of per-sequence token arrays (their valid lengths differ), device ms)"""
        n = len(self.ctxs)
        tok = np.ascontiguousarray(tokens, dtype=np.int32); pos = np.ascontiguousarray(start_pos, dtype=np.int32)
        out = np.empty((n, max_steps), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        ms = C.c_float(0)
        fin = np.zeros(n, dtype=np.int32)
        _chk(self.L.lnb_batch_decode_until(self.h, tok.ctypes.data_as(C.POINTER(C.c_int32)), pos.ctypes.data_as(C.POINTER(C.c_int32)), max_steps, _p(out), _p(cnt), _p(fin), C.byref(ms)))
        self.finished = [bool(f) for f in fin]               # per sequence: a stop id ended it (start_pos[s] < 0 keeps an ended sequence frozen in the next chunk)
        return [out[s, :cnt[s]].copy() for s in range(n)], ms.value

    def profile_kernel(self, which, pos, iters):
        ms = C.c_float(0)
        _chk(self.L.lnb_batch_profile_kernel(self.h, which, pos, iters, C.byref(ms)))
        return ms.value

    def set_state(self, tokens, start_pos):
        """positions (and optionally next input tokens) of the sequences before a run of Pipeline.tick_batch steps"""
        pos = np.ascontiguousarray(start_pos, dtype=np.int32)
        tok = None if tokens is None else np.ascontiguousarray(tokens, dtype=np.int32)
        _chk(self.L.lnb_batch_set_state(self.h, None if tok is None else tok.ctypes.data_as(C.POINTER(C.c_int32)), pos.ctypes.data_as(C.POINTER(C.c_int32))))
        return self

    def boundary_ptr(self, which):
        """device address of what a batched tick exchanges: 0 = hidden states [n, dim] bf16 (stage input / output), 1 = the n int32 token words"""
        ptr = self.L.lnb_batch_boundary_ptr(self.h, which)
        if not ptr:
            _chk(-1)
        return int(ptr)

    def check_error(self):
        """after Pipeline.sync(): raises if a pipeline tick of this batch met a token outside the vocabulary / an all-NaN logits row"""
        _chk(self.L.lnb_batch_check_error(self.h))
        return self

    def close(self):
        if self.h:
            self.L.lnb_batch_destroy(self.h)
            self.h = C.c_void_p()


class Pipeline:
    """One rank of the layer-sharded pipeline behind the C ABI (lnb_pipeline_*): RCCL send / recv straight from / into the stage's
    device buffers, stage steps as captured graphs, nothing synchronised per tick."""

    def __init__(self, transformer, rank, world, unique_id=None, loopback_group=None, host_transport=False):
        """unique_id: rank 0's 128 RCCL id bytes (world > 1); loopback_group: a name -- the in-process transport instead of RCCL;
        host_transport: no transport at all -- ticks only `run`, the caller moves the boundary buffers (pipeline.run_ticks_batched_torch)"""
        self.L, self.rank, self.world = transformer.L, rank, world
        self.h = C.c_void_p()
        if host_transport:
            _chk(self.L.lnb_pipeline_init_host(transformer.h, rank, world, C.byref(self.h)))
            return
        if loopback_group is not None:
            _chk(self.L.lnb_pipeline_init_loopback(transformer.h, rank, world, loopback_group.encode(), C.byref(self.h)))
            return
        idp = C.c_char_p(bytes(unique_id)) if unique_id is not None else None
        _chk(self.L.lnb_pipeline_init(transformer.h, rank, world, idp, C.byref(self.h)))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _chk(lib().lnb_pipeline_unique_id(buf))
        return bytes(buf.raw)

    def tick(self, run=None, run_rows=0, run_pos=0, run_tokens=None, send=None, send_rows=0, recv=None, recv_rows=0):
        slot = C.c_int(-1)
        tok = None if run_tokens is None else np.ascontiguousarray(run_tokens, dtype=np.int32)
        _chk(self.L.lnb_pipeline_tick(self.h, run.h if run else None, run_rows, run_pos, _p(tok) if tok is not None else None,
                                      send.h if send else None, send_rows, recv.h if recv else None, recv_rows, C.byref(slot)))
        return slot.value

    def tick_batch(self, run=None, send=None, recv=None):
        """the tick with a Batch as the unit: one pass over the stage's weights for all its sequences; -> first token-log slot (last rank)"""
        slot = C.c_int(-1)
        _chk(self.L.lnb_pipeline_tick_batch(self.h, run.h if run else None, send.h if send else None, recv.h if recv else None, C.byref(slot)))
        return slot.value

    def sync(self):
        _chk(self.L.lnb_pipeline_sync(self.h))

    def comm_count(self):
        """ranks the exchange spans, as the transport itself reports it (ncclCommCount)"""
        n = C.c_int(0)
        _chk(self.L.lnb_pipeline_comm_count(self.h, C.byref(n)))
        return n.value

    def read_tokens(self, first_slot, n):
        out = np.empty(n, dtype=np.int32)
        _chk(self.L.lnb_pipeline_read_tokens(self.h, first_slot, n, _p(out)))
        return out

    def close(self):
        if self.h:
            self.L.lnb_pipeline_destroy(self.h)
            self.h = C.c_void_p()


class InferenceEngine:
    """Greedy generation loop of src/inference/inference.go:173-254 over the device path:
    prefill(prompt) through Forward, then the decode steps as hipGraph replays on the device."""

    def __init__(self, transformer, seq_len, stop_token_ids=()):
        self.t, self.seq_len, self.stop = transformer, seq_len, set(stop_token_ids)

    def GenerateTokens(self, prompt_tokens, max_new=None):
        prompt = list(prompt_tokens)
        if len(prompt) >= self.seq_len:
            raise LnbError("context SequenceLength %d must be higher than prompt tokens length %d" % (self.seq_len, len(prompt)))
        ctx = InferenceContext(self.t, self.seq_len)
        try:
            n_new = self.seq_len - len(prompt) if max_new is None else min(max_new, self.seq_len - len(prompt))
            _, first = ctx.Forward(prompt, 0, want_logits=False)
            out = [first]
            if first not in self.stop and n_new > 1:
                more, _ = ctx.decode_greedy(first, len(prompt), n_new - 1)
                for t in more:
                    out.append(int(t))
                    if int(t) in self.stop:
                        break
            return out
        finally:
            ctx.close()


def op_linear(x_u16, w_u16, rw=0, device=0):
    x = np.ascontiguousarray(x_u16, dtype=np.uint16)
    w = np.ascontiguousarray(w_u16, dtype=np.uint16)
    rows, k = x.shape
    n = w.shape[0]
    y = np.empty((rows, n), dtype=np.uint16)
    _chk(lib().lnb_op_linear(device, _p(x), _p(w), _p(y), rows, n, k, rw))
    return y


def op_rmsnorm_linear(x_u16, norm_w_u16, eps, w_u16, rw=0, device=0):
    x = np.ascontiguousarray(x_u16, dtype=np.uint16)
    w = np.ascontiguousarray(w_u16, dtype=np.uint16)
    nw = np.ascontiguousarray(norm_w_u16, dtype=np.uint16)
    rows, k = x.shape
    n = w.shape[0]
    y = np.empty((rows, n), dtype=np.uint16)
    _chk(lib().lnb_op_rmsnorm_linear(device, _p(x), _p(nw), np.float32(eps), _p(w), _p(y), rows, n, k, rw))
    return y


def op_linear_mode(x_u16, w_u16, mode, norm_w_u16=None, eps=1e-5, rw=0, device=0):
    x = np.ascontiguousarray(x_u16, dtype=np.uint16)
    w = np.ascontiguousarray(w_u16, dtype=np.uint16)
    nw = None if norm_w_u16 is None else np.ascontiguousarray(norm_w_u16, dtype=np.uint16)
    rows, k = x.shape
    n = w.shape[0]
    y = np.empty((rows, n), dtype=np.uint16)
    _chk(lib().lnb_op_linear_mode(device, _p(x), None if nw is None else _p(nw), np.float32(eps), _p(w), _p(y), rows, n, k, rw, mode))
    return y


def op_argmax(logits_u16, device=0):
    a = np.ascontiguousarray(logits_u16, dtype=np.uint16).ravel()
    out = C.c_int32(-2)
    _chk(lib().lnb_op_argmax(device, _p(a), a.size, C.byref(out)))
    return out.value


def op_exp_table(divisor=1.0, device=0):
    """out[s] = exp(float64(trunc_bf16(float32(bf16 s) / divisor))) for all 65536 raw scores, as the attention kernels evaluate it (lnb_op_exp_table)"""
    out = np.empty(65536, dtype=np.float64)
    _chk(lib().lnb_op_exp_table(device, C.c_float(divisor), _p(out)))
    return out


_M64 = (1 << 64) - 1


def _splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def synth_tokens(seed, n, vocab):
    """Synthetic prompt of DESIGN.md ("Synthetic weights"): tok[i] = splitmix64(seed ^ i*golden) mod vocab."""
    return np.array([_splitmix64(seed ^ ((i * 0x9E3779B97F4A7C15) & _M64)) % vocab for i in range(n)], dtype=np.int32)

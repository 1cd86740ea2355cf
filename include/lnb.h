/*
 * lnb.h -- C ABI of liblnb_hip.so: the MI355X-native replacement for the LlamaTransformer.Forward
 * hot path of adalkiran/llama-nuts-and-bolts (reference file:line citations are relative to that repo).
 *
 * This is the drop-in boundary: the reference has no plugin registry, the seam is
 * the Go method itself, so each entry point below is what the body of one Go function becomes through
 * cgo (binding shown in INTEGRATION.md).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Conventions: every function returns 0 on success, <0 on error; lnb_last_error() returns a
 * thread-local message (the Go side wraps it in errors.New, mirroring the reference's fmt.Errorf paths).
 * The library owns DEVICE memory only: host pointers are never retained after a call returns
 * (the reference's weight tensors are sub-slices of an mmap, src/torch/types.go:51-55).
 * Handles: lnb_model is immutable after lnb_model_finalize and may be shared by several lnb_ctx
 * (reference: the transformer is read-only after construction, one InferenceContext per GenerateString
 * call, src/inference/inference.go:174); an lnb_ctx must be used by one thread at a time.  The entry points
 * may be mixed on one context (ticks, lnb_forward, lnb_decode_greedy, lnb_profile_kernel): each one re-establishes the device-side position.
 */
#ifndef LNB_H
#define LNB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct lnb_model lnb_model;
typedef struct lnb_ctx lnb_ctx;

/* mirrors model.ModelArgs (src/model/modelargs.go:12-27; defaults :29-44) */
typedef struct lnb_model_args {
    int32_t dim;                 /* 4096 */
    int32_t n_layers;            /* 32 */
    int32_t n_heads;             /* 32 */
    int32_t n_kv_heads;          /* 8; <0 => n_heads (llamatransformer.go:73-75) */
    int32_t vocab_size;          /* 128256 */
    int32_t multiple_of;         /* 1024 */
    double  ffn_dim_multiplier;  /* 1.3; <= -1 => unset (llamatransformer.go:573-575) */
    float   norm_eps;            /* 1e-5 */
    int32_t use_scaled_rope;     /* 1 */
    double  rope_theta;          /* 500000; <=0 => 500000 (llamatransformer.go:80-82) */
    int32_t max_seq_len;         /* 2048; RoPE table has 2*max_seq_len rows (llamatransformer.go:109) */
} lnb_model_args;

/* ABI version of this header.  A binding (cgo, ctypes, JNI ...) compares it with lnb_abi_version() of the library it loaded and refuses a mismatch:
 * round 5 changed lnb_batch_decode_until's signature (`finished` in the middle) -- a stale binding would have passed a float* where the library writes
 * n int32 values.  Bumped whenever an existing entry point changes its arguments; additions do not bump it.
 *   6: lnb_batch_decode_until(..., n_generated, finished, ms_out); lnb_runtime_info; lnb_abi_version itself. */
#define LNB_ABI_VERSION 6
int lnb_abi_version(void);

const char* lnb_last_error(void);
int lnb_device_count(int* out_count);
/* Runtime facts a host needs before it trusts a run with SEVERAL contexts in flight (one InferenceContext per generation, one goroutine each:
 * src/inference/inference.go:163-174).  The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and
 * streams that share a queue serialise; while it is being loaded this library exports 16 unless the host set a value or LNB_KEEP_HW_QUEUES=1 --
 * but the runtime reads the variable at ITS initialisation, so a host that used HIP before loading the library keeps its 4 queues (two
 * contexts in flight then run slower than one).  hw_queues_measured (probe_queues != 0, 4-20 ms): streams that really ran concurrently.
 * A host warns when contexts in flight > min(hw_queues_expected, hw_queues_measured or expected) (bench.py, go/inferencecontext_hip.go do). */
typedef struct lnb_runtime_info_t {
    int32_t abi_version;
    int32_t device, n_cus;
    int32_t shader_clock_khz, memory_clock_khz, wall_clock_khz;   /* hipDeviceAttributeClockRate / MemoryClockRate / WallClockRate */
    int32_t hw_queues_env;                 /* GPU_MAX_HW_QUEUES in the process environment now (0: unset = the runtime's default 4) */
    int32_t hw_queues_set_by_library;      /* 1: the value is this library's load-time default */
    int32_t hip_initialised_before_load;   /* 1: /dev/kfd was already open when the library was loaded: the runtime had read the variable before */
    int32_t hw_queues_expected;            /* what the runtime will have honoured given the three fields above */
    int32_t hw_queues_measured;            /* probe: 32 spinning one-wave kernels on 32 streams ran in 32 / this many rounds (0: not probed) */
    float   probe_ms;
    char    device_name[64];
    char    arch[32];
} lnb_runtime_info_t;
int lnb_runtime_info(int device, int probe_queues, lnb_runtime_info_t* out);
/* preflight of a multi-GPU host (bench.py --gpus N prints it per rank before any timing): the device behind an index (any out pointer may be
 * NULL) and whether `device` can map `peer`'s memory -- the peer-to-peer path RCCL's ncclSend / ncclRecv ride on (xGMI inside a node) */
int lnb_device_info(int device, char* name, int name_cap, int64_t* hbm_bytes, int* n_cus, char* arch, int arch_cap);
int lnb_device_can_access_peer(int device, int peer, int* out);
/* The PCI bus id of a device index ("0000:c1:00.0"): what tells two ranks apart that both say "device 0" because their launcher gave each of them
 * one visible GPU (a multi-GPU preflight compares these, not the indices). */
int lnb_device_pci_bus_id(int device, char* out, int cap);

/* ---- model: replaces model.NewLlamaTransformer (src/model/llamatransformer.go:64-113) --------------
 * A model handle owns the device copy of one pipeline STAGE: transformer blocks [layer_begin,layer_end),
 * plus tok_embeddings when layer_begin == 0 and norm + output when layer_end == n_layers.
 * layer_begin=0, layer_end=n_layers is the whole model on one GPU. */
int lnb_model_create(const lnb_model_args* args, int device, int layer_begin, int layer_end, lnb_model** out);
/* The same with the range in THIRDS of a block: 3l = attention part of block l (attention_norm, wq|wk|wv, attention, wo + residual,
 * llamatransformer.go:222-232), 3l+1 = gate/up part (ffn_norm, w1|w3, SiLU*up, :237, :601-617), 3l+2 = down part (w2 + residual,
 * :619, :248).  A pipeline stage may start or end inside a block: after the attention part the live state is again one [S, dim]
 * vector; after the gate/up part it is that vector plus the [S, ffn_hidden] activations (lnb_ctx_hidden_ptr(c, 2)).  The parts cost
 * about the same HBM time, so pipeline.stage_parts balances the stages to a third of a block.
 * lnb_model_create(.., lb, le, ..) == lnb_model_create_parts(.., 3*lb, 3*le, ..). */
int lnb_model_create_parts(const lnb_model_args* args, int device, int part_begin, int part_end, lnb_model** out);
int lnb_model_destroy(lnb_model* m);

/* FFN hidden size derivation (llamatransformer.go:569-577) */
int lnb_model_ffn_hidden_dim(const lnb_model_args* args);

/* Bind one checkpoint tensor by its Meta key (the names the reference binds with getTensor/getLayerTensor,
 * llamatransformer.go:84,98,105,191,202,273-283,580-587; shape check = loader.go:183-192).
 * host_bf16 is the reference layout: row-major [out_features,in_features] bf16 bits; it is copied to the
 * device and re-tiled; the caller keeps ownership.  Tensors of layers outside this stage are rejected. */
int lnb_model_set_tensor(lnb_model* m, const char* name, const uint16_t* host_bf16, const int64_t* shape, int rank);
/* read a tensor back in the reference layout (tests / debugging) */
int lnb_model_get_tensor(lnb_model* m, const char* name, uint16_t* host_bf16, int64_t nelem);
/* the tensors this model stage binds: index 0..n-1 -> checkpoint name, expected shape (rank 1: [n]; rank 2: [out,in]) */
int lnb_model_num_tensors(const lnb_model* m);
int lnb_model_tensor_info(const lnb_model* m, int index, const char** name, int64_t* shape2, int* rank);
/* random-init every tensor of this stage on the device with the counter-based generator of DESIGN.md
 * ("Synthetic weights"): no checkpoint is needed for benchmarking */
int lnb_model_fill_synthetic(lnb_model* m, uint64_t seed);
/* builds PrecomputedFreqsCis (llamatransformer.go:109,694-751) and the SiLU table (src/ml/activations.go:15-20);
 * rope_rows <= 0 selects the reference's 2*max_seq_len rows, a larger value extends the table by the same formula */
int lnb_model_finalize(lnb_model* m, int rope_rows);
/* LlamaTransformer.PrecomputedFreqsCis as [rows][head_dim/2][2] f32 (exported field, llamatransformer.go:24) */
int lnb_model_rope_table(lnb_model* m, float* out, int64_t nfloats, int* rows_out);
int64_t lnb_model_weight_bytes(lnb_model* m);

/* ---- context: replaces model.NewInferenceContext (src/model/inferencecontext.go:17-46) ---------------
 * device-resident, zero-filled CacheK/CacheV [seq_len, n_kv_heads, head_dim] bf16 per owned layer.
 * Context length: up to about 23000 positions (the long-context decode attention keeps 4 bytes per position in the LDS); head_dim 32, 64
 * or 128.  Calls of 2..15 rows use a kernel that stages 12 bytes per position and fail beyond ~7800 positions (head_dim 128; ~10800 at 64):
 * one-token calls and calls of 16 or more rows have no such limit. */
int lnb_ctx_create(lnb_model* m, int seq_len, lnb_ctx** out);
int lnb_ctx_destroy(lnb_ctx* c);
int lnb_ctx_reset(lnb_ctx* c);                                       /* zero the caches again */
/* InferenceContext.CacheK/CacheV[layer] (exported, poked by llamatransformer_simulated_test.go:527-538) */
int lnb_ctx_read_kv(lnb_ctx* c, int layer, int which /*0=K 1=V*/, uint16_t* host_bf16);
/* Arithmetic mode of a context.  LNB_MODE_EXACT (default): every matmul output is the reference's single k-ordered f32 chain
 * (src/ml/operations_lineartransform.go:46-65), every intermediate bit-identical to the Go CPU path.  LNB_MODE_FAST: the same
 * operators and bf16 truncation points with split-K f32 sums (decode) and bf16 matrix-core GEMMs (prefill): HBM / MFMA bound instead
 * of add-latency bound.  NOT a parity mode: per operator it stays within one bf16 ulp of the chain, but after 32 blocks the logits are
 * NOT within the north star's 1e-2 of the reference's (measured on the 8B shape: max |dlogit| 0.578, argmax differs in 13.9 % of
 * teacher-forced steps -- NOTES.md 6.2) and token ids diverge.  Opt-in, per context, switchable between calls; the KV cache is
 * shared by both modes. */
enum { LNB_MODE_EXACT = 0, LNB_MODE_FAST = 1 };
int lnb_ctx_set_mode(lnb_ctx* c, int mode);
int lnb_ctx_get_mode(const lnb_ctx* c);
/* Kernel FORMS of the exact one-token steps of a context (same arithmetic, same bits).  LNB_SCHED_LATENCY (default): one generation owns the
 * chip -- every launch takes all CUs with eight or nine waves and 91..124 KB of LDS per workgroup.  LNB_SCHED_THROUGHPUT: for several
 * generations in flight on one GPU, one context and stream each (the reference's one InferenceContext per generation, inference.go:174; what
 * every pipeline rank runs): every workgroup stays at or below 57 KB of LDS, so that a chain-bound launch of one context shares the CUs with
 * the HBM-bound gate|up launch of another.  Slower for a single stream, faster in aggregate (bench.py: sequences_in_flight).  Switchable
 * between calls; the context's captured graphs are dropped. */
enum { LNB_SCHED_LATENCY = 0, LNB_SCHED_THROUGHPUT = 1 };
int lnb_ctx_set_schedule(lnb_ctx* c, int sched);
int lnb_ctx_get_schedule(const lnb_ctx* c);
/* Decode attention form (both bit-identical to the reference arithmetic): one-token calls whose context exceeds long_threshold
 * positions use the long-context kernels (scores over all CUs, PV per (head, 16-dim slice), softmax denominator certified against
 * the reference's serial f64 sum instead of walked).  long_threshold < 0: keep (default 512, env LNB_ATTN_LONG_T).  force_zseq is a set of
 * test / experiment switches: bit 0 (1) = always walk the serial sum; bit 3 (8) = run both phases in ONE launch (attn_one_kernel, round 6: the
 * (head, slice) workgroups exchange the scores inside the launch behind a BOUNDED poll -- a workgroup whose peers are not resident computes their
 * share itself, so two contexts can never wait for each other; same bits; measured slower than the two launches, hence opt-in, also by
 * LNB_ATTN_ONE=1; latency schedule only); bit 2 (4) = one launch with every poll timing out; bit 1 (2) = two launches whatever the environment says.
 * lnb_ctx_zseq_count: rows that could not be certified and walked the serial sum. */
int lnb_ctx_set_attention(lnb_ctx* c, int long_threshold, int force_zseq);
int lnb_ctx_zseq_count(lnb_ctx* c, int* out);
/* Diagnostics of the fused RMSNorm (llamatransformer.go:222, :237, :166: RMSNorm in front of wq|wk|wv, w1|w3, output): rows of one-token calls
 * whose sum of squares left the branch-free item walk for the slower record walk (same bits either way; ~2.5 % of gaussian rows). */
int lnb_ctx_norm_fallbacks(lnb_ctx* c, int* out);
/* Which matrix-core attention the context's LAST multi-row call (llamatransformer.go:409-514 with S >= 16, exact mode) ran: 3 = attn_mfma3_kernel (the scores
 * computed once, their 16-bit exp-table indices kept in the context's scratch), 1 = attn_mfma_kernel (the scores computed twice: the scratch was refused --
 * above LNB_ATTN_SIDX_MB, or no device memory for it), 0 = none (no such call yet, fewer than 16 rows, head_dim 32, tolerance mode).  Same bits either way;
 * a bench prints it so that a silently refused scratch shows.  The scratch is 512 bytes per (head, 16 query rows, 16 positions) -- 1.07 GB for a 4096-row prompt of the
 * 8B shape --, one per context, grown on demand; a context's first one-token call after the prompt frees it when it is larger than LNB_ATTN_SIDX_KEEP_MB (default 256). */
int lnb_ctx_prefill_attention_form(const lnb_ctx* c, int* out);
/* optional per-layer progress hook = infContext.Logf("Transformer block layer %d / %d was run, took %.4f sec(s)")
 * (llamatransformer.go:157-163); forces a per-layer stream sync, so it is off by default */
typedef void (*lnb_layer_cb)(int layer_1based, int n_layers, double secs, void* user);
int lnb_ctx_set_layer_callback(lnb_ctx* c, lnb_layer_cb cb, void* user);

/* ---- forward: replaces (*LlamaTransformer).Forward (src/model/llamatransformer.go:145-180) -------------
 * tokens: [seq] int32 (caller-owned, not retained, inference.go:195-202); start_pos as in the reference.
 * logits_out: caller-allocated [seq, vocab_size] f32, bf16-representable values (llamatransformer.go:170-177);
 *   NULL => only the last row is evaluated and only argmax_last_out is produced.
 * argmax_last_out: ml.Argmax of the last row (first maximum wins, operations_impl.go:529-541); may be NULL.
 * Errors (same conditions as the reference): seq == 0 "empty token array" (llamatransformer.go:146-148);
 * start_pos+seq beyond the RoPE table or the KV cache (tensor.go:275-279; the reference silently drops the
 * SetSlice error at llamatransformer.go:402-403 and then fails in Slice :409 -- here it fails up front);
 * seq > 1 with (start_pos+seq) % seq != 0 "two tensor shapes cannot be broadcasted" (tensor.go:414-428).
 * Requires a whole-model handle (layer_begin == 0 && layer_end == n_layers). */
int lnb_forward(lnb_ctx* c, const int32_t* tokens, int seq, int start_pos, float* logits_out, int32_t* argmax_last_out);

/* ---- greedy loop on the device: the decode half of InferenceEngine.generateTokensInternal
 * (src/inference/inference.go:194-252).  Starting from `token` at position start_pos (its KV is computed by
 * the first step), runs n_steps one-token Forward+Argmax steps as replays of one captured hipGraph without
 * any host round trip; out_tokens[i] is the token generated by step i.  ms_out (optional) = device time of
 * the n_steps measured with HIP events on the library's stream. */
int lnb_decode_greedy(lnb_ctx* c, int32_t token, int start_pos, int n_steps, int32_t* out_tokens, float* ms_out);
/* (lnb_decode_greedy and lnb_batch_decode always produce n_steps tokens: stop ids of the context are NOT compared by them -- use the
 * _until forms, which report how far the run got.) */
/* Stop ids ON THE DEVICE (inference.go:233-252: generation ends with the first token that is one of model.StopTokenIds -- <|eot_id|>,
 * <|eom_id|> -- and that token is emitted).  lnb_ctx_set_stop_ids gives a context up to 8 of them (0: none); the token-feedback kernel then
 * freezes the generation (position, token word, token log) at the first match, so a run of any length can be enqueued without a host check
 * per token.  lnb_decode_greedy_until = lnb_decode_greedy that reports how far it got: *n_generated tokens are valid (max_steps unless a
 * stop id ended the run; the stop token is the last of them), *finished (optional) = 1 if one did.  Same tokens whatever the chunking. */
int lnb_ctx_set_stop_ids(lnb_ctx* c, const int32_t* ids, int n);
int lnb_decode_greedy_until(lnb_ctx* c, int32_t token, int start_pos, int max_steps, int32_t* out_tokens, int* n_generated, int* finished, float* ms_out);

/* ---- batched exact decode: several generations per pass over the weights ----------------------------------
 * The reference runs one generation per InferenceContext (src/inference/inference.go:174) and shares the weight matrix across the rows of a
 * call (src/ml/operations_lineartransform.go:173-193).  A batch groups 1..128 contexts of ONE whole-model handle; per step, every
 * sequence's one-token Forward + Argmax (inference.go:194-252) is evaluated in a single pass over the weights: up to 16 sequences are the 16
 * columns of the f32 matrix-core instruction, which computes each column's k-ordered chain exactly; 17..128 sequences are the rows of the
 * prefill's streaming product (1 / 2 / 4 tiles of 16 sequences per wave, same instruction, same chains) -- except that 17..32 sequences keep
 * the column form, as two groups of 16, for the matrices with few output rows (wq|wk|wv, wo, w2).  Per-sequence position, RoPE row, KV
 * append and attention; every sequence's tokens, logits and caches are bit-identical to its single-sequence run (and the CPU reference).
 * lnb_model_enable_batch: builds the weights' second, matrix-core friendly copy on the device (once, after lnb_model_finalize; it costs the
 *   model's matrix bytes again -- 15 GB for the 8B shape; fails cleanly when that does not fit, the other entry points stay usable).  Works
 *   on a whole model and on a pipeline stage of whole blocks.  Since round 5 it is a pure PERFORMANCE option: prompts (lnb_forward with 16 or
 *   more rows) stream the resident weight layouts, and lnb_batch_create on a model without the copy runs every product of the batch as rows of
 *   that streaming kernel, whatever the number of sequences (64..128 sequences: 3-8 % slower than with the copy; up to 32 the column forms,
 *   which read the copy, are the faster ones by more).  Same bits either way.
 * lnb_batch_create: the contexts keep their own KV caches and positions (prefill each with lnb_forward first); seq_len of each context at
 *   most ~7.8 K positions (head_dim 128).  A context must not be used by another call while a batch call that contains it runs.
 * lnb_batch_decode: sequence s continues from tokens[s] at position start_pos[s] (different positions are fine); n_steps greedy steps for
 *   all of them as replays of one captured hipGraph; out_tokens[s * n_steps + i] = token i of sequence s.  Afterwards every context's
 *   cache holds its new rows: lnb_forward / lnb_decode_greedy / another batch may continue it.
 * Lifetime: a batch holds its member contexts' device pointers (tables, captured graphs): lnb_ctx_destroy on a member FAILS while the
 * batch is alive -- destroy the batch first (the Go binding's Close does it in that order). */
typedef struct lnb_batch lnb_batch;
int lnb_model_enable_batch(lnb_model* m);
int64_t lnb_model_batch_bytes(lnb_model* m);
int lnb_batch_create(lnb_ctx* const* ctxs, int n, lnb_batch** out);
int lnb_batch_destroy(lnb_batch* b);
int lnb_batch_decode(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos, int n_steps, int32_t* out_tokens, float* ms_out);
/* ... with per-sequence stop ids (lnb_ctx_set_stop_ids on the member contexts, above): n_generated[s] tokens of row s are valid, the last one
 * the stop token if the sequence finished; finished[s] (optional, may be NULL) = 1 if a stop id ended sequence s -- n_generated[s] == max_steps
 * alone cannot tell.  A finished sequence's position and caches stay where they stopped.  Decoding in chunks: pass start_pos[s] < 0 for a
 * sequence that has finished -- it stays frozen (n_generated[s] = 0, finished[s] = 1) instead of restarting from its stop token; its
 * tokens[s] is ignored.  (Stop ids are a single-GPU feature of the batch: lnb_batch_set_state on a stage of a multi-stage pipeline refuses
 * contexts that carry any -- only the last stage would see the token.) */
int lnb_batch_decode_until(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos, int max_steps, int32_t* out_tokens, int32_t* n_generated, int32_t* finished, float* ms_out);
/* measurement aid: average HIP-event time of one kernel class of the batched step (which as lnb_profile_kernel; a norm launch counts
 * with the product it feeds), every sequence placed at `pos`; overwrites the caches' row `pos` */
int lnb_batch_profile_kernel(lnb_batch* b, int which, int pos, int iters, float* avg_ms_out);

/* ---- pipeline-stage form (layer-sharded multi-GPU) -------------------------------
 * hidden state buffers live on the device and are owned by the ctx: [seq_len, dim] bf16.
 * which: 0 = stage input, 1 = stage output, 2 = the [seq_len, ffn_hidden] gate*up activations, part of the hand-off when the
 * stage boundary lies between a block's gate/up and down parts.  RCCL send/recv (done by the host layer) targets these pointers. */
void* lnb_ctx_hidden_ptr(lnb_ctx* c, int which);
/* Runs this stage's layers.  tokens != NULL only on the first stage (embedding gather); otherwise the input
 * hidden state must already be in hidden_ptr(0).  On the last stage logits_out/argmax_last_out behave as in
 * lnb_forward; on other stages they must be NULL and the result is left in hidden_ptr(1). */
int lnb_forward_stage(lnb_ctx* c, const int32_t* tokens, int seq, int start_pos, float* logits_out, int32_t* argmax_last_out);
/* The same in two halves, for a pipeline rank that exchanges hidden states with its neighbours WHILE the stage computes: _begin
 * enqueues the stage on the ctx's stream and returns (want_argmax: last stage only, norm + output + argmax of the last row, as
 * inference.go:207-211 needs); _end blocks until it has finished and returns the token.  One begin per end. */
int lnb_forward_stage_begin(lnb_ctx* c, const int32_t* tokens, int seq, int start_pos, int want_argmax);
int lnb_forward_stage_end(lnb_ctx* c, int32_t* argmax_last_out);
/* block the calling thread until everything enqueued for this ctx has finished */
int lnb_ctx_synchronize(lnb_ctx* c);
/* raw HIP stream (hipStream_t) the ctx enqueues on, for event timing by the caller */
void* lnb_ctx_stream(lnb_ctx* c);

/* ---- the pipeline's exchange behind the boundary (RCCL point-to-point over xGMI; librccl.so.1 is loaded on first use) -----------
 * The reference runs its blocks in one loop (llamatransformer.go:156-164); here rank r of `world` holds the stage created with
 * lnb_model_create_parts and one lnb_ctx per sequence in flight.  A host layer in any language only needs a way to hand rank 0's
 * 128-byte id to the other ranks (a file, a socket, MPI, torch.distributed ...).
 * lnb_pipeline_tick ENQUEUES and returns (no stream is synchronised):
 *   run   != NULL: the stage step of that sequence -- run_rows rows at position run_pos; run_tokens: host token ids (rank 0, prompt rows)
 *                  or NULL (other ranks: the input is the received hidden state; rank 0, one-token steps: the token received from the
 *                  last rank, already on the device).  One-token steps replay a captured hipGraph.  On the last rank the argmax token is
 *                  appended to a pinned log; *token_slot_out = its slot (lnb_pipeline_read_tokens).
 *   send  != NULL: ncclSend of that context's result to rank+1 (hidden state [send_rows, dim], plus the [send_rows, ffn_hidden]
 *                  activations when the stage ends between a block's gate/up and down parts); on the LAST rank: the 4-byte token to rank 0.
 *   recv  != NULL: ncclRecv from rank-1 into that context's input buffers; on rank 0: the token from the last rank.
 * send and recv go into ONE ncclGroupStart/End on the pipe's exchange stream, ordered against the compute streams by events; the
 * schedule (which sequence runs / is sent / is received in which tick) belongs to the host layer -- every rank must post matching
 * sends and receives in the same tick order (llama-nuts-and-bolts_amd/pipeline.py: run_ticks_native). */
typedef struct lnb_pipe lnb_pipe;
int lnb_pipeline_unique_id(void* id128);                       /* rank 0: ncclGetUniqueId */
int lnb_pipeline_init(lnb_model* stage, int rank, int world, const void* id128, lnb_pipe** out);   /* world == 1: no communicator, the token ring is a device copy */
/* the same pipe with an IN-PROCESS transport instead of RCCL: every stage lives in this process (pipes that name the same `group`), a send
 * meets its receive in a mailbox and becomes a device-to-device copy.  Same ticks, events and graphs; for single-process hosts and for
 * testing a schedule where RCCL cannot run (it wants one GPU per rank).  All stages of a group must live on ONE device and be ticked from
 * ONE host thread in lock-step order (a stage on another device is refused; running a sequence whose input has been requested but not
 * yet posted by its sender is an error, not a read of stale data) */
int lnb_pipeline_init_loopback(lnb_model* stage, int rank, int world, const char* group, lnb_pipe** out);
/* the same pipe WITHOUT a transport, for a host layer that moves what crosses the stage boundary itself (pipeline.py's torch.distributed
 * fallback when RCCL cannot form its communicator: staging tensors + batch_isend_irecv between lnb_pipeline_sync calls): ticks take `run`
 * only (send / recv are refused), everything else -- captured stage steps, positions advancing on the device, the first stage reading the
 * ring's token words, the last stage's token log -- as above.  The buffers to move: lnb_ctx_hidden_ptr (a sequence), lnb_batch_boundary_ptr
 * (a batch).  lnb_pipeline_comm_count reports 0. */
int lnb_pipeline_init_host(lnb_model* stage, int rank, int world, lnb_pipe** out);
int lnb_pipeline_destroy(lnb_pipe* p);
int lnb_pipeline_tick(lnb_pipe* p, lnb_ctx* run, int run_rows, int run_pos, const int32_t* run_tokens,
                      lnb_ctx* send, int send_rows, lnb_ctx* recv, int recv_rows, int* token_slot_out);
/* The same with a BATCH of sequences as the unit that moves through the stages (lnb_batch over this rank's stage model and its contexts of
 * those sequences; lnb_model_enable_batch works on stages of whole blocks).  Prefill each sequence with single-sequence ticks, call
 * lnb_batch_set_state on every rank (tokens: rank 0 only, or NULL to keep what the prefill's token ring left in the contexts), then per step:
 * run = ONE pass over the stage's weights for all the batch's sequences (captured graph; positions advance on the device; on the last rank
 * the n tokens go to the pinned log, *token_slot_out = the first of n consecutive slots), send = hidden states [n, dim] to rank + 1 (last
 * rank: the n token words to rank 0), recv = the mirror image.  Same grouping, events and transports as lnb_pipeline_tick. */
int lnb_batch_set_state(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos);
/* pipeline ticks only enqueue: a token id outside the vocabulary (its embedding row is not gathered) or an all-NaN logits row (argmax -1,
 * operations_impl.go:529-541) is latched in a device word instead of failing a call.  lnb_batch_check_error waits for the batch's stream,
 * returns an error if the word was set since the last check (or the last lnb_batch_set_state) and clears it.  Call it after
 * lnb_pipeline_sync, before trusting lnb_pipeline_read_tokens.  (lnb_batch_decode checks for itself.) */
int lnb_batch_check_error(lnb_batch* b);
/* device pointers of what a batched tick exchanges (owned by the batch).  which: 0 = the hidden states [n, dim] bf16 -- the stage's input
 * before `run`, its output after; 1 = the n int32 token words the last stage's argmax leaves and the first stage of a multi-stage pipe reads */
void* lnb_batch_boundary_ptr(lnb_batch* b, int which);
int lnb_pipeline_tick_batch(lnb_pipe* p, lnb_batch* run, lnb_batch* send, lnb_batch* recv, int* token_slot_out);
int lnb_pipeline_sync(lnb_pipe* p);
/* the number of ranks the exchange spans as the TRANSPORT reports it (ncclCommCount of the communicator; pipes joined to an in-process
 * group; 1 for a one-stage pipe): lets a host check that N processes really formed ONE N-rank communicator */
int lnb_pipeline_comm_count(lnb_pipe* p, int* out);
/* tokens by log slot (token_slot_out of the tick that produced them); synchronises the device.  The log is a ring of the newest 65536
 * tokens: slots count up for the life of the pipe, reading never has to "drain" anything, a slot older than that is refused */
int lnb_pipeline_read_tokens(lnb_pipe* p, int first_slot, int n, int32_t* out);
/* diagnostic: the same grouped ncclSend + ncclRecv as a tick, on a one-rank communicator (to itself), n_bytes device to device and
 * compared -- checks the RCCL binding on a box with a single GPU, where a multi-rank communicator cannot be formed */
int lnb_pipeline_selftest(int device, int n_bytes);

/* ---- measurement aid (bench.py roofline leg): average HIP-event time of ONE kernel class of the decode step.
 * which: 0 attn_norm+QKV+RoPE GEMV, 1 attention, 2 wo GEMV, 3 ffn_norm+w1|w3 GEMV, 4 w2 GEMV, 5 norm+output GEMV,
 * 6 the five kernels of a whole block.  Consecutive launches cycle through this stage's layers so every launch
 * streams its weights from HBM instead of the 256 MiB Infinity Cache.  The KV cache content at `pos` is overwritten. */
int lnb_profile_kernel(lnb_ctx* c, int which, int pos, int iters, float* avg_ms_out);
/* ... and of the gate|up (w1|w3) and down (w2) kernels of a block launched on TWO streams, w2 `w2_delay_us` behind w1|w3: the time a
 * w1|w3 -> w2 streaming stage (llamatransformer.go:593-624) would have to beat, measured without building it (w2 reads stale activations:
 * only the time means anything; w2_delay_us < 0: w2 first, w1|w3 that long behind it).  w2_lds_pad: extra dynamic LDS for the w2 launch (forces one workgroup of each kernel per CU). */
int lnb_profile_ffn_pair(lnb_ctx* c, int pos, int iters, int w2_delay_us, int w2_lds_pad, float* avg_ms_out);
/* ... and the in-kernel cycle stamps of ONE launch of a GEMV class (which 0, 2, 3, 4 or 5): out[8 waves][16] doubles, per wave = {workgroups
 * that reported, avg total shader cycles, max total, avg barrier wait, avg "x staged / prologue end", avg "norm fold or walk" (wo / w2 chain
 * waves: chain start), avg phase stamps 0..6, stamp 7 = the same launch on the constant-rate wall clock}; *wall_clock_khz = that clock's rate
 * (hipDeviceAttributeWallClockRate).  bench.py turns them into the measured prologue / per-step / boundary terms of roofline.measured_model. */
int lnb_profile_kernel_stamps(lnb_ctx* c, int which, int pos, double* out, int* wall_clock_khz);

/* ---- single-op entry points (the src/ml operators on the hot path), used by the parity tests --------
 * y[rows,n] = trunc(sum_k x[rows,k]*w[n,k])  == ml.LinearTransformation (operations_impl.go:427-447);
 * host buffers in the reference layout; rw in {0(auto),16,32,64} selects the tiling */
int lnb_op_linear(int device, const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int n_out, int k_in, int rw);
/* RMSNorm.Forward (llamatransformer.go:633-639) followed by a linear layer, as the fused kernel computes it */
int lnb_op_rmsnorm_linear(int device, const uint16_t* x, const uint16_t* norm_w, float eps, const uint16_t* w,
                          uint16_t* y, int rows, int n_out, int k_in, int rw);

/* either operator in a given arithmetic mode (norm_w == NULL: plain linear); LNB_MODE_FAST runs the split-K kernels */
int lnb_op_linear_mode(int device, const uint16_t* x, const uint16_t* norm_w, float eps, const uint16_t* w, uint16_t* y,
                       int rows, int n_out, int k_in, int rw, int mode);
/* ml.Argmax (operations_impl.go:513-548) of n bf16 values: strict '<' scan from -MaxFloat32, so the FIRST maximum wins and
 * NaN / -inf are never selected (-1 when nothing qualifies); the kernel the device greedy loop uses (inference.go:207-211) */
int lnb_op_argmax(int device, const uint16_t* logits_bf16, int n, int32_t* out);
/* The softmax numerator as the device evaluates it, over ALL 65536 possible bf16 raw scores s: out[s] = exp(float64(trunc_bf16(float32(s) / divisor)))
 * (llamatransformer.go:464 DivToScalar, operations_impl.go:498 math.Exp) -- the table every attention kernel either looks up or computes inline with the same
 * instructions.  divisor = 1: the device's f64 exp on every bf16 value, which a test compares bit for bit with the host libm the oracle uses. */
int lnb_op_exp_table(int device, float divisor, double* out65536);

/* ---- weight ingestion: replaces torch.TorchModelReader + model.loadModelArgsFromFile (SURVEY.md 8f "next" #2) -------
 * src/torch/torchmodelreader.go:39-145, src/torch/types.go:9-56, src/pickle/pickledispatch.go:13-78,
 * src/common/memorymapper_unix.go:18-41, src/model/loader.go:183-192, src/model/modelargs.go:12-65.
 * A checkpoint handle is a read-only mmap of a PyTorch zip file (Meta's consolidated.00.pth): entries must be STORED,
 * exactly one *.pkl, tensors rebuilt by torch._utils._rebuild_tensor_v2 over torch.BFloat16Storage (Half/Float storages
 * are listed but not loadable into a model).  Tensor data pointers point INTO the mmap and stay valid until
 * lnb_checkpoint_close; lnb_model_set_tensor / lnb_model_load_checkpoint copy from there straight to the device. */
typedef struct lnb_checkpoint lnb_checkpoint;
enum { LNB_DTYPE_BF16 = 0, LNB_DTYPE_F16 = 1, LNB_DTYPE_F32 = 2 };
int lnb_checkpoint_open(const char* path, lnb_checkpoint** out);
void lnb_checkpoint_close(lnb_checkpoint* c);
int lnb_checkpoint_num_tensors(const lnb_checkpoint* c);
/* index of a tensor by its key, -1 if absent */
int lnb_checkpoint_find(const lnb_checkpoint* c, const char* name);
/* any out pointer may be NULL; shape has room for 4 dims; fails for non-contiguous tensors */
int lnb_checkpoint_tensor(const lnb_checkpoint* c, int index, const char** name, int* dtype, int64_t* shape4, int* rank,
                          const void** data, int64_t* nbytes);
/* bind every tensor the model stage owns (getTensor semantics: "tensor \"x\" not found", "... has incorrect shape;
 * expected [a b], got [c d]"); the model still needs lnb_model_finalize afterwards */
int lnb_model_load_checkpoint(lnb_model* m, const lnb_checkpoint* c);
/* params.json -> lnb_model_args with NewModelArgs' defaults for absent keys (n_kv_heads -1, vocab_size -1, multiple_of 256,
 * ffn_dim_multiplier -1, norm_eps 1e-5, rope_theta 500000, use_scaled_rope false, max_seq_len 2048) */
int lnb_model_args_from_json(const char* params_json_path, lnb_model_args* out);

/* ---- prompt tokenisation (SURVEY.md 8f "next" #3): tiktoken reader + split pattern + byte-pair merge + chat template --------
 * src/tiktoken/tiktokenreader.go:12-85, src/model/vocabulary.go:22-50, src/inference/tokenize.go:27-197.  Host code only.
 * tokenizer.model = lines "<base64 token> <rank>"; 256 special tokens follow the mergeable ranks.  Pieces are BYTES. */
typedef struct lnb_tokenizer lnb_tokenizer;
int lnb_tokenizer_load(const char* tokenizer_model_path, lnb_tokenizer** out);
void lnb_tokenizer_free(lnb_tokenizer* t);
int lnb_tokenizer_vocab_size(const lnb_tokenizer* t);
/* ids of <|begin_of_text|>, <|end_of_text|>, <|eot_id|>, <|eom_id|> (the last two are the stop ids); any pointer may be NULL */
int lnb_tokenizer_special(const lnb_tokenizer* t, int32_t* bos, int32_t* eos, int32_t* eot, int32_t* eom);
int lnb_tokenizer_token_id(const lnb_tokenizer* t, const char* bytes, int len);            /* -1 if absent */
int lnb_tokenizer_piece(const lnb_tokenizer* t, int32_t id, const char** bytes, int* len);  /* Vocabulary.IdToToken */
/* InferenceEngine.TokenizeString: returns the number of tokens written to out (capacity cap), < 0 on error */
int lnb_tokenizer_encode(const lnb_tokenizer* t, const char* text, int len, int32_t* out, int cap);
/* InferenceEngine.Tokenize(promptParts): <|begin_of_text|>, then per non-empty part <|start_header_id|> header <|end_header_id|>
 * "\n\n" content <|eot_id|>, then the open assistant header */
int lnb_tokenizer_encode_chat(const lnb_tokenizer* t, const char* const* headers, const char* const* contents, int n_parts,
                              int32_t* out, int cap);
/* Detokeniser = InferenceEngine.TokenToString (src/inference/tokenize.go:197-239) with its generationDecodingContext.waitingBytes
 * (inference.go:35): a piece that is not valid UTF-8 by itself (byte-fallback tokens; pieces ending inside a character) is buffered until the
 * buffered bytes are valid UTF-8, then ONE rune is released; valid pieces pass through.  lnb_tokenizer_decode_stream returns the number of
 * bytes written to out (0 while waiting; *added_to_waiting = the reference's third result), < 0 on error.  One stream per generation.
 * Not restated: processEmoji's aliases and its buffering of combining marks / ZWJ sequences (emoji.go) -- those pieces come out as text. */
typedef struct lnb_detok lnb_detok;
int lnb_tokenizer_stream_create(const lnb_tokenizer* t, lnb_detok** out);
void lnb_tokenizer_stream_free(lnb_detok* d);
int lnb_tokenizer_decode_stream(lnb_detok* d, int32_t token_id, char* out, int cap, int* added_to_waiting);
int lnb_tokenizer_stream_pending(const lnb_detok* d, const char** bytes, int* len);       /* the bytes still waiting */

#ifdef __cplusplus
}
#endif
#endif

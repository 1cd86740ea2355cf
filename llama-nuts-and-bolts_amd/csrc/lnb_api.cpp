// lnb_api.cpp -- C ABI (include/lnb.h) over the gfx950 kernels of lnb_kernels.hip.
// Host-side orchestration of LlamaTransformer.Forward (reference: src/model/llamatransformer.go:145-180):
// device-resident weights (re-tiled once), device-resident KV cache, 5 launches per transformer block,
// and a hipGraph-replayed greedy decode loop whose position/token state lives on the device.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <dirent.h>
#include <unistd.h>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/lnb.h"
#include "lnb_device.h"
#include "lnb_rccl.h"

extern "C" {
hipError_t lnbk_gemv(const GemvParams* p, int rw, int nch, int epi, int norm, hipStream_t st);
hipError_t lnbk_attn(const AttnParams* p, hipStream_t st);
int lnbk_attn_short_max_T(int hd);
size_t lnbk_attn_long_lds(int seq_len);
size_t lnbk_attn_one_lds(int seq_len, int hd);
hipError_t lnbk_exp_table(double* tab, float divisor, hipStream_t st);
hipError_t lnbk_gemm(const GemmParams* p, int epi, hipStream_t st);
hipError_t lnbk_rmsnorm_rows(const uint16_t* x, const uint16_t* w, uint16_t* out, int S, int K, float eps, hipStream_t st);
hipError_t lnbk_embed(const uint16_t* emb, const int32_t* tokens, uint16_t* x, int S, int dim, int vocab, int* err, hipStream_t st);
hipError_t lnbk_argmax(const uint16_t* logits, int V, int32_t* next_token, StepState* state, int32_t* out_tokens, int out_cap, int advance, hipStream_t st);
hipError_t lnbk_set_state(StepState* state, int pos, int n_out, int honour_stop, hipStream_t st);
hipError_t lnbk_set_stop(StepState* state, const int32_t* ids, int n, hipStream_t st);
hipError_t lnbk_advance_state(StepState* state, int rows, hipStream_t st);
hipError_t lnbk_tile(const uint16_t* src, uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH, int gather, hipStream_t st);
hipError_t lnbk_synth_fill(uint16_t* dst, int rows, int K, int row_off, int chain, int RW, int NCH, uint64_t seed, uint32_t tensor_id, int kind, float sigma, hipStream_t st);
hipError_t lnbk_init(void);
hipError_t lnbk_fast_gemv(const GemvParams* p, int rw, int nch, int epi, int norm, hipStream_t st);
hipError_t lnbk_fast_gemm(const GemmParams* p, int epi, hipStream_t st);
hipError_t lnbk_fast_attn(const AttnParams* p, hipStream_t st);
hipError_t lnbk_fast_init(void);
hipError_t lnbk_m16_from_tiled(const uint16_t* src, uint16_t* dst, int rows, int K, int RW, int NCH, hipStream_t st);
hipError_t lnbk_stream(const StreamParams* p, int epi, int acc2, int num_cus, hipStream_t st);
hipError_t lnbk_batch_rmsnorm(const uint16_t* x, const uint16_t* norm_w, float eps, uint16_t* xt, int K, int nseq, hipStream_t st);
hipError_t lnbk_batch_embed(const uint16_t* emb, const BatchTab* tab, uint16_t* x, int nseq, int dim, int vocab, int* err, hipStream_t st);
hipError_t lnbk_batch_argmax(const uint16_t* logits, int V, const BatchTab* tab, int nseq, int32_t* ring, hipStream_t st);
hipError_t lnbk_batch_set_state(const BatchTab* tab, const int32_t* tokens, const int32_t* pos, int32_t* ring, int honour_stop, hipStream_t st);
hipError_t lnbk_batch_scatter_ring(const BatchTab* tab, const int32_t* ring, hipStream_t st);
hipError_t lnbk_batch_advance(const BatchTab* tab, hipStream_t st);
hipError_t lnbk_batch_prepare(void);
hipError_t lnbk_gemm_stream(const GemmParams* p, int epi, int num_cus, hipStream_t st);
hipError_t lnbk_spin(int us, hipStream_t st);
}

// The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run one after the
// other.  This library gives every context its own stream precisely so that several generations overlap on one GPU (one InferenceContext per
// generation, inference.go:174; 2N contexts per pipeline rank): with the default, two contexts can land on ONE queue (2 in flight: 218 instead of
// 278 tokens/s) and eight share four (306 instead of 337 tokens/s; profiles/r05_hw_queues.log).  The runtime reads the variable when it initialises
// -- at the first HIP call of the process -- so a default set while this library is being loaded is in time unless the host has used HIP before;
// a value the host exported wins.
// Round 6: the default is opt-out (LNB_KEEP_HW_QUEUES=1 leaves the environment alone), what happened is recorded, and
// lnb_runtime_info() reports it -- including whether the HIP runtime had ALREADY been initialised when this library was loaded (a host that touched
// HIP first silently keeps 4 queues: 2 contexts in flight then run SLOWER than one) and, on request, the number of streams that really run
// concurrently (measured, not read from the environment).
static int g_hwq_set_by_library = 0;      // 1: this library put "16" into the environment
static int g_hip_live_at_load = 0;        // 1: /dev/kfd was already open when the library was loaded -> the runtime read the variable before us
static bool kfd_is_open() {
    // the ROCm runtime (HSA) opens /dev/kfd when it initialises, never before: an open descriptor on it = HIP has been used in this process
    DIR* d = opendir("/proc/self/fd");
    if (!d) return false;
    bool found = false;
    char path[64], target[256];
    while (struct dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (!strcmp(target, "/dev/kfd")) { found = true; break; }
    }
    closedir(d);
    return found;
}
__attribute__((constructor)) static void lnb_default_hw_queues() {
    g_hip_live_at_load = kfd_is_open() ? 1 : 0;
    const char* keep = getenv("LNB_KEEP_HW_QUEUES");
    if (keep && *keep && atoi(keep) != 0) return;
    if (!getenv("GPU_MAX_HW_QUEUES")) { setenv("GPU_MAX_HW_QUEUES", "16", 0); g_hwq_set_by_library = 1; }
}

static thread_local char g_err[1024] = "";
static int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
extern "C" const char* lnb_last_error(void) { return g_err; }
// shared with lnb_checkpoint.cpp (same library, not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) void lnb_set_error(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static inline float bf_wide_h(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t bf_trunc_h(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

// ---------------------------------------------------------------------------------------------------
struct TensorRef {
    int rows = 0, cols = 0;        // logical shape in the reference layout (1-D tensors: rows = 1)
    bool tiled = false;
    uint16_t* linear = nullptr;    // when !tiled
    TiledDesc* td = nullptr;       // when tiled
    int row_off = 0, chain = 0;
    uint32_t synth_id = 0; int synth_kind = 0;
    int rank = 2;
};
struct LayerW {
    uint16_t* attn_norm = nullptr; uint16_t* ffn_norm = nullptr;
    TiledDesc wqkv{}, wo{}, w13{}, w2{};
    // batched decode (lnb_model_enable_batch): the same matrices in the 16-row matrix-core layout (M16, lnb_device.h)
    uint16_t *m_wqkv = nullptr, *m_wo = nullptr, *m_w13 = nullptr, *m_w2 = nullptr;
};
struct lnb_model {
    lnb_model_args a{};
    int device = 0, layer_begin = 0, layer_end = 0;      // layers this handle touches
    int part_begin = 0, part_end = 0;                     // ... in thirds of a block: 3l = attention part of layer l, 3l+1 = gate/up, 3l+2 = down
    int head_dim = 0, n_rep = 0, ffn_hidden = 0, q_dim = 0, kv_dim = 0;
    uint16_t* tok_embd = nullptr; uint16_t* norm = nullptr; TiledDesc output{};
    std::vector<LayerW> layers;           // indexed by absolute layer id - layer_begin
    std::map<std::string, TensorRef> tensors;
    float* cis = nullptr; int cis_rows = 0; float* silu = nullptr; double* exp_tab = nullptr;
    std::vector<float> cis_host;
    hipStream_t stream = nullptr;
    bool finalized = false;
    int64_t weight_bytes = 0;
    bool batch_enabled = false; uint16_t* m_output = nullptr; int64_t batch_bytes = 0;
    bool first() const { return part_begin == 0; }
    bool last() const { return part_end == 3 * a.n_layers; }
    bool has_part(int l, int q) const { return 3 * l + q >= part_begin && 3 * l + q < part_end; }
    bool has_attn(int l) const { return has_part(l, 0); }
    bool has_w13(int l) const { return has_part(l, 1); }
    bool has_w2(int l) const { return has_part(l, 2); }
    bool whole(int l) const { return has_attn(l) && has_w13(l) && has_w2(l); }
};
struct lnb_ctx {
    lnb_model* m = nullptr; int seq_len = 0;
    hipStream_t stream = nullptr;
    std::vector<uint16_t*> ck, cv;
    StepState* st = nullptr; int32_t* dtok = nullptr; int32_t* dnext = nullptr; int* derr = nullptr;
    int32_t* dout = nullptr; int dout_cap = 0;
    uint16_t *x = nullptr, *h = nullptr, *q = nullptr, *att = nullptr, *ffn = nullptr, *logits = nullptr;
    uint16_t* xn = nullptr;                // [seq_len][dim] normalised rows for the matrix-core prefill path
    int logits_rows = 0;
    hipGraphExec_t graph = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    lnb_layer_cb cb = nullptr; void* cb_user = nullptr;
    int32_t* h_io = nullptr;               // pinned host words of lnb_forward_stage_begin/_end: [0] argmax, [1] token error, [2..] tokens
    bool pending = false, pending_tokens = false, pending_argmax = false;
    int mode = LNB_MODE_EXACT;             // LNB_MODE_FAST: split-K kernels of lnb_fast.hip (tolerance mode, opt-in)
    int sched = LNB_SCHED_LATENCY;         // LNB_SCHED_THROUGHPUT: the co-residency-friendly forms of the one-token kernels (lnb_ctx_set_schedule)
    int n_stop = 0;                        // host copy of StepState.n_stop (lnb_ctx_set_stop_ids): entry points that cannot honour stop ids refuse such a context
    int batch_users = 0;                   // live lnb_batch handles this context is a member of: their device tables and captured graphs hold its raw pointers
    // long-context decode attention (attn_long_*_kernel): used for one-token calls whose context exceeds attn_long_T
    double* e_buf = nullptr; double* z_part = nullptr; int* zseq_count = nullptr;
    int last_prefill_form = 0;         // lnb_ctx_prefill_attention_form
    uint64_t* score_idx = nullptr; size_t score_idx_bytes = 0; int sidx_jt = 0;      // attn_mfma3_kernel's scratch (the exp-table indices of pass 1), grown on demand by check_call; sidx_jt = 0: this call runs attn_mfma_kernel
    unsigned* attn_cnt = nullptr;          // attn_one_kernel: [n_heads] arrival counters of its in-launch exchange + [n_heads] = polls that ran out
    int attn_long_T = 0; int force_zseq = 0;   // force_zseq: bit 0 = walk the serial softmax denominator, bit 1 = keep the two-launch long-context form, bit 2 = one launch with every poll timing out (tests), bit 3 = one launch
    int attn_short_cap = 0;                // longest context the one-workgroup-per-head kernel can stage in the LDS
    bool attn_long = false;                // selection for the launches being enqueued (set per call / per captured graph)
    hipGraphExec_t graph_long = nullptr;   // the decode step captured with the long-context attention
    // pipeline stage (lnb_pipeline_tick): one-token stage step as a captured graph per attention form, events towards / from the exchange stream
    hipGraphExec_t stage_graph[2] = {nullptr, nullptr};
    hipEvent_t ev_done = nullptr, ev_in = nullptr, ev_sent = nullptr;
    bool in_pending = false, sent_pending = false;
    int dev_pos = -1;                      // position the device-side StepState will hold when the stream reaches this point (-1: unknown)
    int call_T = 0;                        // start_pos + rows of the call being enqueued (0: not known to the host, e.g. inside a replayed graph)
    hipEvent_t ev_h2d = nullptr; bool h2d_pending = false;   // the last copy out of the pinned staging words h_io[2..] (enqueue-only paths)
    bool recv_unmatched = false;           // in-process transport: a receive into this context is posted and its sender has not arrived yet
};
// Every path that rewrites the device-side StepState goes through here, so that the pipeline tick's "the graph left pos+1 behind, skip
// the set_state launch" shortcut (dev_pos) can never act on a position some OTHER entry point has since overwritten (lnb_forward,
// lnb_decode_greedy and lnb_profile_kernel may be mixed with ticks on one context).  known = false: the caller is about to advance
// the state on the device by itself (greedy loop) and the host stops tracking it.
static hipError_t ctx_set_state(lnb_ctx* c, int pos, int n_out, bool known, bool honour_stop = false) {
    c->dev_pos = known ? pos : -1;
    return lnbk_set_state(c->st, pos, n_out, honour_stop ? 1 : 0, c->stream);
}
// the pinned staging words h_io[2..] are reused by every enqueue-only call that takes host tokens: wait for the copy that still reads them
static int staging_acquire(lnb_ctx* c) {
    if (c->h2d_pending) { HIPCHK(hipEventSynchronize(c->ev_h2d)); c->h2d_pending = false; }
    return 0;
}
static int staging_release(lnb_ctx* c) {
    if (!c->ev_h2d) HIPCHK(hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_h2d, c->stream)); c->h2d_pending = true;
    return 0;
}

// every captured graph of a context bakes in its buffers, its arithmetic mode and its attention form: whatever changes one of those drops them all
static void drop_graphs(lnb_ctx* c) {
    if (c->graph) { hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    if (c->graph_long) { hipGraphExecDestroy(c->graph_long); c->graph_long = nullptr; }
    for (int i = 0; i < 2; i++) if (c->stage_graph[i]) { hipGraphExecDestroy(c->stage_graph[i]); c->stage_graph[i] = nullptr; }
}
static int env_int(const char* name, int dflt) { const char* s = getenv(name); return s && *s ? atoi(s) : dflt; }
static int auto_rw(int lane_rows, const char* env, int K = 0, bool plain = false) {
    int v = env_int(env, 0);
    if (v == 16 || v == 32 || v == 64 || (v == 4 && plain && K % 128 == 0 && K <= 16384) || ((v == 56 || v == 28) && lane_rows % v == 0) || (v == 24 && K > 0 && K % 256 == 0)) return v;
    // thin matrices without a fused norm / rope epilogue: the row-broadcast kernel (products stay in registers)
    if (plain && lane_rows <= 16 * 256 && K > 0 && K % 128 == 0 && K <= 16384) return 4;
    // thin matrices: one workgroup (16 or 32 rows) per CU, all resident at once on the 256 CUs -- a second round of
    // workgroups would double the serial chain time
    if (lane_rows <= 16 * 256) return 16;
    if (lane_rows <= 32 * 256) return 32;
    return 64;
}
static int g_num_cus = 256;
static thread_local long long* g_dbg = nullptr;   // (thread-local: arming the stamps on one thread must not reach another thread's launches -- ADVICE r5) LNB_GEMV_TIMING=1: per-wave timing dump of the profiled launch
static thread_local int g_dbg_full = 0;           // ... with every barrier and ring wait timed (perturbs the launch); 0: phase stamps and the exit record only
// one resident workgroup per CU: a matrix with more row blocks than CUs is walked persistently
static void set_grid(GemvParams& g, const TiledDesc& t) { g.dbg = g_dbg; g.dbg_full = g_dbg ? g_dbg_full : 0; g.n_blocks = t.n_blocks; g.n_wg = t.n_blocks < g_num_cus ? t.n_blocks : g_num_cus; }
static int alloc_tiled(TiledDesc& t, int n_rows, int K, int rw, int nch, int64_t& bytes) {
    t.n_rows = n_rows; t.k = K; t.rw = rw; t.nch = nch; t.n_blocks = rw == 4 ? (n_rows + 15) / 16 : (n_rows + rw - 1) / rw;   // rw 4: 16-row workgroups
    size_t n = tiled_elems(n_rows, K, rw, nch) * 2;
    HIPCHK(hipMalloc((void**)&t.w, n));
    HIPCHK(hipMemset(t.w, 0, n));
    bytes += (int64_t)n;
    return 0;
}
static int alloc_linear(uint16_t** p, size_t elems, int64_t& bytes) {
    HIPCHK(hipMalloc((void**)p, elems * 2)); HIPCHK(hipMemset(*p, 0, elems * 2)); bytes += (int64_t)elems * 2; return 0;
}

extern "C" int lnb_device_count(int* out) { int n = 0; HIPCHK(hipGetDeviceCount(&n)); *out = n; return 0; }
extern "C" int lnb_abi_version(void) { return LNB_ABI_VERSION; }
// What the host should know before it trusts a multi-context run: the hardware queues the HIP runtime was told to use, who told it,
// whether that was in time, and -- probe_queues != 0 -- how many streams REALLY run concurrently: 32 one-wave kernels that each hold their stream for
// 2 ms of wall clock, one per stream; with Q queues they finish in ceil(32 / Q) rounds (4-20 ms including the stream set-up).
extern "C" int lnb_runtime_info(int device, int probe_queues, lnb_runtime_info_t* out) {
    if (!out) return fail("null argument");
    memset(out, 0, sizeof *out);
    out->abi_version = LNB_ABI_VERSION;
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail("no HIP device: liblnb_hip.so has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device));
    out->device = device; out->n_cus = prop.multiProcessorCount;
    snprintf(out->device_name, sizeof out->device_name, "%s", prop.name);
    snprintf(out->arch, sizeof out->arch, "%s", prop.gcnArchName);
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, device) == hipSuccess) out->shader_clock_khz = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, device) == hipSuccess) out->memory_clock_khz = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, device) == hipSuccess) out->wall_clock_khz = v;
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    out->hw_queues_env = q && *q ? atoi(q) : 0;
    out->hw_queues_set_by_library = g_hwq_set_by_library;
    out->hip_initialised_before_load = g_hip_live_at_load;
    // what the runtime will have honoured: the environment's value unless HIP was live before this library could set it (then the runtime's default, 4,
    // unless the HOST had exported a value -- which it read at its own initialisation)
    out->hw_queues_expected = (g_hip_live_at_load && g_hwq_set_by_library) ? 4 : (out->hw_queues_env > 0 ? out->hw_queues_env : 4);
    if (probe_queues) {
        constexpr int NSTREAM = 32, SPIN_US = 2000;      // (2 ms per spin: the ~0.3 ms it takes the host to launch on 32 streams must not count as a round)
        hipStream_t sts[NSTREAM] = {}; hipEvent_t e0 = nullptr, e1 = nullptr, ej[NSTREAM] = {};
        hipError_t e = hipSuccess;
        for (int i = 0; i < NSTREAM && e == hipSuccess; i++) { e = hipStreamCreateWithFlags(&sts[i], hipStreamNonBlocking); if (e == hipSuccess) e = hipEventCreateWithFlags(&ej[i], hipEventDisableTiming); }
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2 && e == hipSuccess; rep++) {      // (the first round warms the streams up)
            e = hipEventRecord(e0, sts[0]);
            for (int i = 1; i < NSTREAM && e == hipSuccess; i++) e = hipStreamWaitEvent(sts[i], e0, 0);
            for (int i = 0; i < NSTREAM && e == hipSuccess; i++) e = lnbk_spin(SPIN_US, sts[i]);
            for (int i = 1; i < NSTREAM && e == hipSuccess; i++) { e = hipEventRecord(ej[i], sts[i]); if (e == hipSuccess) e = hipStreamWaitEvent(sts[0], ej[i], 0); }
            if (e == hipSuccess) e = hipEventRecord(e1, sts[0]);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        }
        for (int i = 0; i < NSTREAM; i++) { if (sts[i]) hipStreamDestroy(sts[i]); if (ej[i]) hipEventDestroy(ej[i]); }
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        HIPCHK(e);
        const double rounds = (double)ms * 1000.0 / SPIN_US;     // 32 / Q, plus launch overhead
        int qm = rounds > 0.5 ? (int)((double)NSTREAM / rounds + 0.5) : NSTREAM;
        out->hw_queues_measured = qm < 1 ? 1 : (qm > NSTREAM ? NSTREAM : qm);
        out->probe_ms = ms;
    }
    return 0;
}
// what a multi-GPU host prints before it trusts a run (bench.py --gpus N: the per-rank preflight record): the device behind an index, and
// whether `device` can map `peer`'s memory (hipDeviceCanAccessPeer: xGMI / PCIe peer-to-peer, what RCCL's point-to-point path rides on)
extern "C" int lnb_device_info(int device, char* name, int name_cap, int64_t* hbm_bytes, int* n_cus, char* arch, int arch_cap) {
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d devices)", device, ndev);
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device));
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", prop.name);
    if (arch && arch_cap > 0) snprintf(arch, (size_t)arch_cap, "%s", prop.gcnArchName);
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (n_cus) *n_cus = prop.multiProcessorCount;
    return 0;
}
extern "C" int lnb_device_pci_bus_id(int device, char* out, int cap) {
    if (!out || cap < 16) return fail("null argument or a buffer under 16 bytes");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipDeviceGetPCIBusId(out, cap, device));
    return 0;
}
extern "C" int lnb_device_can_access_peer(int device, int peer, int* out) {
    if (!out) return fail("null argument");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || peer < 0 || peer >= ndev) return fail("device pair (%d, %d) out of range (%d devices)", device, peer, ndev);
    if (device == peer) { *out = 1; return 0; }
    int can = 0; HIPCHK(hipDeviceCanAccessPeer(&can, device, peer));
    *out = can;
    return 0;
}

extern "C" int lnb_model_ffn_hidden_dim(const lnb_model_args* a) {   // llamatransformer.go:569-577
    int h = 4 * a->dim;
    h = (int)(2 * h / 3);
    if (a->ffn_dim_multiplier > -1) h = (int)(a->ffn_dim_multiplier * (double)h);
    h = a->multiple_of * ((h + a->multiple_of - 1) / a->multiple_of);
    return h;
}

static void reg_linear(lnb_model* m, const std::string& name, uint16_t* p, int cols, uint32_t sid, int kind, int rows = 1, int rank = 1) {
    TensorRef r; r.rows = rows; r.cols = cols; r.tiled = false; r.linear = p; r.synth_id = sid; r.synth_kind = kind; r.rank = rank;
    m->tensors[name] = r;
}
static void reg_tiled(lnb_model* m, const std::string& name, TiledDesc* td, int rows, int cols, int row_off, int chain, uint32_t sid) {
    TensorRef r; r.rows = rows; r.cols = cols; r.tiled = true; r.td = td; r.row_off = row_off; r.chain = chain; r.synth_id = sid; r.rank = 2;
    m->tensors[name] = r;
}

extern "C" int lnb_model_create(const lnb_model_args* args, int device, int layer_begin, int layer_end, lnb_model** out) {
    return lnb_model_create_parts(args, device, 3 * layer_begin, 3 * layer_end, out);
}
// Pipeline stages may be cut INSIDE a block, in thirds: 3l = attention part (attention_norm, wq|wk|wv, attention, wo + residual,
// llamatransformer.go:222-232), 3l+1 = gate/up part (ffn_norm, w1|w3, SiLU*up, :237, :601-617), 3l+2 = down part (w2 + residual,
// :619, :248).  After the attention part the live state is again one [S, dim] vector (h); after the gate/up part it is h plus the
// [S, ffn_hidden] activations.  The three parts cost about the same HBM time (48 / 47 / 45 us for the 8B shape), so the stages of a
// pipeline can be balanced to a third of a block (pipeline.stage_parts).
static int model_alloc(lnb_model* m);
extern "C" int lnb_model_create_parts(const lnb_model_args* args, int device, int part_begin, int part_end, lnb_model** out) {
    if (!args || !out) return fail("null argument");
    *out = nullptr;
    const int layer_begin = part_begin / 3, layer_end = (part_end + 2) / 3;
    lnb_model_args a = *args;
    if (a.n_kv_heads < 0) a.n_kv_heads = a.n_heads;                       // llamatransformer.go:73-75
    if (a.rope_theta <= 0) a.rope_theta = 500000.0;                       // :80-82
    if (a.n_layers <= 0) return fail("n_layers must be positive (got %d)", a.n_layers);
    if (a.vocab_size <= 0) return fail("vocab_size must be positive (got %d; params.json without the key gives -1: take it from tok_embeddings)", a.vocab_size);
    if (a.multiple_of <= 0) return fail("multiple_of must be positive (got %d)", a.multiple_of);
    if (a.dim <= 0 || a.n_heads <= 0 || a.n_kv_heads <= 0 || a.dim % a.n_heads || a.n_heads % a.n_kv_heads) return fail("invalid head configuration");
    if (part_begin < 0 || part_end > 3 * a.n_layers || part_begin >= part_end) return fail("invalid layer range [%d,%d)", part_begin / 3, (part_end + 2) / 3);
    int hd = a.dim / a.n_heads;
    if (a.dim % 8 || hd % 8) return fail("dim and head_dim must be multiples of 8");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail("no HIP device: liblnb_hip.so has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    HIPCHK(lnbk_init());
    HIPCHK(lnbk_fast_init());
    HIPCHK(lnbk_batch_prepare());                                         // dynamic-LDS limits of the prefill / batch kernels, once, outside any capture (a prompt on a model without enable_batch reaches gemm_stream_kernel too)
    { hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device)); if (prop.multiProcessorCount > 0) g_num_cus = prop.multiProcessorCount; }
    lnb_model* m = new lnb_model();
    m->a = a; m->device = device; m->layer_begin = layer_begin; m->layer_end = layer_end; m->part_begin = part_begin; m->part_end = part_end;
    m->head_dim = hd; m->n_rep = a.n_heads / a.n_kv_heads; m->ffn_hidden = lnb_model_ffn_hidden_dim(&a);
    m->q_dim = a.n_heads * hd; m->kv_dim = a.n_kv_heads * hd;
    if (m->ffn_hidden <= 0 || m->ffn_hidden % 8) { delete m; return fail("ffn hidden dim must be a positive multiple of 8"); }
    if (model_alloc(m)) { lnb_model_destroy(m); return -1; }             // (the error text survives: destroy does not touch it)
    *out = m;
    return 0;
}
static int model_alloc(lnb_model* m) {
    const lnb_model_args& a = m->a;
    const int layer_begin = m->layer_begin, layer_end = m->layer_end;
    HIPCHK(hipStreamCreate(&m->stream));
    int64_t& wb = m->weight_bytes;
    const int dim = a.dim, F = m->ffn_hidden;
    if (m->first()) {
        if (alloc_linear(&m->tok_embd, (size_t)a.vocab_size * dim, wb)) return -1;
        reg_linear(m, "tok_embeddings.weight", m->tok_embd, dim, 0, 0, a.vocab_size, 2);
    }
    m->layers.resize(layer_end - layer_begin);
    for (int l = layer_begin; l < layer_end; l++) {
        LayerW& L = m->layers[l - layer_begin];
        char nm[128]; uint32_t base = 16u * (uint32_t)(l + 1);
        if (m->has_attn(l)) {
            if (alloc_linear(&L.attn_norm, dim, wb)) return -1;
            // wq|wk|wv: 24-row blocks (gemv_quad_kernel: quad-DPP chain waves, one block on every CU) when the rows divide that way -- the 8B shape
            int rwq = auto_rw(m->q_dim + 2 * m->kv_dim, "LNB_RW_QKV", dim);
            if (env_int("LNB_RW_QKV", 0) == 0 && m->q_dim + 2 * m->kv_dim == 24 * g_num_cus && dim % 256 == 0) rwq = 24;
            if (alloc_tiled(L.wqkv, m->q_dim + 2 * m->kv_dim, dim, rwq, 1, wb)) return -1;
            if (alloc_tiled(L.wo, dim, m->q_dim, auto_rw(dim, "LNB_RW_WO", m->q_dim, true), 1, wb)) return -1;
            snprintf(nm, sizeof nm, "layers.%d.attention_norm.weight", l); reg_linear(m, nm, L.attn_norm, dim, base + 0, 1);
            snprintf(nm, sizeof nm, "layers.%d.attention.wq.weight", l); reg_tiled(m, nm, &L.wqkv, m->q_dim, dim, 0, 0, base + 1);
            snprintf(nm, sizeof nm, "layers.%d.attention.wk.weight", l); reg_tiled(m, nm, &L.wqkv, m->kv_dim, dim, m->q_dim, 0, base + 2);
            snprintf(nm, sizeof nm, "layers.%d.attention.wv.weight", l); reg_tiled(m, nm, &L.wqkv, m->kv_dim, dim, m->q_dim + m->kv_dim, 0, base + 3);
            snprintf(nm, sizeof nm, "layers.%d.attention.wo.weight", l); reg_tiled(m, nm, &L.wo, dim, m->q_dim, 0, 0, base + 4);
        }
        if (m->has_w13(l)) {
            if (alloc_linear(&L.ffn_norm, dim, wb)) return -1;
            {   // two-chain gate|up matrix: when the rows split into exactly one 56-row block per CU, take it (all CUs stream)
                int rw13 = auto_rw(F, "LNB_RW_W13");
                if (env_int("LNB_RW_W13", 0) == 0 && F == 56 * g_num_cus) rw13 = 56;
                if (alloc_tiled(L.w13, F, dim, rw13, 2, wb)) return -1;
            }
            snprintf(nm, sizeof nm, "layers.%d.ffn_norm.weight", l); reg_linear(m, nm, L.ffn_norm, dim, base + 5, 1);
            snprintf(nm, sizeof nm, "layers.%d.feed_forward.w1.weight", l); reg_tiled(m, nm, &L.w13, F, dim, 0, 0, base + 6);
            snprintf(nm, sizeof nm, "layers.%d.feed_forward.w3.weight", l); reg_tiled(m, nm, &L.w13, F, dim, 0, 1, base + 8);
        }
        if (m->has_w2(l)) {
            if (alloc_tiled(L.w2, dim, F, auto_rw(dim, "LNB_RW_W2", F, true), 1, wb)) return -1;
            snprintf(nm, sizeof nm, "layers.%d.feed_forward.w2.weight", l); reg_tiled(m, nm, &L.w2, dim, F, 0, 0, base + 7);
        }
    }
    if (m->last()) {
        if (alloc_linear(&m->norm, dim, wb)) return -1;
        reg_linear(m, "norm.weight", m->norm, dim, 1, 1);
        if (alloc_tiled(m->output, a.vocab_size, dim, auto_rw(a.vocab_size, "LNB_RW_OUT"), 1, wb)) return -1;
        reg_tiled(m, "output.weight", &m->output, a.vocab_size, dim, 0, 0, 2);
    }
    return 0;
}

extern "C" int lnb_model_destroy(lnb_model* m) {
    if (!m) return 0;
    hipSetDevice(m->device);
    if (m->tok_embd) hipFree(m->tok_embd);
    if (m->norm) hipFree(m->norm);
    if (m->output.w) hipFree(m->output.w);
    for (auto& L : m->layers) {
        hipFree(L.attn_norm); hipFree(L.ffn_norm); hipFree(L.wqkv.w); hipFree(L.wo.w); hipFree(L.w13.w); hipFree(L.w2.w);
        hipFree(L.m_wqkv); hipFree(L.m_wo); hipFree(L.m_w13); hipFree(L.m_w2);
    }
    hipFree(m->m_output);
    if (m->cis) hipFree(m->cis);
    if (m->silu) hipFree(m->silu);
    if (m->exp_tab) hipFree(m->exp_tab);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    return 0;
}

extern "C" int64_t lnb_model_weight_bytes(lnb_model* m) { return m ? m->weight_bytes : 0; }

extern "C" int lnb_model_set_tensor(lnb_model* m, const char* name, const uint16_t* host, const int64_t* shape, int rank) {
    if (!m || !name || !host || !shape) return fail("null argument");
    HIPCHK(hipSetDevice(m->device));
    auto it = m->tensors.find(name);
    if (it == m->tensors.end()) return fail("tensor \"%s\" does not belong to this model stage", name);
    if (m->batch_enabled) return fail("tensor \"%s\": the model's batched-decode copy has been built; bind tensors before lnb_model_enable_batch", name);
    TensorRef& r = it->second;
    // shape check as loader.go:183-192
    if (rank != r.rank || (rank == 1 && shape[0] != r.cols) || (rank == 2 && (shape[0] != r.rows || shape[1] != r.cols)))
        return fail("tensor \"%s\": shape mismatch, expected [%d%s%d]", name, r.rank == 1 ? r.cols : r.rows, r.rank == 1 ? "" : " ", r.rank == 1 ? 0 : r.cols);
    size_t n = (size_t)r.rows * r.cols;
    if (!r.tiled) { HIPCHK(hipMemcpy(r.linear, host, n * 2, hipMemcpyHostToDevice)); return 0; }
    uint16_t* stage = nullptr;
    HIPCHK(hipMalloc((void**)&stage, n * 2));
    HIPCHK(hipMemcpy(stage, host, n * 2, hipMemcpyHostToDevice));
    hipError_t e = lnbk_tile(stage, r.td->w, r.rows, r.cols, r.row_off, r.chain, r.td->rw, r.td->nch, 0, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    hipFree(stage);
    HIPCHK(e);
    return 0;
}

// enumerate the tensors this model stage binds (names as in the checkpoint, shapes as getTensor expects them)
extern "C" int lnb_model_num_tensors(const lnb_model* m) { return m ? (int)m->tensors.size() : 0; }
extern "C" int lnb_model_tensor_info(const lnb_model* m, int k, const char** name, int64_t* shape, int* rank) {
    if (!m || k < 0 || k >= (int)m->tensors.size()) return fail("tensor index %d out of range", k);
    auto it = m->tensors.begin();
    std::advance(it, k);
    if (name) *name = it->first.c_str();
    if (rank) *rank = it->second.rank;
    if (shape) { if (it->second.rank == 1) shape[0] = it->second.cols; else { shape[0] = it->second.rows; shape[1] = it->second.cols; } }
    return 0;
}

extern "C" int lnb_model_get_tensor(lnb_model* m, const char* name, uint16_t* host, int64_t nelem) {
    if (!m || !name || !host) return fail("null argument");
    HIPCHK(hipSetDevice(m->device));
    auto it = m->tensors.find(name);
    if (it == m->tensors.end()) return fail("tensor \"%s\" does not belong to this model stage", name);
    TensorRef& r = it->second;
    size_t n = (size_t)r.rows * r.cols;
    if ((int64_t)n != nelem) return fail("tensor \"%s\": expected %zu elements", name, n);
    if (!r.tiled) { HIPCHK(hipMemcpy(host, r.linear, n * 2, hipMemcpyDeviceToHost)); return 0; }
    uint16_t* stage = nullptr;
    HIPCHK(hipMalloc((void**)&stage, n * 2));
    hipError_t e = lnbk_tile(r.td->w, stage, r.rows, r.cols, r.row_off, r.chain, r.td->rw, r.td->nch, 1, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e == hipSuccess) e = hipMemcpy(host, stage, n * 2, hipMemcpyDeviceToHost);
    hipFree(stage);
    HIPCHK(e);
    return 0;
}

extern "C" int lnb_model_fill_synthetic(lnb_model* m, uint64_t seed) {
    if (!m) return fail("null argument");
    if (m->batch_enabled) return fail("the model's batched-decode copy has been built; fill the weights before lnb_model_enable_batch");
    HIPCHK(hipSetDevice(m->device));
    for (auto& kv : m->tensors) {
        TensorRef& r = kv.second;
        if (r.tiled) HIPCHK(lnbk_synth_fill(r.td->w, r.rows, r.cols, r.row_off, r.chain, r.td->rw, r.td->nch, seed, r.synth_id, 0, 0.02f, m->stream));
        else HIPCHK(lnbk_synth_fill(r.linear, r.rows, r.cols, 0, 0, 0, 1, seed, r.synth_id, r.synth_kind, 0.02f, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    return 0;
}

// precomputeFreqsCis + applyScaling (llamatransformer.go:662-751): host-side, f32 arithmetic with bf16
// truncation of freqs, positions and angles; cos/sin in f64.  (Independent restatement: the oracle has its own.)
static void build_rope_table(int head_dim, int rows, double theta, bool scaled, std::vector<float>& cis) {
    const int n = head_dim / 2;
    std::vector<uint16_t> freqs(n);
    const float dimf = (float)head_dim;
    for (int i = 0; i < n; i++) {
        float val = bf_wide_h(bf_trunc_h((float)(2 * i)));
        freqs[i] = bf_trunc_h((float)(1.0 / std::pow(theta, (double)(val / dimf))));
    }
    if (scaled) {
        const float scale_factor = 8.0f, low_freq_factor = 1.0f, high_freq_factor = 4.0f, old_context_len = 8192.0f;
        const float low_freq_wavelen = old_context_len / low_freq_factor, high_freq_wavelen = old_context_len / high_freq_factor;
        for (int i = 0; i < n; i++) {
            volatile float freq = bf_wide_h(freqs[i]);
            volatile float wavelen = (float)(2 * M_PI) / freq;
            float nf;
            if (wavelen < high_freq_wavelen) nf = freq;
            else if (wavelen > low_freq_wavelen) nf = freq / scale_factor;
            else {
                volatile float smooth = (old_context_len / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor);
                volatile float t1 = (1 - smooth) * freq; t1 = t1 / scale_factor;
                volatile float t2 = smooth * freq;
                nf = t1 + t2;
            }
            freqs[i] = bf_trunc_h(nf);
        }
    }
    cis.resize((size_t)rows * n * 2);
    for (int p = 0; p < rows; p++) {
        const float t = bf_wide_h(bf_trunc_h((float)p));
        for (int i = 0; i < n; i++) {
            volatile float prod = t * bf_wide_h(freqs[i]);
            const double ang = (double)bf_wide_h(bf_trunc_h(prod));
            cis[((size_t)p * n + i) * 2 + 0] = (float)std::cos(ang);
            cis[((size_t)p * n + i) * 2 + 1] = (float)std::sin(ang);
        }
    }
}

extern "C" int lnb_model_finalize(lnb_model* m, int rope_rows) {
    if (!m) return fail("null argument");
    HIPCHK(hipSetDevice(m->device));
    m->cis_rows = rope_rows > 0 ? rope_rows : m->a.max_seq_len * 2;           // llamatransformer.go:109
    build_rope_table(m->head_dim, m->cis_rows, m->a.rope_theta, m->a.use_scaled_rope != 0, m->cis_host);
    if (m->cis) hipFree(m->cis);
    HIPCHK(hipMalloc((void**)&m->cis, m->cis_host.size() * 4));
    HIPCHK(hipMemcpy(m->cis, m->cis_host.data(), m->cis_host.size() * 4, hipMemcpyHostToDevice));
    // TABLE_SILU (activations.go:15-25): f32(x / (1 + exp(-x))) in f64 for every bf16 bit pattern
    std::vector<float> silu(1 << 16);
    for (int i = 0; i < (1 << 16); i++) { double x = (double)bf_wide_h((uint16_t)i); silu[i] = (float)(x / (1.0 + std::exp(-x))); }
    if (!m->silu) HIPCHK(hipMalloc((void**)&m->silu, silu.size() * 4));
    HIPCHK(hipMemcpy(m->silu, silu.data(), silu.size() * 4, hipMemcpyHostToDevice));
    if (!m->exp_tab) HIPCHK(hipMalloc((void**)&m->exp_tab, (size_t)(1 << 16) * 8));
    HIPCHK(lnbk_exp_table(m->exp_tab, bf_wide_h(bf_trunc_h((float)std::sqrt((double)m->head_dim))), m->stream));   // (attn_mfma_kernel)
    HIPCHK(hipStreamSynchronize(m->stream));
    m->finalized = true;
    return 0;
}

extern "C" int lnb_model_rope_table(lnb_model* m, float* out, int64_t nfloats, int* rows_out) {
    if (!m || !m->finalized) return fail("model not finalized");
    if (rows_out) *rows_out = m->cis_rows;
    if (out) {
        if ((size_t)nfloats != m->cis_host.size()) return fail("rope table has %zu floats", m->cis_host.size());
        memcpy(out, m->cis_host.data(), m->cis_host.size() * 4);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
static int ctx_alloc(lnb_ctx* c);
extern "C" int lnb_ctx_create(lnb_model* m, int seq_len, lnb_ctx** out) {
    if (!m || !out) return fail("null argument");
    *out = nullptr;
    if (!m->finalized) return fail("model not finalized");
    HIPCHK(hipSetDevice(m->device));
    lnb_ctx* c = new lnb_ctx();
    c->m = m; c->seq_len = seq_len > 0 ? seq_len : m->a.max_seq_len;        // inferencecontext.go:22-26
    // Context length: one-token calls above the attention crossover and calls of 16 or more rows run kernels without a per-position LDS
    // array that would not fit (the long-context PV kernel keeps 4 bytes per position: ~23 K positions); the one-workgroup-per-head kernel
    // (12 bytes per position: ~7.8 K at head_dim 128) then only serves calls of 2..15 rows, which fail beyond its reach (lnb.h)
    if (m->head_dim != 128 && m->head_dim != 64 && m->head_dim != 32) { delete c; return fail("head_dim %d not one of 32/64/128", m->head_dim); }
    if (lnbk_attn_long_lds(c->seq_len) > 160 * 1024)
        { const int sl = c->seq_len; delete c; return fail("seq_len %d too long for the LDS staging of the attention kernels (about 23000 positions)", sl); }
    if (ctx_alloc(c)) { lnb_ctx_destroy(c); return -1; }
    *out = c;
    return 0;
}
static int ctx_alloc(lnb_ctx* c) {
    lnb_model* m = c->m;
    // a NON-blocking stream: it never synchronises with the legacy (null) stream.  Contexts are driven from several host threads at
    // once (one per generation); with a blocking stream, one thread's graph capture made any other thread's null-stream call
    // (hipMemcpy, hipMemset) fail with "would make the legacy stream depend on a capturing blocking stream".  Everything a context
    // does is therefore ordered on ITS stream: asynchronous copies and memsets + a stream synchronise, never the null stream.
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->ev0)); HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipHostMalloc((void**)&c->h_io, ((size_t)c->seq_len + 2) * 4, hipHostMallocDefault));
    const size_t kvn = (size_t)c->seq_len * m->kv_dim;
    for (size_t l = 0; l < m->layers.size(); l++) {
        uint16_t *k = nullptr, *v = nullptr;
        if (m->has_attn(m->layer_begin + (int)l)) {                         // a stage that holds only FFN parts of a block keeps no cache for it
            HIPCHK(hipMalloc((void**)&k, kvn * 2)); c->ck.push_back(k); c->cv.push_back(nullptr);
            HIPCHK(hipMalloc((void**)&v, kvn * 2)); c->cv.back() = v;
            HIPCHK(hipMemsetAsync(k, 0, kvn * 2, c->stream)); HIPCHK(hipMemsetAsync(v, 0, kvn * 2, c->stream));     // ml.Zeros, inferencecontext.go:32-42
        } else { c->ck.push_back(nullptr); c->cv.push_back(nullptr); }
    }
    HIPCHK(hipMalloc((void**)&c->st, sizeof(StepState))); HIPCHK(hipMemsetAsync(c->st, 0, sizeof(StepState), c->stream));
    HIPCHK(hipMalloc((void**)&c->dtok, (size_t)c->seq_len * 4));
    HIPCHK(hipMalloc((void**)&c->dnext, 16)); HIPCHK(hipMalloc((void**)&c->derr, 16)); HIPCHK(hipMemsetAsync(c->derr, 0, 16, c->stream));
    c->dout_cap = c->seq_len; HIPCHK(hipMalloc((void**)&c->dout, (size_t)c->dout_cap * 4));
    const size_t S = c->seq_len;
    HIPCHK(hipMalloc((void**)&c->x, S * m->a.dim * 2)); HIPCHK(hipMalloc((void**)&c->h, S * m->a.dim * 2));
    HIPCHK(hipMalloc((void**)&c->xn, S * m->a.dim * 2));
    HIPCHK(hipMalloc((void**)&c->q, S * m->q_dim * 2)); HIPCHK(hipMalloc((void**)&c->att, S * m->q_dim * 2));
    HIPCHK(hipMalloc((void**)&c->ffn, S * m->ffn_hidden * 2));
    if (m->last()) { HIPCHK(hipMalloc((void**)&c->logits, (size_t)m->a.vocab_size * 2)); c->logits_rows = 1; }
    HIPCHK(hipMalloc((void**)&c->e_buf, (size_t)m->a.n_heads * S * 8));
    HIPCHK(hipMalloc((void**)&c->z_part, (size_t)m->a.n_heads * ((S + 63) / 64 + 8) * 8));     // (two-launch form: one partial per 256 positions; attn_one_kernel: one per 64)
    HIPCHK(hipMalloc((void**)&c->zseq_count, 16)); HIPCHK(hipMemsetAsync(c->zseq_count, 0, 16, c->stream));
    HIPCHK(hipMalloc((void**)&c->attn_cnt, ((size_t)m->a.n_heads + 4) * 4)); HIPCHK(hipMemsetAsync(c->attn_cnt, 0, ((size_t)m->a.n_heads + 4) * 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // crossover measured on MI355X (tools/att_timing.py): the one-workgroup-per-head kernel wins below a few hundred positions
    c->attn_long_T = env_int("LNB_ATTN_LONG_T", 512);
    c->attn_short_cap = lnbk_attn_short_max_T(m->head_dim);
    return 0;
}

extern "C" int lnb_ctx_destroy(lnb_ctx* c) {
    if (!c) return 0;
    // a batch bakes its members' device pointers (state words, token words, caches) into its tables and captured graphs: freeing a member
    // under it would make the next lnb_batch_decode / lnb_pipeline_tick_batch read and write freed memory.  Destroy the batch first.
    if (c->batch_users > 0) return fail("context is a member of %d live batch(es): lnb_batch_destroy them first", c->batch_users);
    hipSetDevice(c->m->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    drop_graphs(c);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_sent) hipEventDestroy(c->ev_sent);
    if (c->ev_h2d) hipEventDestroy(c->ev_h2d);
    hipFree(c->e_buf); hipFree(c->z_part); hipFree(c->zseq_count); hipFree(c->attn_cnt); if (c->score_idx) hipFree(c->score_idx);
    for (auto p : c->ck) if (p) hipFree(p);
    for (auto p : c->cv) if (p) hipFree(p);
    hipFree(c->st); hipFree(c->dtok); hipFree(c->dnext); hipFree(c->derr); hipFree(c->dout);
    if (c->h_io) hipHostFree(c->h_io);
    hipFree(c->x); hipFree(c->h); hipFree(c->xn); hipFree(c->q); hipFree(c->att); hipFree(c->ffn); if (c->logits) hipFree(c->logits);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" int lnb_ctx_reset(lnb_ctx* c) {
    if (!c) return fail("null argument");
    HIPCHK(hipSetDevice(c->m->device));
    const size_t kvn = (size_t)c->seq_len * c->m->kv_dim;
    for (size_t l = 0; l < c->ck.size(); l++) if (c->ck[l]) { HIPCHK(hipMemsetAsync(c->ck[l], 0, kvn * 2, c->stream)); HIPCHK(hipMemsetAsync(c->cv[l], 0, kvn * 2, c->stream)); }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int lnb_ctx_read_kv(lnb_ctx* c, int layer, int which, uint16_t* host) {
    if (!c || !host) return fail("null argument");
    HIPCHK(hipSetDevice(c->m->device));
    if (layer < c->m->layer_begin || layer >= c->m->layer_end || !c->m->has_attn(layer)) return fail("layer %d is not owned by this stage", layer);
    const size_t kvn = (size_t)c->seq_len * c->m->kv_dim;
    if (which) { HIPCHK(hipMemcpyAsync(host, c->cv[layer - c->m->layer_begin], kvn * 2, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return 0; }
    // K lives as [kv head][d/8][position][8] on the device; hand it back in the reference's [position][kv head][d] order
    std::vector<uint16_t> raw(kvn);
    HIPCHK(hipMemcpyAsync(raw.data(), c->ck[layer - c->m->layer_begin], kvn * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int hd = c->m->head_dim, nk = hd >> 3, KVH = c->m->kv_dim / hd;
    for (int j = 0; j < c->seq_len; j++)
        for (int kh = 0; kh < KVH; kh++)
            for (int d = 0; d < hd; d++)
                host[(size_t)j * c->m->kv_dim + kh * hd + d] = raw[(((size_t)kh * nk + (d >> 3)) * c->seq_len + j) * 8 + (d & 7)];
    return 0;
}

// exact-order (default) or tolerance mode for everything this context runs afterwards; the captured decode graph is per mode
extern "C" int lnb_ctx_set_mode(lnb_ctx* c, int mode) {
    if (!c) return fail("null argument");
    if (mode != LNB_MODE_EXACT && mode != LNB_MODE_FAST) return fail("unknown mode %d (LNB_MODE_EXACT = 0, LNB_MODE_FAST = 1)", mode);
    if (mode == c->mode) return 0;
    HIPCHK(hipSetDevice(c->m->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    drop_graphs(c);
    c->mode = mode;
    return 0;
}
extern "C" int lnb_ctx_get_mode(const lnb_ctx* c) { return c ? c->mode : -1; }
// Which FORMS of the exact one-token kernels this context's steps launch (same arithmetic, same bits; tests/test_gpu_round5.py).  The latency
// forms (default) are built for ONE stream owning the chip: every launch takes all 256 CUs with eight or nine waves and 91 - 124 KB of LDS per
// workgroup, so nothing of another stream fits beside it.  A server or a pipeline rank keeps several generations in flight, one context and
// stream each (inference.go:174): there a chain-bound launch of one context should hide under the HBM-bound gate|up launch of another, which
// needs both workgroups on one CU -- the throughput forms keep every workgroup at or below 57 KB (wq|wk|wv: 128-step stages; wo, w2: the
// self-feeding row-broadcast kernel).  Round 4 had lost this (8 prompts in flight 308 -> 272 tokens/s) when it made the latency forms the only ones.
extern "C" int lnb_ctx_set_schedule(lnb_ctx* c, int sched) {
    if (!c) return fail("null argument");
    if (sched != LNB_SCHED_LATENCY && sched != LNB_SCHED_THROUGHPUT) return fail("unknown schedule %d (LNB_SCHED_LATENCY = 0, LNB_SCHED_THROUGHPUT = 1)", sched);
    if (sched == c->sched) return 0;
    HIPCHK(hipSetDevice(c->m->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    drop_graphs(c);
    c->sched = sched;
    return 0;
}
extern "C" int lnb_ctx_get_schedule(const lnb_ctx* c) { return c ? c->sched : -1; }
// Decode attention form: one-token calls at contexts above long_threshold run the chip-wide long-context kernels (bit-identical to the
// one-workgroup-per-head kernel, tests/test_gpu_configs.py).  long_threshold < 0 keeps the current value; force_zseq = 1 makes the
// long-context kernel always walk the reference's serial f64 sum instead of certifying the tree estimate (test hook).
extern "C" int lnb_ctx_set_attention(lnb_ctx* c, int long_threshold, int force_zseq) {
    if (!c) return fail("null argument");
    HIPCHK(hipSetDevice(c->m->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    drop_graphs(c);
    if (long_threshold >= 0) c->attn_long_T = long_threshold;
    c->force_zseq = force_zseq & 15;                         // bit 0: serial denominator; bit 1: two-launch long-context form; bit 2: one launch, every poll times out (tests); bit 3: one launch
    return 0;
}
// how many (head, token, layer) rows of the long-context attention had to walk the serial Z chain because the estimate could not be certified
extern "C" int lnb_ctx_zseq_count(lnb_ctx* c, int* out) {
    if (!c || !out) return fail("null argument");
    HIPCHK(hipSetDevice(c->m->device));
    HIPCHK(hipMemcpyAsync(c->h_io, c->zseq_count, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c->h_io[0];
    return 0;
}
// how many fused-RMSNorm rows (one per norm-fused GEMV launch and row) could not be summed by the branch-free item walk and took the record walk
extern "C" int lnb_ctx_norm_fallbacks(lnb_ctx* c, int* out) {
    if (!c || !out) return fail("null argument");
    HIPCHK(hipSetDevice(c->m->device));
    HIPCHK(hipMemcpyAsync(c->h_io, c->zseq_count + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c->h_io[0];
    return 0;
}
// which matrix-core attention the last multi-row call of the context ran (check_call decides, per call): 3 scores once / 1 scores twice / 0 none
extern "C" int lnb_ctx_prefill_attention_form(const lnb_ctx* c, int* out) {
    if (!c || !out) return fail("null argument");
    *out = c->last_prefill_form;
    return 0;
}
extern "C" int lnb_ctx_set_layer_callback(lnb_ctx* c, lnb_layer_cb cb, void* user) { if (!c) return fail("null argument"); c->cb = cb; c->cb_user = user; return 0; }
extern "C" void* lnb_ctx_hidden_ptr(lnb_ctx* c, int which) { return !c ? nullptr : which == 2 ? (void*)c->ffn : (void*)c->x; }
extern "C" void* lnb_ctx_stream(lnb_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int lnb_ctx_synchronize(lnb_ctx* c) { if (!c) return fail("null argument"); HIPCHK(hipSetDevice(c->m->device)); HIPCHK(hipStreamSynchronize(c->stream)); return 0; }

// ---- the five launches of one transformer block (position comes from c->st on the device) -----------
enum { K_QKV = 0, K_ATTN = 1, K_WO = 2, K_W13 = 3, K_W2 = 4, K_HEAD = 5, K_LAYER = 6 };
// Calls of 16 or more rows (prefill) run the same exact chains on the f32 matrix cores (gemm_mfma_kernel); LNB_PREFILL_MFMA=0
// keeps them on the S = 1 kernels (one launch row per token row), which is what the parity tests compare the two with.
static bool use_mfma(int S) { static const int on = env_int("LNB_PREFILL_MFMA", 1); return on && S >= 16; }
static GemmParams gemm_of(const TiledDesc& t, const uint16_t* x, int K, int n_rows, int S, const StepState* st) {
    GemmParams g{}; g.w = t.w; g.rw = t.rw; g.nch = t.nch; g.x = x; g.K = K; g.n_rows = n_rows; g.S = S; g.st = st; g.w16 = t.w16; return g;
}
// S < 16 rows: the exact-order chain kernels, or (LNB_MODE_FAST) the split-K kernels over the same resident weights
static hipError_t gemv_dispatch(const lnb_ctx* c, const GemvParams* g, int rw, int nch, int epi, int norm, hipStream_t st) {
    return c->mode == LNB_MODE_FAST ? lnbk_fast_gemv(g, rw, nch, epi, norm, st) : lnbk_gemv(g, rw, nch, epi, norm, st);
}
// S >= 16 rows: the exact chains on the f32 matrix cores, or (LNB_MODE_FAST) the bf16 matrix-core GEMM; shapes the fast kernel does not
// take (K not a multiple of 64) stay on the exact one
static hipError_t gemm_dispatch(int mode, const GemmParams* g, int epi, hipStream_t st) {
    // below ~200 rows the bf16 GEMM's 256-row weight tiles leave most CUs without work (wq|wk|wv: 24 workgroups) and the exact
    // kernel with its 16-row tiles is the faster one (128 rows: 29.6 ms against 36.6 ms per Forward): the tolerance mode may always
    // use an exact kernel
    static const int min_rows = env_int("LNB_FAST_GEMM_MIN_ROWS", 192);
    if (mode == LNB_MODE_FAST && g->S >= min_rows) { hipError_t e = lnbk_fast_gemm(g, epi, st); if (e != hipErrorNotSupported) return e; }
    // exact order with the matrix-core copy of the weights present (lnb_model_enable_batch): the streaming feed -- the same chains, the
    // same bits (tests/test_gpu_batch.py), the weights never staged through the LDS; LNB_PREFILL_STREAM=0 keeps the LDS-tiled kernel
    // Round 5: the same kernel also reads the RESIDENT layouts (no copy needed: the row-broadcast units are M16 units in another order, the chain
    // layouts' units are transposed inside the lane quads); LNB_PREFILL_NATIVE=0 keeps the LDS-tiled kernel for models without the copy
    static const int stream_on = env_int("LNB_PREFILL_STREAM", 1), native_on = env_int("LNB_PREFILL_NATIVE", 1);
    if (stream_on && (g->K & 127) == 0 && (g->w16 || (native_on && g->w && (g->rw == 4 ? g->nch == 1 && (epi == EPI_STORE || epi == EPI_RESID) : g->rw >= 16))))
        return lnbk_gemm_stream(g, epi, g_num_cus, st);
    return lnbk_gemm(g, epi, st);
}
static int enqueue_layer_kernel(lnb_ctx* c, int l, int S, int which, hipStream_t st_other = nullptr, int lds_pad = 0) {
    lnb_model* m = c->m; const lnb_model_args& a = m->a; hipStream_t st = st_other ? st_other : c->stream;
    LayerW& L = m->layers[l - m->layer_begin];
    uint16_t* ck = c->ck[l - m->layer_begin]; uint16_t* cv = c->cv[l - m->layer_begin];
    // h = x + attention (llamatransformer.go:232) normally has its own buffer; in a block this stage holds only a part of, it lives
    // in x itself (the residual epilogues read and write element n from the same lane, so out may alias res): the hand-off inside a
    // block is then the same buffer as the hand-off between blocks (plus the ffn activations when the cut is in front of w2)
    uint16_t* const hbuf = m->whole(l) ? c->h : c->x;
    if (use_mfma(S) && which != K_ATTN) {
        switch (which) {
        case K_QKV: {
            HIPCHK(lnbk_rmsnorm_rows(c->x, L.attn_norm, c->xn, S, a.dim, a.norm_eps, st));
            GemmParams g = gemm_of(L.wqkv, c->xn, a.dim, L.wqkv.n_rows, S, c->st);
            g.cis = m->cis; g.q_out = c->q; g.cache_k = ck; g.cache_v = cv; g.seq_len = c->seq_len; g.q_dim = m->q_dim; g.kv_dim = m->kv_dim; g.head_dim = m->head_dim;
            HIPCHK(gemm_dispatch(c->mode, &g, EPI_QKV_ROPE, st)); return 0; }
        case K_WO: { GemmParams g = gemm_of(L.wo, c->att, m->q_dim, a.dim, S, c->st); g.out = hbuf; g.res = c->x; HIPCHK(gemm_dispatch(c->mode, &g, EPI_RESID, st)); return 0; }
        case K_W13: {
            HIPCHK(lnbk_rmsnorm_rows(hbuf, L.ffn_norm, c->xn, S, a.dim, a.norm_eps, st));
            GemmParams g = gemm_of(L.w13, c->xn, a.dim, m->ffn_hidden, S, c->st); g.out = c->ffn; g.silu = m->silu;
            HIPCHK(gemm_dispatch(c->mode, &g, EPI_SILU_MUL, st)); return 0; }
        case K_W2: { GemmParams g = gemm_of(L.w2, c->ffn, m->ffn_hidden, a.dim, S, c->st); g.out = c->x; g.res = hbuf; HIPCHK(gemm_dispatch(c->mode, &g, EPI_RESID, st)); return 0; }
        }
    }
    switch (which) {
    case K_QKV: {   // attn_norm + wq|wk|wv + RoPE + KV append  (llamatransformer.go:222, :297-403)
        GemvParams g{}; g.w = L.wqkv.w; g.x = c->x; g.norm_w = L.attn_norm; g.norm_fb = c->zseq_count + 1; g.eps = a.norm_eps; g.K = a.dim; g.n_rows = L.wqkv.n_rows; g.S = S; g.st = c->st;
        g.cis = m->cis; g.q_out = c->q; g.cache_k = ck; g.cache_v = cv; g.seq_len = c->seq_len; g.q_dim = m->q_dim; g.kv_dim = m->kv_dim; g.head_dim = m->head_dim;
        g.sched = c->sched;
        set_grid(g, L.wqkv); HIPCHK(gemv_dispatch(c, &g, L.wqkv.rw, 1, EPI_QKV_ROPE, 1, st)); return 0; }
    case K_ATTN: {  // scores / softmax / PV  (:409-514)
        AttnParams ap{}; ap.q = c->q; ap.cache_k = ck; ap.cache_v = cv; ap.out = c->att; ap.st = c->st; ap.dbg = g_dbg;
        ap.S = S; ap.H = a.n_heads; ap.KVH = a.n_kv_heads; ap.hd = m->head_dim; ap.seq_len = c->seq_len;
        ap.lds_T = c->seq_len < c->attn_short_cap ? c->seq_len : c->attn_short_cap;
        ap.host_T = c->call_T;
        ap.divisor = bf_wide_h(bf_trunc_h((float)std::sqrt((double)m->head_dim)));           // llamatransformer.go:464
        ap.mfma = use_mfma(S) ? 1 : 0; ap.exp_tab = m->exp_tab;
        ap.longctx = 0; ap.force_zseq = c->force_zseq & 1; ap.e_buf = c->e_buf; ap.z_part = c->z_part; ap.zseq_count = c->zseq_count; ap.cnt = c->attn_cnt;
        if (ap.mfma && c->sidx_jt > 0 && c->call_T > 0 && c->sidx_jt * 16 >= c->call_T) { ap.score_idx = c->score_idx; ap.sidx_jt = c->sidx_jt; }     // (round 6: scores once, attn_mfma3_kernel)
        if (S == 1 && c->attn_long) {
            // ONE launch (attn_one_kernel, round 6): built, bit-exact in every form (tests/test_gpu_round6.py), and NOT the default -- measured on MI355X
            // (profiles/r06_att_timing.log, 8B head geometry): T = 4101: 26.3 us against 23.2 for the two launches; 1024: 14.1 / 11.5; 272: 8.2 against 7.7
            // for the one-workgroup-per-head kernel.  What the launch boundary costs (~1.2 us + the PV kernel's cold reads of e_buf) comes back as the
            // in-launch exchange (write-through stores drained + arrive + poll: 6.6 k cycles) plus a second dependent round trip for e_j (6.3 k).
            // Opt-in: LNB_ATTN_ONE=1, or bit 3 (8) of lnb_ctx_set_attention's flags; only for a context whose stream owns the chip (latency schedule)
            // and whose (head, slice) grid fits the CUs.  Bit 1 (2) of the flags keeps the two launches whatever the environment says.
            static const int one_env = env_int("LNB_ATTN_ONE", 0);
            const bool one = (one_env || (c->force_zseq & 12)) && !(c->force_zseq & 2) && c->sched == LNB_SCHED_LATENCY && m->head_dim % 16 == 0 &&
                             a.n_heads * (m->head_dim / 16) <= g_num_cus && lnbk_attn_one_lds(c->seq_len, m->head_dim) <= (size_t)160 * 1024;
            ap.longctx = one ? ((c->force_zseq & 4) ? 3 : 2) : 1;
        }
        if (c->mode == LNB_MODE_FAST && ap.mfma) {           // tolerance mode prefill: flash form on the bf16 matrix cores
            hipError_t e = lnbk_fast_attn(&ap, st);
            if (e != hipErrorNotSupported) { HIPCHK(e); return 0; }
        }
        HIPCHK(lnbk_attn(&ap, st)); return 0; }
    case K_WO: {    // wo + residual  (:522, :232)
        GemvParams o{}; o.w = L.wo.w; o.x = c->att; o.K = m->q_dim; o.n_rows = a.dim; o.S = S; o.st = c->st; o.out = hbuf; o.res = c->x; o.sched = c->sched;
        set_grid(o, L.wo); HIPCHK(gemv_dispatch(c, &o, L.wo.rw, 1, EPI_RESID, 0, st)); return 0; }
    case K_W13: {   // ffn_norm + w1|w3 + SiLU*up  (:237, :601-617)
        GemvParams f{}; f.w = L.w13.w; f.x = hbuf; f.norm_w = L.ffn_norm; f.norm_fb = c->zseq_count + 1; f.eps = a.norm_eps; f.K = a.dim; f.n_rows = m->ffn_hidden; f.S = S; f.st = c->st;
        f.out = c->ffn; f.silu = m->silu; f.sched = c->sched;
        set_grid(f, L.w13); HIPCHK(gemv_dispatch(c, &f, L.w13.rw, 2, EPI_SILU_MUL, 1, st)); return 0; }
    case K_W2: {    // w2 + residual  (:619, :248)
        GemvParams d{}; d.w = L.w2.w; d.x = c->ffn; d.K = m->ffn_hidden; d.n_rows = a.dim; d.S = S; d.st = c->st; d.out = c->x; d.res = hbuf; d.lds_pad = lds_pad; d.sched = c->sched; d.prio = env_int("LNB_W2_PRIO", 0);
        set_grid(d, L.w2); HIPCHK(gemv_dispatch(c, &d, L.w2.rw, 1, EPI_RESID, 0, st)); return 0; }
    }
    return fail("bad kernel id");
}
static int enqueue_layers(lnb_ctx* c, int S, bool with_cb) {
    lnb_model* m = c->m;
    for (int l = m->layer_begin; l < m->layer_end; l++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int k = K_QKV; k <= K_W2; k++) {
            if (!(k <= K_WO ? m->has_attn(l) : k == K_W13 ? m->has_w13(l) : m->has_w2(l))) continue;   // a stage may hold only parts of its first / last block
            if (enqueue_layer_kernel(c, l, S, k)) return -1;
        }
        if (with_cb && c->cb) {
            HIPCHK(hipStreamSynchronize(c->stream));
            double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            c->cb(l + 1, m->a.n_layers, secs, c->cb_user);                    // llamatransformer.go:163
        }
    }
    return 0;
}
// final RMSNorm + output projection of `rows` rows starting at row `first` (llamatransformer.go:166-170)
static int enqueue_head(lnb_ctx* c, int first, int rows) {
    lnb_model* m = c->m;
    if (use_mfma(rows)) {
        HIPCHK(lnbk_rmsnorm_rows(c->x + (size_t)first * m->a.dim, m->norm, c->xn, rows, m->a.dim, m->a.norm_eps, c->stream));
        GemmParams gm = gemm_of(m->output, c->xn, m->a.dim, m->a.vocab_size, rows, c->st); gm.out = c->logits;
        HIPCHK(gemm_dispatch(c->mode, &gm, EPI_STORE, c->stream));
        return 0;
    }
    GemvParams g{}; g.w = m->output.w; g.x = c->x + (size_t)first * m->a.dim; g.norm_w = m->norm; g.norm_fb = c->zseq_count + 1; g.eps = m->a.norm_eps; g.K = m->a.dim;
    g.n_rows = m->a.vocab_size; g.S = rows; g.st = c->st; g.out = c->logits;
    set_grid(g, m->output);
    HIPCHK(gemv_dispatch(c, &g, m->output.rw, 1, EPI_STORE, 1, c->stream));
    return 0;
}

// one-token call at context T: the long-context kernels above the crossover, and always beyond the short kernel's LDS reach
static bool want_long_attention(const lnb_ctx* c, int seq, int start_pos) {
    return seq == 1 && (start_pos + 1 > c->attn_long_T || start_pos + 1 > c->attn_short_cap);
}
static int check_call(lnb_ctx* c, int seq, int start_pos) {
    if (seq == 0) return fail("empty token array");                                            // llamatransformer.go:146-148
    if (seq < 0 || start_pos < 0) return fail("negative sequence length or start position");
    const int T = start_pos + seq;
    if (T > c->m->cis_rows) return fail("incompatible locStart, locEnd values and tensor (position %d beyond the %d-row RoPE table)", T, c->m->cis_rows);
    if (T > c->seq_len) return fail("incompatible locStart, locEnd values and tensor (position %d beyond the KV cache of %d)", T, c->seq_len);
    if (seq > 1 && T % seq != 0) return fail("two tensor shapes cannot be broadcasted: [%d %d %d] and [%d %d]", c->m->a.n_heads, seq, T, seq, seq);
    // calls of 2.. rows that do NOT run on the matrix-core attention (fewer than 16 rows, or head_dim 32, which attn_mfma_kernel does not
    // take) go through the row-per-workgroup kernel, whose LDS arrays are sized for attn_short_cap positions
    const bool mfma_attn = use_mfma(seq) && (c->m->head_dim == 64 || c->m->head_dim == 128);
    c->call_T = T;
    // round 6: the matrix-core attention of an exact prefill keeps the table indices of its first pass (attn_mfma3_kernel): 512 bytes per (head, 16 query rows, 16
    // positions), one buffer per context, reused by every layer.  Grown here, on the host side of the call (never inside a capture); above LNB_ATTN_SIDX_MB
    // (default 4096; 0 = never) the call runs attn_mfma_kernel, which computes the scores twice instead.
    c->sidx_jt = 0;
    bool stage_has_attn = false;                             // (a pipeline stage cut inside a block may hold FFN parts only: no attention, no scratch)
    for (int l = c->m->layer_begin; l < c->m->layer_end && !stage_has_attn; l++) stage_has_attn = c->m->has_attn(l);
    // a one-token call on a context that still holds a LARGE scratch from its prompt (1.07 GB for 4096 rows of the 8B shape) gives it back: the decode steps never read it,
    // and a host that prefills many contexts would otherwise keep a gigabyte per context for their whole life.  Up to LNB_ATTN_SIDX_KEEP_MB (default 256) it stays for the next prompt.
    if (seq == 1 && c->score_idx && c->score_idx_bytes > ((size_t)env_int("LNB_ATTN_SIDX_KEEP_MB", 256) << 20)) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->score_idx); c->score_idx = nullptr; c->score_idx_bytes = 0;
    }
    if (mfma_attn && stage_has_attn && c->mode != LNB_MODE_FAST) {
        const int cap_mb = env_int("LNB_ATTN_SIDX_MB", 4096);                 // (read per call: a test switches it inside one process)
        const size_t jt = (size_t)(T + 15) / 16, need = (size_t)c->m->a.n_heads * (size_t)((seq + 15) / 16) * jt * 512;
        if (cap_mb > 0 && need <= ((size_t)cap_mb << 20)) {
            if (need > c->score_idx_bytes) {
                HIPCHK(hipStreamSynchronize(c->stream));
                if (c->score_idx) { hipFree(c->score_idx); c->score_idx = nullptr; c->score_idx_bytes = 0; }
                if (hipMalloc((void**)&c->score_idx, need) == hipSuccess) c->score_idx_bytes = need;
                else { (void)hipGetLastError(); c->score_idx = nullptr; }             // (no room: the two-pass kernel needs none)
            }
            if (c->score_idx) c->sidx_jt = (int)jt;
        }
    }
    if (seq > 1) c->last_prefill_form = (!mfma_attn || !stage_has_attn || c->mode == LNB_MODE_FAST) ? 0 : c->sidx_jt > 0 ? 3 : 1;
    if (seq > 1 && !mfma_attn && T > c->attn_short_cap)
        return fail("a call of %d rows (2..15, or any multi-row call at head_dim 32) at context %d: the row-per-workgroup attention kernel stages "
                    "at most %d positions in the LDS; use one-token calls or 16 or more rows there", seq, T, c->attn_short_cap);
    return 0;
}

// lnb_forward_stage split in two so that a pipeline rank can put its exchange with the neighbouring ranks in flight while the stage
// computes: _begin enqueues the whole stage (embedding gather on the first stage, the blocks, norm + output + argmax on the last
// when want_argmax) on the ctx's stream and returns; _end waits for it and reports the token / the vocabulary check.
extern "C" int lnb_forward_stage_begin(lnb_ctx* c, const int32_t* tokens, int seq, int start_pos, int want_argmax) {
    if (!c) return fail("null argument");
    lnb_model* m = c->m; const int V = m->a.vocab_size;
    HIPCHK(hipSetDevice(m->device));
    if (c->pending) return fail("lnb_forward_stage_begin: the previous call has not been ended");
    if (check_call(c, seq, start_pos)) return -1;
    if (tokens && !m->first()) return fail("tokens given to a stage that does not own tok_embeddings");
    if (!tokens && m->first()) return fail("first stage needs tokens");
    if (want_argmax && !m->last()) return fail("logits requested from a stage that does not own output.weight");
    hipStream_t st = c->stream;
    c->attn_long = want_long_attention(c, seq, start_pos);
    HIPCHK(ctx_set_state(c, start_pos, 0, true));
    if (tokens) {
        if (staging_acquire(c)) return -1;
        memcpy(c->h_io + 2, tokens, (size_t)seq * 4);        // the caller's array need not outlive this call
        HIPCHK(hipMemcpyAsync(c->dtok, c->h_io + 2, (size_t)seq * 4, hipMemcpyHostToDevice, st));
        if (staging_release(c)) return -1;
        HIPCHK(hipMemsetAsync(c->derr, 0, 4, st));
        HIPCHK(lnbk_embed(m->tok_embd, c->dtok, c->x, seq, m->a.dim, V, c->derr, st));       // Fwd_Get_Rows :118
        HIPCHK(hipMemcpyAsync(c->h_io + 1, c->derr, 4, hipMemcpyDeviceToHost, st));
    }
    if (enqueue_layers(c, seq, false)) return -1;
    if (want_argmax) {
        if (c->logits_rows < 1) { HIPCHK(hipMalloc((void**)&c->logits, (size_t)V * 2)); c->logits_rows = 1; }
        if (enqueue_head(c, seq - 1, 1)) return -1;
        HIPCHK(lnbk_argmax(c->logits, V, c->dnext, c->st, c->dout, c->dout_cap, 0, st));        // inference.go:207-211
        HIPCHK(hipMemcpyAsync(c->h_io, c->dnext, 4, hipMemcpyDeviceToHost, st));
    }
    c->pending = true; c->pending_tokens = tokens != nullptr; c->pending_argmax = want_argmax != 0;
    return 0;
}
extern "C" int lnb_forward_stage_end(lnb_ctx* c, int32_t* argmax_last_out) {
    if (!c) return fail("null argument");
    if (!c->pending) return fail("lnb_forward_stage_end without a begin");
    HIPCHK(hipSetDevice(c->m->device));
    c->pending = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pending_tokens && c->h_io[1]) return fail("token id at index %d is outside the vocabulary", c->h_io[1] - 1);
    if (argmax_last_out) {
        if (!c->pending_argmax) return fail("lnb_forward_stage_end: no argmax was requested at begin");
        *argmax_last_out = c->h_io[0];
    }
    return 0;
}

extern "C" int lnb_forward_stage(lnb_ctx* c, const int32_t* tokens, int seq, int start_pos, float* logits_out, int32_t* argmax_last_out) {
    if (!c) return fail("null argument");
    lnb_model* m = c->m; const int V = m->a.vocab_size;
    HIPCHK(hipSetDevice(m->device));
    if (check_call(c, seq, start_pos)) return -1;
    if (tokens && !m->first()) return fail("tokens given to a stage that does not own tok_embeddings");
    if (!tokens && m->first()) return fail("first stage needs tokens");
    if ((logits_out || argmax_last_out) && !m->last()) return fail("logits requested from a stage that does not own output.weight");
    hipStream_t st = c->stream;
    c->attn_long = want_long_attention(c, seq, start_pos);
    HIPCHK(ctx_set_state(c, start_pos, 0, true));
    if (tokens) {
        HIPCHK(hipMemcpyAsync(c->dtok, tokens, (size_t)seq * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(c->derr, 0, 4, st));
        HIPCHK(lnbk_embed(m->tok_embd, c->dtok, c->x, seq, m->a.dim, V, c->derr, st));       // Fwd_Get_Rows :118
    }
    if (enqueue_layers(c, seq, true)) return -1;
    if (m->last() && (logits_out || argmax_last_out)) {
        const int rows = logits_out ? seq : 1;
        if (rows > c->logits_rows) {
            HIPCHK(hipStreamSynchronize(st));
            // the captured decode graph has the old buffer baked into its head GEMV / argmax nodes: drop it with the buffer
            drop_graphs(c);
            hipFree(c->logits); c->logits = nullptr; c->logits_rows = 0;
            HIPCHK(hipMalloc((void**)&c->logits, (size_t)rows * V * 2)); c->logits_rows = rows;
        }
        if (enqueue_head(c, logits_out ? 0 : seq - 1, rows)) return -1;
        HIPCHK(lnbk_argmax(c->logits + (size_t)(rows - 1) * V, V, c->dnext, c->st, c->dout, c->dout_cap, 0, st));   // inference.go:207-211
        if (logits_out) {
            std::vector<uint16_t> hb((size_t)rows * V);
            HIPCHK(hipMemcpyAsync(hb.data(), c->logits, hb.size() * 2, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (size_t i = 0; i < hb.size(); i++) logits_out[i] = bf_wide_h(hb[i]);              // output.ToFloat32() :175
        }
        if (argmax_last_out) HIPCHK(hipMemcpyAsync(argmax_last_out, c->dnext, 4, hipMemcpyDeviceToHost, st));
    }
    if (tokens) HIPCHK(hipMemcpyAsync(c->h_io + 1, c->derr, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (tokens && c->h_io[1]) return fail("token id at index %d is outside the vocabulary", c->h_io[1] - 1);
    return 0;
}

extern "C" int lnb_forward(lnb_ctx* c, const int32_t* tokens, int seq, int start_pos, float* logits_out, int32_t* argmax_last_out) {
    if (!c || !tokens) return fail("null argument");
    if (!c->m->first() || !c->m->last()) return fail("lnb_forward needs a whole-model handle; use lnb_forward_stage for pipeline stages");
    return lnb_forward_stage(c, tokens, seq, start_pos, logits_out, argmax_last_out);
}

// one decode step = embed(token on device) -> layers -> head -> argmax that feeds the next step
static int enqueue_decode_step(lnb_ctx* c) {
    lnb_model* m = c->m;
    // LNB_MEASURE_SKIP_TOKEN_KERNELS=1 (timing only, the tokens are garbage): the step without its embedding gather and argmax launches =
    // the most that fusing them into the neighbouring products could return (NOTES 5.8)
    const bool skip = env_int("LNB_MEASURE_SKIP_TOKEN_KERNELS", 0) != 0;
    if (!skip) HIPCHK(lnbk_embed(m->tok_embd, c->dtok, c->x, 1, m->a.dim, m->a.vocab_size, c->derr, c->stream));
    if (enqueue_layers(c, 1, false)) return -1;
    if (enqueue_head(c, 0, 1)) return -1;
    if (!skip) HIPCHK(lnbk_argmax(c->logits, m->a.vocab_size, c->dtok, c->st, c->dout, c->dout_cap, 1, c->stream));
    return 0;
}

// Stop ids on the device (inference.go:233-252: generation ends with the first token that is one of model.StopTokenIds, and that token is
// emitted): the argmax kernel compares every generated token with the context's stop ids and freezes the generation -- position, token
// word, token log -- when one matches; whatever was enqueued behind it recomputes the same step and changes nothing.  0 ids = never stops.
extern "C" int lnb_ctx_set_stop_ids(lnb_ctx* c, const int32_t* ids, int n) {
    if (!c || (n > 0 && !ids)) return fail("null argument");
    if (n < 0 || n > LNB_MAX_STOP_IDS) return fail("a context takes 0..%d stop ids (got %d)", LNB_MAX_STOP_IDS, n);
    HIPCHK(hipSetDevice(c->m->device));
    HIPCHK(lnbk_set_stop(c->st, ids, n, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_stop = n;
    return 0;
}
static int decode_greedy_impl(lnb_ctx* c, int32_t token, int start_pos, int n_steps, int32_t* out_tokens, int* n_generated, int* finished, float* ms_out);
extern "C" int lnb_decode_greedy(lnb_ctx* c, int32_t token, int start_pos, int n_steps, int32_t* out_tokens, float* ms_out) {
    return decode_greedy_impl(c, token, start_pos, n_steps, out_tokens, nullptr, nullptr, ms_out);
}
// the same loop, reporting how far it got: *n_generated tokens are valid in out_tokens (max_steps unless a stop id ended the generation;
// the stop token is the last of them), *finished = 1 if a stop id did
extern "C" int lnb_decode_greedy_until(lnb_ctx* c, int32_t token, int start_pos, int max_steps, int32_t* out_tokens, int* n_generated, int* finished, float* ms_out) {
    if (!n_generated) return fail("null argument");
    return decode_greedy_impl(c, token, start_pos, max_steps, out_tokens, n_generated, finished, ms_out);
}
static int decode_greedy_impl(lnb_ctx* c, int32_t token, int start_pos, int n_steps, int32_t* out_tokens, int* n_generated, int* finished, float* ms_out) {
    if (!c || !out_tokens) return fail("null argument");
    lnb_model* m = c->m;
    if (!m->first() || !m->last()) return fail("lnb_decode_greedy needs a whole-model handle");
    HIPCHK(hipSetDevice(m->device));
    if (n_steps <= 0) return fail("n_steps must be positive");
    if (n_steps > c->dout_cap) return fail("n_steps %d exceeds the context length %d", n_steps, c->dout_cap);
    if (check_call(c, 1, start_pos) || check_call(c, 1, start_pos + n_steps - 1)) return -1;
    if (token < 0 || token >= m->a.vocab_size) return fail("token id at index 0 is outside the vocabulary");
    hipStream_t st = c->stream;
    const bool use_graph = env_int("LNB_NO_GRAPH", 0) == 0;
    // one captured graph per attention form: the step at context T replays the long-context one when T exceeds the crossover
    auto capture = [&](hipGraphExec_t* slot, bool longctx) -> int {
        if (*slot) return 0;
        hipGraph_t g = nullptr;
        c->attn_long = longctx;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_decode_step(c);
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) { if (g) hipGraphDestroy(g); return -1; }
        HIPCHK(e);
        HIPCHK(hipGraphInstantiate(slot, g, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(g));
        return 0;
    };
    const bool any_short = !want_long_attention(c, 1, start_pos), any_long = want_long_attention(c, 1, start_pos + n_steps - 1);
    if (use_graph && any_short && capture(&c->graph, false)) return -1;
    if (use_graph && any_long && capture(&c->graph_long, true)) return -1;
    HIPCHK(hipMemcpyAsync(c->dtok, &token, 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c->derr, 0, 4, st));
    // (the argmax kernel advances the position on the device from here on.)  Stop ids are compared only on behalf of a caller that learns how far
    // the run got: lnb_decode_greedy promises n_steps tokens and ignores them -- it used to freeze silently at a stop id and hand back stale
    // log entries behind it
    HIPCHK(ctx_set_state(c, start_pos, 0, false, n_generated != nullptr));
    c->call_T = 0;                                           // (the captured graphs serve every position: nothing host-side to validate)
    HIPCHK(hipEventRecord(c->ev0, st));
    for (int i = 0; i < n_steps; i++) {
        const bool longctx = want_long_attention(c, 1, start_pos + i);
        if (use_graph) HIPCHK(hipGraphLaunch(longctx ? c->graph_long : c->graph, st));
        else { c->attn_long = longctx; if (enqueue_decode_step(c)) return -1; }
    }
    HIPCHK(hipEventRecord(c->ev1, st));
    HIPCHK(hipMemcpyAsync(out_tokens, c->dout, (size_t)n_steps * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->h_io + 1, c->derr, 4, hipMemcpyDeviceToHost, st));
    StepState hs{};
    if (n_generated) HIPCHK(hipMemcpyAsync(&hs, c->st, sizeof(StepState), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (ms_out) HIPCHK(hipEventElapsedTime(ms_out, c->ev0, c->ev1));
    if (c->h_io[1]) return fail("generated token id is outside the vocabulary");
    if (n_generated) { *n_generated = hs.n_out; if (finished) *finished = hs.finished; c->dev_pos = -1; }     // (a stopped run leaves the device position short of start_pos + n_steps)
    return 0;
}

// One launch of a GEMV class with the per-wave debug buffer armed (GemvParams.dbg): out[w * 16 + ..] for wave w = {workgroups that reported, avg
// total cycles, max total, avg barrier wait, avg [2] (x staged / prologue end), avg [3] (fold or walk; rowcast_lds chain waves: chain start),
// avg stamps 0..7 (stamp 7 = the launch on the constant-rate wall clock)}.  The device position must have been set by the caller.
static int gemv_stamps(lnb_ctx* c, int which, double* out, int full) {
    lnb_model* m = c->m; hipStream_t st = c->stream;
    const size_t n = (size_t)4096 * 8 * 4, n2 = (size_t)4096 * 8 * 8;       // + 8 phase stamps per wave behind the four totals (LNB_STAMP in lnb_kernels.hip)
    long long* dbuf = nullptr;
    HIPCHK(hipMalloc((void**)&dbuf, (n + n2) * 8));
    { hipError_t e = hipMemsetAsync(dbuf, 0, (n + n2) * 8, st); if (e != hipSuccess) { hipFree(dbuf); HIPCHK(e); } }
    const int nl = m->layer_end - m->layer_begin;
    if (nl <= 0 && which != K_HEAD) { hipFree(dbuf); return fail("this stage holds no transformer block: only kernel class %d (norm + output) can be stamped", K_HEAD); }   // (i % nl below)
    // three plain launches of the class on other layers first, back to back with the stamped one: it then runs as a launch inside a decode step does
    // (instruction cache, clocks and memory pipeline warm) instead of as the first launch after an idle gap
    int rc = 0;
    for (int i = 4; i < 7 && !rc; i++) rc = which == K_HEAD ? enqueue_head(c, 0, 1) : enqueue_layer_kernel(c, m->layer_begin + i % nl, 1, which);
    g_dbg = dbuf; g_dbg_full = full;
    if (!rc) rc = which == K_HEAD ? enqueue_head(c, 0, 1) : enqueue_layer_kernel(c, m->layer_begin + 7 % nl, 1, which);
    g_dbg = nullptr; g_dbg_full = 0;
    if (rc) { hipFree(dbuf); return -1; }
    std::vector<long long> h(n + n2);
    {   // (no early return between here and the free: ADVICE r5)
        hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(h.data(), dbuf, (n + n2) * 8, hipMemcpyDeviceToHost);
        hipFree(dbuf);
        HIPCHK(e);
    }
    for (int w = 0; w < 8; w++) {
        double* o = out + w * 16; for (int k = 0; k < 16; k++) o[k] = 0;
        int cnt = 0;
        for (int g = 0; g < 4096; g++) {
            const long long* d = &h[((size_t)g * 8 + w) * 4]; const long long* s8 = &h[n + ((size_t)g * 8 + w) * 8];
            if (d[0] <= 0) continue;
            cnt++; o[1] += (double)d[0]; if ((double)d[0] > o[2]) o[2] = (double)d[0]; o[3] += (double)d[1]; o[4] += (double)d[2]; o[5] += (double)d[3];
            for (int k = 0; k < 8; k++) o[6 + k] += (double)s8[k];
        }
        o[0] = cnt;
        if (cnt) { o[1] /= cnt; o[3] /= cnt; o[4] /= cnt; o[5] /= cnt; for (int k = 0; k < 8; k++) o[6 + k] /= cnt; }
    }
    return 0;
}
// measurement aid (bench.py: roofline.measured_model): the in-kernel cycle stamps of ONE launch of a GEMV class (which: 0, 2, 3, 4, 5 as in
// lnb_profile_kernel) at position pos; out = 8 waves x 16 doubles as gemv_stamps lays them out; *wall_clock_khz = the rate of stamp 7's clock
extern "C" int lnb_profile_kernel_stamps(lnb_ctx* c, int which, int pos, double* out, int* wall_clock_khz) {
    if (!c || !out) return fail("null argument");
    lnb_model* m = c->m;
    HIPCHK(hipSetDevice(m->device));
    if (which < 0 || which > K_HEAD || which == K_ATTN) return fail("bad arguments");
    if (check_call(c, 1, pos)) return -1;
    if (which == K_HEAD && !m->last()) return fail("this stage does not own output.weight");
    c->attn_long = want_long_attention(c, 1, pos);
    HIPCHK(ctx_set_state(c, pos, 0, true));
    if (wall_clock_khz) { int khz = 0; HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m->device)); *wall_clock_khz = khz; }
    return gemv_stamps(c, which, out, 0);                    // light: the launch runs as it does in production
}

// measurement aid (tools/ffn_overlap.py, profiles/r06_ffn_stream.md): what a w1|w3 -> w2 streaming stage could reach, measured before it is built.  The gate|up
// kernel and the down kernel of a block are launched on two streams, the down kernel `w2_delay_us` microseconds behind the gate|up kernel (a one-wave
// spin kernel in front of it holds its stream): delay 0 = plain co-residency, delay ~ "the first row band is complete" = the timeline of a band
// pipeline, with no dependency stall at all (w2 reads STALE activations -- the results are meaningless, only the time is).  w2_lds_pad pads w2's
// LDS request (r3's way of forcing exactly one workgroup of each kernel per CU).  Both kernels take the forms of the context's schedule.
extern "C" int lnb_profile_ffn_pair(lnb_ctx* c, int pos, int iters, int w2_delay_us, int w2_lds_pad, float* avg_ms_out) {
    if (!c || !avg_ms_out) return fail("null argument");
    lnb_model* m = c->m;
    HIPCHK(hipSetDevice(m->device));
    if (iters <= 0) return fail("bad arguments");
    if (check_call(c, 1, pos)) return -1;
    const int nl2 = m->layer_end - m->layer_begin;
    if (nl2 <= 0) return fail("this stage holds no transformer block");
    hipStream_t st = c->stream, st2 = nullptr; hipEvent_t ef = nullptr, ej = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    c->attn_long = false;
    HIPCHK(ctx_set_state(c, pos, 0, true));
    auto pair = [&](int i) -> int {
        const int l = m->layer_begin + i % nl2;
        HIPCHK(hipEventRecord(ef, st)); HIPCHK(hipStreamWaitEvent(st2, ef, 0));
        if (w2_delay_us >= 0) {
            if (enqueue_layer_kernel(c, l, 1, K_W13)) return -1;
            if (w2_delay_us > 0) HIPCHK(lnbk_spin(w2_delay_us, st2));
            if (enqueue_layer_kernel(c, l, 1, K_W2, st2, w2_lds_pad)) return -1;
        } else {                                             // w2 first (its waves are the older ones), gate|up -delay behind it
            if (enqueue_layer_kernel(c, l, 1, K_W2, st2, w2_lds_pad)) return -1;
            if (w2_delay_us < -1) HIPCHK(lnbk_spin(-w2_delay_us, st));
            if (enqueue_layer_kernel(c, l, 1, K_W13)) return -1;
        }
        HIPCHK(hipEventRecord(ej, st2)); HIPCHK(hipStreamWaitEvent(st, ej, 0));
        return 0;
    };
    int rc = 0;
    for (int i = 0; i < 3 && !rc; i++) rc = pair(i);
    if (!rc) { HIPCHK(hipEventRecord(c->ev0, st)); for (int i = 0; i < iters && !rc; i++) rc = pair(i + 3); HIPCHK(hipEventRecord(c->ev1, st)); }
    hipError_t e = hipDeviceSynchronize();
    float ms2 = 0; if (!rc && e == hipSuccess) e = hipEventElapsedTime(&ms2, c->ev0, c->ev1);
    hipEventDestroy(ef); hipEventDestroy(ej); hipStreamDestroy(st2);
    if (rc) return -1;
    HIPCHK(e);
    *avg_ms_out = ms2 / (float)iters;
    return 0;
}

extern "C" int lnb_profile_kernel(lnb_ctx* c, int which, int pos, int iters, float* avg_ms_out) {
    if (!c || !avg_ms_out) return fail("null argument");
    lnb_model* m = c->m;
    HIPCHK(hipSetDevice(m->device));
    // which 7 / 8 (measurement only): the gate|up kernel and the down kernel of a block launched CONCURRENTLY on two streams (7: w2 first, i.e.
    // its waves are the older ones on every SIMD; 8: w1|w3 first), w2's LDS request padded so that exactly one workgroup of each kernel sits on
    // every CU -- what co-residency would cost a w1|w3 -> w2 row-band pipeline (NOTES.md 6.1); results are NOT meaningful (w2 reads stale input)
    if (iters <= 0 || which < 0 || which > K_LAYER + 2) return fail("bad arguments");
    if (check_call(c, 1, pos)) return -1;
    if (which == K_HEAD && !m->last()) return fail("this stage does not own output.weight");
    hipStream_t st = c->stream;
    if (which > K_LAYER) {
        hipStream_t st2 = nullptr; hipEvent_t ef = nullptr, ej = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        c->attn_long = false;
        HIPCHK(ctx_set_state(c, pos, 0, true));
        const int nl2 = m->layer_end - m->layer_begin, pad = env_int("LNB_W2_LDS_PAD", 28 * 1024);
        auto pair = [&](int i) -> int {
            const int l = m->layer_begin + i % nl2;
            HIPCHK(hipEventRecord(ef, st)); HIPCHK(hipStreamWaitEvent(st2, ef, 0));
            if (which == K_LAYER + 1) { if (enqueue_layer_kernel(c, l, 1, K_W2, st2, pad)) return -1; if (enqueue_layer_kernel(c, l, 1, K_W13)) return -1; }
            else { if (enqueue_layer_kernel(c, l, 1, K_W13)) return -1; if (enqueue_layer_kernel(c, l, 1, K_W2, st2, pad)) return -1; }
            HIPCHK(hipEventRecord(ej, st2)); HIPCHK(hipStreamWaitEvent(st, ej, 0));
            return 0;
        };
        int rc = 0;
        for (int i = 0; i < 3 && !rc; i++) rc = pair(i);
        if (!rc) { HIPCHK(hipEventRecord(c->ev0, st)); for (int i = 0; i < iters && !rc; i++) rc = pair(i + 3); HIPCHK(hipEventRecord(c->ev1, st)); }
        hipError_t e = hipDeviceSynchronize();
        float ms2 = 0; if (!rc && e == hipSuccess) e = hipEventElapsedTime(&ms2, c->ev0, c->ev1);
        hipEventDestroy(ef); hipEventDestroy(ej); hipStreamDestroy(st2);
        if (rc) return -1;
        HIPCHK(e);
        *avg_ms_out = ms2 / (float)iters;
        return 0;
    }
    c->attn_long = want_long_attention(c, 1, pos);
    HIPCHK(ctx_set_state(c, pos, 0, true));
    const int nl = m->layer_end - m->layer_begin;
    // consecutive launches walk through the layers so that every launch streams its weights from HBM
    // (one layer's 235 MB gate/up matrix would otherwise sit in the 256 MiB Infinity Cache)
    auto run = [&](int i) -> int {
        int l = m->layer_begin + (env_int("LNB_PROFILE_SAME_LAYER", 0) ? 0 : (i % nl));
        if (which == K_HEAD) return enqueue_head(c, 0, 1);
        if (which == K_LAYER) { for (int k = K_QKV; k <= K_W2; k++) if (enqueue_layer_kernel(c, l, 1, k)) return -1; return 0; }
        return enqueue_layer_kernel(c, l, 1, which);
    };
    for (int i = 0; i < 3; i++) if (run(i)) return -1;
    HIPCHK(hipEventRecord(c->ev0, st));
    for (int i = 0; i < iters; i++) if (run(i + 3)) return -1;
    HIPCHK(hipEventRecord(c->ev1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms_out = ms / (float)iters;
    if (env_int("LNB_GEMV_TIMING", 0) && which == K_ATTN) {
        long long* dbuf = nullptr;
        HIPCHK(hipMalloc((void**)&dbuf, 64 * 8)); HIPCHK(hipMemsetAsync(dbuf, 0, 64 * 8, st));
        g_dbg = dbuf;
        int rc = run(7);
        g_dbg = nullptr;
        if (rc) { hipFree(dbuf); return -1; }
        HIPCHK(hipStreamSynchronize(st));
        long long h[64] = {0};
        HIPCHK(hipMemcpy(h, dbuf, sizeof h, hipMemcpyDeviceToHost));
        hipFree(dbuf);
        for (int w = 0; w < 4; w++) {
            fprintf(stderr, "[timing] attention pos %d wave %d phases (s_memtime ticks):", pos, w);
            for (int k = 1; k < 7; k++) fprintf(stderr, " %lld", h[w * 16 + k] - h[w * 16 + k - 1]);
            fprintf(stderr, " | first pass: chain %lld exp %lld", h[w * 16 + 7] - h[w * 16 + 1], h[w * 16 + 8] - h[w * 16 + 7]);
            fprintf(stderr, "\n");
        }
    }
    if (env_int("LNB_GEMV_TIMING", 0) && which != K_ATTN && which != K_LAYER) {
        double v[8 * 16];
        if (gemv_stamps(c, which, v, 1)) return -1;
        fprintf(stderr, "[timing] kernel class %d: per-wave s_memtime ticks (avg over workgroups)\n", which);
        for (int w = 0; w < 8; w++) {
            const double* d = v + w * 16;
            if (d[0] > 0) fprintf(stderr, "[timing]   wave %d: n=%d total=%.0f (max %.0f) barrier_wait=%.0f x_or_vmwait=%.0f rms_fold_or_walk=%.0f\n", w, (int)d[0], d[1], d[2], d[3], d[4], d[5]);
            if (d[0] > 0) { fprintf(stderr, "[timing]     stamps (since kernel start):"); for (int k = 0; k < 8; k++) fprintf(stderr, " %.0f", d[6 + k]); fprintf(stderr, "\n"); }
        }
    }
    return 0;
}


// ---- batched exact decode: several independent sequences per pass over the weights --------------------------------------------------
// The reference runs one generation per InferenceContext (src/inference/inference.go:174) and shares the weight matrix across the rows of
// a call (src/ml/operations_lineartransform.go:173-193).  A batch groups up to LNB_BATCH_MAX (128) contexts of ONE whole-model handle: per step every
// sequence's one-token Forward + Argmax happens in a single pass over the weights -- the sequences are the 16 columns of
// v_mfma_f32_16x16x4_f32, which evaluates each column's k-ordered chain exactly (lnb_batch_kernels.h) -- with per-sequence position, RoPE
// row, KV append and attention.  Every sequence's tokens and caches are bit-identical to its single-sequence run.
struct lnb_batch {
    lnb_model* m = nullptr; int n = 0; std::vector<lnb_ctx*> ctxs;
    hipStream_t stream = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    BatchTab* tab = nullptr; BatchKV* kv = nullptr;
    uint16_t *x = nullptr, *h = nullptr, *xt = nullptr, *q = nullptr, *att_xt = nullptr, *ffn_xt = nullptr, *logits = nullptr;
    int* derr = nullptr; int32_t *d_tokens = nullptr, *d_pos = nullptr; int32_t* h_io = nullptr;   // pinned: [0..MAX) tokens, [MAX..2 MAX) positions, [2 MAX] error word (MAX = LNB_BATCH_MAX)
    hipGraphExec_t graph = nullptr; int lds_T = 0; bool counted = false;
    bool rows_form = false;                // the model carries no matrix-core copy: every product runs as ROWS of gemm_stream_kernel on the resident layouts, whatever n
    // pipeline stage (lnb_pipeline_tick_batch): the contiguous token words exchanged between the last and the first stage, the stage step
    // as a captured graph, events towards / from the exchange stream (as lnb_ctx has them for single-sequence ticks)
    int32_t* ring = nullptr; hipGraphExec_t stage_graph = nullptr;
    hipEvent_t ev_done = nullptr, ev_in = nullptr, ev_sent = nullptr; bool in_pending = false, sent_pending = false, recv_unmatched = false;
};
static int m16_copy(lnb_model* m, const TiledDesc& t, int rows, uint16_t** out) {
    const size_t bytes = m16_elems(rows, t.k, t.nch) * 2;
    HIPCHK(hipMalloc((void**)out, bytes));
    HIPCHK(hipMemsetAsync(*out, 0, bytes, m->stream));
    HIPCHK(lnbk_m16_from_tiled(t.w, *out, rows, t.k, t.rw, t.nch, m->stream));
    m->batch_bytes += (int64_t)bytes;
    return 0;
}
extern "C" int lnb_model_enable_batch(lnb_model* m) {
    if (!m) return fail("null argument");
    if (!m->finalized) return fail("model not finalized");
    if (m->batch_enabled) return 0;
    if (m->part_begin % 3 || m->part_end % 3) return fail("batched decode needs a stage of whole blocks (this one is cut inside a block: parts [%d, %d))", m->part_begin, m->part_end);
    if (m->a.dim % 128 || m->q_dim % 128 || m->ffn_hidden % 128)
        return fail("batched decode streams the weights in 128-step chunks: dim (%d), n_heads*head_dim (%d) and the FFN hidden size (%d) must be multiples of 128", m->a.dim, m->q_dim, m->ffn_hidden);
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(lnbk_batch_prepare());
    int rc = 0;
    for (auto& L : m->layers) {
        rc |= m16_copy(m, L.wqkv, L.wqkv.n_rows, &L.m_wqkv) | m16_copy(m, L.wo, m->a.dim, &L.m_wo) | m16_copy(m, L.w13, m->ffn_hidden, &L.m_w13) | m16_copy(m, L.w2, m->a.dim, &L.m_w2);
        if (rc) break;
    }
    if (!rc && m->last()) rc = m16_copy(m, m->output, m->a.vocab_size, &m->m_output);
    hipError_t e = hipStreamSynchronize(m->stream);
    if (!rc && e == hipSuccess) {                            // the prefill products find the copy through the matrix descriptors
        for (auto& L : m->layers) { L.wqkv.w16 = L.m_wqkv; L.wo.w16 = L.m_wo; L.w13.w16 = L.m_w13; L.w2.w16 = L.m_w2; }
        m->output.w16 = m->m_output;
    }
    if (rc || e != hipSuccess) {                             // (out of memory on a model that fills the HBM: the single-sequence paths stay usable)
        for (auto& L : m->layers) { hipFree(L.m_wqkv); hipFree(L.m_wo); hipFree(L.m_w13); hipFree(L.m_w2); L.m_wqkv = L.m_wo = L.m_w13 = L.m_w2 = nullptr; }
        hipFree(m->m_output); m->m_output = nullptr; m->batch_bytes = 0;
        (void)hipGetLastError();                             // the failed hipMalloc leaves the thread's last-error set: every launcher ends with hipGetLastError()
        if (!rc) return fail("lnb_model_enable_batch: %s", hipGetErrorString(e));
        return -1;
    }
    m->batch_enabled = true;
    return 0;
}
extern "C" int64_t lnb_model_batch_bytes(lnb_model* m) { return m ? m->batch_bytes : 0; }

extern "C" int lnb_batch_destroy(lnb_batch* b) {
    if (!b) return 0;
    hipSetDevice(b->m->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    if (b->counted) for (lnb_ctx* c : b->ctxs) c->batch_users--;
    if (b->graph) hipGraphExecDestroy(b->graph);
    hipFree(b->tab); hipFree(b->kv); hipFree(b->x); hipFree(b->h); hipFree(b->xt); hipFree(b->q); hipFree(b->att_xt); hipFree(b->ffn_xt); hipFree(b->logits);
    hipFree(b->derr); hipFree(b->d_tokens); hipFree(b->d_pos); hipFree(b->ring);
    if (b->stage_graph) hipGraphExecDestroy(b->stage_graph);
    if (b->ev_done) hipEventDestroy(b->ev_done);
    if (b->ev_in) hipEventDestroy(b->ev_in);
    if (b->ev_sent) hipEventDestroy(b->ev_sent);
    if (b->h_io) hipHostFree(b->h_io);
    if (b->ev0) hipEventDestroy(b->ev0);
    if (b->ev1) hipEventDestroy(b->ev1);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
    return 0;
}
static int batch_alloc(lnb_batch* b) {
    lnb_model* m = b->m; const int n = b->n;
    HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&b->ev0)); HIPCHK(hipEventCreate(&b->ev1));
    HIPCHK(hipHostMalloc((void**)&b->h_io, (2 * LNB_BATCH_MAX + 8) * 4, hipHostMallocDefault));
    BatchTab t{}; t.n = n;
    std::vector<BatchKV> kv(m->layers.size());
    for (int s = 0; s < LNB_BATCH_MAX; s++) {
        lnb_ctx* c = b->ctxs[s < n ? s : 0];                // (unused columns point at sequence 0's words: never dereferenced, never null)
        t.st[s] = c->st; t.dtok[s] = c->dtok; t.dout[s] = c->dout; t.dout_cap[s] = c->dout_cap; t.seq_len[s] = c->seq_len;
        for (size_t l = 0; l < m->layers.size(); l++) { kv[l].ck[s] = c->ck[l]; kv[l].cv[s] = c->cv[l]; }
    }
    HIPCHK(hipMalloc((void**)&b->tab, sizeof t)); HIPCHK(hipMemcpyAsync(b->tab, &t, sizeof t, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMalloc((void**)&b->kv, kv.size() * sizeof(BatchKV)));
    HIPCHK(hipMemcpyAsync(b->kv, kv.data(), kv.size() * sizeof(BatchKV), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));                 // (the host copies above are stack / vector memory)
    const size_t N = (size_t)((std::max(n, LNB_STREAM_COLS) + 15) / 16) * 16, dim = m->a.dim;      // (whole column groups of 16: the 17..32-sequence form keeps B-operand layouts per group)
    auto zalloc = [&](uint16_t** p, size_t elems) -> int { HIPCHK(hipMalloc((void**)p, elems * 2)); HIPCHK(hipMemsetAsync(*p, 0, elems * 2, b->stream)); return 0; };
    // 1..16 sequences: activations in the B-operand layout [K][16 sequences], the columns past n stay zero for ever; 17..128: plain rows [n][K]
    if (zalloc(&b->x, N * dim) || zalloc(&b->h, N * dim) || zalloc(&b->xt, N * dim) || zalloc(&b->q, N * m->q_dim) || zalloc(&b->att_xt, N * m->q_dim) ||
        zalloc(&b->ffn_xt, N * m->ffn_hidden) || (m->last() && zalloc(&b->logits, N * (size_t)m->a.vocab_size))) return -1;
    HIPCHK(hipMalloc((void**)&b->derr, 16)); HIPCHK(hipMemsetAsync(b->derr, 0, 16, b->stream));
    HIPCHK(hipMalloc((void**)&b->d_tokens, LNB_BATCH_MAX * 4)); HIPCHK(hipMalloc((void**)&b->d_pos, LNB_BATCH_MAX * 4));
    HIPCHK(hipMalloc((void**)&b->ring, LNB_BATCH_MAX * 4)); HIPCHK(hipMemsetAsync(b->ring, 0, LNB_BATCH_MAX * 4, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
extern "C" int lnb_batch_create(lnb_ctx* const* ctxs, int n, lnb_batch** out) {
    if (!ctxs || !out) return fail("null argument");
    *out = nullptr;
    if (n < 1 || n > LNB_BATCH_MAX) return fail("a batch holds 1..%d sequences (got %d)", LNB_BATCH_MAX, n);
    for (int s = 0; s < n; s++) {
        if (!ctxs[s]) return fail("null context at index %d", s);
        if (ctxs[s]->m != ctxs[0]->m) return fail("context %d belongs to another model handle", s);
        for (int r = 0; r < s; r++) if (ctxs[r] == ctxs[s]) return fail("context %d appears twice in the batch", s);
        if (ctxs[s]->mode != LNB_MODE_EXACT) return fail("context %d is in the tolerance mode: batched decode is exact-order only", s);
    }
    lnb_model* m = ctxs[0]->m;
    HIPCHK(hipSetDevice(m->device));
    if (!m->batch_enabled) {
        // Round 5: a batch on a model WITHOUT the matrix-core copy runs every product as rows of the streaming product, which reads the RESIDENT
        // weight layouts (gemm_stream_kernel, SRC 1 / 2) -- any number of sequences.  The copy (lnb_model_enable_batch) is a pure performance
        // option: the column forms of up to 32 sequences (mfma_stream_kernel / mfma_pair_kernel) read it and are faster there.
        if (m->part_begin % 3 || m->part_end % 3) return fail("batched decode needs a stage of whole blocks (this one is cut inside a block: parts [%d, %d))", m->part_begin, m->part_end);
        if (m->a.dim % 128 || m->q_dim % 128 || m->ffn_hidden % 128)
            return fail("batched decode streams the weights in 128-step chunks: dim (%d), n_heads*head_dim (%d) and the FFN hidden size (%d) must be multiples of 128", m->a.dim, m->q_dim, m->ffn_hidden);
        HIPCHK(lnbk_batch_prepare());
    }
    lnb_batch* b = new lnb_batch();
    b->m = m; b->n = n; b->ctxs.assign(ctxs, ctxs + n); b->rows_form = !m->batch_enabled;
    for (int s = 0; s < n; s++) {
        if (ctxs[s]->seq_len > ctxs[s]->attn_short_cap) {
            const int sl = ctxs[s]->seq_len, cap = ctxs[s]->attn_short_cap; delete b;
            return fail("context %d: seq_len %d is beyond the %d positions the batched attention stages in the LDS", s, sl, cap);
        }
        b->lds_T = std::max(b->lds_T, ctxs[s]->seq_len);
    }
    if (batch_alloc(b)) { lnb_batch_destroy(b); return -1; }
    for (lnb_ctx* c : b->ctxs) c->batch_users++;
    b->counted = true;
    *out = b;
    return 0;
}
static StreamParams stream_of(const lnb_batch* b, const uint16_t* w, const uint16_t* xt, int K, int n_rows, int nch) {
    StreamParams p{}; p.w = w; p.xt = xt; p.K = K; p.n_rows = n_rows; p.nch = nch; p.n_chains = ((n_rows + 15) / 16) * nch; p.nseq = b->n; p.dbg = g_dbg;
    return p;
}
static bool stream_acc2(const StreamParams& p) { return p.nch == 2 || p.n_chains > 4 * g_num_cus; }   // thin matrices: one tile per wave, every tile on its own SIMD
// More than 16 sequences: the batch's rows through the prefill's streaming product (gemm_stream_kernel: weights M16 -> A operand, 1 / 2 / 4
// batch tiles of 16 sequences per wave), plain row-major activations.  Same chains per sequence; EPI_QKV_ROPE and the attention take each
// row's position and caches from the batch tables.  (xt / att_xt / ffn_xt hold rows here, not the B-operand layout.)
// (w16 == nullptr -- no matrix-core copy: the resident layout t feeds the same kernel, gemm_stream_kernel SRC 1 / 2)
static GemmParams wide_of(const lnb_batch* b, const TiledDesc& t, const uint16_t* w16, const uint16_t* x, int K, int n_rows, int nch) {
    GemmParams g{}; g.w16 = w16; g.w = t.w; g.rw = t.rw; g.nch = nch; g.x = x; g.K = K; g.n_rows = n_rows; g.S = b->n;
    return g;
}
// 17 .. 32 sequences (LNB_BATCH_GROUPS=0: off): the thin matrices -- one k-ordered chain per 16-row tile and 16 columns -- as TWO column groups of
// mfma_pair_kernel on disjoint CUs (256-384 tiles x 2 groups: every CU carries two chains) instead of rows of gemm_stream_kernel (one chain per
// wave, its operands unpacked by the same wave); the fat matrices stay rows.  The activations between them change layout at their producers:
// norm -> xt groups (batch_rmsnorm_xt_kernel), attention -> out_xt groups, SiLU*up epilogue -> out_xt groups.
static bool batch_groups(const lnb_batch* b) {
    static const int on = env_int("LNB_BATCH_GROUPS", 1);
    return on && !b->rows_form && b->n > LNB_STREAM_COLS && b->n <= 2 * LNB_STREAM_COLS;
}
static int enqueue_batch_kernel_wide(lnb_batch* b, int l, int which) {
    lnb_model* m = b->m; const lnb_model_args& a = m->a; hipStream_t st = b->stream;
    const int n = b->n, dim = a.dim, F = m->ffn_hidden;
    if (batch_groups(b) && which != K_HEAD) {
        LayerW& L = m->layers[l - m->layer_begin];
        const int G = (n + LNB_STREAM_COLS - 1) / LNB_STREAM_COLS;
        auto pair_of = [&](const uint16_t* w, const uint16_t* xt, int K, int n_rows) {
            StreamParams p{}; p.w = w; p.xt = xt; p.K = K; p.n_rows = n_rows; p.nch = 1; p.n_chains = (n_rows + 15) / 16; p.nseq = n; p.n_groups = G; p.dbg = nullptr;
            return p;
        };
        switch (which) {
        case K_QKV: {
            HIPCHK(lnbk_batch_rmsnorm(b->x, L.attn_norm, a.norm_eps, b->xt, dim, n, st));
            StreamParams p = pair_of(L.m_wqkv, b->xt, dim, L.wqkv.n_rows);
            p.cis = m->cis; p.q_out = b->q; p.tab = b->tab; p.kv = b->kv + (l - m->layer_begin); p.q_dim = m->q_dim; p.kv_dim = m->kv_dim; p.head_dim = m->head_dim;
            HIPCHK(lnbk_stream(&p, EPI_QKV_ROPE, 0, g_num_cus, st)); return 0; }
        case K_ATTN: {
            AttnParams ap{}; ap.q = b->q; ap.out_xt = b->att_xt; ap.btab = b->tab; ap.bkv = b->kv + (l - m->layer_begin); ap.dbg = nullptr;
            ap.S = n; ap.H = a.n_heads; ap.KVH = a.n_kv_heads; ap.hd = m->head_dim; ap.seq_len = b->lds_T; ap.lds_T = b->lds_T; ap.host_T = 0;
            ap.divisor = bf_wide_h(bf_trunc_h((float)std::sqrt((double)m->head_dim)));
            ap.force_zseq = 0; ap.zseq_count = b->ctxs[0]->zseq_count; ap.exp_tab = m->exp_tab;
            HIPCHK(lnbk_attn(&ap, st)); return 0; }
        case K_WO: {
            StreamParams p = pair_of(L.m_wo, b->att_xt, m->q_dim, dim); p.out = b->h; p.res = b->x;
            HIPCHK(lnbk_stream(&p, EPI_RESID, 0, g_num_cus, st)); return 0; }
        case K_W13: {
            HIPCHK(lnbk_rmsnorm_rows(b->h, L.ffn_norm, b->xt, n, dim, a.norm_eps, st));
            GemmParams g = wide_of(b, L.w13, L.m_w13, b->xt, dim, F, 2); g.out = nullptr; g.out_xt = b->ffn_xt; g.silu = m->silu;
            HIPCHK(lnbk_gemm_stream(&g, EPI_SILU_MUL, g_num_cus, st)); return 0; }
        case K_W2: {
            StreamParams p = pair_of(L.m_w2, b->ffn_xt, F, dim); p.out = b->x; p.res = b->h;
            HIPCHK(lnbk_stream(&p, EPI_RESID, 0, g_num_cus, st)); return 0; }
        }
        return fail("bad kernel id");
    }
    if (which == K_HEAD) {
        HIPCHK(lnbk_rmsnorm_rows(b->x, m->norm, b->xt, n, dim, a.norm_eps, st));
        GemmParams g = wide_of(b, m->output, m->m_output, b->xt, dim, a.vocab_size, 1); g.out = b->logits;
        HIPCHK(lnbk_gemm_stream(&g, EPI_STORE, g_num_cus, st));
        return 0;
    }
    LayerW& L = m->layers[l - m->layer_begin];
    switch (which) {
    case K_QKV: {
        HIPCHK(lnbk_rmsnorm_rows(b->x, L.attn_norm, b->xt, n, dim, a.norm_eps, st));
        GemmParams g = wide_of(b, L.wqkv, L.m_wqkv, b->xt, dim, L.wqkv.n_rows, 1);
        g.cis = m->cis; g.q_out = b->q; g.btab = b->tab; g.bkv = b->kv + (l - m->layer_begin); g.q_dim = m->q_dim; g.kv_dim = m->kv_dim; g.head_dim = m->head_dim;
        HIPCHK(lnbk_gemm_stream(&g, EPI_QKV_ROPE, g_num_cus, st)); return 0; }
    case K_ATTN: {
        AttnParams ap{}; ap.q = b->q; ap.out = b->att_xt; ap.out_xt = nullptr; ap.btab = b->tab; ap.bkv = b->kv + (l - m->layer_begin); ap.dbg = nullptr;
        ap.S = n; ap.H = a.n_heads; ap.KVH = a.n_kv_heads; ap.hd = m->head_dim; ap.seq_len = b->lds_T; ap.lds_T = b->lds_T; ap.host_T = 0;
        ap.divisor = bf_wide_h(bf_trunc_h((float)std::sqrt((double)m->head_dim)));
        ap.force_zseq = 0; ap.zseq_count = b->ctxs[0]->zseq_count; ap.exp_tab = m->exp_tab;     // (attn_gqa_kernel looks exp up)
        HIPCHK(lnbk_attn(&ap, st)); return 0; }
    case K_WO: {
        GemmParams g = wide_of(b, L.wo, L.m_wo, b->att_xt, m->q_dim, dim, 1); g.out = b->h; g.res = b->x;
        HIPCHK(lnbk_gemm_stream(&g, EPI_RESID, g_num_cus, st)); return 0; }
    case K_W13: {
        HIPCHK(lnbk_rmsnorm_rows(b->h, L.ffn_norm, b->xt, n, dim, a.norm_eps, st));
        GemmParams g = wide_of(b, L.w13, L.m_w13, b->xt, dim, F, 2); g.out = b->ffn_xt; g.silu = m->silu;
        HIPCHK(lnbk_gemm_stream(&g, EPI_SILU_MUL, g_num_cus, st)); return 0; }
    case K_W2: {
        GemmParams g = wide_of(b, L.w2, L.m_w2, b->ffn_xt, F, dim, 1); g.out = b->x; g.res = b->h;
        HIPCHK(lnbk_gemm_stream(&g, EPI_RESID, g_num_cus, st)); return 0; }
    }
    return fail("bad kernel id");
}
// which: K_QKV (attention norm + wq|wk|wv + RoPE + KV append), K_ATTN, K_WO, K_W13 (ffn norm + w1|w3 + SiLU*up), K_W2, K_HEAD (norm + output)
static int enqueue_batch_kernel(lnb_batch* b, int l, int which) {
    if (b->n > LNB_STREAM_COLS || b->rows_form) return enqueue_batch_kernel_wide(b, l, which);
    lnb_model* m = b->m; const lnb_model_args& a = m->a; hipStream_t st = b->stream;
    const int n = b->n, dim = a.dim, F = m->ffn_hidden;
    if (which == K_HEAD) {
        HIPCHK(lnbk_batch_rmsnorm(b->x, m->norm, a.norm_eps, b->xt, dim, n, st));
        StreamParams p = stream_of(b, m->m_output, b->xt, dim, a.vocab_size, 1); p.out = b->logits;
        HIPCHK(lnbk_stream(&p, EPI_STORE, stream_acc2(p), g_num_cus, st));
        return 0;
    }
    LayerW& L = m->layers[l - m->layer_begin];
    switch (which) {
    case K_QKV: {
        HIPCHK(lnbk_batch_rmsnorm(b->x, L.attn_norm, a.norm_eps, b->xt, dim, n, st));
        StreamParams p = stream_of(b, L.m_wqkv, b->xt, dim, L.wqkv.n_rows, 1);
        p.cis = m->cis; p.q_out = b->q; p.tab = b->tab; p.kv = b->kv + (l - m->layer_begin); p.q_dim = m->q_dim; p.kv_dim = m->kv_dim; p.head_dim = m->head_dim;
        HIPCHK(lnbk_stream(&p, EPI_QKV_ROPE, stream_acc2(p), g_num_cus, st)); return 0; }
    case K_ATTN: {
        AttnParams ap{}; ap.q = b->q; ap.out_xt = b->att_xt; ap.btab = b->tab; ap.bkv = b->kv + (l - m->layer_begin); ap.dbg = nullptr;
        ap.S = n; ap.H = a.n_heads; ap.KVH = a.n_kv_heads; ap.hd = m->head_dim; ap.seq_len = b->lds_T; ap.lds_T = b->lds_T; ap.host_T = 0;
        ap.divisor = bf_wide_h(bf_trunc_h((float)std::sqrt((double)m->head_dim)));
        ap.force_zseq = 0; ap.zseq_count = b->ctxs[0]->zseq_count; ap.exp_tab = m->exp_tab;     // (attn_gqa_kernel looks exp up)
        HIPCHK(lnbk_attn(&ap, st)); return 0; }
    case K_WO: {
        StreamParams p = stream_of(b, L.m_wo, b->att_xt, m->q_dim, dim, 1); p.out = b->h; p.res = b->x;
        HIPCHK(lnbk_stream(&p, EPI_RESID, stream_acc2(p), g_num_cus, st)); return 0; }
    case K_W13: {
        HIPCHK(lnbk_batch_rmsnorm(b->h, L.ffn_norm, a.norm_eps, b->xt, dim, n, st));
        StreamParams p = stream_of(b, L.m_w13, b->xt, dim, F, 2); p.out_xt = b->ffn_xt; p.silu = m->silu;
        HIPCHK(lnbk_stream(&p, EPI_SILU_MUL, 1, g_num_cus, st)); return 0; }
    case K_W2: {
        StreamParams p = stream_of(b, L.m_w2, b->ffn_xt, F, dim, 1); p.out = b->x; p.res = b->h;
        HIPCHK(lnbk_stream(&p, EPI_RESID, stream_acc2(p), g_num_cus, st)); return 0; }
    }
    return fail("bad kernel id");
}
// One batched step of this model STAGE: the embedding gather on the first stage (otherwise the hidden states [n, dim] are already in b->x:
// received from the stage before), the owned blocks, and on the last stage norm + output + argmax (which advances every sequence's
// position); a stage without the head advances the positions itself.  ring_in: the first stage of a multi-stage pipeline takes the tokens
// from the contiguous words the last stage sent (lnb_pipeline_tick_batch).
static int enqueue_batch_step(lnb_batch* b, bool ring_in = false) {
    lnb_model* m = b->m;
    if (m->first()) {
        if (ring_in) HIPCHK(lnbk_batch_scatter_ring(b->tab, b->ring, b->stream));
        HIPCHK(lnbk_batch_embed(m->tok_embd, b->tab, b->x, b->n, m->a.dim, m->a.vocab_size, b->derr, b->stream));
    }
    for (int l = m->layer_begin; l < m->layer_end; l++)
        for (int k = K_QKV; k <= K_W2; k++) if (enqueue_batch_kernel(b, l, k)) return -1;
    if (m->last()) {
        if (enqueue_batch_kernel(b, 0, K_HEAD)) return -1;
        HIPCHK(lnbk_batch_argmax(b->logits, m->a.vocab_size, b->tab, b->n, b->ring, b->stream));
    } else HIPCHK(lnbk_batch_advance(b->tab, b->stream));
    return 0;
}
static int batch_decode_impl(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos, int n_steps, int32_t* out_tokens, int32_t* n_generated, int32_t* finished, float* ms_out);
extern "C" int lnb_batch_decode(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos, int n_steps, int32_t* out_tokens, float* ms_out) {
    return batch_decode_impl(b, tokens, start_pos, n_steps, out_tokens, nullptr, nullptr, ms_out);     // n_steps tokens per sequence: stop ids are not compared
}
// with per-sequence stop ids (lnb_ctx_set_stop_ids on the member contexts): n_generated[s] tokens of row s of out_tokens are valid; a finished
// sequence's column keeps computing its last step (the pass over the weights is shared), its state and caches stay where they stopped
// finished[s] (optional) = 1 if a stop id ended sequence s: n_generated[s] == max_steps alone cannot tell a run that stopped on its last step from
// one that did not.  A caller that decodes in chunks passes start_pos[s] < 0 for a sequence that has finished: it stays frozen (nothing of it is
// touched, n_generated[s] = 0, finished[s] = 1) instead of being restarted from its stop token.
extern "C" int lnb_batch_decode_until(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos, int max_steps, int32_t* out_tokens, int32_t* n_generated, int32_t* finished, float* ms_out) {
    if (!n_generated) return fail("null argument");
    return batch_decode_impl(b, tokens, start_pos, max_steps, out_tokens, n_generated, finished, ms_out);
}
static int batch_decode_impl(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos, int n_steps, int32_t* out_tokens, int32_t* n_generated, int32_t* finished, float* ms_out) {
    if (!b || !tokens || !start_pos || !out_tokens) return fail("null argument");
    lnb_model* m = b->m;
    if (!m->first() || !m->last()) return fail("lnb_batch_decode needs a whole-model handle; pipeline stages run their batches through lnb_pipeline_tick_batch");
    HIPCHK(hipSetDevice(m->device));
    if (n_steps <= 0) return fail("n_steps must be positive");
    for (int s = 0; s < b->n; s++) {
        lnb_ctx* c = b->ctxs[s];
        if (n_steps > c->dout_cap) return fail("sequence %d: n_steps %d exceeds the context length %d", s, n_steps, c->dout_cap);
        const bool frozen = start_pos[s] < 0;                 // an ended sequence of a chunked run (lnb_batch_decode_until): nothing of it is touched
        if (frozen && !n_generated) return fail("sequence %d: negative start position (only lnb_batch_decode_until takes one, for a sequence that has finished)", s);
        if (!frozen && (check_call(c, 1, start_pos[s]) || check_call(c, 1, start_pos[s] + n_steps - 1))) return -1;
        if (!frozen && (tokens[s] < 0 || tokens[s] >= m->a.vocab_size)) return fail("sequence %d: token id at index 0 is outside the vocabulary", s);
        if (c->pending) return fail("sequence %d: a lnb_forward_stage_begin has not been ended", s);
        HIPCHK(hipStreamSynchronize(c->stream));             // whatever the context's own stream still does to its caches comes first
        c->dev_pos = -1; c->call_T = 0;                      // the batch advances the context's device-side position by itself
    }
    hipStream_t st = b->stream;
    const bool use_graph = env_int("LNB_NO_GRAPH", 0) == 0;
    if (use_graph && !b->graph) {
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_batch_step(b);
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) { if (g) hipGraphDestroy(g); return -1; }
        HIPCHK(e);
        HIPCHK(hipGraphInstantiate(&b->graph, g, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(g));
    }
    memcpy(b->h_io, tokens, (size_t)b->n * 4); memcpy(b->h_io + LNB_BATCH_MAX, start_pos, (size_t)b->n * 4);
    HIPCHK(hipMemcpyAsync(b->d_tokens, b->h_io, (size_t)b->n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(b->d_pos, b->h_io + LNB_BATCH_MAX, (size_t)b->n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(b->derr, 0, 4, st));
    HIPCHK(lnbk_batch_set_state(b->tab, b->d_tokens, b->d_pos, b->ring, n_generated ? 1 : 0, st));
    HIPCHK(hipEventRecord(b->ev0, st));
    for (int i = 0; i < n_steps; i++) {
        if (use_graph) HIPCHK(hipGraphLaunch(b->graph, st));
        else if (enqueue_batch_step(b)) return -1;
    }
    HIPCHK(hipEventRecord(b->ev1, st));
    for (int s = 0; s < b->n; s++)
        HIPCHK(hipMemcpyAsync(out_tokens + (size_t)s * n_steps, b->ctxs[s]->dout, (size_t)n_steps * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_io + 2 * LNB_BATCH_MAX, b->derr, 4, hipMemcpyDeviceToHost, st));
    std::vector<StepState> hs(n_generated ? b->n : 0);
    for (int s = 0; s < (int)hs.size(); s++) HIPCHK(hipMemcpyAsync(&hs[s], b->ctxs[s]->st, sizeof(StepState), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (ms_out) HIPCHK(hipEventElapsedTime(ms_out, b->ev0, b->ev1));
    if (b->h_io[2 * LNB_BATCH_MAX]) return fail("sequence %d: generated token id is outside the vocabulary", b->h_io[2 * LNB_BATCH_MAX] - 1);
    for (int s = 0; s < (int)hs.size(); s++) { n_generated[s] = hs[s].n_out; if (finished) finished[s] = hs[s].finished; }
    return 0;
}
// what a batched tick exchanges, for a host layer that moves it itself (lnb_pipeline_init_host)
extern "C" void* lnb_batch_boundary_ptr(lnb_batch* b, int which) {
    if (!b) { fail("null argument"); return nullptr; }
    if (which == 0) return b->x;
    if (which == 1) return b->ring;
    fail("lnb_batch_boundary_ptr: which = %d (0: hidden states [n, dim] bf16, 1: the n token words)", which);
    return nullptr;
}
extern "C" int lnb_batch_check_error(lnb_batch* b) {
    if (!b) return fail("null argument");
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    int h = 0;
    HIPCHK(hipMemcpy(&h, b->derr, 4, hipMemcpyDeviceToHost));
    if (h) {
        HIPCHK(hipMemset(b->derr, 0, 4));
        return fail("batched pipeline step: a token id outside the vocabulary (or an all-NaN logits row) in one of the batch's sequences since the last check");
    }
    return 0;
}
// Every sequence's position (and, on the first stage, optionally its next input token) before a run of lnb_pipeline_tick_batch steps: the
// positions then advance on the device with every step.  tokens == NULL keeps what each context's token word holds (what the
// single-sequence prefill ticks left there).
extern "C" int lnb_batch_set_state(lnb_batch* b, const int32_t* tokens, const int32_t* start_pos) {
    if (!b || !start_pos) return fail("null argument");
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipMemsetAsync(b->derr, 0, 4, b->stream));        // a fresh run: forget what an earlier one latched
    const bool one_stage = b->m->first() && b->m->last();
    for (int s = 0; s < b->n; s++) {
        lnb_ctx* c = b->ctxs[s];
        if (check_call(c, 1, start_pos[s])) return -1;
        if (tokens && (tokens[s] < 0 || tokens[s] >= b->m->a.vocab_size)) return fail("sequence %d: token id at index 0 is outside the vocabulary", s);
        // only the LAST stage's argmax sees a stop id: the other stages would go on advancing their positions, re-embedding the stale ring token
        // and appending KV rows behind the stop.  Stop ids in a multi-stage batched pipeline belong to the host: it reads the tokens.
        if (c->n_stop > 0 && !one_stage)
            return fail("sequence %d: its context carries %d stop ids, which a batched tick of a multi-stage pipeline cannot honour (only the last stage sees the "
                        "token): clear them (lnb_ctx_set_stop_ids(ctx, NULL, 0)) and end the sequence on the host", s, c->n_stop);
        c->dev_pos = -1; c->call_T = 0;
    }
    HIPCHK(hipDeviceSynchronize());                          // a setup call: whatever the contexts' streams and the pipe's exchange stream still do
                                                             // (the prefill's token hand-off into the contexts' token words) comes first
    if (tokens) memcpy(b->h_io, tokens, (size_t)b->n * 4);
    memcpy(b->h_io + LNB_BATCH_MAX, start_pos, (size_t)b->n * 4);
    if (tokens) HIPCHK(hipMemcpyAsync(b->d_tokens, b->h_io, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_pos, b->h_io + LNB_BATCH_MAX, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(lnbk_batch_set_state(b->tab, tokens ? b->d_tokens : nullptr, b->d_pos, b->ring, one_stage ? 1 : 0, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
// measurement aid (bench.py): average HIP-event time of ONE kernel class of the batched step (which as in lnb_profile_kernel; the norm
// launches count with the product they feed), consecutive launches cycling through the layers; every sequence is placed at `pos`
extern "C" int lnb_batch_profile_kernel(lnb_batch* b, int which, int pos, int iters, float* avg_ms_out) {
    if (!b || !avg_ms_out) return fail("null argument");
    lnb_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    if (iters <= 0 || which < 0 || which > K_LAYER) return fail("bad arguments");
    if (which == K_HEAD && !m->last()) return fail("this stage does not own output.weight");
    for (int s = 0; s < b->n; s++) { if (check_call(b->ctxs[s], 1, pos)) return -1; b->h_io[s] = 0; b->h_io[LNB_BATCH_MAX + s] = pos; b->ctxs[s]->dev_pos = -1; }
    hipStream_t st = b->stream;
    HIPCHK(hipMemcpyAsync(b->d_tokens, b->h_io, (size_t)b->n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(b->d_pos, b->h_io + LNB_BATCH_MAX, (size_t)b->n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(lnbk_batch_set_state(b->tab, b->d_tokens, b->d_pos, b->ring, 0, st));
    const int nl = m->layer_end - m->layer_begin;
    auto run = [&](int i) -> int {
        const int l = m->layer_begin + i % nl;
        if (which == K_HEAD) return enqueue_batch_kernel(b, 0, K_HEAD);
        if (which == K_LAYER) { for (int k = K_QKV; k <= K_W2; k++) if (enqueue_batch_kernel(b, l, k)) return -1; return 0; }
        return enqueue_batch_kernel(b, l, which);
    };
    for (int i = 0; i < 3; i++) if (run(i)) return -1;
    HIPCHK(hipEventRecord(b->ev0, st));
    for (int i = 0; i < iters; i++) if (run(i + 3)) return -1;
    HIPCHK(hipEventRecord(b->ev1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *avg_ms_out = ms / (float)iters;
    return 0;
}

// ---- layer-sharded pipeline: the exchange behind the C ABI ---------------------------------------------------------------------
// The reference runs its 32 blocks in one loop (llamatransformer.go:156-164); a pipeline cuts that loop over the GPUs of a node.  Rank r
// holds a stage (lnb_model_create_parts) and one lnb_ctx per sequence in flight; a TICK enqueues, without ever blocking the host:
//   * the stage step of one sequence on that context's stream (one-token steps replay a captured hipGraph; the position lives on the
//     device), gated by the event of the receive that delivered its input;
//   * ONE RCCL group on the pipe's exchange stream: ncclSend of the previous result to rank r+1 straight from the context's hidden
//     state buffer (the last rank sends the 4-byte argmax token to rank 0 straight from device memory) and ncclRecv of the next input
//     from rank r-1 straight into the target context's buffer (rank 0: the token, into the device word the embedding gather reads).
// No staging copies, no host round trip for the token ring, no stream synchronisation per tick: the host only enqueues, and reads the
// generated tokens from a pinned log after lnb_pipeline_sync.  RCCL is loaded on first use (lnb_pipeline.cpp).
// In-process transport (lnb_pipeline_init_loopback): every stage of a pipeline lives in ONE process (all on one GPU, or one per GPU) and a
// "send" meets its "receive" in a mailbox; whichever side is posted second enqueues the device copy.  Same tick code, same events, same
// graphs as the RCCL transport -- what it replaces is only ncclSend / ncclRecv.  Used to test host schedules where RCCL cannot run
// (two ranks need two GPUs) and by single-process multi-GPU hosts.
struct LoopMsg { lnb_ctx* ctx; int rows; lnb_batch* bat = nullptr; };   // a single-sequence hand-off (ctx) or a batch's (bat)
struct LoopGroup { std::map<std::pair<int, int>, std::deque<LoopMsg>> sends, recvs; int users = 0; int device = -1; };
static std::map<std::string, LoopGroup> g_loops;
static std::mutex g_loops_mu;

struct lnb_pipe {
    lnb_model* m = nullptr; int rank = 0, world = 1;
    LoopGroup* loop = nullptr; std::string loop_tag;
    void* comm = nullptr; const lnb_rccl_api* api = nullptr;
    hipStream_t xs = nullptr;              // exchange stream
    int32_t* h_tok = nullptr; int tok_cap = 0, tok_n = 0;   // pinned RING of the tokens the last stage produced, in tick order: slot s lives at s % tok_cap
    bool use_graph = true;
    bool host = false;                     // no transport: the host layer moves the boundary buffers (lnb_pipeline_init_host)
};
static int pipe_log_cap() { const int v = env_int("LNB_PIPELINE_LOG_CAP", 1 << 16); return v < 4 ? 4 : v; }   // (the env knob is for the wrap-around test)
#define NCCLCHK(p_, expr) do { int r_ = (expr); if (r_ != 0) return fail("%s failed: %s (%s:%d)", #expr, (p_)->api->GetErrorString(r_), __FILE__, __LINE__); } while (0)

extern "C" int lnb_pipeline_unique_id(void* id128) {
    if (!id128) return fail("null argument");
    const lnb_rccl_api* api = lnb_rccl_load();
    if (!api) return -1;
    lnb_nccl_id id; memset(&id, 0, sizeof id);
    int r = api->GetUniqueId(&id);
    if (r != 0) return fail("ncclGetUniqueId failed: %s", api->GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return 0;
}
extern "C" int lnb_pipeline_init(lnb_model* m, int rank, int world, const void* id128, lnb_pipe** out) {
    if (!m || !out) return fail("null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail("rank %d out of range for %d pipeline stages", rank, world);
    if (!m->finalized) return fail("model not finalized");
    if ((rank == 0) != m->first()) return fail("pipeline rank %d: only the first stage owns tok_embeddings (this stage starts at block part %d)", rank, m->part_begin);
    if ((rank == world - 1) != m->last()) return fail("pipeline rank %d of %d: only the last stage owns norm + output (this stage ends at block part %d)", rank, world, m->part_end);
    if (world > 1 && !id128) return fail("null unique id");
    HIPCHK(hipSetDevice(m->device));
    lnb_pipe* p = new lnb_pipe();
    p->m = m; p->rank = rank; p->world = world; p->use_graph = env_int("LNB_PIPELINE_GRAPH", 1) != 0;
    if (world > 1) {
        p->api = lnb_rccl_load();
        if (!p->api) { delete p; return -1; }
        lnb_nccl_id id; memcpy(&id, id128, sizeof id);
        int r = p->api->CommInitRank(&p->comm, world, id, rank);
        if (r != 0) { fail("ncclCommInitRank failed: %s", p->api->GetErrorString(r)); delete p; return -1; }
    }
    hipError_t e = hipStreamCreateWithFlags(&p->xs, hipStreamNonBlocking);
    if (e == hipSuccess) { p->tok_cap = pipe_log_cap(); e = hipHostMalloc((void**)&p->h_tok, (size_t)p->tok_cap * 4, hipHostMallocDefault); }
    if (e != hipSuccess) { fail("pipeline init: %s", hipGetErrorString(e)); if (p->comm) p->api->CommDestroy(p->comm); if (p->xs) hipStreamDestroy(p->xs); delete p; return -1; }
    *out = p;
    return 0;
}
extern "C" int lnb_pipeline_init_loopback(lnb_model* m, int rank, int world, const char* group, lnb_pipe** out) {
    if (!m || !out || !group) return fail("null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail("rank %d out of range for %d pipeline stages", rank, world);
    if (!m->finalized) return fail("model not finalized");
    if ((rank == 0) != m->first()) return fail("pipeline rank %d: only the first stage owns tok_embeddings (this stage starts at block part %d)", rank, m->part_begin);
    if ((rank == world - 1) != m->last()) return fail("pipeline rank %d of %d: only the last stage owns norm + output (this stage ends at block part %d)", rank, world, m->part_end);
    HIPCHK(hipSetDevice(m->device));
    lnb_pipe* p = new lnb_pipe();
    p->m = m; p->rank = rank; p->world = world; p->use_graph = env_int("LNB_PIPELINE_GRAPH", 1) != 0;
    hipError_t e = hipStreamCreateWithFlags(&p->xs, hipStreamNonBlocking);
    if (e == hipSuccess) { p->tok_cap = pipe_log_cap(); e = hipHostMalloc((void**)&p->h_tok, (size_t)p->tok_cap * 4, hipHostMallocDefault); }
    if (e != hipSuccess) { fail("pipeline init: %s", hipGetErrorString(e)); if (p->xs) hipStreamDestroy(p->xs); delete p; return -1; }
    if (world > 1) {
        // The mailbox transport records a context's events on whichever pipe's exchange stream posts second, and a context's flags are
        // plain fields: every stage of a group must live on ONE device and be ticked from ONE host thread in lock-step order (what the
        // tests and single-process hosts do).  One stage per GPU is the RCCL transport's job (lnb_pipeline_init).
        std::lock_guard<std::mutex> lock(g_loops_mu);
        LoopGroup& g = g_loops[group];
        if (g.users > 0 && g.device != m->device) {
            const int have = g.device;
            hipStreamDestroy(p->xs); hipHostFree(p->h_tok); delete p;
            return fail("in-process pipeline group \"%s\" lives on device %d: a stage on device %d cannot join it (use lnb_pipeline_init for one stage per GPU)", group, have, m->device);
        }
        g.device = m->device;
        p->loop_tag = group; p->loop = &g; p->loop->users++;
    }
    *out = p;
    return 0;
}
// The pipe WITHOUT a transport: stage steps, graphs, positions and the token log as in the other two, but what crosses the stage boundary is
// moved by the host layer (pipeline.py's torch.distributed fallback: staging tensors + batch_isend_irecv) between lnb_pipeline_sync calls.
extern "C" int lnb_pipeline_init_host(lnb_model* m, int rank, int world, lnb_pipe** out) {
    if (!m || !out) return fail("null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail("rank %d out of range for %d pipeline stages", rank, world);
    if (!m->finalized) return fail("model not finalized");
    if ((rank == 0) != m->first()) return fail("pipeline rank %d: only the first stage owns tok_embeddings (this stage starts at block part %d)", rank, m->part_begin);
    if ((rank == world - 1) != m->last()) return fail("pipeline rank %d of %d: only the last stage owns norm + output (this stage ends at block part %d)", rank, world, m->part_end);
    HIPCHK(hipSetDevice(m->device));
    lnb_pipe* p = new lnb_pipe();
    p->m = m; p->rank = rank; p->world = world; p->host = true; p->use_graph = env_int("LNB_PIPELINE_GRAPH", 1) != 0;
    hipError_t e = hipStreamCreateWithFlags(&p->xs, hipStreamNonBlocking);
    if (e == hipSuccess) { p->tok_cap = pipe_log_cap(); e = hipHostMalloc((void**)&p->h_tok, (size_t)p->tok_cap * 4, hipHostMallocDefault); }
    if (e != hipSuccess) { fail("pipeline init: %s", hipGetErrorString(e)); if (p->xs) hipStreamDestroy(p->xs); delete p; return -1; }
    *out = p;
    return 0;
}
extern "C" int lnb_pipeline_destroy(lnb_pipe* p) {
    if (!p) return 0;
    hipSetDevice(p->m->device);
    hipDeviceSynchronize();
    if (p->loop) { std::lock_guard<std::mutex> lock(g_loops_mu); if (--p->loop->users == 0) g_loops.erase(p->loop_tag); }
    if (p->comm) p->api->CommDestroy(p->comm);
    if (p->xs) hipStreamDestroy(p->xs);
    if (p->h_tok) hipHostFree(p->h_tok);
    delete p;
    return 0;
}
static int pipe_events(lnb_ctx* c) {
    if (!c->ev_done) { HIPCHK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
                       HIPCHK(hipEventCreateWithFlags(&c->ev_sent, hipEventDisableTiming)); }
    return 0;
}
// one stage step on the context's stream; tokens != NULL: host tokens (prefill on rank 0); NULL on rank 0: the token already sits in
// c->dtok (received from the last rank).  The last stage leaves the argmax token in c->dnext.
static int enqueue_stage_step(lnb_pipe* p, lnb_ctx* c, const int32_t* tokens, int rows, int pos) {
    lnb_model* m = c->m; const int V = m->a.vocab_size; hipStream_t st = c->stream;
    const bool longctx = want_long_attention(c, rows, pos);
    auto body = [&](bool host_tokens) -> int {
        if (m->first()) {
            if (host_tokens) {                               // a second multi-row tick (chunked prefill) must not overwrite words a queued copy still reads
                if (staging_acquire(c)) return -1;
                memcpy(c->h_io + 2, tokens, (size_t)rows * 4);
                HIPCHK(hipMemcpyAsync(c->dtok, c->h_io + 2, (size_t)rows * 4, hipMemcpyHostToDevice, st));
                if (staging_release(c)) return -1;
            }
            HIPCHK(lnbk_embed(m->tok_embd, c->dtok, c->x, rows, m->a.dim, V, c->derr, st));   // Fwd_Get_Rows :118
        }
        if (enqueue_layers(c, rows, false)) return -1;
        if (m->last()) {
            if (enqueue_head(c, rows - 1, 1)) return -1;
            HIPCHK(lnbk_argmax(c->logits, V, c->dnext, c->st, c->dout, c->dout_cap, 0, st));   // inference.go:207-211
        }
        return 0;
    };
    c->attn_long = longctx;
    if (rows == 1 && !tokens && p->use_graph) {
        hipGraphExec_t* slot = &c->stage_graph[longctx ? 1 : 0];
        if (!*slot) {
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = body(false);
            if (!rc && lnbk_advance_state(c->st, 1, st) != hipSuccess) rc = fail("advance_state launch failed");
            hipError_t e = hipStreamEndCapture(st, &g);
            if (rc) { if (g) hipGraphDestroy(g); return -1; }
            HIPCHK(e);
            HIPCHK(hipGraphInstantiate(slot, g, nullptr, nullptr, 0));
            HIPCHK(hipGraphDestroy(g));
        }
        if (c->dev_pos != pos) HIPCHK(ctx_set_state(c, pos, 0, true));
        c->call_T = 0;
        HIPCHK(hipGraphLaunch(*slot, st));
        c->dev_pos = pos + 1;                                // (the graph ends with advance_state)
        return 0;
    }
    HIPCHK(ctx_set_state(c, pos, 0, true));
    return body(tokens != nullptr);
}
// What crosses the boundary behind this stage (towards rank+1) / in front of it: the [rows, dim] hidden state, plus the [rows, ffn_hidden]
// gate*up activations when the cut lies between a block's gate/up and down parts.
static int pipe_xfer(lnb_pipe* p, lnb_ctx* c, int rows, bool sending) {
    lnb_model* m = c->m;
    const int edge = sending ? m->part_end : m->part_begin, peer = sending ? p->rank + 1 : p->rank - 1;
    const size_t nx = (size_t)rows * m->a.dim * 2, nf = (size_t)rows * m->ffn_hidden * 2;
    if (sending) { NCCLCHK(p, p->api->Send(c->x, nx, LNB_NCCL_INT8, peer, p->comm, p->xs)); if (edge % 3 == 2) NCCLCHK(p, p->api->Send(c->ffn, nf, LNB_NCCL_INT8, peer, p->comm, p->xs)); }
    else { NCCLCHK(p, p->api->Recv(c->x, nx, LNB_NCCL_INT8, peer, p->comm, p->xs)); if (edge % 3 == 2) NCCLCHK(p, p->api->Recv(c->ffn, nf, LNB_NCCL_INT8, peer, p->comm, p->xs)); }
    return 0;
}
// mailbox transport: `mine` is posted now; if its counterpart is already waiting, enqueue the copy on this pipe's exchange stream
static int loop_copy(lnb_pipe* p, const LoopMsg& s, const LoopMsg& r, bool token) {
    HIPCHK(hipStreamWaitEvent(p->xs, s.ctx->ev_done, 0));   // the result has been computed
    HIPCHK(hipStreamWaitEvent(p->xs, r.ctx->ev_done, 0));   // nothing still reads the buffer being overwritten
    if (token) HIPCHK(hipMemcpyAsync(r.ctx->dtok, s.ctx->dnext, 4, hipMemcpyDeviceToDevice, p->xs));
    else {
        if (s.rows != r.rows) return fail("loopback exchange: %d rows sent, %d rows expected", s.rows, r.rows);
        lnb_model* sm = s.ctx->m;
        HIPCHK(hipMemcpyAsync(r.ctx->x, s.ctx->x, (size_t)s.rows * sm->a.dim * 2, hipMemcpyDeviceToDevice, p->xs));
        if (sm->part_end % 3 == 2) HIPCHK(hipMemcpyAsync(r.ctx->ffn, s.ctx->ffn, (size_t)s.rows * sm->ffn_hidden * 2, hipMemcpyDeviceToDevice, p->xs));
    }
    HIPCHK(hipEventRecord(s.ctx->ev_sent, p->xs)); s.ctx->sent_pending = true;
    HIPCHK(hipEventRecord(r.ctx->ev_in, p->xs)); r.ctx->in_pending = true;
    return 0;
}
static int loop_post(lnb_pipe* p, bool sending, lnb_ctx* c, int rows, int peer, bool token) {
    std::lock_guard<std::mutex> lock(g_loops_mu);
    const std::pair<int, int> key = sending ? std::make_pair(p->rank, peer) : std::make_pair(peer, p->rank);   // (source rank, destination rank)
    auto& mine = sending ? p->loop->sends[key] : p->loop->recvs[key];
    auto& theirs = sending ? p->loop->recvs[key] : p->loop->sends[key];
    const LoopMsg msg{c, rows};
    if (theirs.empty()) { mine.push_back(msg); if (!sending) c->recv_unmatched = true; return 0; }
    const LoopMsg other = theirs.front(); theirs.pop_front();
    if (other.bat) return fail("in-process pipeline: a single-sequence hand-off met a batch hand-off (the stages must post the same ticks in the same order)");
    (sending ? other.ctx : c)->recv_unmatched = false;
    return sending ? loop_copy(p, msg, other, token) : loop_copy(p, other, msg, token);
}
extern "C" int lnb_pipeline_tick(lnb_pipe* p, lnb_ctx* run, int run_rows, int run_pos, const int32_t* run_tokens,
                                 lnb_ctx* send, int send_rows, lnb_ctx* recv, int recv_rows, int* token_slot_out) {
    if (!p) return fail("null argument");
    lnb_model* m = p->m;
    HIPCHK(hipSetDevice(m->device));
    const bool first = p->rank == 0, last = p->rank == p->world - 1;
    for (lnb_ctx* c : {run, send, recv}) if (c) { if (c->m != m) return fail("context of another model stage"); if (pipe_events(c)) return -1; }
    if (p->host && (send || recv)) return fail("this pipe has no transport (lnb_pipeline_init_host): the host layer moves lnb_ctx_hidden_ptr's buffers itself; send and recv must be NULL");
    if (token_slot_out) *token_slot_out = -1;
    if (run) {
        if (check_call(run, run_rows, run_pos)) return -1;
        if (run_tokens && !first) return fail("tokens given to a stage that does not own tok_embeddings");
        if (!run_tokens && first && run_rows != 1) return fail("the first stage needs host tokens for a multi-row step");
        if (run->recv_unmatched) return fail("in-process pipeline: this sequence's input has been requested but the sending stage has not posted it yet "
                                             "(the mailbox transport needs the stages ticked in lock-step order from one thread)");
        if (run->in_pending) { HIPCHK(hipStreamWaitEvent(run->stream, run->ev_in, 0)); run->in_pending = false; }      // its input has arrived
        if (run->sent_pending) { HIPCHK(hipStreamWaitEvent(run->stream, run->ev_sent, 0)); run->sent_pending = false; }  // its previous output has left
        if (enqueue_stage_step(p, run, run_tokens, run_rows, run_pos)) return -1;
        if (last) {
            if (p->tok_n == 0x7FFFFFFF) return fail("pipeline token log: slot counter exhausted (2^31 tokens): create a new pipe");
            HIPCHK(hipMemcpyAsync(p->h_tok + p->tok_n % p->tok_cap, run->dnext, 4, hipMemcpyDeviceToHost, run->stream));   // (the ring keeps the newest tok_cap tokens)
            if (token_slot_out) *token_slot_out = p->tok_n;
            p->tok_n++;
            if (p->world == 1) HIPCHK(hipMemcpyAsync(run->dtok, run->dnext, 4, hipMemcpyDeviceToDevice, run->stream));    // the ring of a one-stage pipe
        }
        HIPCHK(hipEventRecord(run->ev_done, run->stream));
    }
    if (p->world > 1 && p->loop && (send || recv)) {         // in-process transport
        if (recv) HIPCHK(hipEventRecord(recv->ev_done, recv->stream));
        if (send && loop_post(p, true, send, send_rows, last ? 0 : p->rank + 1, last)) return -1;
        if (recv && loop_post(p, false, recv, recv_rows, first ? p->world - 1 : p->rank - 1, first)) return -1;
        return 0;
    }
    if (p->world > 1 && (send || recv)) {
        if (send) HIPCHK(hipStreamWaitEvent(p->xs, send->ev_done, 0));                 // the result being sent has been computed
        if (recv) { HIPCHK(hipEventRecord(recv->ev_done, recv->stream)); HIPCHK(hipStreamWaitEvent(p->xs, recv->ev_done, 0)); }   // nothing still reads the buffer being overwritten
        NCCLCHK(p, p->api->GroupStart());
        int rc = 0;
        if (send) {
            if (last) { int r_ = p->api->Send(send->dnext, 4, LNB_NCCL_INT8, 0, p->comm, p->xs); if (r_) rc = fail("ncclSend failed: %s", p->api->GetErrorString(r_)); }
            else rc = pipe_xfer(p, send, send_rows, true);
        }
        if (!rc && recv) {
            if (first) { int r_ = p->api->Recv(recv->dtok, 4, LNB_NCCL_INT8, p->world - 1, p->comm, p->xs); if (r_) rc = fail("ncclRecv failed: %s", p->api->GetErrorString(r_)); }
            else rc = pipe_xfer(p, recv, recv_rows, false);
        }
        int re = p->api->GroupEnd();
        if (rc) return -1;
        if (re != 0) return fail("ncclGroupEnd failed: %s", p->api->GetErrorString(re));
        if (send) { HIPCHK(hipEventRecord(send->ev_sent, p->xs)); send->sent_pending = true; }
        if (recv) { HIPCHK(hipEventRecord(recv->ev_in, p->xs)); recv->in_pending = true; }
    }
    return 0;
}

// ---- the pipeline with BATCHES as the unit that moves through the stages ---------------------------------------------------------------
// Every rank holds, per group of sequences in flight, one lnb_batch over its stage's contexts of those sequences (prefill them one by one
// with the single-sequence ticks above, then lnb_batch_set_state on every rank).  A batched tick enqueues, like lnb_pipeline_tick:
//   run : one decode step of the group on this stage -- ONE pass over the stage's weights for all its sequences (captured graph); the
//         positions advance on the device; on the last rank the n tokens are appended to the pinned log (*token_slot_out = first slot);
//   send: the group's hidden states [n, dim] to rank + 1 (the last rank: its n token words to rank 0);   recv: the mirror image.
static int batch_events(lnb_batch* b) {
    if (!b->ev_done) { HIPCHK(hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
                       HIPCHK(hipEventCreateWithFlags(&b->ev_sent, hipEventDisableTiming)); }
    return 0;
}
static int loop_copy_batch(lnb_pipe* p, lnb_batch* sb, lnb_batch* rb, bool token) {
    if (sb->n != rb->n) return fail("loopback exchange: a batch of %d sequences sent, %d expected", sb->n, rb->n);
    HIPCHK(hipStreamWaitEvent(p->xs, sb->ev_done, 0));
    HIPCHK(hipStreamWaitEvent(p->xs, rb->ev_done, 0));
    if (token) HIPCHK(hipMemcpyAsync(rb->ring, sb->ring, (size_t)sb->n * 4, hipMemcpyDeviceToDevice, p->xs));
    else HIPCHK(hipMemcpyAsync(rb->x, sb->x, (size_t)sb->n * sb->m->a.dim * 2, hipMemcpyDeviceToDevice, p->xs));
    HIPCHK(hipEventRecord(sb->ev_sent, p->xs)); sb->sent_pending = true;
    HIPCHK(hipEventRecord(rb->ev_in, p->xs)); rb->in_pending = true;
    return 0;
}
static int loop_post_batch(lnb_pipe* p, bool sending, lnb_batch* b, int peer, bool token) {
    std::lock_guard<std::mutex> lock(g_loops_mu);
    const std::pair<int, int> key = sending ? std::make_pair(p->rank, peer) : std::make_pair(peer, p->rank);
    auto& mine = sending ? p->loop->sends[key] : p->loop->recvs[key];
    auto& theirs = sending ? p->loop->recvs[key] : p->loop->sends[key];
    LoopMsg msg{nullptr, b->n}; msg.bat = b;
    if (theirs.empty()) { mine.push_back(msg); if (!sending) b->recv_unmatched = true; return 0; }
    const LoopMsg other = theirs.front(); theirs.pop_front();
    if (!other.bat) return fail("in-process pipeline: a batch hand-off met a single-sequence hand-off (the stages must post the same ticks in the same order)");
    (sending ? other.bat : b)->recv_unmatched = false;
    return sending ? loop_copy_batch(p, b, other.bat, token) : loop_copy_batch(p, other.bat, b, token);
}
extern "C" int lnb_pipeline_tick_batch(lnb_pipe* p, lnb_batch* run, lnb_batch* send, lnb_batch* recv, int* token_slot_out) {
    if (!p) return fail("null argument");
    lnb_model* m = p->m;
    HIPCHK(hipSetDevice(m->device));
    const bool first = p->rank == 0, last = p->rank == p->world - 1;
    for (lnb_batch* b : {run, send, recv}) if (b) { if (b->m != m) return fail("batch of another model stage"); if (batch_events(b)) return -1; }
    if (p->host && (send || recv)) return fail("this pipe has no transport (lnb_pipeline_init_host): the host layer moves lnb_batch_boundary_ptr's buffers itself; send and recv must be NULL");
    if (token_slot_out) *token_slot_out = -1;
    if (run) {
        hipStream_t st = run->stream;
        if (run->recv_unmatched) return fail("in-process pipeline: this batch's input has been requested but the sending stage has not posted it yet "
                                             "(the mailbox transport needs the stages ticked in lock-step order from one thread)");
        if (run->in_pending) { HIPCHK(hipStreamWaitEvent(st, run->ev_in, 0)); run->in_pending = false; }
        if (run->sent_pending) { HIPCHK(hipStreamWaitEvent(st, run->ev_sent, 0)); run->sent_pending = false; }
        const bool ring_in = first && p->world > 1;
        if (p->use_graph) {
            if (!run->stage_graph) {
                hipGraph_t g = nullptr;
                HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                int rc = enqueue_batch_step(run, ring_in);
                hipError_t e = hipStreamEndCapture(st, &g);
                if (rc) { if (g) hipGraphDestroy(g); return -1; }
                HIPCHK(e);
                HIPCHK(hipGraphInstantiate(&run->stage_graph, g, nullptr, nullptr, 0));
                HIPCHK(hipGraphDestroy(g));
            }
            HIPCHK(hipGraphLaunch(run->stage_graph, st));
        } else if (enqueue_batch_step(run, ring_in)) return -1;
        if (last) {
            if (p->tok_n > 0x7FFFFFFF - LNB_BATCH_MAX) return fail("pipeline token log: slot counter exhausted (2^31 tokens): create a new pipe");
            const int slot = p->tok_n, at = slot % p->tok_cap, fit = std::min(run->n, p->tok_cap - at);   // (the ring keeps the newest tok_cap tokens)
            HIPCHK(hipMemcpyAsync(p->h_tok + at, run->ring, (size_t)fit * 4, hipMemcpyDeviceToHost, st));
            if (fit < run->n) HIPCHK(hipMemcpyAsync(p->h_tok, run->ring + fit, (size_t)(run->n - fit) * 4, hipMemcpyDeviceToHost, st));
            if (token_slot_out) *token_slot_out = slot;
            p->tok_n += run->n;
        }
        HIPCHK(hipEventRecord(run->ev_done, st));
    }
    if (p->world > 1 && p->loop && (send || recv)) {         // in-process transport
        if (recv) HIPCHK(hipEventRecord(recv->ev_done, recv->stream));
        if (send && loop_post_batch(p, true, send, last ? 0 : p->rank + 1, last)) return -1;
        if (recv && loop_post_batch(p, false, recv, first ? p->world - 1 : p->rank - 1, first)) return -1;
        return 0;
    }
    if (p->world > 1 && (send || recv)) {
        if (send) HIPCHK(hipStreamWaitEvent(p->xs, send->ev_done, 0));
        if (recv) { HIPCHK(hipEventRecord(recv->ev_done, recv->stream)); HIPCHK(hipStreamWaitEvent(p->xs, recv->ev_done, 0)); }
        NCCLCHK(p, p->api->GroupStart());
        int r1 = 0, r2 = 0;
        if (send) r1 = last ? p->api->Send(send->ring, (size_t)send->n * 4, LNB_NCCL_INT8, 0, p->comm, p->xs)
                            : p->api->Send(send->x, (size_t)send->n * m->a.dim * 2, LNB_NCCL_INT8, p->rank + 1, p->comm, p->xs);
        if (recv) r2 = first ? p->api->Recv(recv->ring, (size_t)recv->n * 4, LNB_NCCL_INT8, p->world - 1, p->comm, p->xs)
                             : p->api->Recv(recv->x, (size_t)recv->n * m->a.dim * 2, LNB_NCCL_INT8, p->rank - 1, p->comm, p->xs);
        const int re = p->api->GroupEnd();
        if (r1 || r2 || re) return fail("RCCL batch exchange failed: %s", p->api->GetErrorString(r1 ? r1 : r2 ? r2 : re));
        if (send) { HIPCHK(hipEventRecord(send->ev_sent, p->xs)); send->sent_pending = true; }
        if (recv) { HIPCHK(hipEventRecord(recv->ev_in, p->xs)); recv->in_pending = true; }
    }
    return 0;
}
// The RCCL binding on real hardware without a second GPU: a one-rank communicator and the pipeline's own grouped exchange -- one
// ncclSend + one ncclRecv, here both to rank 0 (itself) -- on a non-blocking stream, device buffer to device buffer, then a byte
// compare.  Exercises dlopen, the by-value unique id, the group calls and a RCCL kernel launch from a process that never loaded torch.
extern "C" int lnb_pipeline_selftest(int device, int n_bytes) {
    if (n_bytes <= 0 || n_bytes > (1 << 26)) return fail("n_bytes out of range");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail("no HIP device: liblnb_hip.so has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    const lnb_rccl_api* api = lnb_rccl_load();
    if (!api) return -1;
    lnb_nccl_id id; memset(&id, 0, sizeof id);
    int r = api->GetUniqueId(&id);
    if (r != 0) return fail("ncclGetUniqueId failed: %s", api->GetErrorString(r));
    void* comm = nullptr;
    r = api->CommInitRank(&comm, 1, id, 0);
    if (r != 0) return fail("ncclCommInitRank failed: %s", api->GetErrorString(r));
    std::vector<unsigned char> h((size_t)n_bytes), back((size_t)n_bytes);
    for (int i = 0; i < n_bytes; i++) h[i] = (unsigned char)((i * 131 + 7) >> 3);
    unsigned char *src = nullptr, *dst = nullptr; hipStream_t xs = nullptr;
    int rc = 0;
    hipError_t e = hipMalloc((void**)&src, (size_t)n_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&dst, (size_t)n_bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&xs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpy(src, h.data(), (size_t)n_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dst, 0, (size_t)n_bytes);
    if (e != hipSuccess) rc = fail("selftest setup: %s", hipGetErrorString(e));
    if (!rc) {
        int r0 = api->GroupStart(), r1 = api->Send(src, (size_t)n_bytes, LNB_NCCL_INT8, 0, comm, xs), r2 = api->Recv(dst, (size_t)n_bytes, LNB_NCCL_INT8, 0, comm, xs), r3 = api->GroupEnd();
        const int bad = r0 ? r0 : r1 ? r1 : r2 ? r2 : r3;
        if (bad) rc = fail("RCCL self exchange failed: %s", api->GetErrorString(bad));
    }
    if (!rc && (e = hipStreamSynchronize(xs)) != hipSuccess) rc = fail("selftest sync: %s", hipGetErrorString(e));
    if (!rc && (e = hipMemcpy(back.data(), dst, (size_t)n_bytes, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail("selftest readback: %s", hipGetErrorString(e));
    if (!rc && memcmp(back.data(), h.data(), (size_t)n_bytes) != 0) rc = fail("RCCL self exchange delivered different bytes");
    api->CommDestroy(comm);
    if (xs) hipStreamDestroy(xs);
    if (src) hipFree(src);
    if (dst) hipFree(dst);
    return rc;
}
// how many ranks the exchange spans, asked of the transport itself: ncclCommCount of the RCCL communicator (world > 1), the number of pipes
// that joined the in-process group (loopback), 1 for a one-stage pipe (no communicator exists)
extern "C" int lnb_pipeline_comm_count(lnb_pipe* p, int* out) {
    if (!p || !out) return fail("null argument");
    *out = 1;
    if (p->world == 1) return 0;
    if (p->host) { *out = 0; return 0; }                     // no transport of its own
    if (p->loop) { std::lock_guard<std::mutex> lock(g_loops_mu); *out = p->loop->users; return 0; }
    int n = 0;
    NCCLCHK(p, p->api->CommCount(p->comm, &n));
    *out = n;
    return 0;
}
// block until everything enqueued so far (stage steps and exchanges) has finished
extern "C" int lnb_pipeline_sync(lnb_pipe* p) {
    if (!p) return fail("null argument");
    HIPCHK(hipSetDevice(p->m->device));
    HIPCHK(hipDeviceSynchronize());
    return 0;
}
// tokens the last stage produced, by log slot (lnb_pipeline_tick's token_slot_out); valid after lnb_pipeline_sync
extern "C" int lnb_pipeline_read_tokens(lnb_pipe* p, int first_slot, int n, int32_t* out) {
    if (!p || !out) return fail("null argument");
    if (first_slot < 0 || n < 0 || n > p->tok_n - first_slot) return fail("token slots [%d, %d) out of range (%d logged)", first_slot, first_slot + n, p->tok_n);
    if (p->tok_n - first_slot > p->tok_cap) return fail("token slot %d has been overwritten: the log keeps the newest %d tokens (%d logged)", first_slot, p->tok_cap, p->tok_n);
    HIPCHK(hipSetDevice(p->m->device));
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < n; i++) out[i] = p->h_tok[(first_slot + i) % p->tok_cap];
    return 0;
}

// ---- single-op entry points for the parity tests ------------------------------------------------------
static int op_linear_impl(int device, const uint16_t* x, const uint16_t* norm_w, float eps, const uint16_t* w, uint16_t* y,
                          int rows, int n_out, int k_in, int rw, int mode = LNB_MODE_EXACT) {
    if (!x || !w || !y) return fail("null argument");
    if (rows <= 0 || n_out <= 0 || k_in <= 0) return fail("empty operand");
    if (k_in % 8) return fail("in_features must be a multiple of 8");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail("no HIP device: liblnb_hip.so has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    HIPCHK(lnbk_init());
    HIPCHK(lnbk_fast_init());
    if (mode != LNB_MODE_EXACT && mode != LNB_MODE_FAST) return fail("unknown mode %d", mode);
    if (rw == 0) rw = auto_rw(n_out, "LNB_RW_OP", k_in, norm_w == nullptr);
    if (rw != 16 && rw != 32 && rw != 64 && !(rw == 4 && !norm_w && k_in % 128 == 0 && k_in <= 16384) && !(rw == 24 && k_in % 256 == 0))
        return fail("rw must be 16, 32 or 64 (or 4: row-broadcast layout, no fused norm, in_features a multiple of 128; or 24: quad layout, in_features a multiple of 256)");
    if ((size_t)k_in * 4 > 120 * 1024) return fail("in_features %d does not fit the LDS staging", k_in);
    TiledDesc t{}; int64_t bytes = 0;
    if (alloc_tiled(t, n_out, k_in, rw, 1, bytes)) return -1;
    uint16_t *dx = nullptr, *dw = nullptr, *dy = nullptr, *dn = nullptr; StepState* st = nullptr;
    HIPCHK(hipMalloc((void**)&dx, (size_t)rows * k_in * 2)); HIPCHK(hipMalloc((void**)&dw, (size_t)n_out * k_in * 2));
    HIPCHK(hipMalloc((void**)&dy, (size_t)rows * n_out * 2)); HIPCHK(hipMalloc((void**)&st, sizeof(StepState)));
    HIPCHK(hipMemset(st, 0, sizeof(StepState)));
    HIPCHK(hipMemcpy(dx, x, (size_t)rows * k_in * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw, w, (size_t)n_out * k_in * 2, hipMemcpyHostToDevice));
    if (norm_w) { HIPCHK(hipMalloc((void**)&dn, (size_t)k_in * 2)); HIPCHK(hipMemcpy(dn, norm_w, (size_t)k_in * 2, hipMemcpyHostToDevice)); }
    HIPCHK(lnbk_tile(dw, t.w, n_out, k_in, 0, 0, rw, 1, 0, nullptr));
    if (use_mfma(rows) && env_int("LNB_OP_STREAM", 0) && k_in % 128 == 0) {      // (tests: the same operator through gemm_stream_kernel)
        HIPCHK(lnbk_batch_prepare());
        const size_t mb = m16_elems(n_out, k_in, 1) * 2;
        HIPCHK(hipMalloc((void**)&t.w16, mb)); HIPCHK(hipMemset(t.w16, 0, mb));
        HIPCHK(lnbk_m16_from_tiled(t.w, t.w16, n_out, k_in, rw, 1, nullptr));
    }
    if (use_mfma(rows)) {                                   // 16 or more rows: the matrix-core path of the prefill
        uint16_t* dxn = nullptr;
        if (norm_w) { HIPCHK(hipMalloc((void**)&dxn, (size_t)rows * k_in * 2)); HIPCHK(lnbk_rmsnorm_rows(dx, dn, dxn, rows, k_in, eps, nullptr)); }
        GemmParams gm = gemm_of(t, norm_w ? dxn : dx, k_in, n_out, rows, st); gm.out = dy;
        HIPCHK(gemm_dispatch(mode, &gm, EPI_STORE, nullptr));
        HIPCHK(hipDeviceSynchronize());
        if (dxn) hipFree(dxn);
    } else {
    GemvParams g{}; g.w = t.w; g.x = dx; g.norm_w = dn; g.eps = eps; g.K = k_in; g.n_rows = n_out; g.S = rows; g.st = st; g.out = dy;
    set_grid(g, t);
    if (mode == LNB_MODE_FAST) HIPCHK(lnbk_fast_gemv(&g, rw, 1, EPI_STORE, norm_w ? 1 : 0, nullptr));
    else HIPCHK(lnbk_gemv(&g, rw, 1, EPI_STORE, norm_w ? 1 : 0, nullptr));
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(y, dy, (size_t)rows * n_out * 2, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(st); hipFree(t.w); if (dn) hipFree(dn); if (t.w16) hipFree(t.w16);
    return 0;
}
// ml.Argmax (operations_impl.go:513-548) of one row of bf16 logits through argmax_kernel, the kernel of the greedy loop
extern "C" int lnb_op_argmax(int device, const uint16_t* logits_bf16, int n, int32_t* out) {
    if (!logits_bf16 || !out) return fail("null argument");
    if (n <= 0) return fail("empty operand");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail("no HIP device: liblnb_hip.so has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    uint16_t* d = nullptr; int32_t* dn = nullptr; StepState* st = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)n * 2 + 16)); HIPCHK(hipMalloc((void**)&dn, 16)); HIPCHK(hipMalloc((void**)&st, sizeof(StepState)));
    HIPCHK(hipMemset(st, 0, sizeof(StepState)));
    HIPCHK(hipMemcpy(d, logits_bf16, (size_t)n * 2, hipMemcpyHostToDevice));
    hipError_t e = lnbk_argmax(d, n, dn, st, nullptr, 0, 0, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dn, 4, hipMemcpyDeviceToHost);
    hipFree(d); hipFree(dn); hipFree(st);
    HIPCHK(e);
    return 0;
}
extern "C" int lnb_op_exp_table(int device, float divisor, double* out65536) {
    if (!out65536) return fail("null argument");
    if (!(divisor > 0.0f)) return fail("divisor must be positive");
    int ndev = 0; HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail("no HIP device: liblnb_hip.so has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    double* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)65536 * 8));
    hipError_t e = lnbk_exp_table(d, divisor, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out65536, d, (size_t)65536 * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    HIPCHK(e);
    return 0;
}
extern "C" int lnb_op_linear(int device, const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int n_out, int k_in, int rw) {
    return op_linear_impl(device, x, nullptr, 0.0f, w, y, rows, n_out, k_in, rw);
}
// the same two operators in a given arithmetic mode (LNB_MODE_FAST: the split-K kernels; norm_w may be NULL)
extern "C" int lnb_op_linear_mode(int device, const uint16_t* x, const uint16_t* norm_w, float eps, const uint16_t* w, uint16_t* y,
                                  int rows, int n_out, int k_in, int rw, int mode) {
    return op_linear_impl(device, x, norm_w, eps, w, y, rows, n_out, k_in, rw, mode);
}
extern "C" int lnb_op_rmsnorm_linear(int device, const uint16_t* x, const uint16_t* norm_w, float eps, const uint16_t* w, uint16_t* y,
                                     int rows, int n_out, int k_in, int rw) {
    if (!norm_w) return fail("null argument");
    return op_linear_impl(device, x, norm_w, eps, w, y, rows, n_out, k_in, rw);
}

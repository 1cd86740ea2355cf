// lnb_host.hpp -- C++ host-side mirror of the reference's Go API for the LlamaTransformer.Forward path, written over
// the C ABI of include/lnb.h (the reference is compiled Go; no Go toolchain exists on this image, so the host layer
// the north star asks for in Go is provided in C++ with the SAME names, argument meaning and error behaviour, and the
// cgo form of the same calls is in llama-nuts-and-bolts_amd/go/ and INTEGRATION.md).
//
//   reference (Go)                                              here (C++)
//   model.ModelArgs                 src/model/modelargs.go:12    lnb::ModelArgs
//   model.Model{Tensors,ModelArgs}  src/model/model.go           lnb::Model  (name -> host bf16 tensor, as the loader leaves it)
//   model.LoadModel                 src/model/loader.go:18-70    lnb::LoadModel(modelDir)  (pth zip + pickle + params.json; no tokenizer)
//   tiktoken.Load + Vocabulary      tiktokenreader.go:39         lnb::Vocabulary (TokenizeString / Tokenize over lnb_tokenizer_*)
//   model.NewLlamaTransformer       llamatransformer.go:64       lnb::LlamaTransformer::New(model, device)
//   (*LlamaTransformer).Forward     llamatransformer.go:145      lnb::LlamaTransformer::Forward(ctx, tokens, startPos, &logits)
//   model.NewInferenceContext       inferencecontext.go:17       lnb::InferenceContext(transformer, inferenceArgs, logFn)
//   inference.NewInferenceEngine    inference.go:50              lnb::InferenceEngine(transformer, inferenceArgs, logFn)
//   generateTokensInternal          inference.go:173             lnb::InferenceEngine::GenerateTokens(prompt, onToken)
// Errors: the Go functions return (value, error); here a std::runtime_error carries the same message text.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <cstdio>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/lnb.h"

namespace lnb {

using TokenId = int32_t;

struct ModelArgs {                       // modelargs.go:12-27, defaults :29-44
    int Dim = 4096, N_Layers = 32, N_Heads = 32, N_KVHeads = -1, VocabSize = -1, MultipleOf = 256;
    double FFNDimMultiplier = -1;
    float NormEpsilon = 1e-5f;
    bool UseScaledRope = false;
    double RopeTheta = 500000;
    int MaxSequenceLength = 2048;
    lnb_model_args c() const {
        lnb_model_args a{};
        a.dim = Dim; a.n_layers = N_Layers; a.n_heads = N_Heads; a.n_kv_heads = N_KVHeads; a.vocab_size = VocabSize;
        a.multiple_of = MultipleOf; a.ffn_dim_multiplier = FFNDimMultiplier; a.norm_eps = NormEpsilon;
        a.use_scaled_rope = UseScaledRope ? 1 : 0; a.rope_theta = RopeTheta; a.max_seq_len = MaxSequenceLength;
        return a;
    }
};

struct HostTensor {                      // ml.Tensor of DT_BF16 as the loader produces it (row-major, mmap-backed in the reference)
    std::vector<int64_t> Size;
    const uint16_t* RawData = nullptr;
};

struct PromptPart { std::string Header, Content; };     // src/inference/tokenize.go:21-25

class Vocabulary {                       // model.Vocabulary + the tokenizer half of InferenceEngine (tokenize.go:27-197)
public:
    explicit Vocabulary(const std::string& tokenizerModelPath) {
        if (lnb_tokenizer_load(tokenizerModelPath.c_str(), &h_) != 0) throw std::runtime_error(lnb_last_error());
        lnb_tokenizer_special(h_, &BeginOfSentenceId, &EndOfSentenceId, &eot_, &eom_);
    }
    ~Vocabulary() { if (h_) lnb_tokenizer_free(h_); }
    Vocabulary(const Vocabulary&) = delete;
    Vocabulary& operator=(const Vocabulary&) = delete;
    int Size() const { return lnb_tokenizer_vocab_size(h_); }
    std::set<int32_t> StopTokenIds() const { return {eom_, eot_}; }                      // tiktokenreader.go:80
    std::vector<int32_t> TokenizeString(const std::string& text) const {                 // tokenize.go:181-197
        std::vector<int32_t> out(4 * text.size() + 16);
        const int n = lnb_tokenizer_encode(h_, text.data(), (int)text.size(), out.data(), (int)out.size());
        if (n < 0) throw std::runtime_error(lnb_last_error());
        out.resize(n); return out;
    }
    std::vector<int32_t> Tokenize(const std::vector<PromptPart>& parts) const {          // tokenize.go:27-94
        std::vector<const char*> hs, cs; size_t bytes = 0;
        for (auto& p : parts) { hs.push_back(p.Header.c_str()); cs.push_back(p.Content.c_str()); bytes += p.Header.size() + p.Content.size(); }
        std::vector<int32_t> out(4 * bytes + 64 * (parts.size() + 2));
        const int n = lnb_tokenizer_encode_chat(h_, hs.data(), cs.data(), (int)parts.size(), out.data(), (int)out.size());
        if (n < 0) throw std::runtime_error(lnb_last_error());
        out.resize(n); return out;
    }
    std::string IdToToken(int32_t id) const {
        const char* b = nullptr; int n = 0;
        if (lnb_tokenizer_piece(h_, id, &b, &n) != 0) throw std::runtime_error(lnb_last_error());
        return std::string(b, (size_t)n);
    }
    int32_t BeginOfSentenceId = -1, EndOfSentenceId = -1;
private:
    lnb_tokenizer* h_ = nullptr; int32_t eot_ = -1, eom_ = -1;
};

struct Model {                           // model.Model: what LoadModel returns (src/model/loader.go:18-70)
    ModelArgs Args;
    std::shared_ptr<Vocabulary> Vocab;   // present when <dir>/tokenizer.model exists
    std::map<std::string, HostTensor> Tensors;
    std::set<TokenId> StopTokenIds;      // Vocabulary.StopTokenIds (src/tiktoken/tiktokenreader.go:48-82)
    bool Synthetic = false;              // no checkpoint: random-init on the device
    uint64_t SyntheticSeed = 1234;
};

struct InferenceArgs { int SequenceLength = 0; };   // src/common/inferenceargs.go:3-11

inline void check(int rc) { if (rc != 0) throw std::runtime_error(lnb_last_error()); }
// the library behind this header must be the one it was written against (LNB_ABI_VERSION), and a host that keeps several generations in flight wants to
// know how many hardware queues its streams really get (inference.go:163-174: one goroutine + one InferenceContext per generation)
inline void CheckABI() { if (lnb_abi_version() != LNB_ABI_VERSION) throw std::runtime_error("liblnb_hip.so reports another ABI version than include/lnb.h"); }
inline lnb_runtime_info_t RuntimeInfo(int device = 0, bool probeQueues = false) { lnb_runtime_info_t ri; check(lnb_runtime_info(device, probeQueues ? 1 : 0, &ri)); return ri; }

// model.LoadModel (src/model/loader.go:18-70): <dir>/consolidated.00.pth (mmap'ed, tensors are views into it for the lifetime
// of the returned Model, like the reference's never-unmapped mmap, src/torch/types.go:51-55) + <dir>/params.json.
// The tokenizer is out of scope here: VocabSize, which the reference takes from it (loader.go:101-108),
// is taken from the first dimension of tok_embeddings.weight when params.json does not give it.
inline std::shared_ptr<Model> LoadModel(const std::string& modelDir) {
    lnb_checkpoint* ck = nullptr;
    check(lnb_checkpoint_open((modelDir + "/consolidated.00.pth").c_str(), &ck));
    std::shared_ptr<lnb_checkpoint> keep(ck, lnb_checkpoint_close);
    std::shared_ptr<Model> m(new Model(), [keep](Model* p) { delete p; });       // the mmap lives as long as the Model
    lnb_model_args a{};
    check(lnb_model_args_from_json((modelDir + "/params.json").c_str(), &a));
    m->Args.Dim = a.dim; m->Args.N_Layers = a.n_layers; m->Args.N_Heads = a.n_heads; m->Args.N_KVHeads = a.n_kv_heads;
    m->Args.VocabSize = a.vocab_size; m->Args.MultipleOf = a.multiple_of; m->Args.FFNDimMultiplier = a.ffn_dim_multiplier;
    m->Args.NormEpsilon = a.norm_eps; m->Args.UseScaledRope = a.use_scaled_rope != 0; m->Args.RopeTheta = a.rope_theta;
    m->Args.MaxSequenceLength = a.max_seq_len;
    for (int i = 0; i < lnb_checkpoint_num_tensors(ck); i++) {
        const char* name = nullptr; int dtype = 0, rank = 0; int64_t shape[4] = {0, 0, 0, 0}, nbytes = 0; const void* data = nullptr;
        check(lnb_checkpoint_tensor(ck, i, &name, &dtype, shape, &rank, &data, &nbytes));
        if (dtype != LNB_DTYPE_BF16) continue;                               // only torch.BFloat16Storage (src/torch/types.go:15)
        HostTensor t; t.Size.assign(shape, shape + rank); t.RawData = (const uint16_t*)data;
        m->Tensors[name] = t;
    }
    {   // tokenizer.model is optional here; with it, VocabSize follows loader.go:101-108
        FILE* tf = fopen((modelDir + "/tokenizer.model").c_str(), "rb");
        if (tf) {
            fclose(tf);
            m->Vocab = std::make_shared<Vocabulary>(modelDir + "/tokenizer.model");
            m->StopTokenIds = m->Vocab->StopTokenIds();
            if (m->Args.VocabSize < 1) m->Args.VocabSize = m->Vocab->Size();
            else if (m->Args.VocabSize != m->Vocab->Size())
                throw std::runtime_error("VocabSize=" + std::to_string(m->Args.VocabSize) + " and vocabulary model length=" + std::to_string(m->Vocab->Size()) + " aren't equal");
        }
    }
    if (m->Args.VocabSize < 1) {
        auto it = m->Tensors.find("tok_embeddings.weight");
        if (it == m->Tensors.end()) throw std::runtime_error("tensor \"tok_embeddings.weight\" not found");
        m->Args.VocabSize = (int)it->second.Size[0];
    }
    return m;
}

class LlamaTransformer {
public:
    // model.NewLlamaTransformer: binds every tensor by name and shape (getTensor/getLayerTensor, loader.go:183-192),
    // builds PrecomputedFreqsCis (llamatransformer.go:109)
    static LlamaTransformer* New(const Model& model, int device = 0) {
        auto* t = new LlamaTransformer();
        t->args_ = model.Args;
        lnb_model_args a = model.Args.c();
        check(lnb_model_create(&a, device, 0, model.Args.N_Layers, &t->h_));
        try {
            if (model.Synthetic) check(lnb_model_fill_synthetic(t->h_, model.SyntheticSeed));
            else for (const auto& kv : model.Tensors)
                check(lnb_model_set_tensor(t->h_, kv.first.c_str(), kv.second.RawData, kv.second.Size.data(), (int)kv.second.Size.size()));
            check(lnb_model_finalize(t->h_, 0));
        } catch (...) { lnb_model_destroy(t->h_); delete t; throw; }
        return t;
    }
    ~LlamaTransformer() { if (h_) lnb_model_destroy(h_); }
    std::vector<float> PrecomputedFreqsCis() const {                 // exported field, llamatransformer.go:24
        int rows = 0; check(lnb_model_rope_table(h_, nullptr, 0, &rows));
        std::vector<float> out((size_t)rows * (args_.Dim / args_.N_Heads));
        check(lnb_model_rope_table(h_, out.data(), (int64_t)out.size(), &rows));
        return out;
    }
    const ModelArgs& Args() const { return args_; }
    lnb_model* handle() const { return h_; }
private:
    LlamaTransformer() = default;
    ModelArgs args_; lnb_model* h_ = nullptr;
};

using LogFn = std::function<void(const std::string&)>;

class InferenceContext {                 // inferencecontext.go:8-52
public:
    InferenceContext(const LlamaTransformer& t, InferenceArgs ia, LogFn logFn = nullptr) : t_(t), logFn_(std::move(logFn)) {
        SequenceLength = ia.SequenceLength > 0 ? ia.SequenceLength : t.Args().MaxSequenceLength;
        check(lnb_ctx_create(t.handle(), SequenceLength, &h_));
        if (logFn_) check(lnb_ctx_set_layer_callback(h_, &InferenceContext::layer_cb, this));
    }
    ~InferenceContext() { if (h_) lnb_ctx_destroy(h_); }
    int SequenceLength = 0;
    std::vector<uint16_t> CacheK(int layer) const { return kv(layer, 0); }   // exported fields, poked by the reference's tests
    std::vector<uint16_t> CacheV(int layer) const { return kv(layer, 1); }
    lnb_ctx* handle() const { return h_; }
    // hosts that keep several generations in flight on one GPU (one InferenceContext each, inference.go:174): the co-residency-friendly forms of
    // the one-token kernels (lnb_ctx_set_schedule); same tokens either way
    void SetThroughputSchedule(bool on) { check(lnb_ctx_set_schedule(h_, on ? LNB_SCHED_THROUGHPUT : LNB_SCHED_LATENCY)); }
private:
    static void layer_cb(int layer, int n, double secs, void* user) {      // infContext.Logf(...), llamatransformer.go:163
        auto* self = (InferenceContext*)user;
        char buf[160]; snprintf(buf, sizeof buf, "Transformer block layer %d / %d was run, took %.4f sec(s)", layer, n, secs);
        self->logFn_(buf);
    }
    std::vector<uint16_t> kv(int layer, int which) const {
        const ModelArgs& a = t_.Args(); const int kvh = a.N_KVHeads < 0 ? a.N_Heads : a.N_KVHeads;
        std::vector<uint16_t> out((size_t)SequenceLength * kvh * (a.Dim / a.N_Heads));
        check(lnb_ctx_read_kv(h_, layer, which, out.data()));
        return out;
    }
    const LlamaTransformer& t_; LogFn logFn_; lnb_ctx* h_ = nullptr;
};

// (*LlamaTransformer).Forward(infContext, inputTokens, startPos) -> logits [seq, vocab] f32 (llamatransformer.go:145-180)
inline std::vector<float> Forward(const LlamaTransformer& t, InferenceContext& ctx, const std::vector<TokenId>& inputTokens, int startPos) {
    std::vector<float> logits(inputTokens.size() * (size_t)t.Args().VocabSize);
    check(lnb_forward(ctx.handle(), inputTokens.data(), (int)inputTokens.size(), startPos, logits.data(), nullptr));
    return logits;
}

enum GenerationState { GSInProgress = 1, GSFinishedByReachingEOS = 2, GSFinishedByReachingSeqLen = 3 };   // inference.go:13-17

class InferenceEngine {                  // inference.go:40-56
public:
    InferenceEngine(const Model& model, const LlamaTransformer& t, InferenceArgs ia, LogFn logFn = nullptr)
        : model_(model), t_(t), ia_(ia), logFn_(std::move(logFn)) {}
    InferenceContext* CreateInferenceContext() { return new InferenceContext(t_, ia_, logFn_); }       // :256-258
    // generateTokensInternal (:173-254): prefill through Forward, then the decode loop (Forward(1 token) + Argmax +
    // token feedback) as hipGraph replays on the device; onToken receives (state, token) like generatedTokensCh.
    // chunk = tokens enqueued per lnb_decode_greedy_until call.  The stop ids are checked ON THE DEVICE (lnb_ctx_set_stop_ids: the token-feedback
    // kernel freezes the generation at the first stop token), so the chunk size changes nothing but the latency of the callback: the tokens
    // are the same for chunk 1 and chunk 32 (tests/native/host_mirror_test.cpp).
    void GenerateTokens(const std::vector<TokenId>& promptTokens, const std::function<void(GenerationState, TokenId)>& onToken, int chunk = 32) {
        std::unique_ptr<InferenceContext> ctx(CreateInferenceContext());
        {
            std::vector<TokenId> stops(model_.StopTokenIds.begin(), model_.StopTokenIds.end());
            if (stops.size() > 8) stops.resize(8);
            check(lnb_ctx_set_stop_ids(ctx->handle(), stops.empty() ? nullptr : stops.data(), (int)stops.size()));
        }
        const int promptLength = (int)promptTokens.size();
        if (promptLength >= ctx->SequenceLength)
            throw std::runtime_error("context SequenceLength " + std::to_string(ctx->SequenceLength) +
                                     " must be higher than prompt tokens length " + std::to_string(promptLength));
        TokenId next = -1;
        check(lnb_forward(ctx->handle(), promptTokens.data(), promptLength, 0, nullptr, &next));
        int curPos = promptLength;
        auto emit = [&](TokenId tok) -> bool {                        // stop conditions, :233-252
            if (model_.StopTokenIds.count(tok)) { onToken(GSFinishedByReachingEOS, tok); return false; }
            if (curPos + 1 == ctx->SequenceLength) { onToken(GSFinishedByReachingSeqLen, tok); return false; }
            onToken(GSInProgress, tok); return true;
        };
        if (!emit(next)) return;
        if (chunk < 1) chunk = 1;
        while (true) {
            const int remaining = ctx->SequenceLength - 1 - curPos;
            const int n = remaining < chunk ? remaining : chunk;
            if (n <= 0) return;
            std::vector<TokenId> out(n);
            int got = 0, fin = 0;
            check(lnb_decode_greedy_until(ctx->handle(), next, curPos, n, out.data(), &got, &fin, nullptr));
            for (int i = 0; i < got; i++) { curPos++; next = out[i]; if (!emit(next)) return; }
            if (fin) return;                                          // (emit() has reported the stop token: StopTokenIds are the device's stop ids)
        }
    }
private:
    const Model& model_; const LlamaTransformer& t_; InferenceArgs ia_; LogFn logFn_;
};

}  // namespace lnb

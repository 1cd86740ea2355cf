// Drives the C++ host mirror (llama-nuts-and-bolts_amd/host/lnb_host.hpp) the way cmd/main.go drives the Go API:
// build a (synthetic) model, create the engine, generate.  Prints the generated token ids (one line) so the python
// tests can compare them with the oracle.  Without a GPU NewLlamaTransformer must fail loudly (no CPU fallback).
#include "../../llama-nuts-and-bolts_amd/host/lnb_host.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    // LNB_MODEL_DIR=<dir with consolidated.00.pth + params.json>: the reference's LoadModel path instead of synthetic weights
    try { lnb::CheckABI(); } catch (const std::exception& e) { printf("error: %s\n", e.what()); return 4; }
    std::shared_ptr<lnb::Model> loaded;
    lnb::Model synthetic;
    if (const char* dir = getenv("LNB_MODEL_DIR")) {
        try { loaded = lnb::LoadModel(dir); } catch (const std::exception& e) { printf("error: %s\n", e.what()); return 3; }
    }
    lnb::Model& model = loaded ? *loaded : synthetic;
    if (!loaded) {
    model.Args.Dim = 256; model.Args.N_Layers = 2; model.Args.N_Heads = 4; model.Args.N_KVHeads = 2; model.Args.VocabSize = 1024;
    model.Args.MultipleOf = 64; model.Args.FFNDimMultiplier = 1.3; model.Args.UseScaledRope = true;
    model.Synthetic = true; model.SyntheticSeed = 1234;
    }
    if (const char* stops = getenv("LNB_STOP_IDS")) {        // "a,b": model.StopTokenIds (the synthetic model has no tokenizer to take them from)
        for (const char* q = stops; *q;) { model.StopTokenIds.insert(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; }
    }
    const int chunk = getenv("LNB_CHUNK") ? atoi(getenv("LNB_CHUNK")) : 32;      // tokens enqueued per device call (the stop check is on the device)
    int seq_len = argc > 1 ? atoi(argv[1]) : 40;
    try {
        std::unique_ptr<lnb::LlamaTransformer> t(lnb::LlamaTransformer::New(model, 0));
        int layers_logged = 0;
        lnb::InferenceEngine engine(model, *t, lnb::InferenceArgs{seq_len}, [&](const std::string&) { layers_logged++; });
        std::vector<lnb::TokenId> prompt;
        for (int i = 2; i < argc; i++) prompt.push_back(atoi(argv[i]));
        printf("tokens:");
        engine.GenerateTokens(prompt, [&](lnb::GenerationState st, lnb::TokenId tok) { printf(" %d", tok); if (st != lnb::GSInProgress) printf(" state=%d", (int)st); }, chunk);
        printf("\nlayers_logged: %d\n", layers_logged);
    } catch (const std::exception& e) {
        printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}

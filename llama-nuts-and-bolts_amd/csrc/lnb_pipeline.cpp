// lnb_pipeline.cpp -- RCCL, loaded on first use.  The layer-sharded pipeline (lnb_pipeline_* in lnb_api.cpp) is the only user; a
// single-GPU process never maps the 570 MB library.  The soname is looked up first, so a host process that already carries an RCCL
// (PyTorch bundles one) shares it; /opt/rocm/lib is the fallback.  "nccl" on ROCm IS RCCL: the p2p send/recv run over xGMI.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include "lnb_rccl.h"

extern "C" __attribute__((visibility("hidden"))) void lnb_set_error(const char* msg);

static lnb_rccl_api g_api;
static bool g_loaded = false;
static std::mutex g_mu;

extern "C" __attribute__((visibility("hidden"))) const lnb_rccl_api* lnb_rccl_load(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_loaded) return &g_api;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        char msg[512]; snprintf(msg, sizeof msg, "librccl.so.1 could not be loaded: %s", dlerror());
        lnb_set_error(msg);
        return nullptr;
    }
    struct { const char* name; void** slot; } syms[] = {
        {"ncclGetUniqueId", (void**)&g_api.GetUniqueId}, {"ncclCommInitRank", (void**)&g_api.CommInitRank},
        {"ncclCommDestroy", (void**)&g_api.CommDestroy}, {"ncclGroupStart", (void**)&g_api.GroupStart},
        {"ncclGroupEnd", (void**)&g_api.GroupEnd}, {"ncclSend", (void**)&g_api.Send}, {"ncclRecv", (void**)&g_api.Recv},
        {"ncclGetErrorString", (void**)&g_api.GetErrorString}, {"ncclGetVersion", (void**)&g_api.GetVersion},
        {"ncclCommCount", (void**)&g_api.CommCount},
    };
    for (auto& s : syms) {
        *s.slot = dlsym(h, s.name);
        if (!*s.slot) {
            char msg[256]; snprintf(msg, sizeof msg, "librccl.so.1 has no symbol %s", s.name);
            lnb_set_error(msg);
            return nullptr;
        }
    }
    g_loaded = true;
    return &g_api;
}

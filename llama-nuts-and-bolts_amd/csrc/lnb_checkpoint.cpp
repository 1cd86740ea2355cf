// lnb_checkpoint.cpp -- weight ingestion for the MI355X LlamaTransformer path ("next" #2).
//
// Reads a PyTorch zip checkpoint (Meta's consolidated.00.pth) WITHOUT PyTorch: the file is mmap'ed, the zip central
// directory is parsed (ZIP64 aware: the 8B checkpoint is 16 GB), the single *.pkl entry is run through a small pickle
// virtual machine, and every tensor comes out as (name, dtype, shape, pointer into the mmap) -- ready to be handed to
// lnb_model_set_tensor, which copies host -> HBM and re-tiles.  Host code only; nothing here touches the GPU.
//
// Behaviour restated (never copied) from adalkiran/llama-nuts-and-bolts:
//   src/torch/torchmodelreader.go:39-145  one *.pkl per archive, storages are the STORED zip entries <pkl stem>/<key>,
//                                         persistent id = ("storage", kind, key, location, numel)
//   src/torch/types.go:9-56               classes: torch._utils._rebuild_tensor_v2, torch.BFloat16Storage
//   src/pickle/pickledispatch.go:13-78    the pickle protocol-2 opcode subset torch.save emits
//   src/pickle/types.go:7-9               collections.OrderedDict (the backward-hooks argument)
//   src/common/memorymapper_unix.go:18-41 read-only mmap, tensors are sub-slices of it
//   src/model/loader.go:183-192           getTensor: "not found" / "incorrect shape" errors
//   src/model/modelargs.go:12-65          params.json -> ModelArgs with the reference's defaults
// Deliberate differences: the tensor's storage_offset is honoured (the reference ignores it, types.go:23-36, which is
// only right when every tensor owns its storage, as in Meta's files); a few more opcodes torch may emit are accepted
// (NONE, SETITEM, LONG_BINGET, LONG1, BINFLOAT); non-contiguous tensors are reported instead of silently mis-read.
#include "../../include/lnb.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

extern "C" void lnb_set_error(const char* msg);          // lnb_api.cpp: thread-local message behind lnb_last_error()
static int cfail(const char* fmt, ...) {
    char buf[768];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    lnb_set_error(buf);
    return -1;
}

namespace {
struct ZipEntry { std::string name; uint64_t data_off = 0, size = 0; };

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct Value;
typedef std::shared_ptr<Value> VP;
enum Kind { V_NONE, V_BOOL, V_INT, V_FLOAT, V_STR, V_TUPLE, V_DICT, V_MARK, V_CLASS, V_STORAGE, V_TENSOR };
enum ClassId { C_ORDERED_DICT, C_REBUILD_TENSOR_V2, C_STORAGE_BF16, C_STORAGE_F16, C_STORAGE_F32 };
struct Value {
    Kind kind = V_NONE;
    int64_t i = 0; double f = 0; std::string s;
    std::vector<VP> items;                                   // tuple items, or dict as key, value, key, value, ...
    // storage / tensor
    int dtype = 0; const uint8_t* data = nullptr; int64_t numel = 0;
    std::vector<int64_t> shape, stride;
    bool contiguous = true;
};
VP mk(Kind k) { auto v = std::make_shared<Value>(); v->kind = k; return v; }
}  // namespace

struct lnb_checkpoint {
    int fd = -1; const uint8_t* base = nullptr; uint64_t size = 0;
    std::vector<ZipEntry> entries;
    std::string data_base;                                   // "<archive>/data" (pkl path without ".pkl")
    std::vector<std::string> names; std::vector<VP> tensors; // in pickle order
    std::map<std::string, int> index;
    ~lnb_checkpoint() { if (base) munmap((void*)base, size); if (fd >= 0) close(fd); }
};

// ---- zip -------------------------------------------------------------------------------------------------------------
static int parse_zip(lnb_checkpoint* c) {
    const uint8_t* b = c->base; const uint64_t n = c->size;
    if (n < 22) return cfail("not a zip archive (too small)");
    // end of central directory: scan back over a possible comment
    int64_t eocd = -1;
    for (int64_t p = (int64_t)n - 22; p >= 0 && p >= (int64_t)n - 22 - 65535; p--)
        if (rd32(b + p) == 0x06054b50u) { eocd = p; break; }
    if (eocd < 0) return cfail("not a zip archive (no end-of-central-directory record)");
    uint64_t cd_count = rd16(b + eocd + 10), cd_size = rd32(b + eocd + 12), cd_off = rd32(b + eocd + 16);
    if (eocd >= 20 && rd32(b + eocd - 20) == 0x07064b50u) {                      // ZIP64 locator -> ZIP64 EOCD
        const uint64_t e64 = rd64(b + eocd - 20 + 8);
        if (e64 > n || n - e64 < 56 || rd32(b + e64) != 0x06064b50u) return cfail("corrupt zip64 end-of-central-directory record");
        cd_count = rd64(b + e64 + 32); cd_size = rd64(b + e64 + 40); cd_off = rd64(b + e64 + 48);
    }
    if (cd_off > n || cd_size > n - cd_off) return cfail("corrupt zip central directory");   // (every bound is checked without overflow:
                                                                                             //  the file is untrusted and read through an mmap)
    uint64_t p = cd_off;
    for (uint64_t k = 0; k < cd_count; k++) {
        if (p > n || n - p < 46 || rd32(b + p) != 0x02014b50u) return cfail("corrupt zip central directory entry %llu", (unsigned long long)k);
        const uint16_t method = rd16(b + p + 10), nlen = rd16(b + p + 28), xlen = rd16(b + p + 30), clen = rd16(b + p + 32);
        if (n - p - 46 < (uint64_t)nlen + xlen + clen) return cfail("corrupt zip central directory entry %llu", (unsigned long long)k);
        uint64_t csize = rd32(b + p + 20), usize = rd32(b + p + 24), lho = rd32(b + p + 42);
        ZipEntry e; e.name.assign((const char*)b + p + 46, nlen);
        // zip64 extended information: only the fields that were 0xFFFFFFFF, in this order
        const uint8_t* x = b + p + 46 + nlen; const uint8_t* xe = x + xlen;
        while (x + 4 <= xe) {
            const uint16_t id = rd16(x), sz = rd16(x + 2); const uint8_t* d = x + 4;
            if (id == 0x0001) {
                if (usize == 0xFFFFFFFFu && d + 8 <= xe) { usize = rd64(d); d += 8; }
                if (csize == 0xFFFFFFFFu && d + 8 <= xe) { csize = rd64(d); d += 8; }
                if (lho == 0xFFFFFFFFu && d + 8 <= xe) { lho = rd64(d); d += 8; }
            }
            x += 4 + sz;
        }
        if (method != 0) return cfail("zip entry \"%s\" is compressed (method %d): torch checkpoints store their entries", e.name.c_str(), method);
        if (lho > n || n - lho < 30 || rd32(b + lho) != 0x04034b50u) return cfail("corrupt zip local header of \"%s\"", e.name.c_str());
        e.data_off = lho + 30 + rd16(b + lho + 26) + rd16(b + lho + 28);       // the LOCAL name/extra lengths (torch pads here)
        e.size = usize;
        if (e.data_off > n || e.size > n - e.data_off) return cfail("zip entry \"%s\" runs past the end of the file", e.name.c_str());
        c->entries.push_back(e);
        p += 46 + nlen + xlen + clen;
    }
    return 0;
}
static const ZipEntry* find_entry(const lnb_checkpoint* c, const std::string& name) {
    for (auto& e : c->entries) if (e.name == name) return &e;
    return nullptr;
}

// ---- pickle ----------------------------------------------------------------------------------------------------------
namespace {
struct Unpickler {
    lnb_checkpoint* c; const uint8_t* p; const uint8_t* end;
    std::vector<VP> stack; std::map<int64_t, VP> memo;
    bool need(size_t k) { return (size_t)(end - p) >= k; }
    int line(std::string& out) {
        const uint8_t* q = p;
        while (q < end && *q != '\n') q++;
        if (q == end) return cfail("pickle: unterminated line");
        out.assign((const char*)p, q - p); p = q + 1; return 0;
    }
    int pop_to_mark(std::vector<VP>& items) {
        size_t m = stack.size();
        while (m > 0 && stack[m - 1]->kind != V_MARK) m--;
        if (m == 0) return cfail("pickle: MARK not found");
        items.assign(stack.begin() + m, stack.end());
        stack.resize(m - 1);
        return 0;
    }
    int find_class(const std::string& mod, const std::string& name, VP& out) {
        // src/torch/torchmodelreader.go:99-108 + src/pickle/types.go:7-9
        static const struct { const char* m; const char* n; int id; } K[] = {
            {"collections", "OrderedDict", C_ORDERED_DICT}, {"torch._utils", "_rebuild_tensor_v2", C_REBUILD_TENSOR_V2},
            {"torch", "BFloat16Storage", C_STORAGE_BF16}, {"torch", "HalfStorage", C_STORAGE_F16}, {"torch", "FloatStorage", C_STORAGE_F32}};
        for (auto& k : K) if (mod == k.m && name == k.n) { out = mk(V_CLASS); out->i = k.id; return 0; }
        return cfail("unknown class \"%s.%s\" not found", mod.c_str(), name.c_str());
    }
    int persistent_load(const VP& pid, VP& out) {
        // ("storage", StorageKind, key, location, numel)  (torchmodelreader.go:110-144)
        if (pid->kind != V_TUPLE || pid->items.size() < 5 || pid->items[0]->kind != V_STR || pid->items[0]->s != "storage")
            return cfail("pid[0] must have value \"storage\"");
        const VP& kind = pid->items[1];
        if (kind->kind != V_CLASS || kind->i < C_STORAGE_BF16) return cfail("pid[1] must be type of StorageKind");
        if (pid->items[2]->kind != V_STR || pid->items[4]->kind != V_INT) return cfail("malformed storage persistent id");
        const std::string fn = c->data_base + "/" + pid->items[2]->s;
        const ZipEntry* e = find_entry(c, fn);
        if (!e) return cfail("file \"%s\" not found in Torch model file", fn.c_str());
        out = mk(V_STORAGE);
        out->dtype = kind->i == C_STORAGE_BF16 ? LNB_DTYPE_BF16 : (kind->i == C_STORAGE_F16 ? LNB_DTYPE_F16 : LNB_DTYPE_F32);
        out->numel = pid->items[4]->i; out->data = c->base + e->data_off; out->s = fn;
        const int64_t isz = out->dtype == LNB_DTYPE_F32 ? 4 : 2;
        if (out->numel < 0 || (uint64_t)out->numel > e->size / (uint64_t)isz) return cfail("storage \"%s\": %lld elements do not fit the %llu-byte zip entry", fn.c_str(), (long long)out->numel, (unsigned long long)e->size);
        return 0;
    }
    int reduce(const VP& fn, const VP& args, VP& out) {
        if (fn->kind != V_CLASS || args->kind != V_TUPLE) return cfail("pickle: REDUCE of a non-class object");
        if (fn->i == C_ORDERED_DICT) { out = mk(V_DICT); return 0; }
        if (fn->i == C_REBUILD_TENSOR_V2) {
            // (storage, storage_offset, size, stride, requires_grad, backward_hooks[, metadata])  (types.go:23-36)
            const auto& a = args->items;
            if (a.size() < 4 || a[0]->kind != V_STORAGE || a[1]->kind != V_INT || a[2]->kind != V_TUPLE || a[3]->kind != V_TUPLE)
                return cfail("cannot convert the arguments of torch._utils._rebuild_tensor_v2");
            out = mk(V_TENSOR);
            out->dtype = a[0]->dtype; out->s = a[0]->s;
            const int64_t isz = out->dtype == LNB_DTYPE_F32 ? 4 : 2, off = a[1]->i;
            // every product / sum below is checked against the storage size before it can overflow
            const int64_t cap = a[0]->numel;
            int64_t numel = 1, maxoff = 0;
            for (auto& d : a[2]->items) {
                if (d->kind != V_INT || d->i < 0) return cfail("tensor size must be non-negative integers");
                out->shape.push_back(d->i);
                if (d->i != 0 && numel > cap / d->i) return cfail("tensor view runs past its storage \"%s\"", a[0]->s.c_str());
                numel *= d->i;
            }
            for (auto& d : a[3]->items) { if (d->kind != V_INT || d->i < 0) return cfail("tensor stride must be non-negative integers"); out->stride.push_back(d->i); }
            if (out->stride.size() != out->shape.size()) return cfail("tensor size and stride ranks differ");
            int64_t expect = 1;
            for (int k = (int)out->shape.size() - 1; k >= 0; k--) {
                if (out->shape[k] > 1 && out->stride[k] != expect) out->contiguous = false;
                if (out->shape[k] > 1) {
                    if (out->stride[k] > cap / (out->shape[k] - 1)) return cfail("tensor view runs past its storage \"%s\"", a[0]->s.c_str());
                    maxoff += (out->shape[k] - 1) * out->stride[k];
                    if (maxoff > cap) return cfail("tensor view runs past its storage \"%s\"", a[0]->s.c_str());
                }
                expect *= out->shape[k] > 0 ? out->shape[k] : 1;
            }
            if (off < 0 || off > cap || (numel > 0 && maxoff >= cap - off)) return cfail("tensor view runs past its storage \"%s\"", a[0]->s.c_str());
            out->numel = numel; out->data = a[0]->data + off * isz;
            return 0;
        }
        return cfail("pickle: storage classes are not callable");
    }
    int run(VP& result) {
        for (;;) {
            if (!need(1)) return cfail("pickle: unexpected end of stream");
            const uint8_t op = *p++;
            switch (op) {
            case 0x80: if (!need(1)) return cfail("pickle: truncated"); if (*p++ > 5) return cfail("unsupported pickle protocol: %d", p[-1]); break;   // PROTO
            case '}': stack.push_back(mk(V_DICT)); break;
            case ')': stack.push_back(mk(V_TUPLE)); break;
            case '(': stack.push_back(mk(V_MARK)); break;
            case 'N': stack.push_back(mk(V_NONE)); break;
            case 0x88: case 0x89: { VP v = mk(V_BOOL); v->i = op == 0x88; stack.push_back(v); break; }
            case 'K': { if (!need(1)) return cfail("pickle: truncated"); VP v = mk(V_INT); v->i = *p++; stack.push_back(v); break; }
            case 'M': { if (!need(2)) return cfail("pickle: truncated"); VP v = mk(V_INT); v->i = rd16(p); p += 2; stack.push_back(v); break; }
            case 'J': { if (!need(4)) return cfail("pickle: truncated"); VP v = mk(V_INT); v->i = (int32_t)rd32(p); p += 4; stack.push_back(v); break; }
            case 0x8a: {                                                       // LONG1: little-endian two's complement
                if (!need(1)) return cfail("pickle: truncated");
                const int nb = *p++; if (nb > 8 || !need(nb)) return cfail("pickle: LONG1 wider than 8 bytes");
                uint64_t u = 0; for (int k = 0; k < nb; k++) u |= (uint64_t)p[k] << (8 * k);
                if (nb > 0 && nb < 8 && (p[nb - 1] & 0x80)) u |= ~0ull << (8 * nb);
                p += nb; VP v = mk(V_INT); v->i = (int64_t)u; stack.push_back(v); break;
            }
            case 'G': { if (!need(8)) return cfail("pickle: truncated"); uint64_t u = 0; for (int k = 0; k < 8; k++) u = (u << 8) | p[k]; p += 8;
                        VP v = mk(V_FLOAT); memcpy(&v->f, &u, 8); stack.push_back(v); break; }
            case 'X': case 'T': { if (!need(4)) return cfail("pickle: truncated"); const uint32_t n = rd32(p); p += 4; if (!need(n)) return cfail("pickle: truncated string");
                                  VP v = mk(V_STR); v->s.assign((const char*)p, n); p += n; stack.push_back(v); break; }
            case 'U': { if (!need(1)) return cfail("pickle: truncated"); const uint32_t n = *p++; if (!need(n)) return cfail("pickle: truncated string");
                        VP v = mk(V_STR); v->s.assign((const char*)p, n); p += n; stack.push_back(v); break; }
            case 'c': { std::string mod, name; if (line(mod) || line(name)) return -1; VP v; if (find_class(mod, name, v)) return -1; stack.push_back(v); break; }
            case 'q': { if (!need(1) || stack.empty()) return cfail("pickle: bad BINPUT"); memo[*p++] = stack.back(); break; }
            case 'r': { if (!need(4) || stack.empty()) return cfail("pickle: bad LONG_BINPUT"); memo[rd32(p)] = stack.back(); p += 4; break; }
            case 'h': { if (!need(1)) return cfail("pickle: truncated"); auto it = memo.find(*p++); if (it == memo.end()) return cfail("pickle: memo miss"); stack.push_back(it->second); break; }
            case 'j': { if (!need(4)) return cfail("pickle: truncated"); auto it = memo.find(rd32(p)); p += 4; if (it == memo.end()) return cfail("pickle: memo miss"); stack.push_back(it->second); break; }
            case 't': { VP v = mk(V_TUPLE); if (pop_to_mark(v->items)) return -1; stack.push_back(v); break; }
            case 0x85: case 0x86: case 0x87: {
                const size_t n = op - 0x84; if (stack.size() < n) return cfail("pickle: stack underflow");
                VP v = mk(V_TUPLE); v->items.assign(stack.end() - n, stack.end()); stack.resize(stack.size() - n); stack.push_back(v); break;
            }
            case 'Q': { if (stack.empty()) return cfail("pickle: stack underflow"); VP pid = stack.back(); stack.pop_back(); VP v; if (persistent_load(pid, v)) return -1; stack.push_back(v); break; }
            case 'R': { if (stack.size() < 2) return cfail("pickle: stack underflow"); VP args = stack.back(); stack.pop_back(); VP fn = stack.back(); stack.pop_back();
                        VP v; if (reduce(fn, args, v)) return -1; stack.push_back(v); break; }
            case 'u': { std::vector<VP> kv; if (pop_to_mark(kv)) return -1; if (stack.empty() || stack.back()->kind != V_DICT || (kv.size() & 1)) return cfail("pickle: bad SETITEMS");
                        auto& d = stack.back()->items; d.insert(d.end(), kv.begin(), kv.end()); break; }
            case 's': { if (stack.size() < 3 || stack[stack.size() - 3]->kind != V_DICT) return cfail("pickle: bad SETITEM");
                        VP val = stack.back(); stack.pop_back(); VP key = stack.back(); stack.pop_back(); stack.back()->items.push_back(key); stack.back()->items.push_back(val); break; }
            case '.': if (stack.empty()) return cfail("pickle: empty stack at STOP"); result = stack.back(); return 0;
            default: return cfail("unsupported Pickle op code: 0x%X '%c'", op, (op >= 32 && op < 127) ? op : '?');   // pickledispatch.go:97
            }
        }
    }
};
}  // namespace

// ---- C ABI -----------------------------------------------------------------------------------------------------------
extern "C" int lnb_checkpoint_open(const char* path, lnb_checkpoint** out) {
    if (!path || !out) return cfail("null argument");
    std::unique_ptr<lnb_checkpoint> c(new lnb_checkpoint());
    c->fd = open(path, O_RDONLY);
    if (c->fd < 0) return cfail("open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(c->fd, &st) != 0) return cfail("stat %s: %s", path, strerror(errno));
    c->size = (uint64_t)st.st_size;
    if (c->size == 0) return cfail("%s is empty", path);
    void* m = mmap(nullptr, c->size, PROT_READ, MAP_PRIVATE, c->fd, 0);      // memorymapper_unix.go:18-41
    if (m == MAP_FAILED) return cfail("mmap %s: %s", path, strerror(errno));
    c->base = (const uint8_t*)m;
    if (parse_zip(c.get())) return -1;
    const ZipEntry* pkl = nullptr; int npkl = 0;
    for (auto& e : c->entries) if (e.name.size() > 4 && e.name.compare(e.name.size() - 4, 4, ".pkl") == 0) { pkl = &e; npkl++; }
    if (npkl != 1) return cfail("no .pkl file found in Torch model file \"%s\"", path);      // torchmodelreader.go:48-50
    c->data_base = pkl->name.substr(0, pkl->name.size() - 4);
    Unpickler u{c.get(), c->base + pkl->data_off, c->base + pkl->data_off + pkl->size, {}, {}};
    VP root;
    if (u.run(root)) return -1;
    if (root->kind != V_DICT) return cfail("the checkpoint's top-level object is not a dict of tensors");
    for (size_t k = 0; k + 1 < root->items.size(); k += 2) {
        const VP& key = root->items[k]; const VP& val = root->items[k + 1];
        if (key->kind != V_STR || val->kind != V_TENSOR) continue;          // (non-tensor entries are not weights)
        auto it = c->index.find(key->s);
        if (it != c->index.end()) { c->tensors[it->second] = val; continue; }   // PickleDict.Set: last value wins, order kept
        c->index[key->s] = (int)c->names.size(); c->names.push_back(key->s); c->tensors.push_back(val);
    }
    *out = c.release();
    return 0;
}
extern "C" void lnb_checkpoint_close(lnb_checkpoint* c) { delete c; }
extern "C" int lnb_checkpoint_num_tensors(const lnb_checkpoint* c) { return c ? (int)c->names.size() : 0; }
extern "C" int lnb_checkpoint_find(const lnb_checkpoint* c, const char* name) {
    if (!c || !name) return -1;
    auto it = c->index.find(name);
    return it == c->index.end() ? -1 : it->second;
}
extern "C" int lnb_checkpoint_tensor(const lnb_checkpoint* c, int i, const char** name, int* dtype, int64_t* shape, int* rank,
                                     const void** data, int64_t* nbytes) {
    if (!c || i < 0 || i >= (int)c->names.size()) return cfail("tensor index %d out of range", i);
    const Value& t = *c->tensors[i];
    if (!t.contiguous) return cfail("tensor \"%s\" is not contiguous (stride does not match a row-major layout)", c->names[i].c_str());
    if (t.shape.size() > 4) return cfail("tensor \"%s\" has rank %d (> 4)", c->names[i].c_str(), (int)t.shape.size());
    if (name) *name = c->names[i].c_str();
    if (dtype) *dtype = t.dtype;
    if (rank) *rank = (int)t.shape.size();
    if (shape) for (size_t k = 0; k < t.shape.size(); k++) shape[k] = t.shape[k];
    if (data) *data = t.data;
    if (nbytes) *nbytes = t.numel * (t.dtype == LNB_DTYPE_F32 ? 4 : 2);
    return 0;
}

// every tensor this model (pipeline stage) owns, bound by the reference's names with the reference's shape check
extern "C" int lnb_model_load_checkpoint(lnb_model* m, const lnb_checkpoint* c) {
    if (!m || !c) return cfail("null argument");
    const int n = lnb_model_num_tensors(m);
    for (int k = 0; k < n; k++) {
        const char* name = nullptr; int64_t want[2] = {0, 0}; int wrank = 0;
        if (lnb_model_tensor_info(m, k, &name, want, &wrank)) return -1;
        const int i = lnb_checkpoint_find(c, name);
        if (i < 0) return cfail("tensor \"%s\" not found", name);                                           // loader.go:185-187
        int dtype = 0, rank = 0; int64_t shape[4] = {0, 0, 0, 0}, nbytes = 0; const void* data = nullptr;
        if (lnb_checkpoint_tensor(c, i, nullptr, &dtype, shape, &rank, &data, &nbytes)) return -1;
        bool same = rank == wrank;
        for (int d = 0; same && d < rank; d++) same = shape[d] == want[d];
        if (!same) {                                                                                           // loader.go:188-190
            std::string a = "[", b = "[";
            for (int d = 0; d < wrank; d++) a += (d ? " " : "") + std::to_string(want[d]);
            for (int d = 0; d < rank; d++) b += (d ? " " : "") + std::to_string(shape[d]);
            return cfail("tensor \"%s\" has incorrect shape; expected %s], got %s]", name, a.c_str(), b.c_str());
        }
        if (dtype != LNB_DTYPE_BF16) return cfail("tensor \"%s\" is not bfloat16 (only torch.BFloat16Storage is supported, src/torch/types.go:15)", name);
        if (lnb_model_set_tensor(m, name, (const uint16_t*)data, shape, rank)) return -1;
    }
    return 0;
}

// ---- params.json -> lnb_model_args (src/model/modelargs.go:12-65) -----------------------------------------------------
static bool json_number(const std::string& js, const char* key, double& out) {
    const std::string pat = std::string("\"") + key + "\"";
    size_t p = js.find(pat);
    if (p == std::string::npos) return false;
    p = js.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    p++;
    while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) p++;
    if (js.compare(p, 4, "true") == 0) { out = 1; return true; }
    if (js.compare(p, 5, "false") == 0) { out = 0; return true; }
    if (js.compare(p, 4, "null") == 0) return false;
    char* e = nullptr;
    out = strtod(js.c_str() + p, &e);
    return e != js.c_str() + p;
}
extern "C" int lnb_model_args_from_json(const char* path, lnb_model_args* a) {
    if (!path || !a) return cfail("null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return cfail("open %s: %s", path, strerror(errno));
    std::string js; char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) js.append(buf, n);
    fclose(f);
    // NewModelArgs defaults (modelargs.go:29-44)
    a->dim = 4096; a->n_layers = 32; a->n_heads = 32; a->n_kv_heads = -1; a->vocab_size = -1; a->multiple_of = 256;
    a->ffn_dim_multiplier = -1; a->norm_eps = 1e-5f; a->rope_theta = 500000; a->use_scaled_rope = 0; a->max_seq_len = 2048;
    double v;
    if (js.find('{') == std::string::npos) return cfail("%s: not a JSON object", path);
    if (json_number(js, "dim", v)) a->dim = (int32_t)v;
    if (json_number(js, "n_layers", v)) a->n_layers = (int32_t)v;
    if (json_number(js, "n_heads", v)) a->n_heads = (int32_t)v;
    if (json_number(js, "n_kv_heads", v)) a->n_kv_heads = (int32_t)v;
    if (json_number(js, "vocab_size", v)) a->vocab_size = (int32_t)v;
    if (json_number(js, "multiple_of", v)) a->multiple_of = (int32_t)v;
    if (json_number(js, "ffn_dim_multiplier", v)) a->ffn_dim_multiplier = v;
    if (json_number(js, "norm_eps", v)) a->norm_eps = (float)v;
    if (json_number(js, "use_scaled_rope", v)) a->use_scaled_rope = v != 0;
    if (json_number(js, "rope_theta", v)) a->rope_theta = v;
    return 0;
}

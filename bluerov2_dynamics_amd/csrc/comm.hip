// comm.hip -- the one collective of the path, straight on RCCL (no torch): the per-GPU G^T[G|Y] blocks of a sharded EDMDc
// fit are summed over the ranks of a node (xGMI).  The reference is single-process (nothing to cite but the data template
// training/train_sim_brov2_koopmanEDMDc.py:153-214); the contract is SURVEY.md 8(b)/(e).
//
// librccl is bound at run time with dlopen: libbrov2.so must keep loading on a host without RCCL, and inside a PyTorch
// process the copy PyTorch already mapped (same SONAME librccl.so.1) is the one to use -- two RCCL instances in one process
// would each bring their own bootstrap threads and IPC handles.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/brov2.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2 API: /opt/rocm/include/rccl/rccl.h:40-43,187,448,467)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
constexpr int kNcclSuccess = 0, kNcclSum = 0, kNcclMax = 2, kNcclInt64 = 4, kNcclUint64 = 5, kNcclFloat64 = 8;
static_assert(sizeof(NcclUniqueId) == BROV_COMM_ID_BYTES, "ncclUniqueId size");

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    Rccl& r = g_rccl;
    const char* env = std::getenv("BROV2_RCCL_LIBRARY");
    const bool forced = env && env[0];             // an explicit library is the ONLY candidate (no fall-back to the defaults)
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    auto open = [&](const char* n, int flags) {
        r.handle = dlopen(n, flags);
        if (!r.handle && !(flags & RTLD_NOLOAD)) {
            const char* e = dlerror();                 // dlerror() clears the message: read it exactly once
            if (why.empty()) why = e ? e : "?";
        }
        return r.handle != nullptr;
    };
    if (forced) {
        open(env, RTLD_NOW | RTLD_LOCAL);
    } else {
        // a copy that is already mapped (PyTorch's) wins
        for (const char* n : {"librccl.so.1", "librccl.so"})
            if (open(n, RTLD_NOW | RTLD_NOLOAD)) break;
        for (size_t i = 0; !r.handle && i < sizeof names / sizeof names[0]; ++i) open(names[i], RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) { r.err = std::string("librccl not found: ") + (why.empty() ? "?" : why); return; }
    auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + s; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.err.empty() ? &g_rccl : nullptr;
}

}  // namespace

struct brov_comm {
    NcclComm comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    std::string err;
};

namespace {
int cfail(brov_comm* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
int nccl_fail(brov_comm* c, int rc, const char* what) {
    const Rccl* r = rccl();
    return cfail(c, BROV_ERR_COMM, std::string(what) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(rc) : "rccl error"));
}
struct DevGuard {
    int prev = -1; bool sw = false;
    explicit DevGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) { (void)hipSetDevice(dev); sw = true; } }
    ~DevGuard() { if (sw && prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

int brov_comm_available(void) { return rccl() ? 1 : 0; }

int brov_comm_unique_id(unsigned char id[BROV_COMM_ID_BYTES]) {
    if (!id) return BROV_ERR_ARG;
    const Rccl* r = rccl();
    if (!r) return BROV_ERR_COMM;
    NcclUniqueId u;
    if (r->GetUniqueId(&u) != kNcclSuccess) return BROV_ERR_COMM;
    std::memcpy(id, u.internal, BROV_COMM_ID_BYTES);
    return BROV_OK;
}

int brov_comm_init_rank(int device_id, const unsigned char id[BROV_COMM_ID_BYTES], int nranks, int rank, brov_comm** out) {
    if (!out) return BROV_ERR_ARG;
    *out = nullptr;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return BROV_ERR_ARG;
    const Rccl* r = rccl();
    if (!r) return BROV_ERR_COMM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BROV_ERR_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return BROV_ERR_ARG;
    brov_comm* c = new (std::nothrow) brov_comm();
    if (!c) return BROV_ERR_NOMEM;
    c->nranks = nranks; c->rank = rank; c->device = device_id;
    DevGuard g(device_id);            // ncclCommInitRank binds the communicator to the current device
    NcclUniqueId u;
    std::memcpy(u.internal, id, BROV_COMM_ID_BYTES);
    const int rc = r->CommInitRank(&c->comm, nranks, u, rank);
    if (rc != kNcclSuccess) { delete c; return BROV_ERR_COMM; }
    *out = c;
    return BROV_OK;
}

void brov_comm_destroy(brov_comm* c) {
    if (!c) return;
    const Rccl* r = rccl();
    if (r && c->comm) { DevGuard g(c->device); (void)r->CommDestroy(c->comm); }
    delete c;
}

int brov_comm_nranks(const brov_comm* c) { return c ? c->nranks : BROV_ERR_ARG; }
int brov_comm_rank(const brov_comm* c) { return c ? c->rank : BROV_ERR_ARG; }
const char* brov_comm_last_error(const brov_comm* c) { return c ? c->err.c_str() : (g_rccl.err.empty() ? "null comm" : g_rccl.err.c_str()); }

// In-place sum over the ranks of both Gram blocks as ONE grouped RCCL operation on `hip_stream` (asynchronous).
int edmdc_gram_allreduce_dev(brov_comm* c, double* d_GtG, int64_t n_gtg, double* d_GtY, int64_t n_gty, void* hip_stream) {
    if (!c || n_gtg < 0 || n_gty < 0 || (n_gtg && !d_GtG) || (n_gty && !d_GtY)) return cfail(c, BROV_ERR_ARG, "edmdc_gram_allreduce_dev: bad argument");
    const Rccl* r = rccl();
    if (!r) return cfail(c, BROV_ERR_COMM, g_rccl.err);
    DevGuard g(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    int rc = r->GroupStart();
    if (rc != kNcclSuccess) return nccl_fail(c, rc, "ncclGroupStart");
    int rc1 = kNcclSuccess, rc2 = kNcclSuccess;
    if (n_gtg) rc1 = r->AllReduce(d_GtG, d_GtG, (size_t)n_gtg, kNcclFloat64, kNcclSum, c->comm, st);
    if (n_gty) rc2 = r->AllReduce(d_GtY, d_GtY, (size_t)n_gty, kNcclFloat64, kNcclSum, c->comm, st);
    rc = r->GroupEnd();
    if (rc1 != kNcclSuccess) return nccl_fail(c, rc1, "ncclAllReduce(GtG)");
    if (rc2 != kNcclSuccess) return nccl_fail(c, rc2, "ncclAllReduce(GtY)");
    if (rc != kNcclSuccess) return nccl_fail(c, rc, "ncclGroupEnd");
    return BROV_OK;
}

// In-place all-reduce of `count` 64-bit words on `hip_stream` (asynchronous): op 0 = sum of int64, op 1 = maximum of uint64 -- the
// two exchanges of the sharded Lloyd loop (edmdc_set_kmeans_allreduce / edmdc_kmeans_use_comm): integer member sums, coordinate ranges.
int brov_comm_allreduce_words(brov_comm* c, void* d_buf, int64_t count, int op, void* hip_stream) {
    if (!c || count < 0 || (count && !d_buf) || (op != 0 && op != 1)) return cfail(c, BROV_ERR_ARG, "brov_comm_allreduce_words: bad argument");
    const Rccl* r = rccl();
    if (!r) return cfail(c, BROV_ERR_COMM, g_rccl.err);
    if (count == 0) return BROV_OK;
    DevGuard g(c->device);
    const int rc = r->AllReduce(d_buf, d_buf, (size_t)count, op == 0 ? kNcclInt64 : kNcclUint64, op == 0 ? kNcclSum : kNcclMax, c->comm,
                                reinterpret_cast<hipStream_t>(hip_stream));
    if (rc != kNcclSuccess) return nccl_fail(c, rc, "ncclAllReduce(words)");
    return BROV_OK;
}

}  // extern "C"

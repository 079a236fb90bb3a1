// comm.hip -- part of libhmcmt_hip.so: the all-gather of the chains' sample blocks over RCCL (xGMI inside a node).
//
// Replaces, for a host that is not Python/torch, what parallelHMCSampler does with the workers' results: the reference
// fetches every worker's (hmcmodel, hmcstats, hmcdata) to the master with remotecall_fetch
// (HMCMT/src/HMCSampler/parallelHMC.jl:23-45); here one process per GPU holds its chains' blocks and ONE collective --
// ncclAllGather of `count` doubles per rank -- leaves every rank with every block.  librccl.so is loaded on first use
// (dlopen): the hot-path library itself does not depend on it.  One communicator = one GPU of one process.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include "../../include/hmcmt.h"

namespace {

// the subset of rccl.h this file needs (opaque handles, ABI-stable enums: ncclFloat64 = 8, ncclSuccess = 0)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueIdBytes { char internal[HMCMT_COMM_ID_BYTES]; };
typedef int (*fnGetUniqueId)(ncclUniqueIdBytes*);
typedef int (*fnCommInitRank)(ncclComm_t*, int, ncclUniqueIdBytes, int);
typedef int (*fnAllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
typedef int (*fnCommDestroy)(ncclComm_t);
typedef const char* (*fnGetErrorString)(int);
constexpr int NCCL_FLOAT64 = 8;

struct Rccl {
    void* h = nullptr;
    fnGetUniqueId getUniqueId = nullptr;
    fnCommInitRank commInitRank = nullptr;
    fnAllGather allGather = nullptr;
    fnCommDestroy commDestroy = nullptr;
    fnGetErrorString errorString = nullptr;
    std::string err;
};
Rccl g_rccl;
std::string g_commCreateError;

bool load_rccl() {
    if (g_rccl.h) return true;
    // The RCCL that belongs to the HIP runtime this process runs on: the one in the directory libamdhip64 was loaded from.
    // (A Python process may hold two ROCm stacks -- /opt/rocm and the copies bundled in the torch wheel; whichever
    // libamdhip64 was loaded first serves everybody, and an RCCL of the OTHER stack on top of it hangs in
    // ncclCommInitRank: met in tests/test_gpu_parity.py, where conftest.py loads /opt/rocm's runtime before torch comes in.)
    std::string dir;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t sl = dir.rfind('/');
        dir = sl == std::string::npos ? std::string() : dir.substr(0, sl + 1);
    }
    const std::string names[] = {dir + "librccl.so.1", dir + "librccl.so", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;                   // dlerror() of the last attempt (the call returns the message once and clears it)
    const char* forced = getenv("HMCMT_RCCL_PATH");       // this library and no other (deployments with RCCL elsewhere; the loader's test)
    if (forced && *forced) {
        g_rccl.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.h) { const char* e = dlerror(); why = e ? e : "?"; }
    } else
    for (const std::string& n : names) {
        if (n.empty() || (n[0] != '/' && !dir.empty() && &n < &names[2])) continue;
        g_rccl.h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) break;
        const char* e = dlerror();
        why = e ? e : "?";
    }
    if (!g_rccl.h) { g_rccl.err = std::string("cannot load librccl.so: ") + (why.empty() ? "?" : why); return false; }
    g_rccl.getUniqueId = (fnGetUniqueId)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.commInitRank = (fnCommInitRank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.allGather = (fnAllGather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.commDestroy = (fnCommDestroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.errorString = (fnGetErrorString)dlsym(g_rccl.h, "ncclGetErrorString");
    if (!g_rccl.getUniqueId || !g_rccl.commInitRank || !g_rccl.allGather || !g_rccl.commDestroy) {
        g_rccl.err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        dlclose(g_rccl.h); g_rccl.h = nullptr;
        return false;
    }
    return true;
}
std::string nccl_msg(const char* what, int rc) {
    return std::string(what) + ": " + (g_rccl.errorString ? g_rccl.errorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}

}  // namespace

struct hmcmt_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;      // staging for host buffers
    size_t capSend = 0, capRecv = 0;
    std::string err;
};

extern "C" {

int hmcmt_comm_id(void* id) {
    if (!id) return HMCMT_EINVAL;
    if (!load_rccl()) { g_commCreateError = g_rccl.err; return HMCMT_ENODEV; }
    ncclUniqueIdBytes u;
    const int rc = g_rccl.getUniqueId(&u);
    if (rc) { g_commCreateError = nccl_msg("ncclGetUniqueId", rc); return HMCMT_EHIP; }
    std::memcpy(id, u.internal, HMCMT_COMM_ID_BYTES);
    return 0;
}

int hmcmt_comm_create(hmcmt_comm** out, int32_t device_id, int32_t nranks, int32_t rank, const void* id) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) { g_commCreateError = "hmcmt_comm_create: bad arguments"; return HMCMT_EINVAL; }
    *out = nullptr;
    if (!load_rccl()) { g_commCreateError = g_rccl.err; return HMCMT_ENODEV; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) { g_commCreateError = "hmcmt_comm_create: no such HIP device"; return HMCMT_ENODEV; }
    if (hipSetDevice(device_id) != hipSuccess) { g_commCreateError = "hipSetDevice failed"; return HMCMT_EHIP; }
    hmcmt_comm* c = new hmcmt_comm;
    c->device = device_id; c->nranks = nranks; c->rank = rank;
    ncclUniqueIdBytes u;
    std::memcpy(u.internal, id, HMCMT_COMM_ID_BYTES);
    const int rc = g_rccl.commInitRank(&c->comm, nranks, u, rank);
    if (rc) { g_commCreateError = nccl_msg("ncclCommInitRank", rc); delete c; return HMCMT_EHIP; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_commCreateError = "hipStreamCreate failed"; g_rccl.commDestroy(c->comm); delete c; return HMCMT_EHIP; }
    *out = c;
    return 0;
}

const char* hmcmt_comm_last_error(const hmcmt_comm* c) { return c ? c->err.c_str() : g_commCreateError.c_str(); }

int hmcmt_allgather_samples(hmcmt_comm* c, const double* send, double* recv, int64_t count, int32_t on_device) {
    if (!c || !send || !recv || count < 0) return HMCMT_EINVAL;
    if (count == 0) return 0;
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return HMCMT_EHIP; }
    const size_t nb = (size_t)count * sizeof(double);
    const double* ds = send;
    double* dr = recv;
    if (!on_device) {
        if (c->capSend < nb) { if (c->d_send) hipFree(c->d_send); c->capSend = 0; if (hipMalloc((void**)&c->d_send, nb) != hipSuccess) { c->err = "device allocation failed"; return HMCMT_ENOMEM; } c->capSend = nb; }
        if (c->capRecv < nb * c->nranks) { if (c->d_recv) hipFree(c->d_recv); c->capRecv = 0; if (hipMalloc((void**)&c->d_recv, nb * c->nranks) != hipSuccess) { c->err = "device allocation failed"; return HMCMT_ENOMEM; } c->capRecv = nb * c->nranks; }
        if (hipMemcpyAsync(c->d_send, send, nb, hipMemcpyHostToDevice, c->stream) != hipSuccess) { c->err = "copy to the device failed"; return HMCMT_EHIP; }
        ds = c->d_send; dr = c->d_recv;
    }
    const int rc = g_rccl.allGather(ds, dr, (size_t)count, NCCL_FLOAT64, c->comm, c->stream);
    if (rc) { c->err = nccl_msg("ncclAllGather", rc); return HMCMT_EHIP; }
    if (!on_device && hipMemcpyAsync(recv, c->d_recv, nb * c->nranks, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { c->err = "copy from the device failed"; return HMCMT_EHIP; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "the collective failed (stream error)"; return HMCMT_EHIP; }
    return 0;
}

int hmcmt_comm_destroy(hmcmt_comm* c) {
    if (!c) return HMCMT_EINVAL;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.commDestroy(c->comm);
    if (c->d_send) hipFree(c->d_send);
    if (c->d_recv) hipFree(c->d_recv);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

}  // extern "C"

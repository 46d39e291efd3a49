// comm_rccl.hip -- the halo exchange of the sharded modes behind the C-ABI (SURVEY.md section 8b / 8e:
// "exabm4d_halo_exchange(... ncclComm_t ...)"; north_star: "RCCL over xGMI only for halo exchange at chunk
// borders").  Host code only.  RCCL is dlopen()ed on first use: libexabm4d.so does not link it, so hosts
// that never shard (the bm4d() drop-in, single-GPU pipelines) need no RCCL at all.
//
// One process per GPU; the caller creates one communicator per context from a 128-byte unique id that rank 0
// generates and the host layer hands round (distributed.py: over MASTER_ADDR / MASTER_PORT, no torch).  The
// exchange itself is the pattern SURVEY.md 8e names -- ncclGroupStart; ncclSend / ncclRecv with each slab
// neighbour; ncclGroupEnd -- as bytes (the planes are uint16 counts or fp32 estimates; RCCL has no 16-bit
// integer type), enqueued on the CONTEXT'S stream: it is ordered against the kernels that produce the planes
// it sends and against those that read the planes it receives, without a host synchronisation.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/exabm4d.h"
#include "exabm4d_common.h"

extern "C" int exabm4d_internal_fail(exabm4d_ctx* ctx, int code, const char* msg);     // exabm4d_api.hip
extern "C" hipStream_t exabm4d_internal_stream(exabm4d_ctx* ctx);
extern "C" int exabm4d_internal_device(exabm4d_ctx* ctx);

namespace {
// the slice of rccl.h this file uses (the library is third-party and stays outside the link line)
struct ncclUniqueId_t {
    char internal[128];
};
typedef void* ncclComm_h;
typedef int ncclResult_i;            // ncclSuccess = 0
constexpr int kNcclInt8 = 0, kNcclFloat64 = 8, kNcclMax = 2;

struct Rccl {
    void* handle = nullptr;
    ncclResult_i (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    ncclResult_i (*CommInitRank)(ncclComm_h*, int, ncclUniqueId_t, int) = nullptr;
    ncclResult_i (*CommDestroy)(ncclComm_h) = nullptr;
    ncclResult_i (*GroupStart)() = nullptr;
    ncclResult_i (*GroupEnd)() = nullptr;
    ncclResult_i (*Send)(const void*, size_t, int, int, ncclComm_h, hipStream_t) = nullptr;
    ncclResult_i (*Recv)(void*, size_t, int, int, ncclComm_h, hipStream_t) = nullptr;
    ncclResult_i (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_h, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_i) = nullptr;
    std::string err;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {std::getenv("EXABM4D_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
        g_rccl.err = dlerror();
    }
    if (!g_rccl.handle) return;
    auto sym = [&](const char* s) {
        void* p = dlsym(g_rccl.handle, s);
        if (!p) g_rccl.err = std::string("librccl lacks ") + s;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd ||
        !g_rccl.Send || !g_rccl.Recv || !g_rccl.AllReduce) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
    }
}
bool rccl_ready(exabm4d_ctx* ctx, int* rc) {
    std::call_once(g_once, load_rccl);
    if (g_rccl.handle) return true;
    *rc = exabm4d_internal_fail(ctx, EXABM4D_ERR_UNSUPPORTED,
                                ("RCCL is not available (set EXABM4D_RCCL_LIB to librccl.so): " + g_rccl.err).c_str());
    return false;
}
int nccl_fail(exabm4d_ctx* ctx, ncclResult_i r, const char* what) {
    std::string m = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return exabm4d_internal_fail(ctx, EXABM4D_ERR_HIP, m.c_str());
}
}  // namespace

struct exabm4d_comm {
    ncclComm_h comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    double* word = nullptr;          // device scratch of exabm4d_comm_max_f64_host
};

extern "C" {

int exabm4d_comm_unique_id(uint8_t id[EXABM4D_COMM_ID_BYTES]) {
    int rc = 0;
    if (!id) return exabm4d_internal_fail(nullptr, EXABM4D_ERR_INVALID, "id is NULL");
    if (!rccl_ready(nullptr, &rc)) return rc;
    ncclUniqueId_t u;
    const ncclResult_i r = g_rccl.GetUniqueId(&u);
    if (r != 0) return nccl_fail(nullptr, r, "ncclGetUniqueId");
    static_assert(sizeof(u) == EXABM4D_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id, &u, sizeof u);
    return EXABM4D_OK;
}

int exabm4d_comm_create(exabm4d_ctx* ctx, int nranks, int rank, const uint8_t id[EXABM4D_COMM_ID_BYTES],
                        exabm4d_comm** out) {
    int rc = 0;
    if (!ctx || !id || !out) return exabm4d_internal_fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks)
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_INVALID, "comm: need 0 <= rank < nranks");
    if (!rccl_ready(ctx, &rc)) return rc;
    if (hipSetDevice(exabm4d_internal_device(ctx)) != hipSuccess)
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_HIP, "comm: hipSetDevice");
    ncclUniqueId_t u;
    std::memcpy(&u, id, sizeof u);
    exabm4d_comm* c = new (std::nothrow) exabm4d_comm();
    if (!c) return exabm4d_internal_fail(ctx, EXABM4D_ERR_NOMEM, "out of host memory");
    const ncclResult_i r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);      // collective over the ranks
    if (r != 0) {
        delete c;
        return nccl_fail(ctx, r, "ncclCommInitRank");
    }
    if (hipMalloc((void**)&c->word, sizeof(double)) != hipSuccess) {
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_NOMEM, "comm: device scratch");
    }
    c->nranks = nranks;
    c->rank = rank;
    c->device = exabm4d_internal_device(ctx);
    *out = c;
    return EXABM4D_OK;
}

int exabm4d_comm_destroy(exabm4d_comm* comm) {
    if (!comm) return EXABM4D_OK;
    if (comm->word) (void)hipFree(comm->word);
    if (comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->comm);
    delete comm;
    return EXABM4D_OK;
}

int exabm4d_halo_exchange_dev(exabm4d_ctx* ctx, exabm4d_comm* comm, int lo_peer, const void* send_lo, void* recv_lo,
                              size_t bytes_lo, int hi_peer, const void* send_hi, void* recv_hi, size_t bytes_hi) {
    if (!ctx || !comm || !comm->comm) return exabm4d_internal_fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (comm->device != exabm4d_internal_device(ctx))
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_INVALID, "halo exchange: communicator of another device");
    const bool lo = lo_peer >= 0 && bytes_lo > 0, hi = hi_peer >= 0 && bytes_hi > 0;
    if ((lo && (lo_peer >= comm->nranks || !send_lo || !recv_lo)) || (hi && (hi_peer >= comm->nranks || !send_hi || !recv_hi)))
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_INVALID, "halo exchange: bad peer or NULL buffer");
    if (!lo && !hi) return EXABM4D_OK;
    if (hipSetDevice(comm->device) != hipSuccess) return exabm4d_internal_fail(ctx, EXABM4D_ERR_HIP, "hipSetDevice");
    hipStream_t s = exabm4d_internal_stream(ctx);
    ncclResult_i r = g_rccl.GroupStart();
    if (r != 0) return nccl_fail(ctx, r, "ncclGroupStart");
    // every rank posts its sends and receives in the same order (lower neighbour first): the matching of two
    // messages between the same pair of ranks is by order
    if (lo && r == 0) r = g_rccl.Send(send_lo, bytes_lo, kNcclInt8, lo_peer, comm->comm, s);
    if (lo && r == 0) r = g_rccl.Recv(recv_lo, bytes_lo, kNcclInt8, lo_peer, comm->comm, s);
    if (hi && r == 0) r = g_rccl.Send(send_hi, bytes_hi, kNcclInt8, hi_peer, comm->comm, s);
    if (hi && r == 0) r = g_rccl.Recv(recv_hi, bytes_hi, kNcclInt8, hi_peer, comm->comm, s);
    const ncclResult_i e = g_rccl.GroupEnd();
    if (r != 0) return nccl_fail(ctx, r, "ncclSend / ncclRecv");
    if (e != 0) return nccl_fail(ctx, e, "ncclGroupEnd");
    return EXABM4D_OK;
}

int exabm4d_comm_max_f64_host(exabm4d_ctx* ctx, exabm4d_comm* comm, double* value) {
    if (!ctx || !comm || !comm->comm || !value) return exabm4d_internal_fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (hipSetDevice(comm->device) != hipSuccess) return exabm4d_internal_fail(ctx, EXABM4D_ERR_HIP, "hipSetDevice");
    hipStream_t s = exabm4d_internal_stream(ctx);
    if (hipMemcpyAsync(comm->word, value, sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess)
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_HIP, "comm max: upload");
    const ncclResult_i r = g_rccl.AllReduce(comm->word, comm->word, 1, kNcclFloat64, kNcclMax, comm->comm, s);
    if (r != 0) return nccl_fail(ctx, r, "ncclAllReduce");
    if (hipMemcpyAsync(value, comm->word, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return exabm4d_internal_fail(ctx, EXABM4D_ERR_HIP, "comm max: download");
    return EXABM4D_OK;
}

}  // extern "C"

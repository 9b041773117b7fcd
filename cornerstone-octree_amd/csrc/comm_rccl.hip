// cstone_hip_comm_ops served by RCCL, natively in C++ -- what the multi-rank Domain::sync (domain_mr.hip) uses on a node
// of MI355X: one process per GPU, every collective enqueued on the context's stream (no host synchronisation, no Python
// in the path).  Replaces the MPI transport of the reference: MPI_Allreduce (R/sfc/box_mpi.hpp:104,
// R/tree/update_mpi.hpp:59), the Isend / Probe / Recv loops of the particle exchange
// (R/domain/domaindecomp_mpi_gpu.cuh:86-185) and of the halo exchange (R/halos/exchange_halos_gpu.cuh:35-119).
//   all_reduce   -> ncclAllReduce (f64 | u32, sum | min), in place
//   all_gather   -> ncclAllGather of bytes
//   all_to_all_v -> one group of ncclSend / ncclRecv per peer (xGMI is point to point: every pair has its own link);
//                   the rank's own segment is a device-to-device copy
// librccl.so is opened on first use: a single-GPU client of libcstone_hip does not load it.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "ctx.hpp"

struct cstone_hip_comm_rccl
{
    cstone_hip_ctx* ctx = nullptr;
    ncclComm_t comm     = nullptr;
    int rank = 0, size = 1;
};

namespace cship
{
namespace
{

struct RcclApi
{
    void* handle = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId*)                                                                  = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                           = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t)                                                                     = nullptr;
    ncclResult_t (*allReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*allGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)              = nullptr;
    ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                     = nullptr;
    ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                           = nullptr;
    ncclResult_t (*groupStart)()                                                                                = nullptr;
    ncclResult_t (*groupEnd)()                                                                                  = nullptr;
    const char* (*getErrorString)(ncclResult_t)                                                                 = nullptr;
};

RcclApi* rccl(std::string* why)
{
    static RcclApi api;
    static std::string error;
    static bool tried = false;
    if (!tried)
    {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { error = std::string("cannot open librccl.so: ") + dlerror(); }
        else
        {
            auto sym = [&](const char* n)
            {
                void* p = dlsym(api.handle, n);
                if (!p && error.empty()) error = std::string("librccl.so lacks ") + n;
                return p;
            };
            api.getUniqueId    = (decltype(api.getUniqueId))sym("ncclGetUniqueId");
            api.commInitRank   = (decltype(api.commInitRank))sym("ncclCommInitRank");
            api.commDestroy    = (decltype(api.commDestroy))sym("ncclCommDestroy");
            api.allReduce      = (decltype(api.allReduce))sym("ncclAllReduce");
            api.allGather      = (decltype(api.allGather))sym("ncclAllGather");
            api.send           = (decltype(api.send))sym("ncclSend");
            api.recv           = (decltype(api.recv))sym("ncclRecv");
            api.groupStart     = (decltype(api.groupStart))sym("ncclGroupStart");
            api.groupEnd       = (decltype(api.groupEnd))sym("ncclGroupEnd");
            api.getErrorString = (decltype(api.getErrorString))sym("ncclGetErrorString");
        }
    }
    if (!error.empty())
    {
        if (why) *why = error;
        return nullptr;
    }
    return &api;
}

int ncclFail(cstone_hip_comm_rccl* c, const char* what, ncclResult_t r)
{
    RcclApi* api = rccl(nullptr);
    fail(c->ctx, CSTONE_E_INTERNAL, "%s: %s", what, api && api->getErrorString ? api->getErrorString(r) : "RCCL error");
    return int(r) ? int(r) : 1;
}

int allReduceCb(void* user, void* buf, size_t count, int dtype, int op)
{
    auto* c = static_cast<cstone_hip_comm_rccl*>(user);
    if (count == 0) return 0;
    ncclResult_t r = rccl(nullptr)->allReduce(buf, buf, count, dtype == 0 ? ncclFloat64 : ncclUint32,
                                              op == 0 ? ncclSum : ncclMin, c->comm, c->ctx->stream);
    return r == ncclSuccess ? 0 : ncclFail(c, "ncclAllReduce", r);
}

int allGatherCb(void* user, const void* send, void* recv, size_t bytes)
{
    auto* c = static_cast<cstone_hip_comm_rccl*>(user);
    if (bytes == 0) return 0;
    ncclResult_t r = rccl(nullptr)->allGather(send, recv, bytes, ncclUint8, c->comm, c->ctx->stream);
    return r == ncclSuccess ? 0 : ncclFail(c, "ncclAllGather", r);
}

int allToAllVCb(void* user, const void* send, const size_t* sendBytes, void* recv, const size_t* recvBytes)
{
    auto* c      = static_cast<cstone_hip_comm_rccl*>(user);
    RcclApi* api = rccl(nullptr);
    const char* s = static_cast<const char*>(send);
    char* r       = static_cast<char*>(recv);
    size_t sOff = 0, rOff = 0, ownSend = 0, ownRecv = 0;
    ncclResult_t rc = api->groupStart();
    if (rc != ncclSuccess) return ncclFail(c, "ncclGroupStart", rc);
    for (int p = 0; p < c->size && rc == ncclSuccess; ++p)
    {
        if (p == c->rank) { ownSend = sOff, ownRecv = rOff; }
        else
        {
            if (sendBytes[p]) rc = api->send(s + sOff, sendBytes[p], ncclUint8, p, c->comm, c->ctx->stream);
            if (rc == ncclSuccess && recvBytes[p]) rc = api->recv(r + rOff, recvBytes[p], ncclUint8, p, c->comm, c->ctx->stream);
        }
        sOff += sendBytes[p], rOff += recvBytes[p];
    }
    ncclResult_t end = api->groupEnd();
    if (rc != ncclSuccess) return ncclFail(c, "ncclSend / ncclRecv", rc);
    if (end != ncclSuccess) return ncclFail(c, "ncclGroupEnd", end);
    const size_t own = sendBytes[c->rank] < recvBytes[c->rank] ? sendBytes[c->rank] : recvBytes[c->rank];
    if (own && hipMemcpyAsync(r + ownRecv, s + ownSend, own, hipMemcpyDeviceToDevice, c->ctx->stream) != hipSuccess)
    {
        fail(c->ctx, CSTONE_E_HIP, "all_to_all_v: copy of the rank's own segment failed");
        return 1;
    }
    return 0;
}

} // namespace
} // namespace cship

using namespace cship;

extern "C"
{

int cstone_hip_comm_rccl_unique_id(cstone_hip_ctx* ctx, void* id128_host)
{
    if (!ctx || !id128_host) return fail(ctx, CSTONE_E_ARG, "comm_rccl_unique_id: null argument");
    std::string why;
    RcclApi* api = rccl(&why);
    if (!api) return fail(ctx, CSTONE_E_INTERNAL, "comm_rccl_unique_id: %s", why.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
    ncclResult_t r = api->getUniqueId(static_cast<ncclUniqueId*>(id128_host));
    if (r != ncclSuccess) return fail(ctx, CSTONE_E_INTERNAL, "ncclGetUniqueId: %s", api->getErrorString(r));
    return CSTONE_OK;
}

int cstone_hip_comm_rccl_create(cstone_hip_ctx* ctx, const void* id128_host, int rank, int num_ranks,
                                cstone_hip_comm_rccl** out)
{
    if (!ctx || !id128_host || !out || num_ranks < 1 || rank < 0 || rank >= num_ranks)
        return fail(ctx, CSTONE_E_ARG, "comm_rccl_create: bad argument");
    *out = nullptr;
    std::string why;
    RcclApi* api = rccl(&why);
    if (!api) return fail(ctx, CSTONE_E_INTERNAL, "comm_rccl_create: %s", why.c_str());
    CS_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    __builtin_memcpy(&id, id128_host, sizeof id);
    auto* c = new cstone_hip_comm_rccl;
    c->ctx = ctx, c->rank = rank, c->size = num_ranks;
    ncclResult_t r = api->commInitRank(&c->comm, num_ranks, id, rank);
    if (r != ncclSuccess)
    {
        delete c;
        return fail(ctx, CSTONE_E_INTERNAL, "ncclCommInitRank: %s", api->getErrorString(r));
    }
    *out = c;
    return CSTONE_OK;
}

int cstone_hip_comm_rccl_ops(cstone_hip_comm_rccl* comm, cstone_hip_comm_ops* ops)
{
    if (!comm || !ops) return CSTONE_E_ARG;
    ops->user         = comm;
    ops->all_reduce   = allReduceCb;
    ops->all_gather   = allGatherCb;
    ops->all_to_all_v = allToAllVCb;
    return CSTONE_OK;
}

int cstone_hip_comm_rccl_destroy(cstone_hip_comm_rccl* comm)
{
    if (!comm) return CSTONE_OK;
    if (ctxAlive(comm->ctx)) (void)hipStreamSynchronize(comm->ctx->stream);
    else (void)hipDeviceSynchronize(); // (the context went first: nothing of it is touched)
    RcclApi* api = rccl(nullptr);
    if (api && comm->comm) (void)api->commDestroy(comm->comm);
    delete comm;
    return CSTONE_OK;
}

} // extern "C"

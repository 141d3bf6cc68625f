// RCCL communicator attached to a context (one process per GPU; collectives
// run over xGMI).  The reference has no communication layer; this exists for the
// row-partitioned multi-GPU PCG ("Multi-GPU").
#include <rccl/rccl.h>

#include <cstring>

#include "fv_internal.h"

static_assert(sizeof(ncclUniqueId) == FV_COMM_ID_BYTES, "ncclUniqueId size");

#define FV_NCCL(ctx, call)                                                                                     \
    do {                                                                                                       \
        ncclResult_t r__ = (call);                                                                             \
        if (r__ != ncclSuccess) {                                                                              \
            fv_set_error(ctx, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r__), __FILE__, __LINE__);    \
            return FV_ERR_COMM;                                                                                \
        }                                                                                                      \
    } while (0)

extern "C" int fv_comm_unique_id(char id[FV_COMM_ID_BYTES])
{
    if (!id)
        return FV_ERR_ARG;
    ncclUniqueId uid;
    FV_NCCL(nullptr, ncclGetUniqueId(&uid));
    memcpy(id, &uid, sizeof uid);
    return FV_OK;
}

extern "C" int fv_comm_init(fv_ctx *ctx, int nranks, int rank, const char id[FV_COMM_ID_BYTES])
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm)
        fv_comm_destroy(ctx);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm;
    FV_NCCL(ctx, ncclCommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return FV_OK;
}

extern "C" int fv_comm_destroy(fv_ctx *ctx)
{
    if (!ctx)
        return FV_ERR_ARG;
    if (ctx->comm) {
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->nranks = 1;
    ctx->rank = 0;
    return FV_OK;
}

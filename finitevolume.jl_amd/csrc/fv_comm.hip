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

// Halo exchange of one vector: ship sendbuf (grouped by destination) and receive
// every peer's contribution straight into the halo slots (recv_base, grouped by
// owner).  One grouped ncclSend/ncclRecv batch on `stream`; over xGMI these are
// point-to-point transfers between the two neighbouring slabs of a structured grid.
int fv_comm_halo_exchange(fv_ctx *ctx, const fv_dist *d, const double *sendbuf, double *recv_base, hipStream_t stream)
{
    if (d->nranks <= 1)
        return FV_OK;
    if (!ctx->comm || ctx->nranks != d->nranks || ctx->rank != d->rank) {
        fv_set_error(ctx, "distributed problem (%d ranks) without a matching communicator: call fv_comm_init first", d->nranks);
        return FV_ERR_COMM;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    FV_NCCL(ctx, ncclGroupStart());
    int64_t soff = 0, roff = 0;
    for (int q = 0; q < d->nranks; q++) {
        const int64_t sc = d->send_counts[(size_t)q], rcnt = d->recv_counts[(size_t)q];
        if (q != d->rank) {
            if (sc > 0)
                FV_NCCL(ctx, ncclSend(sendbuf + soff, (size_t)sc, ncclDouble, q, comm, stream));
            if (rcnt > 0)
                FV_NCCL(ctx, ncclRecv(recv_base + roff, (size_t)rcnt, ncclDouble, q, comm, stream));
        }
        soff += sc;
        roff += rcnt;
    }
    FV_NCCL(ctx, ncclGroupEnd());
    return FV_OK;
}

int fv_comm_allreduce_sum(fv_ctx *ctx, const fv_dist *d, double *buf, int count, hipStream_t stream)
{
    if (d->nranks <= 1)
        return FV_OK;
    if (!ctx->comm || ctx->nranks != d->nranks) {
        fv_set_error(ctx, "distributed problem (%d ranks) without a matching communicator: call fv_comm_init first", d->nranks);
        return FV_ERR_COMM;
    }
    FV_NCCL(ctx, ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, stream));
    return FV_OK;
}

// RCCL communicator attached to a context (one process per GPU; collectives
// run over xGMI).  The reference has no communication layer; this exists for the
// row-partitioned multi-GPU PCG ("Multi-GPU").
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

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

// between ncclGroupStart and ncclGroupEnd: a failing call must not leave the group open behind it
#define FV_NCCL_IN_GROUP(ctx, call)                                                                            \
    do {                                                                                                       \
        ncclResult_t r__ = (call);                                                                             \
        if (r__ != ncclSuccess) {                                                                              \
            fv_set_error(ctx, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r__), __FILE__, __LINE__);    \
            (void)ncclGroupEnd();                                                                              \
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

// ------------------------------------------------------------------ loopback transport (one device, several ranks)
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so a one-GPU box cannot run the multi-rank protocol
// through it.  For rehearsals and tests, N host threads of ONE process, each with its own context on the same
// device, can form a local group instead: halos move by device-to-device copies between the ranks' buffers and the
// scalar reductions are summed on the host in rank order.  Same call sequence, same plan, same kernels as the RCCL
// path; only the transport differs (and nothing overlaps).  Not meant for production runs.
namespace {
struct LocalGroup {
    int nranks = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    int members = 0;
    std::vector<const double *> sendbuf;
    std::vector<const fv_dist *> plan;
    std::vector<std::vector<double>> red;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return generation != gen; });
    }
};
std::mutex g_groups_mutex;
std::map<int, LocalGroup *> g_groups;
} // namespace

extern "C" int fv_comm_init_local(fv_ctx *ctx, int nranks, int rank, int group_id)
{
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks)
        return FV_ERR_ARG;
    fv_comm_destroy(ctx);
    std::lock_guard<std::mutex> lk(g_groups_mutex);
    LocalGroup *&g = g_groups[group_id];
    if (!g) {
        g = new LocalGroup();
        g->nranks = nranks;
        g->sendbuf.assign((size_t)nranks, nullptr);
        g->plan.assign((size_t)nranks, nullptr);
        g->red.assign((size_t)nranks, std::vector<double>(8, 0.0));
    }
    if (g->nranks != nranks) {
        fv_set_error(ctx, "local group %d was created for %d ranks, not %d", group_id, g->nranks, nranks);
        return FV_ERR_ARG;
    }
    g->members++;
    ctx->local_group = g;
    ctx->local_group_id = group_id;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return FV_OK;
}

static int local_halo_exchange(fv_ctx *ctx, const fv_dist *d, const double *sendbuf, double *recv_base, hipStream_t stream)
{
    LocalGroup *g = static_cast<LocalGroup *>(ctx->local_group);
    FV_HIP(ctx, hipStreamSynchronize(stream)); // my packed values are in place
    g->sendbuf[(size_t)d->rank] = sendbuf;
    g->plan[(size_t)d->rank] = d;
    g->barrier();
    int64_t roff = 0;
    for (int q = 0; q < d->nranks; q++) {
        const int64_t rcnt = d->recv_counts[(size_t)q];
        if (q != d->rank && rcnt > 0) {
            const fv_dist *dq = g->plan[(size_t)q];
            int64_t off = 0; // where rank q keeps what it sends to me: its send list is grouped by destination
            for (int t = 0; t < d->rank; t++)
                off += dq->send_counts[(size_t)t];
            if (dq->send_counts[(size_t)d->rank] != rcnt) {
                fv_set_error(ctx, "loopback halo exchange: rank %d sends %lld values to rank %d, which expects %lld", q,
                             (long long)dq->send_counts[(size_t)d->rank], d->rank, (long long)rcnt);
                g->barrier();
                return FV_ERR_COMM;
            }
            FV_HIP(ctx, hipMemcpyAsync(recv_base + roff, g->sendbuf[(size_t)q] + off, (size_t)rcnt * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        roff += rcnt;
    }
    FV_HIP(ctx, hipStreamSynchronize(stream));
    g->barrier(); // nobody repacks before everybody has copied
    return FV_OK;
}

static int local_allreduce(fv_ctx *ctx, const fv_dist *d, double *buf, int count, hipStream_t stream)
{
    LocalGroup *g = static_cast<LocalGroup *>(ctx->local_group);
    if (count < 0)
        return FV_ERR_ARG;
    std::vector<double> &mine = g->red[(size_t)d->rank];
    if ((int)mine.size() < count) // (the few-double sums of the PCG fit the initial 8; the gathered AMG levels reduce whole coarse vectors)
        mine.resize((size_t)count);
    FV_HIP(ctx, hipMemcpyAsync(mine.data(), buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, stream));
    FV_HIP(ctx, hipStreamSynchronize(stream));
    g->barrier();
    std::vector<double> sum((size_t)count, 0.0);
    for (int q = 0; q < d->nranks; q++) { // rank order: every rank gets the same bits
        const double *theirs = g->red[(size_t)q].data();
        for (int k = 0; k < count; k++)
            sum[(size_t)k] += theirs[k];
    }
    g->barrier(); // everybody has read before anybody overwrites (or regrows) its slot in the next reduction
    FV_HIP(ctx, hipMemcpyAsync(buf, sum.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, stream));
    FV_HIP(ctx, hipStreamSynchronize(stream)); // `sum` is on this stack frame
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
    if (ctx->local_group) {
        std::lock_guard<std::mutex> lk(g_groups_mutex);
        LocalGroup *g = static_cast<LocalGroup *>(ctx->local_group);
        if (--g->members == 0) {
            g_groups.erase(ctx->local_group_id);
            delete g;
        }
        ctx->local_group = nullptr;
    }
    ctx->nranks = 1;
    ctx->rank = 0;
    return FV_OK;
}

// Halo exchange of one vector: ship sendbuf (grouped by destination) and receive
// every peer's contribution straight into the halo slots (recv_base, grouped by
// owner).  One grouped ncclSend/ncclRecv batch on `stream`; over xGMI these are
// point-to-point transfers between the two neighbouring slabs of a structured grid.
// ------------------------------------------------------------------ diagnostics of a distributed step
// Where does a step's time go on a rank that waits for others?  With fv_comm_diag(ctx, 1) every all-reduce, every halo
// exchange, the compute stream's wait for the halo and the two SpMV passes of a row block are bracketed by HIP events on the
// stream they run on; fv_comm_diag_get sums them.  An event between two launches is a barrier (~10 us each at 10^8 cells): for
// a diagnosis pass after the timed region, not for the timed region itself.
constexpr size_t FV_DIAG_MAX_EVENTS = 1 << 14;
int fv_diag_mark(fv_ctx *ctx, int cat, hipStream_t stream)
{
    if (!ctx->diag || cat < 0 || cat >= 5 || ctx->diag_ev[cat].size() >= FV_DIAG_MAX_EVENTS)
        return FV_OK;
    hipEvent_t e;
    FV_HIP(ctx, hipEventCreate(&e));
    ctx->diag_ev[cat].push_back(e);
    FV_HIP(ctx, hipEventRecord(e, stream));
    return FV_OK;
}

static int diag_harvest(fv_ctx *ctx)
{
    for (int cat = 0; cat < 5; cat++) {
        std::vector<hipEvent_t> &v = ctx->diag_ev[cat];
        for (size_t k = 0; k + 1 < v.size(); k += 2) {
            float ms = 0.f;
            if (hipEventSynchronize(v[k + 1]) == hipSuccess && hipEventElapsedTime(&ms, v[k], v[k + 1]) == hipSuccess) {
                ctx->diag_ms[cat] += ms;
                ctx->diag_n[cat]++;
            }
        }
        for (hipEvent_t e : v)
            (void)hipEventDestroy(e);
        v.clear();
    }
    return FV_OK;
}

extern "C" int fv_comm_diag(fv_ctx *ctx, int enable)
{
    if (!ctx)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(diag_harvest(ctx));
    if (enable)
        for (int cat = 0; cat < 5; cat++) {
            ctx->diag_ms[cat] = 0.0;
            ctx->diag_n[cat] = 0;
        }
    ctx->diag = enable != 0;
    return FV_OK;
}

extern "C" int fv_comm_diag_get(fv_ctx *ctx, double total_ms[5], int64_t counts[5])
{
    if (!ctx || !total_ms || !counts)
        return FV_ERR_ARG;
    FV_HIP(ctx, hipSetDevice(ctx->device));
    FV_TRY(diag_harvest(ctx));
    for (int cat = 0; cat < 5; cat++) {
        total_ms[cat] = ctx->diag_ms[cat];
        counts[cat] = ctx->diag_n[cat];
    }
    return FV_OK;
}

int fv_comm_halo_exchange(fv_ctx *ctx, const fv_dist *d, const double *sendbuf, double *recv_base, hipStream_t stream)
{
    ctx->n_halo++;
    if (d->nranks <= 1)
        return FV_OK;
    FV_TRY(fv_diag_mark(ctx, 1, stream));
    struct MarkEnd { // the closing event, on every way out
        fv_ctx *c;
        hipStream_t s;
        ~MarkEnd() { (void)fv_diag_mark(c, 1, s); }
    } mark_end{ctx, stream};
    if (ctx->local_group && ctx->nranks == d->nranks && ctx->rank == d->rank)
        return local_halo_exchange(ctx, d, sendbuf, recv_base, stream);
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
                FV_NCCL_IN_GROUP(ctx, ncclSend(sendbuf + soff, (size_t)sc, ncclDouble, q, comm, stream));
            if (rcnt > 0)
                FV_NCCL_IN_GROUP(ctx, ncclRecv(recv_base + roff, (size_t)rcnt, ncclDouble, q, comm, stream));
        }
        soff += sc;
        roff += rcnt;
    }
    FV_NCCL(ctx, ncclGroupEnd());
    return FV_OK;
}

// Health check of the transport: every rank sends `count` doubles to the next rank and receives from the previous one
// (to itself in a one-rank communicator) in one ncclSend/ncclRecv group on the halo stream — the call pattern of
// fv_comm_halo_exchange — and all ranks sum a vector with ncclAllReduce; *ok = 1 when both arrive as expected.
extern "C" int fv_comm_selftest(fv_ctx *ctx, int64_t count, int *ok)
{
    if (!ctx || !ok || count < 1)
        return FV_ERR_ARG;
    *ok = 0;
    if (!ctx->comm) {
        fv_set_error(ctx, "fv_comm_selftest: no RCCL communicator (fv_comm_init)");
        return FV_ERR_COMM;
    }
    FV_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> host((size_t)count);
    for (int64_t i = 0; i < count; i++)
        host[(size_t)i] = 1000.0 * ctx->rank + (double)(i % 997);
    DevBuf<double> a, b;
    FV_TRY(a.alloc(ctx, (size_t)count));
    FV_TRY(b.alloc(ctx, (size_t)count));
    FV_HIP(ctx, fv_memcpy_sync(ctx, a.p, host.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    // cleared on the stream the receive runs on: stream2 is non-blocking, a memset on the null stream could land after the data
    FV_HIP(ctx, hipMemsetAsync(b.p, 0, (size_t)count * sizeof(double), ctx->stream2));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const int next = (ctx->rank + 1) % ctx->nranks, prev = (ctx->rank + ctx->nranks - 1) % ctx->nranks;
    FV_NCCL(ctx, ncclGroupStart());
    FV_NCCL_IN_GROUP(ctx, ncclSend(a.p, (size_t)count, ncclDouble, next, comm, ctx->stream2));
    FV_NCCL_IN_GROUP(ctx, ncclRecv(b.p, (size_t)count, ncclDouble, prev, comm, ctx->stream2));
    FV_NCCL(ctx, ncclGroupEnd());
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    std::vector<double> got((size_t)count);
    FV_HIP(ctx, fv_memcpy_sync(ctx, got.data(), b.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    bool good = true;
    for (int64_t i = 0; i < count; i++)
        good = good && got[(size_t)i] == 1000.0 * prev + (double)(i % 997);
    FV_NCCL(ctx, ncclAllReduce(a.p, a.p, (size_t)count, ncclDouble, ncclSum, comm, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FV_HIP(ctx, fv_memcpy_sync(ctx, got.data(), a.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    const double ranks = (double)ctx->nranks;
    for (int64_t i = 0; i < count; i++)
        good = good && got[(size_t)i] == 1000.0 * (ranks * (ranks - 1.0) / 2.0) + ranks * (double)(i % 997);
    *ok = good ? 1 : 0;
    return FV_OK;
}

int g_comm_single_rank_collectives = 0; // fv_tune key 21: issue the all-reduces of a one-rank run too (tests: the RCCL call path on one GPU)

extern "C" int fv_comm_stats(fv_ctx *ctx, int64_t *allreduces, int64_t *halo_exchanges, int reset)
{
    if (!ctx)
        return FV_ERR_ARG;
    if (allreduces)
        *allreduces = ctx->n_allreduce;
    if (halo_exchanges)
        *halo_exchanges = ctx->n_halo;
    if (reset)
        ctx->n_allreduce = ctx->n_halo = 0;
    return FV_OK;
}

int fv_comm_allreduce_sum(fv_ctx *ctx, const fv_dist *d, double *buf, int count, hipStream_t stream)
{
    ctx->n_allreduce++; // counted also for one rank (where nothing is sent): the call pattern is what tests look at
    if (d->nranks <= 1 && !(g_comm_single_rank_collectives && ctx->comm && ctx->nranks == 1))
        return FV_OK;
    if (ctx->local_group && ctx->nranks == d->nranks) {
        FV_TRY(fv_diag_mark(ctx, 0, stream));
        const int rc = local_allreduce(ctx, d, buf, count, stream);
        FV_TRY(fv_diag_mark(ctx, 0, stream));
        return rc;
    }
    if (!ctx->comm || ctx->nranks != d->nranks) {
        fv_set_error(ctx, "distributed problem (%d ranks) without a matching communicator: call fv_comm_init first", d->nranks);
        return FV_ERR_COMM;
    }
    FV_TRY(fv_diag_mark(ctx, 0, stream));
    FV_NCCL(ctx, ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, stream));
    FV_TRY(fv_diag_mark(ctx, 0, stream));
    return FV_OK;
}

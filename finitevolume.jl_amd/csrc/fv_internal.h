// Internal declarations shared by the HIP translation units of libfvhip.so.
// gfx950 (MI355X, wave64) only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "fvhip.h"
#include "fv_tune.h"

constexpr int FV_BLOCK = 256;        // 4 waves of 64
constexpr int FV_MAX_PARTIALS = 2048; // 8 blocks/CU x 256 CUs: one partial per block (SpMV kernels: all blocks resident)
#ifndef FV_VEC_MAX_BLOCKS
#define FV_VEC_MAX_BLOCKS 2048
#endif
constexpr int FV_VEC_PARTIALS = FV_VEC_MAX_BLOCKS; // blocks (= partial sums) of the streaming vector kernels
constexpr int FV_VEC_PAD = 32;        // slack behind every vector (double2 tails read past n)

struct fv_ctx {
    int device = 0;
    int num_cus = 256;
    int64_t total_mem = 0;
    std::string name;
    hipStream_t stream = nullptr;  // compute
    hipStream_t stream2 = nullptr; // halo / collectives
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_halo = nullptr, ev_comp = nullptr;
    unsigned long long err_seq = 0; // when `err` was set (fv_last_error prefers a later context-free message)
    void *pinned = nullptr; // small pinned scratch for scalar read-back
    size_t pinned_bytes = 0;
    std::string err;
    // RCCL
    void *comm = nullptr;
    void *local_group = nullptr; // loopback transport for single-device rehearsals (fv_comm_init_local)
    int local_group_id = 0;
    int nranks = 1, rank = 0;
    int64_t n_allreduce = 0, n_halo = 0; // collectives issued through this context (fv_comm_stats)
    int opt_reorder = -1; // FV_OPT_REORDER of this context; -1: the process-wide default (fv_tune key 31)
    int opt_lean = 2;     // FV_OPT_LEAN_SETUP: 0 never, 1 every regular-grid problem that can, 2 those whose CSR would not fit int32 indices
    // fv_comm_diag: HIP event pairs around the pieces of a distributed step — [0] all-reduces, [1] the halo exchange on the second
    // stream, [2] the compute stream's stall at the wait for the halo, [3] interior SpMV pass, [4] boundary pass
    bool diag = false;
    std::vector<hipEvent_t> diag_ev[5]; // pairs, in order of recording
    double diag_ms[5] = {0, 0, 0, 0, 0};
    int64_t diag_n[5] = {0, 0, 0, 0, 0};
};

void fv_set_error(fv_ctx *ctx, const char *fmt, ...);
// fv_spmv.hip: the wave-stream CSR SpMV on caller-owned arrays (two padding entries behind vals / colind)
int fv_csr_stream_spmv(fv_ctx *ctx, int64_t n, const int32_t *rowptr, const int32_t *colind, const double *vals, const double *x, double *y,
                       const double *D, double sigma);

#define FV_HIP(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e__ = (call);                                                                                \
        if (e__ != hipSuccess) {                                                                                \
            fv_set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);      \
            return FV_ERR_HIP;                                                                                  \
        }                                                                                                       \
    } while (0)

#define FV_TRY(expr)                                                                                            \
    do {                                                                                                        \
        int rc__ = (expr);                                                                                      \
        if (rc__ != FV_OK)                                                                                      \
            return rc__;                                                                                        \
    } while (0)

#define FV_LAUNCH_CHECK(ctx) FV_HIP(ctx, hipGetLastError())

// Large arrays are handed out at staggered offsets inside their allocations (an experiment of round 2, frozen at zero: bytes per step of the
// stagger, 0 = off): the streaming kernels walk up to nine arrays at the same index at the same time, and when all of
// them start at the same offset of their (2 MiB-aligned) allocations they land on the same HBM channels together.
extern int g_alloc_skew_bytes;
extern int g_alloc_skew_count;

// Device allocations of DevBuf go through these two (fv_ctx.hip).  Between fv_pool_begin / fv_pool_end on the calling thread a
// released block is kept and handed to a later request it fits (a set-up phase such as the AMG hierarchy allocates and frees
// hundreds of scratch arrays of shrinking size: hipMalloc / hipFree — a device-wide synchronisation each — cost more than its
// kernels); everything such a phase does runs on the context's one stream, which orders the re-use.
hipError_t fv_dev_malloc(void **p, size_t bytes);
void fv_dev_free(void *p);
void fv_pool_begin();
void fv_pool_end();

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    void *base = nullptr; // what hipMalloc returned (p = base + stagger)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (base)
            fv_dev_free(base);
        p = nullptr;
        base = nullptr;
        n = 0;
    }
    int alloc(fv_ctx *ctx, size_t count)
    {
        release();
        if (count == 0)
            count = 1;
        size_t skew = 0;
        if (g_alloc_skew_bytes > 0 && count * sizeof(T) >= ((size_t)1 << 22)) // arrays of 4 MiB and more
            skew = (size_t)(g_alloc_skew_count++ % 16) * (size_t)g_alloc_skew_bytes;
        hipError_t e = fv_dev_malloc(&base, count * sizeof(T) + skew);
        if (e != hipSuccess) {
            p = nullptr;
            base = nullptr;
            fv_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
            return FV_ERR_NOMEM;
        }
        p = reinterpret_cast<T *>(static_cast<char *>(base) + skew);
        n = count;
        return FV_OK;
    }
    void swap(DevBuf &o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(base, o.base);
    }
    int zero(fv_ctx *ctx) { FV_HIP(ctx, hipMemsetAsync(p, 0, n * sizeof(T), ctx->stream)); return FV_OK; }
};

// never 0: every kernel bounds-checks, and a zero-sized grid is a launch error
inline unsigned fv_blocks(int64_t n, int per_block = FV_BLOCK)
{
    const int64_t b = (n + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : b);
}

constexpr int FV_AUTO_SWITCH_ITERS = 50;

// Device-side PCG scalars (one instance per problem).
struct PcgScalars {
    double rz[2];  // r.M^-1.r, indexed by iteration parity
    double rr;     // ||r||^2 (recurrence)
    double tol2;   // (rtol*||b||)^2
    double bnorm2; // ||b||^2
    double pq;     // last p.Ap (diagnostic)
    int32_t iters;
    int32_t done;  // 0 running, 1 converged, 2 breakdown (p.Ap <= 0 or NaN), 3 a chain of unpolled one-iteration steps broke here
    int32_t chain_step; // done == 3: index (within the burst) of the step that needs more than its one iteration
    uint32_t zero_mask; // bursts of chained steps: bit j set = step j of the burst was already converged at its set-up (0 iterations)
    double tol2x[2]; // tol2 of the chained steps by parity of their index (row-block bursts: a step's verdict is taken while the next step's scalars are written)    // the many-iteration loop through the fused kernel (fv_fused_iteration): the x-update of iteration i is applied by the pass of
    // iteration i + 1, which reads that direction anyway.  xlag >= 0: the iteration whose x += alpha_last * lag_p is still to be
    // applied (the loop's last one: pcg_xflush_kernel does it once the loop has stopped); -1: x is up to date.
    double alpha_last;
    const double *lag_p;
    int32_t xlag, pad_;
    // the single-launch solver (fv_small.hip): the number of the launch that wrote this block — a launch that gave up at its grid
    // barrier leaves the previous number, so one copy of the block tells the host both the result and that there is one
    uint32_t small_seq, pad2_;
};

// Plan and buffers of a row block in a distributed run (built by fv_dist_setup).
struct fv_dist {
    int nranks = 1, rank = 0;
    int64_t lo = 0, hi = 0, nhalo = 0, nsend = 0, entry_lo = 0, n_int = 0, n_bnd = 0;
    std::vector<int64_t> bounds, recv_counts, send_counts; // per rank
    DevBuf<int32_t> halo_cols;  // global column of every halo slot, ascending
    DevBuf<int32_t> send_idx;   // local rows to ship, grouped by destination rank, ascending
    DevBuf<int32_t> groups_int; // 64-row groups that touch no halo slot
    DevBuf<int32_t> groups_bnd; // the others
    DevBuf<double> sendbuf;
    DevBuf<double> red;   // 8 scalars: all-reduce buffer
    // interior / boundary groups split by storage form (built at the first distributed SpMV)
    DevBuf<int32_t> int_dia, int_csr, bnd_dia, bnd_csr;
    int64_t n_int_dia = 0, n_int_csr = 0, n_bnd_dia = 0, n_bnd_csr = 0;
    int64_t int_lo = 0, int_hi = 0; // the interior groups as a slice range [lo, hi) when they are contiguous (else empty)
    bool split_built = false;
    int64_t fused_bursts = 0; // bursts since the call began (the agreement is renewed every 16th)
    int fused_agreed = -1; // per fv_dist_run_fixed call: -1 not yet asked, 1 every rank can run the fused step, 0 at least one cannot
};

struct fv_amg; // fv_amg.hip
void fv_amg_free(fv_amg *a);

// the distinct values of the storage term D, when there are few (storage_form in fv_pcg.hip), and how K2S is handed D:
// its stream, or (D = nullptr) one-byte codes into the table, or (code = nullptr too) the single value tab.v[0]
constexpr int FV_STORAGE_CODES = 16;
struct StorageTable {
    double v[FV_STORAGE_CODES];
};
// the three upper diagonals of the symmetric copy as 5-bit codes per row (fv_matrix_codes): tables of their distinct values
constexpr int FV_MATRIX_CODES = 32;
struct MatrixTables {
    double v[3 * FV_MATRIX_CODES];
};
struct StorageArg {
    const double *D;
    const uint8_t *code;
    StorageTable tab;
};

// Rows of a regular-grid operator formed on the fly (fv_lean.h): geometry, node maps, the caller's conductivities, and the diagonal
// as the assembly left it in diagA (+ sigma D where a shift is folded in)
struct GridRows {
    int64_t n1, n2, n3;
    double dx, dy, dz;
    const int32_t *nodemap, *f2n;
    const double *K;
    const int64_t *meta;
    int64_t nK;
    int logt;
    const double *diagA, *D;
    double sigma;
};
constexpr int DIA_K = 8; // distinct column offsets a 64-row slice of the sliced-DIA form can have
struct fv_problem;
void fv_vec_release_spares(fv_problem *p); // fv_place.hip

struct fv_problem {
    fv_ctx *ctx = nullptr;
    int64_t N = 0, F = 0, n = 0, nnz = 0, ndir = 0, E = 0;
    int64_t nhalo = 0;             // row block of a distributed operator: halo slots appended to every vector
    fv_dist *dist = nullptr;       // plan + buffers of the distributed run (fv_dist.hip)
    bool from_grid = false, from_csc = false, assembled = false, transient_ready = false;
    int64_t ns[3] = {0, 0, 0};
    int64_t slab_lo = -1, slab_hi = -1; // fv_problem_create_regulargrid_slab: the planes whose rows are complete
    // FV_OPT_LEAN_SETUP: a regular-grid problem without face arrays, incident lists and CSR — the storage forms of the solver are
    // filled from rows formed on the fly (fv_lean.hip); what needs the CSR or the faces (fv_get_csc, AMG, row blocks, gradients) fails loudly
    bool lean = false;
    double lean_d[3] = {0, 0, 0}; // grid spacing, as regulargrid_kernel forms it (axis[1] - axis[0])
    double lean_mins[3] = {0, 0, 0}, lean_maxs[3] = {0, 0, 0}; // the box (fv_problem_get_grid generates the face list again when asked for it)
    DevBuf<double> lean_K;        // the conductivities of the last fv_assemble as handed over (1, or one per face / per metaindex target)
    DevBuf<int64_t> lean_meta;    // ... and its metaindex (1-based), when one came
    int64_t lean_nK = 0;
    int lean_logt = 0;
    int dia_alloc = 0;            // lean: dia_vals / dia_pos hold 0 nothing yet, 1 the symmetric form's rest slices only, 2 every DIA slice

    // mesh (0-based int32 on device)
    DevBuf<int32_t> node1, node2;
    DevBuf<double> aol, gridvol;
    DevBuf<int32_t> nodemap; // N: >= 0 free index, < 0: -(dirichlet position + 1), last occurrence wins
    DevBuf<int32_t> f2n;     // n: free index -> node
    DevBuf<int32_t> dnodes0; // ndir: Dirichlet nodes, 0-based, caller order
    // Locality re-numbering of the free cells (fv_reorder_free, face-list meshes that arrive numbered at random): every
    // device array indexed by free cell — nodemap's values, f2n, the CSR, b, D, the state vectors — is in the internal
    // numbering; perm[canonical] = internal, iperm[internal] = canonical, canonical = rank among the free nodes
    // (FiniteVolume.jl:32-44).  Free-indexed data crosses the C ABI in the canonical numbering (fv_free_in / fv_free_out,
    // fv_get_csc, fv_problem_get_free_maps), so callers never see the difference.
    DevBuf<int32_t> perm, iperm;
    DevBuf<double> stage; // n doubles: staging of permuted transfers
    bool reordered = false;
    double reorder_mean_before = 0.0, reorder_mean_after = 0.0, reorder_seconds = 0.0;

    // symbolic structure of assembleA
    DevBuf<int32_t> incptr;    // n+1: incident (face,end) entries per free row
    DevBuf<uint32_t> inc_face; // E: face<<1 | end, ascending per row == face order
    DevBuf<uint32_t> inc_slot; // E: bit31 = first contribution to its slot; low bits = slot - rowptr[row]; 0x7fffffff = other end is Dirichlet
    DevBuf<int32_t> rowptr, colind, diagpos;
    // SpMV traversal: 64-row groups listed band by band, plane after plane (empty = natural order)
    DevBuf<int32_t> group_order;
    bool order_built = false;
    int64_t order_stride = 0;

    // sliced-DIA copy of the grid-like slices (fv_pcg.hip): per 64-row slice the distinct column offsets
    // (0 = slice stays with the CSR kernel), the lists of DIA slices / CSR groups, lane-major values
    DevBuf<uint8_t> sl_noff;
    DevBuf<int32_t> sl_off, dia_list, csr_list, dia_pos;
    DevBuf<int32_t> dia_list_ord; // the DIA slices in traversal order (plane-blocked), when the operator has a plane stride
    DevBuf<double> dia_vals;
    int64_t ndia = 0, ncsr_groups = 0, dia_epoch = -1, dia_nblocks = 0; // (dia_nblocks: blocks of 64 values of all DIA slices)
    // SELL-64 copy of the 64-row groups the CSR kernel would serve (irregular meshes after the locality re-numbering): per group
    // `width` lane-major blocks of 64 values + 64 sixteen-bit column offsets relative to the row, the diagonal first (fv_spmv.hip)
    DevBuf<double> sell_vals;
    DevBuf<int16_t> sell_dcol;
    DevBuf<int32_t> sell_ptr, sell_list, sell_rest; // first block of every group; groups in the form / CSR groups that are not
    DevBuf<uint8_t> sell_w;
    int64_t sell_n = 0, sell_nrest = 0, sell_blocks = 0, sell_vals_epoch = -1;
    double sell_tag = 0.0;
    int sell_state = -1; // -1 not looked at, 0 not usable, 1 built
    double dia_tag = 0.0;
    bool dia_built = false;
    bool dia_partial = false; // dia_vals holds only the slices of sym_rest (the symmetric marching kernel does the others)

    // symmetric DIA copy of a plane-structured operator (fv_spmv.hip, symdia_*): the stored diagonal and the three upper
    // diagonals at the operator-wide offsets sym_d[0] < sym_d[1] < sym_d[2] (= the plane stride) as four zero-padded arrays
    // of sym_ld doubles; row 0 of each array sits sym_front doubles into it.  The lower arm a(i, i-d) is read as the upper
    // value of row i-d (A is symmetric bit for bit, checked when the copy is filled).
    DevBuf<double> sym_vals;
    DevBuf<uint8_t> sym_ok;   // per 64-row slice: every stored offset of the slice is 0 or +-sym_d[k]
    DevBuf<int32_t> sym_rest; // the DIA slices where that does not hold (slice-by-slice kernel)
    int64_t sym_d[3] = {0, 0, 0}, sym_ld = 0, sym_front = 0, sym_nrest = 0, sym_epoch = -1;
    bool sym_big = false; // more than 2^32 bytes per array: only the kernels with 64-bit row indices / plane bases may serve the copy
    double sym_tag = 0.0;
    // slices whose diagonal the symmetric kernel re-derives from the six arms it holds (bit 1 of sym_ok; symdia_rowsum_kernel)
    int64_t sym_nderived = 0;
    int sym_rowsum_switch = -1;   // fv_tune key 37 as it was when the flags were set
    int sym_shift_mode = 0;       // 0: the derived diagonal carries no shift; 1: + sym_shift.v[code of the row]; 2: + sym_shift.v[0]
    StorageTable sym_shift = {};  // sigma x the distinct values of D, for the sigma folded into the copy
    // ... and, where each of the three upper diagonals takes at most FV_MATRIX_CODES distinct values (a homogeneous conductivity on a
    // regular grid), one 16-bit word per row: U1 | U2 << 5 | U3 << 10 as codes into sym_mtab.  sym_mcode_n: 0 not applicable, > 0 built
    // (for the copy's current epoch and tag), -1 not looked at yet
    DevBuf<uint16_t> sym_mcode;
    MatrixTables sym_mtab = {};
    int sym_mcode_n = -1;
    int64_t sym_mcode_epoch = -1;
    int last_form = -1; // FV_SPMV_* of the most recent spmv_apply (fv_spmv_form)
    // the fused step's chunk kernel (fv_fused.hip): per row, storage code | diagonal code << 4 (build_chunk_codes, fv_spmv.hip); built with
    // the symmetric copy's values (same assembly, same folded shift, same storage codes); kc_state: 0 not applicable, 1 built
    DevBuf<uint8_t> sym_reg;       // per 64-row slice: offsets all among 0, +-sym_d[k] whatever the marching kernels' windows need (the first / last plane too)
    DevBuf<int32_t> sym_rest_irr;  // the DIA slices that are not even that (the chunk traversal with kc_ends leaves only these to the slice-by-slice kernel)
    int64_t sym_nrest_irr = 0;
    bool kc_ends = false;          // the chunk traversal's centre planes include the first and the last plane
    int kc_ends_switch = -1;       // fv_tune key 60's end-plane choice as it was when the codes were built
    DevBuf<uint8_t> kc_code;
    StorageTable kc_dtab = {};
    int kc_state = 0, kc_ndiag = 0; // kc_state 2: the matrix as doubles — bit 4 = diagonal from its stream, bit 5 = product formed by the traversal (chunk_code_stream_kernel)
    int64_t kc_nstream = 0;         // ... rows whose diagonal that traversal loads from the stored diagonal
    int sym_state = -1; // -1 not looked at yet, 0 not applicable (no such structure, or not symmetric), 1 built

    // numeric
    DevBuf<double> cond, vals, b, diagA, dheads;

    // fixed-dt runs: copy of vals with sigma*D folded into the stored diagonal
    DevBuf<double> vals_shifted;
    double shifted_sigma = 0.0;
    int64_t shifted_epoch = -1, assemble_epoch = 0;
    int fold_ok = -1; // -1 unknown, 0 some free row stores no diagonal, 1 ok
    bool minv_valid = false; // cached Jacobi diagonal 1/(diag(A) + sigma D)
    double minv_sigma = 0.0;
    int64_t minv_epoch = -1;

    // transient
    double Ss = 1.0;
    DevBuf<double> D; // Ss * volumes[free]
    std::vector<double *> slots;
    std::vector<void *> slot_bases; // what hipMalloc returned for every state vector ever created (slots are staggered and swapped)
    std::vector<char> slot_used;
    int64_t storage_epoch = 0;  // bumped by fv_transient_begin (D changed)
    int32_t pingpong_slot = -1; // hidden state vector the fixed-dt run alternates with the caller's slot
    // A fixed-dt run that ends in the carried / speculated regime leaves everything the next call needs to go on as if
    // the two were one (the final residual in r, the prepared set-up of the next step, the previous state in the
    // ping-pong vector): a caller stepping in chunks then pays the fresh residual (an SpMV + a vector pass) once per
    // refresh period instead of once per call.  Valid only for the same slot, dt, assembly, storage and switches, and
    // only until anything else touches the state or the solver's workspace (resume_ok is cleared by fv_pcg_solve's other
    // callers and by every state setter).
    struct FixedRunResume {
        bool ok = false;
        int32_t slot = -1;
        double dt = 0.0, rtol = 0.0; // (rtol too: under a tighter tolerance a state that sat converged at its set-up iterates
                                     // again, and after zero-iteration steps of a burst the direction vectors may sit swapped)
        int64_t assemble_epoch = -1, storage_epoch = -1;
        const double *prev = nullptr; // state the last solve started from
        int64_t steps_since_refresh = 0;
        int refresh = 0, speculate = 0;
    } resume;

    // preconditioner of the PCG: FV_PRECOND_JACOBI (fused into the vector kernels) or FV_PRECOND_AMG (fv_amg.hip)
    int precond = 0;
    // FV_PRECOND_AUTO in implicit steps: Jacobi until a step needs more than FV_AUTO_SWITCH_ITERS iterations, the AMG
    // V-cycle from then on (one V-cycle iteration costs ~3 Jacobi-PCG iterations, measured); reset by fv_precond_set.
    bool auto_steps_amg = false;
    bool amg_gathered = false; // FV_PRECOND_AMG_GATHERED: row blocks share the coarse levels of the whole operator (fv_amg.hip)
    fv_amg *amg = nullptr;

    // PCG workspace
    DevBuf<double> r, pvec, q, minv, rhs, tmp;
    DevBuf<double> cg_u;     // one-reduction CG on row blocks: u = M^-1 r (the SpMV's input there; s = A p lives in pnext)
    DevBuf<double> cg_scal;  // its device scalars: alpha, beta, gamma
    DevBuf<double> pnext;    // p' of a speculatively prepared next step (swapped with pvec when used)
    bool spec_valid = false; // r, pnext and the upper halves of part_rz/rr/bb hold the next step's set-up
    int spec_extra_bb = 0;   // extra rhs.rhs partials behind the speculative half (sparse-b gather)
    DevBuf<int32_t> bnz_idx; // rows where the assembled b is non-zero
    int64_t bnz_count = 0, bnz_epoch = -1;
    // D as one-byte codes into a table of its distinct values (storage_form in fv_pcg.hip): dcode_n = 0 too many values (keep
    // the stream), 1 uniform, else the table size.  Valid for (dcode_epoch, dcode_ptr) = (storage_epoch, D.p)
    DevBuf<uint8_t> dcode;
    StorageTable dtable = {};
    int dcode_n = 0;
    int64_t dcode_epoch = -1;
    const double *dcode_ptr = nullptr;
    int32_t k2s_bytes = 0; // bytes per row the most recent K2S launch streams (fv_update_form)
    // z-form K2S (fv_pcg.hip): where the residual of the state between two steps lives: 0 = in r; 1 / 2 = Jacobi-scaled in
    // pvec / pnext (r = that vector / M^-1, r itself stale)
    int z_where = 0;
    // M^-1 as one-byte codes (fv_minv_codes) and whether the most recent many-iteration loop used them
    DevBuf<uint8_t> mvcode;
    StorageTable mvtable{};
    int mvcode_n = 0;
    int64_t mvcode_epoch = -1, mvcode_sepoch = -1;
    double mvcode_sigma = 0.0;
    const double *mvcode_ptr = nullptr;
    bool loop_minv_coded = false;
    int zf_minv_bits = 3;    // bit 0: M^-1 > 0 fails on some row for sigma > 0, bit 1: for sigma = 0 — for (zf_minv_epoch, zf_storage_epoch) = (assembly, storage)
    bool zf_minv_ok = false;
    double zf_minv_sigma = -1.0;
    int64_t zf_minv_epoch = -2, zf_storage_epoch = -2;
    DevBuf<double> part_pq, part_rz, part_rr, part_bb;
    // the fused step of the one-iteration regime (fv_fused.hip): v = -M^-1 (A z) of the direction in pvec (qv; qv2 receives the
    // next one), its partial sums (two sets by step parity), and whether qv / the sums describe the prepared next step
    DevBuf<double> qv, qv2, fz_part;
    bool vready = false;
    bool burst_fused = false; // row blocks: what the current burst's first step decided for the whole burst
    int vready_parity = 0;   // which set of sums the launch that left qv wrote
    int vready_counts[3] = {0, 0, 0}; // ... and how many pieces of each kind (vector sums, rhs.rhs, z.q)
    int32_t fused_bytes = 0; // bytes per row of the most recent fused launch's storage form (0: none ran)
    int64_t fused_launches = 0, fused_bytes_launch = 0;
    std::vector<fv_trajectory *> trajectories;   // alive trajectories / observation series of this problem: detached (HBM released,
    std::vector<fv_observation *> observations; // problem pointer cleared) by fv_problem_destroy if it comes first
    fv_trajectory *recording = nullptr; // fv_trajectory_record: fixed / adaptive runs push the state of every outer step here
    double record_t = 0.0;              // ... the time of the last recorded state of a fixed-dt run
    DevBuf<double> small_part;     // the single-launch solver of small systems (fv_small.hip): per-block partial sums,
    DevBuf<uint32_t> small_bar;    // its grid barrier's arrival counter and failure flag,
    uint32_t small_bar_base = 0;   // ... the counter's value when the next launch begins
    uint32_t small_seq = 0;        // number of the last launch (PcgScalars::small_seq)
    DevBuf<PcgScalars> small_scal3; // the three solves of a step-doubling attempt in one launch (fv_small_twostep): their scalar blocks
    int64_t small_solves = 0;      // solves it has done
    DevBuf<double> zalt, walt; // the one-launch PCG iteration (fv_ploop_pass): the second scaled-residual and w vectors (z and w ping-pong: halo rows of other blocks read them)
    int ploop_grid = 0;        // ... its grid (= partial sums per quantity)
    // ... and, inside a fixed-dt run, the last update of a step's loop left pending for the next step's set-up to apply
    // (pcg_carry_flush_kernel): z_m = z + alpha w, x_m = x + alpha pp with alpha in the scalar block.  Whoever else comes first flushes it.
    struct PlPending {
        bool valid = false;
        const double *z = nullptr, *w = nullptr, *pp = nullptr, *xin = nullptr;
        double *x = nullptr;
    } pl_pending;
    DevBuf<uint8_t> vcode; // pcg_carry_flush_kernel's code byte: storage code | bit 7 (b is not zero), for (vcode_sepoch, vcode_aepoch) = (storage, assembly)
    int64_t vcode_sepoch = -1, vcode_aepoch = -1;
    int64_t ploop_solves = 0;  // solves whose loop ran that way
    int64_t bytes_total = 0;   // fv_step_form: bytes the launches of every Jacobi-PCG solve on this problem had to move (every array of every launch once), running total
    int32_t ploop_bytes[3] = {0, 0, 0}; // fv_step_form: set-up, first pass, flush of the most recent such solve (bytes per row)
    bool fused_chunked = false; // the most recent fused launch ran on chunks of a plane (fused_chunk_kernel), not on 2-D tiles
    int32_t loop_bytes = 0;  // bytes per row and iteration of the most recent many-iteration solve when its passes ran through the fused kernel (else 0)
    // fv_place.hip: candidates for the vectors the loop writes in every step, timed and not yet handed out
    struct PlacedSpare {
        void *base = nullptr;
        size_t count = 0;
        double rate = 0.0;
    };
    std::vector<PlacedSpare> spares;
    double place_best = 0.0; // the fastest chunked write seen among this problem's candidates (bytes per second)
    int place_probes = 0;
    DevBuf<double> hist;
    DevBuf<PcgScalars> scal;
    int64_t hist_cap = 0;
    int64_t last_iters = 0; // iterations of the previous solve: sizes the first launch chunk of the next one

    // optional per-kernel timing of the PCG loop (fv_profile_enable): HIP event pairs
    // around every K1/K2/K3 launch on the launch stream, harvested at each poll.
    bool profile = false;
    int profile_level = 1; // 1: event pairs around K1, K2 / K2S and K3; 2: around K1 only (two events per iteration instead of six:
                           // every event is a barrier between two launches, ~10 us each at 464^3)
    std::vector<hipEvent_t> prof_ev; // 6 per iteration of a chunk
    double prof_ms[3] = {0, 0, 0};   // spmv_dot, update, pupdate
    int64_t prof_launches[3] = {0, 0, 0};

    ~fv_problem()
    {
        for (void *s : slot_bases)
            if (s)
                (void)hipFree(s);
        fv_vec_release_spares(this);
        for (hipEvent_t e : prof_ev)
            (void)hipEventDestroy(e);
        delete dist;
        fv_amg_free(amg);
    }
};

// ---- fv_scan.hip
// out[i] = sum(in[0..i-1]) for i in [0, n]; out has n+1 entries. total returned on host.
int fv_exclusive_scan_i32(fv_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int64_t *total);

// A blocking copy on the context's own stream.  (hipMemcpy runs on the null stream: its first use in a process creates that stream's
// queue — measured 8-10 ms inside the first product of a problem — and it does not order against the non-blocking ctx->stream.)
inline hipError_t fv_memcpy_sync(fv_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}

// ---- one empty kernel per translation unit.  HIP loads a translation unit's code object at the first launch of one of its kernels:
// 2-8 ms each for the larger ones, 22 ms in all on the path of a first solve (measured: profiles/r04_amg_*).  fv_ctx_create launches
// the empty kernels once per process and device, so that the first solve of a process costs what every later one costs.
#define FV_WARM_TU(name)                                                                                 \
    __global__ void fv_warm_##name##_kernel() {}                                                       \
    void fv_warm_##name(hipStream_t s) { hipLaunchKernelGGL(fv_warm_##name##_kernel, dim3(1), dim3(64), 0, s); }

// ---- fv_grid.hip
int fv_grid_axes(const double mins[3], const double maxs[3], const int64_t ns[3], std::vector<double> ax[3]);
int fv_grid_generate_device(fv_ctx *ctx, const double mins[3], const double maxs[3], const int64_t ns[3], int32_t *node1,
                            int32_t *node2, double *aol, double *volumes, double *coords, int64_t i1_lo = -1, int64_t i1_hi = -1);
int64_t fv_grid_face_offset(const int64_t ns[3], int64_t i1);

// ---- fv_assembly.hip
int fv_build_maps(fv_problem *p, const int64_t *dirichletnodes_host_or_dev);
int fv_build_symbolic(fv_problem *p);
// fv_place.hip: a vector of the stepping loop, chosen by its write class where that shows (hot: written in every step)
int fv_vec_alloc(fv_problem *p, DevBuf<double> &buf, size_t count, bool hot);
int fv_vec_alloc_raw(fv_problem *p, size_t count, bool hot, void **base_out);
size_t fv_vec_skew(size_t bytes); // the stagger to add to the base (FV_ALLOC_SKEW; 0 by default)
extern int g_place;
// fv_lean.hip: the set-up of a lean problem (no faces, no CSR) from rows formed on the fly
GridRows fv_grid_rows(const fv_problem *p, double sigma);
int fv_require_csr(fv_problem *p, const char *what); // FV_ERR_STATE with a message for a lean problem, FV_OK otherwise
int fv_lean_finish(fv_problem *p, const int64_t *dirichletnodes, const double mins[3], const double maxs[3]);
int fv_lean_assemble(fv_problem *p, const double *sources_dev);
int fv_lean_plane_stride(fv_problem *p, int64_t *stride);
int fv_lean_count_far_stride(fv_problem *p, int64_t stride, int64_t *agree);
int fv_lean_dia_pattern(fv_problem *p, uint8_t *sl_noff, int32_t *sl_off, int32_t *is_dia, int32_t *is_csr);
int fv_lean_dia_fill(fv_problem *p, double sigma, int64_t count, const int32_t *list);
// the face arrays a per-face kernel reads: the problem's own, or — a lean problem — generated for the duration of the call
struct FaceArrays {
    const int32_t *node1 = nullptr, *node2 = nullptr;
    const double *cond = nullptr, *aol = nullptr;
    DevBuf<int32_t> t1, t2;
    DevBuf<double> ta, tc;
};
int fv_face_arrays(fv_problem *p, FaceArrays &fa); // fv_assembly.hip
int fv_lean_rows_spmv(fv_problem *p, double tag, const double *x, double *y, const double *shift, double sigma, bool dot, double *partials, const PcgScalars *scal,
                      const int32_t *list, int64_t count, int grid);
int fv_lean_csr32(fv_problem *p, DevBuf<int32_t> &rowptr, DevBuf<int32_t> &colind, DevBuf<double> &vals, DevBuf<int32_t> *diagpos = nullptr);
int fv_lean_get_csc(fv_problem *p, int64_t *colptr, int64_t *rowval, double *nzval);
int fv_lean_symdia_fill(fv_problem *p, double sigma, int32_t d1, int32_t d2, int32_t d3, double *dg, double *u1, double *u2, double *u3);
int fv_widen_indices(fv_ctx *ctx, const int32_t *src, int64_t *dst, int64_t n, int64_t add);
int fv_narrow_indices(fv_ctx *ctx, const int64_t *src, int32_t *dst, int64_t n, int64_t lo, int64_t hi, int *bad);
int fv_compact_flags(fv_ctx *ctx, const int32_t *flag, int64_t n, int32_t *out, int64_t *count); // ascending indices of the set flags
int fv_scatter_nodes(fv_problem *p, const double *ufree_dev, double *head_dev); // freenodes2nodes on device buffers
int fv_gather_free(fv_problem *p, const double *unodes_dev, double *ufree_dev);  // u[freenodes]
// free-indexed vectors across the ABI: src/dst_host_or_dev are in the canonical numbering of the free cells, the device
// side in the problem's internal one (the same unless p->reordered).  count vectors of n doubles, one after the other.
int fv_free_in(fv_problem *p, double *dst_dev, const double *src_host_or_dev, int64_t count = 1);
int fv_free_out(fv_problem *p, double *dst_host_or_dev, const double *src_dev, int64_t count = 1);
// fv_reorder.hip: the same order on the device; *handled = false: graph not suited to it, run the host routine
int fv_device_locality_order(fv_problem *p, int want, int32_t *perm_dev, bool *adopted, bool *handled, double *mean_before, double *mean_after);
// fv_host.cpp
int fv_host_locality_order(int64_t n, int64_t m, const int32_t *ea, const int32_t *eb, int32_t *perm, double *mean_before, double *mean_after);

// ---- fv_pcg.hip
int fv_pcg_prepare(fv_problem *p);
// The linear system handed to the PCG: (A + sigma*D) x = rhs.
struct PcgSystem {
    double sigma = 0.0;
    const double *rhs = nullptr; // explicit right-hand side; for implicit_step: b' (nullptr = 0)
    bool x0_zero = false;        // explicit form only: start from zeros
    // implicit time step from the state held in x:  rhs = b' + D x0/dt  (b' = D*rhs if b_times_D),
    // sigma must be 1/dt.  The initial residual is then b' - A x0 and no rhs vector is formed.
    bool implicit_step = false;
    bool b_times_D = false;
    double dt = 0.0;
    bool fold_shift = false; // use the copy of vals with sigma*D folded into the diagonal (fixed-dt runs)
    // fixed-dt runs, ping-pong state: when x_next is set the first iteration writes x + alpha p THERE and the solve
    // continues in place in x_next (the solution is in x_next iff info->iters > 0, x is then untouched = the old state).
    double *x_next = nullptr;
    // residual carry-over: the workspace r still holds the final recurrence residual of the previous step of the same
    // system (same sigma, same b'), whose initial state was carry_prev.  Then
    //     r0 = rhs_new - (A + sigma D) x = r + sigma D (x - carry_prev)
    // and the step needs no SpMV for its initial residual.
    const double *carry_prev = nullptr;
    // speculate: the first K2 of this step may also prepare the next step's set-up (pcg_update_spec_kernel) when the
    // previous solve took one iteration; use_spec: start from such a prepared set-up if the previous solve left one.
    bool speculate = false, use_spec = false;
    // Bursts of one-iteration steps (fixed-dt runs): chain_index >= 0 enqueues this step (speculative set-up, K1, K2S, K3)
    // WITHOUT polling the device; a step whose one iteration does not converge sets done = 3 and everything enqueued
    // behind it turns into no-ops.  resume_it > 0 continues such a step from its iteration resume_it (no set-up).
    int chain_index = -1;
    int resume_it = 0;
    // chain_more: another chained step follows in the same burst — this step's verdict and that step's scalars then come
    // from one launch (pcg_chain_boundary_kernel) and the next step skips its own set-up kernel
    bool chain_more = false;
    // x0_src (implicit steps on systems the single-launch solver takes, fv_pcg_small_takes): the state the step starts from, when it
    // is not yet in x — the kernel reads it there and writes x, which saves the copy launch in front of every solve of a stepper
    const double *x0_src = nullptr;
    // fixed-dt runs: the next call is the carried step that follows this one — a loop of one-launch iterations may leave its last
    // update pending for that step's set-up (fv_problem::pl_pending)
    bool defer_flush = false;
};
int fv_ploop_flush_pending(fv_problem *p); // applies a pending update (no-op when there is none)
// x holds the initial guess on entry and the solution on return.
int fv_pcg_solve(fv_problem *p, double *x, const PcgSystem &sys, double rtol, int64_t maxiter, fv_solve_info *info, bool time_it);
// how K2S (and the symmetric K1, for the shift folded into a diagonal it re-derives) is handed the storage term D: see
// StorageArg; *bytes_saved = bytes per row against streaming the doubles; ignore_switch: whatever fv_tune key 35 says
int fv_storage_form(fv_problem *p, StorageArg *out, int *bytes_saved, bool ignore_switch);
int fv_pcg_chain_poll(fv_problem *p, int nsteps, int *completed, fv_solve_info *info, uint32_t *zero_mask = nullptr);
// fv_small.hip: Jacobi-PCG of a small system in one persistent launch; *handled = false: not a case for it
int fv_pcg_small(fv_problem *p, double *x, const PcgSystem &sys, double rtol, int64_t maxiter, fv_solve_info *info, bool time_it, bool *handled);
bool fv_pcg_small_takes(const fv_problem *p, const PcgSystem &sys); // whether fv_pcg_small will handle this solve
bool fv_small_twostep_takes(const fv_problem *p, int mode, double dt);
int fv_small_twostep(fv_problem *p, int mode, const double *const rhs[3], bool b_times_D, double *uk, double dt, double *onestep, bool have_onestep,
                     double *two1, double *two, const double *weight, double rtol, int64_t maxiter, fv_solve_info *info, double *err, bool *handled);
int fv_slot_new(fv_problem *p, int32_t *slot); // a state vector of n + nhalo + pad doubles (reuses freed slots)
int fv_spmv_launch(fv_problem *p, const double *x, double *y, double sigma, double *partials_or_null, bool fold = false,
                   int *npartials = nullptr);
int fv_spmv_grid(fv_problem *p);
int fv_dot_device(fv_problem *p, const double *a, const double *b, double *out_host);
int fv_norm2_diff_device(fv_problem *p, const double *a, const double *b, double *out_host);

// the preconditioner an implicit step / a solve of this problem actually runs with
inline int fv_step_precond(const fv_problem *p)
{
    if (p->precond == FV_PRECOND_AUTO)
        return p->auto_steps_amg ? FV_PRECOND_AMG : FV_PRECOND_JACOBI;
    return p->precond;
}

// ---- fv_transient.hip / fv_trajectory.hip: the stepper loop with hooks, trajectories kept in HBM
constexpr int FV_STEP_W = 2; // internal step mode: (D/dt + A) w+ = rhs + D w/dt with the caller's rhs as it is (the adjoint sweep's own state w = gamma / D)
struct FvStepHooks {
    int mode = FV_STEP_FORWARD;
    // unset: the assembled b (forward) / none.  which = 0 .. 2: the buffer the forcing goes to — the three solves of a step-doubling
    // attempt may be enqueued together (fv_small_twostep), their forcings then live side by side
    std::function<int(double t, int which, const double **rhs_dev)> forcing;
    const double *norm_weight = nullptr;                          // step-doubling error = || weight .* (onestep - twostep) ||
    std::function<int(const double *state_dev, double t)> record; // sees the initial state and the state of every outer step
};
int fv_step_raw(fv_problem *p, double *usrc, double *udst, double dt, const double *rhs_dev, int mode, double rtol, int64_t maxiter, fv_solve_info *info);
int fv_stepper_run(fv_problem *p, int32_t slot, double t0, double tfinal, double dt0, bool fixed, double atol, double rtol, int64_t maxiter, int64_t max_outer,
                   double *ts_out, int64_t *n_outer, int64_t *n_solves, fv_solve_info *last_info, const FvStepHooks &h);
void fv_detach_dependents(fv_problem *p); // fv_trajectory.hip
int fv_trajectory_push_device(fv_trajectory *tr, const double *state_dev, double t, const double *scale_dev); // knot = scale .* state (scale may be null)
int fv_norm2_diff_weighted_device(fv_problem *p, const double *a, const double *b, const double *w, double *out_host);
int fv_slot_new(fv_problem *p, int32_t *slot);

// ---- fv_amg.hip
int fv_amg_prepare(fv_problem *p, double sigma);
int fv_amg_apply_device(fv_problem *p, const double *r, double *z, double sigma, bool kcycle = false);
bool fv_amg_kcycle_available(fv_problem *p);
// runs the V-cycle-preconditioned CG from the set-up left in the workspace (r, scal); x is updated in place
int fv_amg_pcg_loop(fv_problem *p, double *x, double sigma, bool fold, int64_t maxiter, PcgScalars *hs);

// y = (A_local + sigma D) x with the row block's own columns only: the caller keeps x's halo slots at zero (fv_pcg.hip)
int fv_dist_local_spmv(fv_problem *p, double *x_with_zero_halo, double *y, double sigma, bool fold, bool want_dot = false);
int fv_dist_full_spmv(fv_problem *p, double *xext, double *y, double sigma, bool fold); // with the halo exchange of xext
int fv_dist_exchange(fv_problem *p, double *xext);
// diagnostics (fv_comm_diag): record an event of category `cat` on `stream` (a pair = two consecutive calls)
int fv_diag_mark(fv_ctx *ctx, int cat, hipStream_t stream);
// ---- fv_comm.hip (RCCL); all are no-ops for a single rank
int fv_comm_halo_exchange(fv_ctx *ctx, const fv_dist *d, const double *sendbuf, double *recv_base, hipStream_t stream);
int fv_comm_allreduce_sum(fv_ctx *ctx, const fv_dist *d, double *buf, int count, hipStream_t stream);

// copy helper: dst/src may be host or device
inline int fv_copy(fv_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0)
        return FV_OK;
    FV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    FV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FV_OK;
}

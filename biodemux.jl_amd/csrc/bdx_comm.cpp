// bdx_comm.cpp — merge_stats across GPUs through the C-ABI (see include/biodemux_hip.h).
//
// The reference merges the per-worker DemuxStats in one process (src/reporting.jl:1-9, called from
// src/core.jl:495 and :628).  Across GPUs the scalar part of that merge is ONE all-reduce (sum, int64)
// of the counter vector every context accumulates in HBM.  This file makes that reachable from any
// host language: it talks to RCCL directly — no torch, no MPI.  Two deployment shapes:
//   * one process, several devices (the Julia host of INTEGRATION.md: one HipWorker per GPU):
//     bdx_comm_init_all = ncclCommInitAll over the contexts' devices, bdx_allreduce_counts_all = the
//     grouped collective driven by one thread;
//   * one process per GPU (torchrun, mpirun): rank 0 makes an id with bdx_comm_get_unique_id, the host
//     ships its 128 bytes to the other ranks by whatever channel it has, every rank calls
//     bdx_comm_init_rank and later bdx_allreduce_counts.
// RCCL is opened lazily with dlopen the first time a communicator is asked for, so the classify path
// has no link-time dependency on it and a process that already carries an RCCL (PyTorch does) keeps
// using that one instance.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only; nothing here is linked against librccl

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "bdx_ctx.h"

struct bdx_comm_state {
    ncclComm_t comm = nullptr;
    int rank = 0;
    int n_ranks = 1;
};

namespace {

struct Rccl {
    void *handle = nullptr;
    bool forced = false;  // bound to the library BDX_RCCL_LIB names (a test stand-in: ranks may share a device)
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    Rccl &r = g_rccl;
    // test seam: BDX_RCCL_LIB names the library to bind instead (tests/fake_rccl.cpp lets several contexts on ONE
    // device run the N > 1 branches below); read once per process, here and nowhere else
    if (const char *forced = getenv("BDX_RCCL_LIB")) {
        r.forced = true;
        r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            const char *e = dlerror();
            r.error = std::string("BDX_RCCL_LIB could not be opened: ") + (e ? e : "unknown error");
            return;
        }
    }
    // an instance that is already part of the process first (RTLD_NOLOAD), then the system's
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *p : paths)
        if (!r.handle) r.handle = dlopen(p, RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) {
        const char *e = dlerror();
        r.error = std::string("RCCL (librccl.so.1) could not be opened: ") + (e ? e : "unknown error");
        return;
    }
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(r.handle, name);
        if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
}

// nullptr (and the message on ctx / the create error) when RCCL is not usable
Rccl *rccl(bdx_ctx *ctx) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty()) {
        bdx_fail(ctx, BDX_E_COMM, "%s", g_rccl.error.c_str());
        return nullptr;
    }
    return &g_rccl;
}

#define NCCL_TRY(ctx, R, call)                                                                             \
    do {                                                                                                   \
        ncclResult_t r__ = (call);                                                                         \
        if (r__ != ncclSuccess) return bdx_fail(ctx, BDX_E_COMM, "%s failed: %s", #call, (R)->GetErrorString(r__)); \
    } while (0)

// group calls: whichever context the host asks afterwards (or none: bdx_last_error(NULL)) has the message
int fail_group(bdx_ctx *const *ctxs, int n, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    bdx_fail(nullptr, code, "%s", buf);
    for (int i = 0; i < n; ++i)
        if (ctxs && ctxs[i]) ctxs[i]->err = buf;
    return code;
}

int ensure_sum_buffer(bdx_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->counts_sum.ensure((size_t)ctx->dev.n_counts * 8));
    return BDX_OK;
}

}  // namespace

void bdx_comm_release(bdx_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    if (ctx->comm->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(ctx->device);
        (void)g_rccl.CommDestroy(ctx->comm->comm);
    }
    delete ctx->comm;
    ctx->comm = nullptr;
}

extern "C" {

int32_t bdx_comm_get_unique_id(void *id_out) {
    if (!id_out) return bdx_fail(nullptr, BDX_E_INVALID, "id_out is NULL");
    Rccl *R = rccl(nullptr);
    if (!R) return BDX_E_COMM;
    static_assert(sizeof(ncclUniqueId) == BDX_COMM_ID_BYTES, "BDX_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    NCCL_TRY(nullptr, R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return BDX_OK;
}

int32_t bdx_comm_init_rank(bdx_ctx *ctx, const void *id, int32_t rank, int32_t n_ranks) {
    if (!ctx) return BDX_E_INVALID;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return bdx_fail(ctx, BDX_E_INVALID, "bad communicator arguments (rank %d of %d)", rank, n_ranks);
    if (ctx->comm) return bdx_fail(ctx, BDX_E_STATE, "the context already has a communicator");
    Rccl *R = rccl(ctx);
    if (!R) return BDX_E_COMM;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof nid);
    bdx_comm_state *st = new (std::nothrow) bdx_comm_state();
    if (!st) return bdx_fail(ctx, BDX_E_DEVICE, "out of host memory");
    ncclResult_t rr = R->CommInitRank(&st->comm, n_ranks, nid, rank);
    if (rr != ncclSuccess) {
        delete st;
        return bdx_fail(ctx, BDX_E_COMM, "ncclCommInitRank failed: %s", R->GetErrorString(rr));
    }
    st->rank = rank;
    st->n_ranks = n_ranks;
    ctx->comm = st;
    // every rank must have been built from the same config: the collectives run with per-rank element counts, and a
    // mismatch would hang or corrupt the sums.  One tiny max-all-reduce of {v, -v} per quantity: all equal <=> max(v) == -max(-v).
    {
        const int npass = ctx->dev.is_dual ? 2 : 1;
        long long probe[6] = {ctx->dev.n_counts, -(long long)ctx->dev.n_counts, ctx->dev.pass[0].n_barcodes, -(long long)ctx->dev.pass[0].n_barcodes,
                              npass > 1 ? ctx->dev.pass[1].n_barcodes : 0, npass > 1 ? -(long long)ctx->dev.pass[1].n_barcodes : 0};
        int rc = ensure_sum_buffer(ctx);
        DevBuf tmp;
        if (rc == BDX_OK && tmp.ensure(sizeof probe) != hipSuccess) rc = bdx_fail(ctx, BDX_E_DEVICE, "hipMalloc failed");
        if (rc == BDX_OK && hipMemcpyAsync(tmp.p, probe, sizeof probe, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = bdx_fail(ctx, BDX_E_DEVICE, "hipMemcpyAsync failed");
        if (rc == BDX_OK) {
            ncclResult_t r2 = R->AllReduce(tmp.p, tmp.p, 6, ncclInt64, ncclMax, st->comm, ctx->stream);
            if (r2 != ncclSuccess) rc = bdx_fail(ctx, BDX_E_COMM, "ncclAllReduce failed: %s", R->GetErrorString(r2));
        }
        long long got[6] = {0, 0, 0, 0, 0, 0};
        if (rc == BDX_OK && (hipMemcpyAsync(got, tmp.p, sizeof got, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                             hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = bdx_fail(ctx, BDX_E_DEVICE, "reading the agreement probe failed");
        tmp.release();
        if (rc == BDX_OK && (got[0] != -got[1] || got[2] != -got[3] || got[4] != -got[5]))
            rc = bdx_fail(ctx, BDX_E_INVALID, "the ranks of one communicator must share the config (counter vectors / barcode counts differ)");
        if (rc != BDX_OK) {
            bdx_comm_release(ctx);
            return rc;
        }
    }
    return BDX_OK;
}

int32_t bdx_comm_init_all(bdx_ctx *const *ctxs, int32_t n) {
    if (!ctxs || n < 1) return bdx_fail(nullptr, BDX_E_INVALID, "bdx_comm_init_all needs at least one context");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return bdx_fail(nullptr, BDX_E_INVALID, "context %d is NULL", i);
        if (ctxs[i]->comm) return fail_group(ctxs, i + 1, BDX_E_STATE, "the context already has a communicator");
        if (ctxs[i]->dev.n_counts != ctxs[0]->dev.n_counts)
            return fail_group(ctxs, i + 1, BDX_E_INVALID, "contexts of one communicator must share the config (counter vectors differ: %d vs %d)",
                              ctxs[i]->dev.n_counts, ctxs[0]->dev.n_counts);
    }
    Rccl *R = rccl(ctxs[0]);
    if (!R) return BDX_E_COMM;
    if (!R->forced)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (ctxs[j]->device == ctxs[i]->device)
                    return fail_group(ctxs, i + 1, BDX_E_INVALID, "contexts %d and %d share device %d (RCCL wants one rank per device)", j, i, ctxs[i]->device);
    std::vector<int> devs((size_t)n);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
    NCCL_TRY(ctxs[0], R, R->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        bdx_comm_state *st = new (std::nothrow) bdx_comm_state();
        if (!st) {  // the communicators not yet handed to a context would leak
            for (int j = i; j < n; ++j)
                if (comms[(size_t)j] && R->CommDestroy) (void)R->CommDestroy(comms[(size_t)j]);
            for (int j = 0; j < i; ++j) bdx_comm_release(ctxs[j]);
            return bdx_fail(ctxs[i], BDX_E_DEVICE, "out of host memory");
        }
        st->comm = comms[(size_t)i];
        st->rank = i;
        st->n_ranks = n;
        ctxs[i]->comm = st;
    }
    return BDX_OK;
}

int32_t bdx_comm_destroy(bdx_ctx *ctx) {
    if (!ctx) return BDX_E_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    bdx_comm_release(ctx);
    return BDX_OK;
}

int32_t bdx_comm_rank(const bdx_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->rank : 0; }
int32_t bdx_comm_size(const bdx_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->n_ranks : 1; }

// ---- the statistics tables (summary = true) travel with the counters --------------------------------
// Their height follows the longest read a rank has seen, so the ranks first agree on the maximum
// (one tiny all-reduce, ncclMax), grow to it, and then every table is summed like the counter vector.
static long long *rows_slot(bdx_ctx *ctx) { return (long long *)((char *)ctx->st_flag.p + 64); }

static int stats_agree_enqueue(bdx_ctx *ctx, Rccl *R) {
    if (!ctx->dev.need_traceback || !ctx->comm) return BDX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long mine = ctx->st_rows;
    HIP_TRY(ctx, hipMemcpyAsync(rows_slot(ctx), &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    NCCL_TRY(ctx, R, R->AllReduce(rows_slot(ctx), rows_slot(ctx), 1, ncclInt64, ncclMax, ctx->comm->comm, ctx->stream));
    return BDX_OK;
}

static int stats_agree_finish(bdx_ctx *ctx) {
    if (!ctx->dev.need_traceback) return BDX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    long long agreed = ctx->st_rows;
    if (ctx->comm) {
        HIP_TRY(ctx, hipMemcpyAsync(&agreed, rows_slot(ctx), sizeof agreed, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    int rc = bdx_stats_reserve(ctx, agreed, true);
    if (rc != BDX_OK) return rc;
    const int npass = ctx->dev.is_dual ? 2 : 1;
    for (int p = 0; p < npass; ++p)
        for (int w = 0; w < 3; ++w) HIP_TRY(ctx, ctx->st_sum[p][w].ensure(bdx_stats_phys_words(ctx, p, w, ctx->st_rows) * 8 + 8));
    ctx->st_sum_rows = ctx->st_rows;
    ctx->st_sum_len_rows = ctx->st_len_rows;
    return BDX_OK;
}

// enqueue on ctx's stream: counts_sum (and the summed statistics tables) = sum over the ranks; the per-rank
// vectors stay as they are
static int enqueue_allreduce(bdx_ctx *ctx, Rccl *R) {
    int rc = ensure_sum_buffer(ctx);
    if (rc != BDX_OK) return rc;
    const size_t n = (size_t)ctx->dev.n_counts;
    const int npass = ctx->dev.is_dual ? 2 : 1;
    if (!ctx->comm) {  // a single-GPU host runs the same call sequence: the sum over one rank
        HIP_TRY(ctx, hipMemcpyAsync(ctx->counts_sum.p, ctx->counts, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (ctx->dev.need_traceback)
            for (int p = 0; p < npass; ++p)
                for (int w = 0; w < 3; ++w) {
                    const size_t words = bdx_stats_phys_words(ctx, p, w, ctx->st_rows);
                    if (words) HIP_TRY(ctx, hipMemcpyAsync(ctx->st_sum[p][w].p, ctx->st_tab[p][w].p, words * 8, hipMemcpyDeviceToDevice, ctx->stream));
                }
        return BDX_OK;
    }
    NCCL_TRY(ctx, R, R->AllReduce(ctx->counts, ctx->counts_sum.p, n, ncclInt64, ncclSum, ctx->comm->comm, ctx->stream));
    if (ctx->dev.need_traceback)
        for (int p = 0; p < npass; ++p)
            for (int w = 0; w < 3; ++w) {
                const size_t words = bdx_stats_phys_words(ctx, p, w, ctx->st_rows);
                if (words) NCCL_TRY(ctx, R, R->AllReduce(ctx->st_tab[p][w].p, ctx->st_sum[p][w].p, words, ncclInt64, ncclSum, ctx->comm->comm, ctx->stream));
            }
    return BDX_OK;
}

int32_t bdx_allreduce_counts(bdx_ctx *ctx) {
    if (!ctx) return BDX_E_INVALID;
    Rccl *R = nullptr;
    if (ctx->comm && !(R = rccl(ctx))) return BDX_E_COMM;
    int rc = stats_agree_enqueue(ctx, R);
    if (rc == BDX_OK) rc = stats_agree_finish(ctx);
    if (rc == BDX_OK) rc = ensure_sum_buffer(ctx);  // (allocations stay outside the group, as in the _all form)
    if (rc != BDX_OK) return rc;
    if (R) NCCL_TRY(ctx, R, R->GroupStart());  // counters + tables as one fused launch
    rc = enqueue_allreduce(ctx, R);
    if (R) {
        ncclResult_t rr = R->GroupEnd();
        if (rc == BDX_OK && rr != ncclSuccess) rc = bdx_fail(ctx, BDX_E_COMM, "ncclGroupEnd failed: %s", R->GetErrorString(rr));
    }
    return rc;
}

int32_t bdx_allreduce_counts_all(bdx_ctx *const *ctxs, int32_t n) {
    if (!ctxs || n < 1) return bdx_fail(nullptr, BDX_E_INVALID, "bdx_allreduce_counts_all needs at least one context");
    bool any_comm = false;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return bdx_fail(nullptr, BDX_E_INVALID, "context %d is NULL", i);
        any_comm |= ctxs[i]->comm != nullptr;
    }
    Rccl *R = nullptr;
    if (any_comm && !(R = rccl(ctxs[0]))) return BDX_E_COMM;
    int rc = BDX_OK;
    // one thread drives several devices: the per-device calls of a collective must be fused into one group
    if (R) NCCL_TRY(ctxs[0], R, R->GroupStart());
    for (int i = 0; i < n && rc == BDX_OK; ++i) rc = stats_agree_enqueue(ctxs[i], R);
    if (R) {
        ncclResult_t rr = R->GroupEnd();
        if (rc == BDX_OK && rr != ncclSuccess) rc = fail_group(ctxs, n, BDX_E_COMM, "ncclGroupEnd failed: %s", R->GetErrorString(rr));
    }
    for (int i = 0; i < n && rc == BDX_OK; ++i) rc = stats_agree_finish(ctxs[i]);  // allocations stay outside the groups
    for (int i = 0; i < n && rc == BDX_OK; ++i) rc = ensure_sum_buffer(ctxs[i]);
    if (rc != BDX_OK) return rc;
    if (R) NCCL_TRY(ctxs[0], R, R->GroupStart());
    for (int i = 0; i < n && rc == BDX_OK; ++i) rc = enqueue_allreduce(ctxs[i], R);
    if (R) {
        ncclResult_t rr = R->GroupEnd();
        if (rc == BDX_OK && rr != ncclSuccess) rc = fail_group(ctxs, n, BDX_E_COMM, "ncclGroupEnd failed: %s", R->GetErrorString(rr));
    }
    return rc;
}

void *bdx_reduced_counts_device_ptr(bdx_ctx *ctx) { return ctx ? ctx->counts_sum.p : nullptr; }

int32_t bdx_get_reduced_counts(bdx_ctx *ctx, int64_t *out, int64_t n) {
    if (!ctx || !out) return BDX_E_INVALID;
    if (n < ctx->dev.n_counts) return bdx_fail(ctx, BDX_E_INVALID, "counts buffer too small: need %d", ctx->dev.n_counts);
    if (!ctx->counts_sum.p) return bdx_fail(ctx, BDX_E_STATE, "bdx_allreduce_counts has not been called");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->counts_sum.p, (size_t)ctx->dev.n_counts * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDX_OK;
}

}  // extern "C"

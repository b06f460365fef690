// TEST INFRASTRUCTURE — a stand-in for librccl inside ONE process, selected with BDX_RCCL_LIB.
//
// A 1-GPU box cannot run the N > 1 branches of csrc/bdx_comm.cpp against the real RCCL (one rank per device).  This
// library implements the eight entry points bdx_comm.cpp binds — with the published NCCL semantics for the calls it
// makes — on top of plain HIP copies, so that several contexts on device 0 can go through bdx_comm_init_all +
// bdx_allreduce_counts_all: every collective inside a ncclGroupStart / ncclGroupEnd pair is matched across the ranks
// of a communicator in posting order, reduced on the host (sum or max over int64) and written back to every rank's
// receive buffer.  Nothing here is shipped or linked into the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Group;
struct FakeComm {
    Group *group;
    int rank, nranks;
};
struct Op {
    const void *send;
    void *recv;
    size_t count;
    ncclRedOp_t op;
    hipStream_t stream;
};
struct Group {
    int nranks = 0;
    std::vector<std::vector<Op>> pending;  // per rank, in posting order
    int live = 0;
};

std::mutex g_mu;
int g_depth = 0;
std::vector<Group *> g_touched;

ncclResult_t run_group(Group *g) {
    // every rank must have posted the same number of collectives with the same shapes
    const size_t nops = g->pending[0].size();
    for (int r = 1; r < g->nranks; ++r)
        if (g->pending[(size_t)r].size() != nops) return ncclInvalidUsage;
    for (size_t k = 0; k < nops; ++k) {
        const size_t count = g->pending[0][k].count;
        const ncclRedOp_t op = g->pending[0][k].op;
        std::vector<long long> acc(count, 0), tmp(count, 0);
        for (int r = 0; r < g->nranks; ++r) {
            const Op &o = g->pending[(size_t)r][k];
            if (o.count != count || o.op != op) return ncclInvalidArgument;
            if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (count && hipMemcpy(tmp.data(), o.send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            for (size_t i = 0; i < count; ++i) {
                if (r == 0)
                    acc[i] = tmp[i];
                else if (op == ncclSum)
                    acc[i] += tmp[i];
                else if (op == ncclMax)
                    acc[i] = tmp[i] > acc[i] ? tmp[i] : acc[i];
                else
                    return ncclInvalidArgument;
            }
        }
        for (int r = 0; r < g->nranks; ++r) {
            const Op &o = g->pending[(size_t)r][k];
            if (count && hipMemcpy(o.recv, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    for (auto &v : g->pending) v.clear();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static int counter = 0;
    memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lk(g_mu);
    const int v = ++counter;
    memcpy(id->internal, "FAKE", 4);
    memcpy(id->internal + 4, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank) {
    if (nranks != 1 || rank != 0) return ncclInvalidUsage;  // (one thread cannot drive a multi-rank rendezvous: use ncclCommInitAll)
    Group *g = new Group();
    g->nranks = 1;
    g->pending.resize(1);
    g->live = 1;
    *comm = (ncclComm_t) new FakeComm{g, 0, 1};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *) {
    if (ndev < 1) return ncclInvalidArgument;
    Group *g = new Group();
    g->nranks = ndev;
    g->pending.resize((size_t)ndev);
    g->live = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = (ncclComm_t) new FakeComm{g, i, ndev};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm *c = (FakeComm *)comm;
    if (!c) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    if (--c->group->live == 0) delete c->group;
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (Group *g : g_touched) {
        const ncclResult_t r = run_group(g);
        if (rc == ncclSuccess) rc = r;
    }
    g_touched.clear();
    return rc;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    FakeComm *c = (FakeComm *)comm;
    if (!c || datatype != ncclInt64) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    c->group->pending[(size_t)c->rank].push_back(Op{sendbuff, recvbuff, count, op, stream});
    if (g_depth > 0) {
        bool seen = false;
        for (Group *g : g_touched) seen |= g == c->group;
        if (!seen) g_touched.push_back(c->group);
        return ncclSuccess;
    }
    // outside a group: only a one-rank communicator can complete from one thread
    if (c->nranks != 1) return ncclInvalidUsage;
    return run_group(c->group);
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess:
            return "no error";
        case ncclInvalidUsage:
            return "invalid usage (fake RCCL)";
        case ncclInvalidArgument:
            return "invalid argument (fake RCCL)";
        case ncclUnhandledCudaError:
            return "unhandled HIP error (fake RCCL)";
        default:
            return "error (fake RCCL)";
    }
}

}  // extern "C"

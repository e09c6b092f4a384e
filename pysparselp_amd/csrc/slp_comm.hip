// slp_comm.hip -- multi-GPU exchange: one RCCL sum-all-reduce of the n partial
// column sums K_g^T y_g per Chambolle-Pock iteration (plus one at setup and a
// few scalars per report).  The reference is single-process; this is new.
//
// RCCL is bound with dlopen on first use: a single-GPU process never loads it,
// and a process that also imports torch does not get a link-time dependency on
// one particular librccl.  xGMI is point-to-point (7 links per GPU): the
// message is n doubles (8 MB at n = 1e6), large enough that RCCL's direct
// reduce-scatter/all-gather schedule over the fully connected node applies; it
// is issued on the library's own stream, in order with the kernels around it.
#include <dlfcn.h>

#include <cstdlib>

#include "slp_common.h"

namespace slp {

struct NcclId { char internal[128]; };
typedef void *NcclComm;
typedef int (*fn_get_unique_id)(NcclId *);
typedef int (*fn_comm_init_rank)(NcclComm *, int, NcclId, int);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(NcclComm);
typedef const char *(*fn_error_string)(int);

static struct {
    void *lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    NcclComm comm = nullptr;
    int nranks = 1, rank = 0;
    long long collectives = 0;  // all-reduces issued since slp_comm_init (slp_comm_collectives)
    DevBuf<double> scratch;
} g;

static void load_rccl() {
    if (g.lib) return;
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
        g.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g.lib) break;
    }
    if (!g.lib) {
        const char *e = dlerror();  // reading it clears it: once
        throw Error(std::string("cannot load RCCL: ") + (e ? e : "?"));
    }
    g.get_unique_id = (fn_get_unique_id)dlsym(g.lib, "ncclGetUniqueId");
    g.comm_init_rank = (fn_comm_init_rank)dlsym(g.lib, "ncclCommInitRank");
    g.all_reduce = (fn_all_reduce)dlsym(g.lib, "ncclAllReduce");
    g.comm_destroy = (fn_comm_destroy)dlsym(g.lib, "ncclCommDestroy");
    g.error_string = (fn_error_string)dlsym(g.lib, "ncclGetErrorString");
    SLP_REQUIRE(g.get_unique_id && g.comm_init_rank && g.all_reduce && g.comm_destroy, "RCCL symbols missing");
}

static void check(int rc, const char *what) {
    if (rc != 0) throw Error(std::string(what) + ": " + (g.error_string ? g.error_string(rc) : "RCCL error"));
}

// SLP_FORCE_DISTRIBUTED=1 runs the partitioned code path (partial sums -> all-reduce -> update) on a single
// rank too, so that it can be exercised on a one-GPU box; the all-reduce is then the identity.
bool comm_active() {
    if (g.comm == nullptr) return false;
    if (g.nranks > 1) return true;
    const char *e = getenv("SLP_FORCE_DISTRIBUTED");
    return e && e[0] == '1';
}

void comm_allreduce_dev(double *buf, i64 count, int op) {
    SLP_REQUIRE(g.comm, "slp_comm_init has not been called");
    if (count <= 0) return;
    ++g.collectives;
    check(g.all_reduce(buf, buf, (size_t)count, /*ncclFloat64*/ 8, op == 1 ? /*ncclMax*/ 2 : /*ncclSum*/ 0, g.comm, ctx().stream),
          "ncclAllReduce");
}

}  // namespace slp

using namespace slp;

extern "C" {

int slp_comm_unique_id(char id[128]) {
    SLP_API_INT({
        load_rccl();
        NcclId nid;
        check(g.get_unique_id(&nid), "ncclGetUniqueId");
        memcpy(id, nid.internal, 128);
    })
}

int slp_comm_init(int nranks, int rank, const char id[128]) {
    SLP_API_INT({
        SLP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && id, "slp_comm_init: bad arguments");
        ctx();
        SLP_REQUIRE(!g.comm, "slp_comm_init: already initialised");
        load_rccl();
        NcclId nid;
        memcpy(nid.internal, id, 128);
        check(g.comm_init_rank(&g.comm, nranks, nid, rank), "ncclCommInitRank");
        g.nranks = nranks;
        g.rank = rank;
        g.scratch.alloc(64);
    })
}

int slp_comm_finalize(void) {
    SLP_API_INT({
        if (g.comm) {
            SLP_HIP(hipStreamSynchronize(ctx().stream));
            g.scratch.release();
            check(g.comm_destroy(g.comm), "ncclCommDestroy");
            g.comm = nullptr;
            g.nranks = 1;
            g.rank = 0;
        }
    })
}

int slp_comm_allreduce_host(double *v, int64_t count, int op) {
    SLP_API_INT({
        SLP_REQUIRE(v && count >= 0 && count <= 64, "slp_comm_allreduce_host: at most 64 values");
        if (!comm_active()) return 0;
        g.scratch.upload(v, (size_t)count);
        comm_allreduce_dev(g.scratch.p, count, op);
        g.scratch.download(v, (size_t)count);
    })
}

long long slp_comm_collectives(void) { return g.collectives; }

int slp_comm_barrier(void) {
    SLP_API_INT({
        if (comm_active()) {
            double z = 0.0;
            g.scratch.upload(&z, 1);
            comm_allreduce_dev(g.scratch.p, 1, 0);
        }
        SLP_HIP(hipStreamSynchronize(ctx().stream));
    })
}

}  // extern "C"

// slp_comm.hip -- multi-GPU exchange: sum-all-reduces of the n partial column
// sums K_g^T y_g (one per Chambolle-Pock iteration, two per matrix-free ADMM
// iteration, one per block-splitting iteration), plus one at setup and a few
// scalars per report.  The reference is single-process; this is new.
//
// Transport 1 (product): RCCL over xGMI.  The prototypes, the unique-id struct
// and the enum values come from <rccl/rccl.h> (compile-time checked); the
// library itself is bound with dlopen on first use, so a single-GPU process
// never loads it and the .so carries no link-time dependency on one librccl.
// xGMI is point-to-point (7 links per GPU): the message is n doubles (8 MB at
// n = 1e6), large enough for RCCL's direct reduce-scatter / all-gather schedule
// over the fully connected node; it is issued on the library's compute stream,
// in order with the kernels around it -- every all-reduce's result is consumed
// by the very next kernel of the iteration, so there is no rank-local pass to
// overlap it with.  The exception is the block-splitting
// ADMM with several blocks per rank: a block's consensus summand is all-reduced
// on a second stream while the next block's projection computes
// (comm_allreduce_dev_async / comm_join).
//
// Transport 2 (tests): a host callback (slp_comm_init_host).  The device buffer
// is copied to the host, the callback reduces it in place across the ranks --
// over gloo, a pipe, or not at all for a recording stub -- and the result is
// copied back.  Lets world_size > 1 runs of the REAL partitioned device code
// share one GPU (RCCL refuses two ranks on one device), and lets tests count
// and inspect the sequence of collectives.
#include <dlfcn.h>

#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>

#include <rccl/rccl.h>

#include "slp_common.h"

namespace slp {

static struct {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclReduceScatter) reduce_scatter = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    ncclComm_t comm = nullptr;
    slp_host_allreduce_fn host_fn = nullptr;  // transport 2
    void *host_user = nullptr;
    std::vector<double> host_buf;
    bool active = false;
    int nranks = 1, rank = 0;
    long long collectives = 0;  // all-reduces issued since slp_comm_init* (slp_comm_collectives)
    DevBuf<double> scratch;
    // asynchronous all-reduces (block groups): a second stream, ordered against the compute stream by events
    hipStream_t side = nullptr;
    hipEvent_t ready[8] = {}, done = nullptr;
    unsigned next_ready = 0;
    bool pending = false;
} g;

// Exchange timing (slp_comm_timing): every collective bracketed by a pair of HIP events on the stream it is issued on --
// recorded only, read after the timed region (slp_comm_timing_read), so nothing synchronises inside it.  Under RCCL the pair
// brackets the collective's kernel: its duration includes the wait for the slowest peer, which is what an iteration pays.
static struct {
    bool on = false;
    std::vector<hipEvent_t> ev;       // 2 per recorded collective (created once, reused by later sessions)
    std::vector<double> bytes;        // payload bytes of each recorded collective (per rank)
    std::vector<int> side;            // issued on the second stream (overlapped with compute)
    size_t used = 0;
    long long dropped = 0;            // collectives beyond the ring
    size_t kMax = 16384;
} xt_;

struct TimedCollective {
    hipStream_t st;
    bool live = false;
    TimedCollective(hipStream_t stream, i64 count, bool on_side) : st(stream) {
        if (!xt_.on) return;
        if (xt_.used >= xt_.kMax) { ++xt_.dropped; return; }
        if (xt_.ev.size() < 2 * (xt_.used + 1)) {
            hipEvent_t a = nullptr, b = nullptr;
            SLP_HIP(hipEventCreate(&a));
            SLP_HIP(hipEventCreate(&b));
            xt_.ev.push_back(a);
            xt_.ev.push_back(b);
        }
        if (xt_.bytes.size() <= xt_.used) { xt_.bytes.resize(xt_.used + 1); xt_.side.resize(xt_.used + 1); }
        xt_.bytes[xt_.used] = (double)count * sizeof(double);
        xt_.side[xt_.used] = on_side ? 1 : 0;
        SLP_HIP(hipEventRecord(xt_.ev[2 * xt_.used], st));
        live = true;
    }
    void done() {
        if (!live) return;
        live = false;
        SLP_HIP(hipEventRecord(xt_.ev[2 * xt_.used + 1], st));
        ++xt_.used;
    }
    ~TimedCollective() { if (live) { (void)hipEventRecord(xt_.ev[2 * xt_.used + 1], st); ++xt_.used; } }
};

// Host transport, asynchronous form (tests: the block groups' overlapped exchange with several ranks on ONE GPU): a worker
// thread takes the jobs in issue order -- waits for the compute stream to reach the point of the call, copies the buffer out on
// the second stream, runs the callback, copies the result back -- while the caller goes on enqueueing the next block's
// projection; comm_join() waits for the queue to drain.  The callback is only ever run by one thread at a time and in the order
// the all-reduces were issued (synchronous ones drain the queue first), so the ranks' sequences of collectives stay aligned.
static struct HostWorker {
    struct Job { double *buf; i64 count; int op; hipEvent_t ready; bool timed; };
    double timed_ms = 0.0, timed_bytes = 0.0, timed_worst = 0.0;   // slp_comm_timing: host clock around a job (copy out, callback, copy back)
    long long timed_n = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    int inflight = 0;
    bool stop = false, running = false;
    std::string error;
} hw;

static void host_worker_main(int device) {
    (void)hipSetDevice(device);
    std::vector<double> stage;
    for (;;) {
        HostWorker::Job job;
        {
            std::unique_lock<std::mutex> lk(hw.mu);
            hw.cv.wait(lk, [] { return hw.stop || !hw.jobs.empty(); });
            if (hw.jobs.empty()) return;  // stop
            job = hw.jobs.front();
            hw.jobs.pop_front();
        }
        std::string err;
        if (stage.size() < (size_t)job.count) stage.resize((size_t)job.count);
        double t_job = 0.0;
        // (capture_mutex: none of these calls while the main thread has a stream capture open, slp_common.h; `ready` was recorded
        // before the job was queued, so waiting for it under the lock cannot wait for the main thread)
        hipError_t e;
        {
            std::lock_guard<std::mutex> no_capture(capture_mutex());
            e = hipEventSynchronize(job.ready);
            t_job = trace_now();
            if (e == hipSuccess) e = hipMemcpyAsync(stage.data(), job.buf, (size_t)job.count * sizeof(double), hipMemcpyDeviceToHost, g.side);
            if (e == hipSuccess) e = hipStreamSynchronize(g.side);
        }
        if (e != hipSuccess) err = std::string("host all-reduce worker: ") + hipGetErrorString(e);
        else if (g.host_fn(stage.data(), job.count, job.op, g.host_user) != 0) err = "host all-reduce callback failed";
        else {
            std::lock_guard<std::mutex> no_capture(capture_mutex());
            e = hipMemcpyAsync(job.buf, stage.data(), (size_t)job.count * sizeof(double), hipMemcpyHostToDevice, g.side);
            if (e == hipSuccess) e = hipStreamSynchronize(g.side);
            if (e != hipSuccess) err = std::string("host all-reduce worker: ") + hipGetErrorString(e);
        }
        {
            std::lock_guard<std::mutex> lk(hw.mu);
            if (!err.empty() && hw.error.empty()) hw.error = err;
            if (job.timed) {
                const double ms = (trace_now() - t_job) * 1e3;
                hw.timed_ms += ms;
                hw.timed_bytes += (double)job.count * sizeof(double);
                hw.timed_worst = ms > hw.timed_worst ? ms : hw.timed_worst;
                ++hw.timed_n;
            }
            --hw.inflight;
        }
        hw.cv.notify_all();
    }
}

// waits until every asynchronous host all-reduce issued so far has landed; throws what the worker ran into
static void host_worker_drain(bool may_throw) {
    std::string err;
    {
        std::unique_lock<std::mutex> lk(hw.mu);
        hw.cv.wait(lk, [] { return hw.inflight == 0; });
        err.swap(hw.error);
    }
    if (may_throw && !err.empty()) throw Error(err);
}

static void host_worker_stop() {
    if (!hw.running) return;
    host_worker_drain(false);
    {
        std::lock_guard<std::mutex> lk(hw.mu);
        hw.stop = true;
    }
    hw.cv.notify_all();
    hw.th.join();
    hw.running = false;
    hw.stop = false;
}

static void side_stream() {
    if (g.side) return;
    SLP_HIP(hipStreamCreateWithFlags(&g.side, hipStreamNonBlocking));
    for (auto &e : g.ready) SLP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SLP_HIP(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
}

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
    g.get_unique_id = (decltype(g.get_unique_id))dlsym(g.lib, "ncclGetUniqueId");
    g.comm_init_rank = (decltype(g.comm_init_rank))dlsym(g.lib, "ncclCommInitRank");
    g.all_reduce = (decltype(g.all_reduce))dlsym(g.lib, "ncclAllReduce");
    g.reduce_scatter = (decltype(g.reduce_scatter))dlsym(g.lib, "ncclReduceScatter");
    g.all_gather = (decltype(g.all_gather))dlsym(g.lib, "ncclAllGather");
    g.comm_destroy = (decltype(g.comm_destroy))dlsym(g.lib, "ncclCommDestroy");
    g.error_string = (decltype(g.error_string))dlsym(g.lib, "ncclGetErrorString");
    SLP_REQUIRE(g.get_unique_id && g.comm_init_rank && g.all_reduce && g.comm_destroy && g.reduce_scatter && g.all_gather, "RCCL symbols missing");
}

static void check(ncclResult_t rc, const char *what) {
    if (rc != ncclSuccess) throw Error(std::string(what) + ": " + (g.error_string ? g.error_string(rc) : "RCCL error"));
}

// SLP_FORCE_DISTRIBUTED=1 runs the partitioned code path (partial sums -> all-reduce -> update) on a single
// rank too, so that it can be exercised on a one-GPU box; the all-reduce is then the identity.
bool comm_active() {
    if (!g.active) return false;
    if (g.nranks > 1) return true;
    const char *e = getenv("SLP_FORCE_DISTRIBUTED");
    return e && e[0] == '1';
}

// In-place all-reduce of `count` doubles at `buf` (device), ordered with the kernels on the compute stream.
// `count` must be the same on every rank (callers derive it from global sizes only).
void comm_allreduce_dev(double *buf, i64 count, int op) {
    SLP_REQUIRE(g.active, "slp_comm_init has not been called");
    if (count <= 0) return;
    ++g.collectives;
    hipStream_t st = ctx().stream;
    if (g.host_fn) {
        // lab (tools/shard_compute_only.py): SLP_COMM_NULL=1 skips the exchange altogether -- WRONG results, for timing the
        // compute side of one rank of N on one GPU without the host transport's PCIe copies in the way
        static const bool null_comm = [] { const char *e = getenv("SLP_COMM_NULL"); return e && e[0] == '1'; }();
        if (null_comm) return;
        if (hw.running) host_worker_drain(true);  // (collectives reach the callback in issue order)
        // opened AFTER the wait for earlier asynchronous jobs: their time is already in hw.timed_ms (host clock), it must not be
        // counted again inside this collective's event pair (exchange.ms_per_iteration counted it twice)
        TimedCollective timed(st, count, false);
        if (g.host_buf.size() < (size_t)count) g.host_buf.resize((size_t)count);
        SLP_HIP(hipMemcpyAsync(g.host_buf.data(), buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        const int rc = g.host_fn(g.host_buf.data(), count, op, g.host_user);
        SLP_REQUIRE(rc == 0, "host all-reduce callback failed");
        SLP_HIP(hipMemcpyAsync(buf, g.host_buf.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
        timed.done();
        SLP_HIP(hipStreamSynchronize(st));  // host_buf is reused by the next call
        return;
    }
    TimedCollective timed(st, count, false);
    check(g.all_reduce(buf, buf, (size_t)count, ncclFloat64, op == 1 ? ncclMax : ncclSum, g.comm, st), "ncclAllReduce");
    timed.done();
}

int comm_rank() { return g.active ? g.rank : 0; }
int comm_size() { return g.active ? g.nranks : 1; }

__global__ void k_comm_keep_slice(i64 total, i64 lo, i64 hi, double *__restrict__ buf) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (i64)gridDim.x * blockDim.x)
        if (j < lo || j >= hi) buf[j] = 0.0;
}

// In-place sum reduce-scatter over `cnt * nranks` doubles at `buf`: afterwards rank k's slice buf[k cnt .. (k+1) cnt) holds the
// sums (the other slices are unspecified), and the in-place all-gather of those slices -- the two halves of an all-reduce, used
// by the sharded ADMM updates (slp_admm_cg.hip): the same bytes over the links, the elementwise work between them on 1/N of the
// variables.  On the compute stream.  Host transport: an all-reduce (of the whole buffer / of the buffer with the foreign slices
// zeroed: x + 0 is exact).
void comm_reduce_scatter_dev(double *buf, i64 cnt) {
    SLP_REQUIRE(g.active, "slp_comm_init has not been called");
    if (cnt <= 0) return;
    if (g.host_fn) { comm_allreduce_dev(buf, cnt * g.nranks, 0); return; }
    ++g.collectives;
    TimedCollective timed(ctx().stream, cnt * g.nranks, false);
    check(g.reduce_scatter(buf, buf + (i64)g.rank * cnt, (size_t)cnt, ncclFloat64, ncclSum, g.comm, ctx().stream), "ncclReduceScatter");
    timed.done();
}

void comm_all_gather_dev(double *buf, i64 cnt) {
    SLP_REQUIRE(g.active, "slp_comm_init has not been called");
    if (cnt <= 0) return;
    if (g.host_fn) {
        const i64 total = cnt * g.nranks;
        { const char *e = getenv("SLP_COMM_NULL"); if (e && e[0] == '1') { ++g.collectives; return; } }
        hipLaunchKernelGGL(k_comm_keep_slice, dim3(grid_for(total, 256)), dim3(256), 0, ctx().stream, total, (i64)g.rank * cnt,
                           (i64)(g.rank + 1) * cnt, buf);
        SLP_HIP(hipGetLastError());
        comm_allreduce_dev(buf, total, 0);
        return;
    }
    ++g.collectives;
    TimedCollective timed(ctx().stream, cnt * g.nranks, false);
    check(g.all_gather(buf + (i64)g.rank * cnt, buf, (size_t)cnt, ncclFloat64, g.comm, ctx().stream), "ncclAllGather");
    timed.done();
}

// The same all-reduce, but on the library's second stream: it starts once everything enqueued so far on the compute stream
// has finished and runs beside whatever the compute stream is given next (the projection of the next block of a group);
// comm_join() makes the compute stream wait for every asynchronous all-reduce issued so far.  The host transport (tests)
// has no asynchronous form: it reduces at once.
void comm_allreduce_dev_async(double *buf, i64 count, int op) {
    SLP_REQUIRE(g.active, "slp_comm_init has not been called");
    if (count <= 0) return;
    if (g.host_fn) {
        const char *sy = getenv("SLP_HOST_ASYNC");  // =0: reduce at once (the transport's first form)
        if (sy && sy[0] == '0') {
            comm_allreduce_dev(buf, count, op);
            return;
        }
        ++g.collectives;
        side_stream();
        hipEvent_t ev = g.ready[g.next_ready++ % 8];
        SLP_HIP(hipEventRecord(ev, ctx().stream));
        if (!hw.running) {
            static const bool at_exit = (atexit([] { host_worker_stop(); }), true);  // a joinable std::thread must not reach its destructor
            (void)at_exit;
            hw.th = std::thread(host_worker_main, ctx().device);
            hw.running = true;
        }
        {
            std::lock_guard<std::mutex> lk(hw.mu);
            hw.jobs.push_back({buf, count, op, ev, xt_.on});
            ++hw.inflight;
        }
        hw.cv.notify_all();
        g.pending = true;
        return;
    }
    ++g.collectives;
    side_stream();
    hipEvent_t ev = g.ready[g.next_ready++ % 8];
    SLP_HIP(hipEventRecord(ev, ctx().stream));
    SLP_HIP(hipStreamWaitEvent(g.side, ev, 0));
    TimedCollective timed(g.side, count, true);
    check(g.all_reduce(buf, buf, (size_t)count, ncclFloat64, op == 1 ? ncclMax : ncclSum, g.comm, g.side), "ncclAllReduce");
    timed.done();
    g.pending = true;
}

// Drains the second stream and forgets pending asynchronous all-reduces (allocator trims, error paths): after this no
// collective can still be writing a buffer that goes back to the cache or to the driver.
void comm_sync_side() {
    if (hw.running) host_worker_drain(false);
    if (g.side) (void)hipStreamSynchronize(g.side);
    g.pending = false;
}

// an asynchronous all-reduce of the COLLECTIVE LIBRARY (not the host transport's worker, which takes capture_mutex) may be running
bool comm_library_collective_pending() { return g.active && g.pending && !g.host_fn; }

void comm_join() {
    if (!g.pending) return;
    if (g.host_fn) {  // the worker has synchronised the second stream behind every copy: nothing left for the compute stream to wait for
        host_worker_drain(true);
        g.pending = false;
        return;
    }
    SLP_HIP(hipEventRecord(g.done, g.side));
    SLP_HIP(hipStreamWaitEvent(ctx().stream, g.done, 0));
    g.pending = false;
}

}  // namespace slp

using namespace slp;

extern "C" {

int slp_comm_unique_id(char id[128]) {
    SLP_API_INT({
        static_assert(sizeof(ncclUniqueId) == 128, "slp_comm_unique_id hands out NCCL_UNIQUE_ID_BYTES = 128 bytes");
        load_rccl();
        ncclUniqueId nid;
        check(g.get_unique_id(&nid), "ncclGetUniqueId");
        memcpy(id, nid.internal, 128);
    })
}

int slp_comm_init(int nranks, int rank, const char id[128]) {
    SLP_API_INT({
        SLP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && id, "slp_comm_init: bad arguments");
        ctx();
        SLP_REQUIRE(!g.active, "slp_comm_init: already initialised");
        load_rccl();
        ncclUniqueId nid;
        memcpy(nid.internal, id, 128);
        check(g.comm_init_rank(&g.comm, nranks, nid, rank), "ncclCommInitRank");
        g.nranks = nranks;
        g.rank = rank;
        g.collectives = 0;
        g.active = true;
        g.scratch.alloc(64);
    })
}

int slp_comm_init_host(int nranks, int rank, slp_host_allreduce_fn fn, void *user) {
    SLP_API_INT({
        SLP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && fn, "slp_comm_init_host: bad arguments");
        ctx();
        SLP_REQUIRE(!g.active, "slp_comm_init_host: already initialised");
        g.host_fn = fn;
        g.host_user = user;
        g.nranks = nranks;
        g.rank = rank;
        g.collectives = 0;
        g.active = true;
        g.scratch.alloc(64);
    })
}

int slp_comm_finalize(void) {
    SLP_API_INT({
        if (g.active) {
            host_worker_stop();
            if (g.side) SLP_HIP(hipStreamSynchronize(g.side));
            SLP_HIP(hipStreamSynchronize(ctx().stream));
            g.scratch.release();
            if (g.comm) check(g.comm_destroy(g.comm), "ncclCommDestroy");
            g.comm = nullptr;
            g.host_fn = nullptr;
            g.host_user = nullptr;
            g.active = false;
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

int slp_comm_timing(int on) {
    SLP_API_INT({
        if (on) {
            if (hw.running) host_worker_drain(true);
            xt_.used = 0; xt_.dropped = 0;
            std::lock_guard<std::mutex> lk(hw.mu);
            hw.timed_ms = hw.timed_bytes = hw.timed_worst = 0.0;
            hw.timed_n = 0;
        }
        xt_.on = on != 0;
    })
}

int slp_comm_timing_read(double out[6]) {
    SLP_API_INT({
        SLP_REQUIRE(out, "slp_comm_timing_read: NULL argument");
        SLP_REQUIRE(!xt_.on, "slp_comm_timing_read: stop the recording first (slp_comm_timing(0))");
        if (hw.running) host_worker_drain(true);
        if (g.side) SLP_HIP(hipStreamSynchronize(g.side));
        SLP_HIP(hipStreamSynchronize(ctx().stream));
        double total = 0.0, bytes = 0.0, worst = 0.0, side_ms = 0.0;
        for (size_t k = 0; k < xt_.used; ++k) {
            float ms = 0.0f;
            SLP_HIP(hipEventElapsedTime(&ms, xt_.ev[2 * k], xt_.ev[2 * k + 1]));
            total += ms;
            bytes += xt_.bytes[k];
            if (ms > worst) worst = ms;
            if (xt_.side[k]) side_ms += ms;
        }
        {   // the host transport's asynchronous worker (tests): host clock around each job, all of it beside the compute stream
            std::lock_guard<std::mutex> lk(hw.mu);
            total += hw.timed_ms; bytes += hw.timed_bytes; side_ms += hw.timed_ms;
            worst = hw.timed_worst > worst ? hw.timed_worst : worst;
        }
        out[0] = (double)xt_.used + (double)hw.timed_n;       // collectives recorded
        out[1] = total;                 // ms inside them (event pairs on the stream each was issued on)
        out[2] = bytes;                 // payload bytes (per rank)
        out[3] = worst;                 // the slowest single collective, ms
        out[4] = side_ms;               // of [1]: issued on the second stream, overlapped with compute
        out[5] = (double)xt_.dropped;    // collectives beyond the ring of event pairs
    })
}

int slp_comm_info(int *nranks, int *rank) {
    if (nranks) *nranks = g.active ? g.nranks : 1;
    if (rank) *rank = g.active ? g.rank : 0;
    return g.active ? 1 : 0;
}

__global__ void k_comm_fill(i64 n, double v, double *__restrict__ buf) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) buf[j] = v;
}

// `reps` sum-all-reduces of `count` doubles on a device buffer through the path the solvers use (comm_allreduce_dev), the first
// one checked: every rank contributes rank + 1 in every entry.  out[0] = ms per all-reduce (HIP events around all of them on the
// compute stream), out[1] = max |error| of the checked one, out[2] = ranks.  tools/rccl_preflight.py.
int slp_comm_bench_allreduce(int64_t count, int reps, double out[3]) {
    SLP_API_INT({
        SLP_REQUIRE(comm_active(), "slp_comm_bench_allreduce: slp_comm_init has not been called");
        SLP_REQUIRE(count > 0 && reps > 0 && out, "slp_comm_bench_allreduce: bad arguments");
        Context &c = ctx();
        DevBuf<double> buf((size_t)count);
        const int grid = grid_for(count, 256);
        hipLaunchKernelGGL(k_comm_fill, dim3(grid), dim3(256), 0, c.stream, (i64)count, (double)(g.rank + 1), buf.p);
        SLP_HIP(hipGetLastError());
        comm_allreduce_dev(buf.p, count, 0);
        const double want = 0.5 * (double)g.nranks * (double)(g.nranks + 1);
        double probe[3] = {0, 0, 0};
        SLP_HIP(hipMemcpyAsync(&probe[0], buf.p, sizeof(double), hipMemcpyDeviceToHost, c.stream));
        SLP_HIP(hipMemcpyAsync(&probe[1], buf.p + count / 2, sizeof(double), hipMemcpyDeviceToHost, c.stream));
        SLP_HIP(hipMemcpyAsync(&probe[2], buf.p + (count - 1), sizeof(double), hipMemcpyDeviceToHost, c.stream));
        SLP_HIP(hipStreamSynchronize(c.stream));
        double err = 0.0;
        for (double v : probe) err = std::max(err, fabs(v - want));
        hipLaunchKernelGGL(k_comm_fill, dim3(grid), dim3(256), 0, c.stream, (i64)count, 0.0, buf.p);   // (zeros stay zeros over the reps)
        SLP_HIP(hipEventRecord(c.ev0, c.stream));
        for (int r = 0; r < reps; ++r) comm_allreduce_dev(buf.p, count, 0);
        SLP_HIP(hipEventRecord(c.ev1, c.stream));
        SLP_HIP(hipEventSynchronize(c.ev1));
        float ms = 0.f;
        SLP_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
        out[0] = (double)ms / (double)reps;
        out[1] = err;
        out[2] = (double)g.nranks;
    })
}

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

// slp_common.h -- shared host-side plumbing of libslp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/slp_hip.h"

// Kernel-lab switches never reach libslp_hip.so.  Timing experiments with parts of a kernel removed (WRONG results by design)
// exist only behind -DSLP_ABLATION (`make ablation` / `make variant`, separate objects, a library of its own that
// pysparselp_amd/_lib.py loads only when SLP_LIB_VARIANT names it: slp_build_flags() != 0).  The lab forms of the tall-cell and
// Gauss-Seidel kernels are patches under tools/lab/patches/; their macros must not appear in an ordinary build:
#if !defined(SLP_ABLATION) && (defined(SLP_TALL_ABL) || defined(SLP_GS_ABLATE) || defined(SLP_GS_BANDS_ABLATE_FETCH) || \
                               defined(SLP_GS_BANDS_ABLATE_PUBLISH) || defined(SLP_GS_BANDS_ABLATE_SC1) || defined(SLP_STRIP_ABLATE))
#error "a kernel-lab macro (SLP_TALL_ABL / SLP_GS_ABLATE / SLP_GS_BANDS_ABLATE_* / SLP_STRIP_ABLATE) without -DSLP_ABLATION: ablated kernels give wrong results and must not be built into libslp_hip.so"
#endif

namespace slp {

typedef int64_t i64;
typedef int32_t i32;

// ---- error plumbing: C++ exceptions inside, int/NULL + slp_last_error outside
void set_error(const std::string &msg);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define SLP_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            throw slp::Error(std::string(#expr) + " failed: " + hipGetErrorString(e_) +   \
                             " (" __FILE__ ":" + std::to_string(__LINE__) + ")");         \
    } while (0)

#define SLP_REQUIRE(cond, msg)                       \
    do {                                             \
        if (!(cond)) throw slp::Error(std::string(msg)); \
    } while (0)

// Wraps the body of an extern "C" entry point that returns int.
#define SLP_API_INT(...)                   \
    try {                                  \
        __VA_ARGS__;                       \
        return 0;                          \
    } catch (const std::exception &e) {    \
        slp::set_error(e.what());          \
        return 1;                          \
    }

#define SLP_API_PTR(...)                   \
    try {                                  \
        __VA_ARGS__;                       \
    } catch (const std::exception &e) {    \
        slp::set_error(e.what());          \
        return nullptr;                    \
    }

// ---- process-wide context: one device, one stream
struct Context {
    bool ready = false;
    int device = -1;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};
Context &ctx();          // throws if slp_init has not succeeded
Context &ctx_unchecked();

// SLP_TRACE=1: driver calls that take longer than 2 ms are reported (setup timeline; see Phase below)
bool trace_enabled();
double trace_now();
void trace_slow(const char *what, size_t bytes, double t0);

// ---- device memory: a caching sub-allocator in front of hipMalloc / hipFree (slp_matrix.hip).
// Returning multi-GB blocks to the driver and asking for new ones is erratic on this platform -- a hipMalloc that follows
// large frees was measured at 1.1 - 3.4 s (profiles/r02_setup_trace_first.txt) against < 1 ms otherwise -- and the setup
// phases allocate and drop tens of GB of temporaries (sort buffers, counts).  Freed blocks are kept, keyed by capacity, and
// handed to the next request they fit (capacity within 2x of the request); slp_trim() or an out-of-memory retry releases
// them.  Stream-ordered like hipFree here: every user of a block is on the library's one stream, and a block is only
// reused by work enqueued later on that stream.
void *dev_alloc(size_t bytes, size_t *capacity);
void dev_free(void *p, size_t capacity);
// blocks of these sizes will be asked for soon: a helper thread takes them from the driver meanwhile (slp_matrix.hip)
void dev_reserve_async(const std::vector<size_t> &sizes);

// ---- device buffer
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t cap = 0;  // bytes of the underlying block
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = 0; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = 0; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) dev_free(p, cap);
        p = nullptr;
        n = 0;
        cap = 0;
    }
    void alloc(size_t count) {
        release();
        // never hand out a NULL device pointer for an empty vector
        p = static_cast<T *>(dev_alloc((count ? count : 1) * sizeof(T), &cap));
        n = count;
    }
    void upload(const T *host, size_t count) {
        if (count > n || !p) alloc(count);
        if (count) SLP_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
        SLP_HIP(hipStreamSynchronize(ctx().stream));  // host buffer may be freed by the caller
    }
    void download(T *host, size_t count) const {
        if (count) SLP_HIP(hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
        SLP_HIP(hipStreamSynchronize(ctx().stream));
    }
    void zero() {
        if (n) SLP_HIP(hipMemsetAsync(p, 0, n * sizeof(T), ctx().stream));
    }
    void copy_from(const DevBuf<T> &o) {
        if (o.n > n || !p) alloc(o.n);
        if (o.n) SLP_HIP(hipMemcpyAsync(p, o.p, o.n * sizeof(T), hipMemcpyDeviceToDevice, ctx().stream));
    }
};

// Held while a stream capture is open, and by the library's helper threads (the host transport's asynchronous worker,
// slp_comm.hip) around their HIP calls: a copy or a synchronisation issued by ANOTHER thread while this one captures
// invalidates the capture on ROCm 7.2 even in hipStreamCaptureModeThreadLocal ("operation failed due to a previous error
// during capture" at the next launch: seen once in ~10 runs of tests/test_gpu_two_ranks.py's block-group test).
inline std::mutex &capture_mutex() {
    static std::mutex m;
    return m;
}

// Launch-bound inner loops (cache-resident LPs: Potts, netlib) are captured once into a hipGraph of
// `unroll` iterations and replayed: a kernel boundary inside a graph costs ~1.5 us instead of a host launch.
struct IterGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int unroll = 0;
    ~IterGraph() { reset(); }
    void reset() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
        unroll = 0;
    }
    // Runs `k` iterations of `body` (which only enqueues kernels on ctx().stream).
    template <class F>
    void run(i64 k, int want_unroll, F body) {
        hipStream_t st = ctx().stream;
        if (want_unroll > k) want_unroll = (int)k;
        const char *off = getenv("SLP_NO_GRAPH");  // eager launches, e.g. under a profiler
        if (!exec && k >= 2 && want_unroll >= 1 && !(off && off[0] == '1')) {
            std::lock_guard<std::mutex> no_helper_calls(capture_mutex());
            SLP_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            try {
                for (int u = 0; u < want_unroll; ++u) body();
            } catch (...) {
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(st, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                throw;
            }
            SLP_HIP(hipStreamEndCapture(st, &graph));
            SLP_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            unroll = want_unroll;
        }
        if (exec) {
            for (; k >= unroll; k -= unroll) SLP_HIP(hipGraphLaunch(exec, st));
        }
        for (; k > 0; --k) body();
    }
};

// Setup phases on stderr (SLP_TRACE=1): host wall-clock per phase with the stream drained on both sides, so that
// allocation / free / host work show up next to kernel time.  Off: no synchronisation, no output.
struct Phase {
    const char *name;
    double t0 = 0.0;
    bool on;
    static bool enabled() { return trace_enabled(); }
    static double now() { return trace_now(); }
    explicit Phase(const char *n) : name(n), on(enabled()) {
        if (on) { (void)hipStreamSynchronize(ctx().stream); t0 = now(); }
    }
    ~Phase() {
        if (on) { (void)hipStreamSynchronize(ctx().stream); fprintf(stderr, "[slp trace] %-34s %9.3f ms\n", name, (now() - t0) * 1e3); }
    }
};

inline int grid_for(i64 work_items, int block, int max_blocks_per_cu = 8) {
    i64 g = (work_items + block - 1) / block;
    i64 cap = (i64)ctx().num_cu * max_blocks_per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- device-resident CSR in both orientations (defined in slp_matrix.hip)
struct CsrDev {
    i64 nrow = 0, ncol = 0, nnz = 0;
    DevBuf<i64> ptr;
    DevBuf<i32> idx;
    DevBuf<double> val;
    i64 max_row_len = 0;
    double mean_row_len() const { return nrow ? (double)nnz / (double)nrow : 0.0; }
};

// tall cells: the packet stream of one workgroup (row block, strip range) -- absolute device addresses, because the
// streams are written in several build passes, each into buffers of its own (slp_tall.hip)
struct TallWg {
    const unsigned int *dir;   // packet headers (8 dwords each)
    const unsigned int *pay;   // payload words
    const double *val;         // fp64 entries (NULL with a value dictionary)
    const double *dict;        // the stream's value table (NULL: the launch's argument)
    i64 npk;                   // packets, including the 2 x depth that are only ever prefetched
    i64 row0;                  // where the block's sums go: out[row0 + r], r < nrows
    i64 x0;                    // first element of x the stream multiplies (a column chunk of a composite copy of A^T), else 0
    i64 ncol;                  // columns of x the stream sees (range check of the x-tile loads)
    int nrows;                 // rows of the block
    int D;                     // entries of the value table
};

// strip-JDS copy of a CSR matrix for the LDS-tiled SpMV (slp_strip.hip)
struct StripJds {
    bool ok = false;
    i64 nrow = 0, ncol = 0, nnz = 0, T = 0, B = 0;
    int S = 1;                    // strip ranges per row block (S > 1: few row blocks, e.g. a 1/8 row partition)
    DevBuf<double> part;          // [S * nrow] partial row sums when S > 1
    mutable DevBuf<double> part2; // [2 * S * nrow] the same for the two-vector product
    DevBuf<i64> base;             // [B*T + 1] first entry of every (row block, strip) cell
    DevBuf<unsigned short> perm;  // [B*T*R] sorted position -> local row
    DevBuf<unsigned char> slen;   // [B*T*R] entry count of the row at a sorted position
    DevBuf<unsigned int> soff;    // [B*T*256] offset of jagged diagonal s inside the cell
    DevBuf<double> val;           // [nnz]
    DevBuf<unsigned short> col;   // [nnz] column inside the strip
    // value-dictionary variant (matrices with few distinct stored values): an entry is (uint16 value id,
    // uint16 column) = 4 B; pairs of entries are packed as {id0, id1, col0, col1}; the D values sit in LDS
    int C = 0;                    // columns per strip of this copy
    int D = 0;                    // 0: fp64 values in `val`; > 0: `ent` + `dict`
    bool wide = false;            // strips of 131072 columns, x gathered from L2 instead of an LDS tile (32-bit columns)
    int rpl = 2;                  // sorted positions per lane: 2 (pairs, 2048-row blocks) or 4 (quads, 4096-row blocks)
    DevBuf<unsigned short> ent;   // [2 * nnz]
    const double *dict = nullptr; // [D] sorted distinct values (owned by the slp_matrix)
    // tall cells (slp_tall.hip): row blocks of tall_R rows x strips of 4096 columns, packets of <= 1024 non-empty rows
    bool tall = false;
    int tall_R = 0;
    DevBuf<TallWg> tall_wg;                          // [B * S] where every workgroup's packet stream lies
    std::vector<DevBuf<unsigned int>> tall_dir;      // 32-byte packet headers, one buffer per build pass
    std::vector<DevBuf<unsigned int>> tall_pay;      // payload words, one buffer per build pass
    std::vector<DevBuf<double>> tall_val;            // fp64 entries: the values at the payload offsets
    size_t tall_bytes = 0;                           // bytes of all of the above (what a product streams)
    // composite of the row chunks of a chunked matrix (slp_chunked.hip): this orientation's copy of every chunk, in order
    std::vector<const StripJds *> parts;
    std::vector<i64> part_off;        // first row of chunk k inside the chunked matrix
    bool parts_cols = false;          // the copy of A^T: the chunks cut its COLUMNS (x is sliced, the sums of a row continue)
    bool fused = false;               // tall_wg holds ONE descriptor table over all parts: a product is a single launch (tall_fuse)
};
constexpr int kTallRmax = 9984;       // most rows of a tall-cell row block (their running sums: 78 KB of LDS)
// sorted distinct stored values of a matrix, when there are at most kDictMax of them
struct ValueDict {
    int state = -1;                       // -1 not looked at, 0 too many distinct values, 1 available
    int D = 0;
    DevBuf<double> values;                // [D] ascending (total order on the bit patterns: -0.0 < +0.0)
    DevBuf<unsigned long long> keys;      // [D] order-preserving integer image of `values`
};
bool value_dictionary(const CsrDev &a, ValueDict &d);
bool strip_wanted(const CsrDev &a, int variant);   // variant: 0 fp64 entries, 1 dictionary pairs, 2 dictionary quads, 3 wide strips
bool strip_build(const CsrDev &a, StripJds &f, const ValueDict *dict, int variant);
void strip_spmv(const StripJds &f, const double *x, double *out);
void product_timing(bool on);            // HIP events around every strip_spmv product while on (slp_strip.hip)
void product_timing_read(double out[3]);  // products, sum of their durations (ms), the longest (ms)
// > 0 while a caller needs every row sum as the single chain of the CSR walk (matrix_spmv in SLP_ORDER_SEQUENTIAL): LDS-strip
// copies built with a strip-range split (S > 1: few row blocks, e.g. a 1/8 row partition) then run one workgroup per row block
extern int g_strip_single_chain;
bool strip_has_tall_split(const StripJds &f);   // a tall-cell copy (or a part of a composite) whose strips are shared by S > 1 workgroups
void strip_spmv2(const StripJds &f, const double *x0, const double *x1, double *out0, double *out1);  // one pass, two vectors
void strip_spmv_with_dict(const StripJds &f, const double *table, const double *x, double *out);     // values table[id] instead of dict[id]
void strip_spmv_pow(const StripJds &f, double pw, const double *x, double *out);                      // fp64 strips: values |v|^pw
bool strip_abs_pow_supported(const StripJds &f);
void strip_spmv_abs_pow(const StripJds &f, double pw, const double *x, double *out);                  // values |v|^pw * 1.0, any copy that supports it
// long rows that are sparse inside every LDS-sized window (slp_tall.hip); transposed: the question / the copy for A^T, taken
// straight from the CSR of A (no transposed CSR is ever formed)
bool tall_wanted(i64 nrow, i64 ncol, i64 nnz);
bool tall_build(const CsrDev &a, bool transposed, StripJds &f, const ValueDict *dict, i64 block_multiple = 0, i64 rows_before = 0,
                i64 rows_total = 0);   // rows_total > 0: the chunk's share of the row blocks of a chunked matrix (tall_geometry)
void tall_spmv(const StripJds &f, const double *x, double *out, int accum);
bool tall_fuse(StripJds &composite);                                          // one descriptor table over all chunks' copies, if they allow it
void tall_spmv_fused(const StripJds &composite, const double *x, double *out);  // the whole product of a fused composite in ONE launch
void tall_spmv_pow(const StripJds &f, double pw, const double *x, double *out, int accum);  // fp64 entries: values |v|^pw * 1.0
size_t strip_format_bytes(const StripJds &f);   // bytes of the copy a product streams (composites: all chunks)

}  // namespace slp

struct slp_matrix {
    slp::CsrDev a;        // rows of A
    slp::CsrDev at;       // rows of A^T (CSC of A), built on the device on first use
    bool have_at = false;
    slp::StripJds fa, fat;            // LDS-tiled copies of a / at, built on first use when they pay
    bool tried_fa = false, tried_fat = false;
    slp::ValueDict vdict;             // shared by both orientations
    slp::DevBuf<double> vx, vy;  // scratch vectors for the host-vector entry points
    int format_policy = 0;       // slp_matrix_set_format: 0 auto, 1 no value dictionary (fp64 entries), 2 CSR kernels only
    bool scaled = false;         // the stored values were row-normalised in place by an ADMM setup (not idempotent)
    int borrowers = 0;           // live solvers created *_on this matrix (they hold raw pointers into its copies)
    int csr_bound = 0;           // those of them whose iterations walk the CSR arrays (slp_matrix_release_csr refuses while > 0)
    bool csr_released = false;   // slp_matrix_release_csr: entries only live in the strip copies (row pointers are kept)
    // chunked matrix (slp_chunked.hip): row chunks whose CSR was dropped as soon as their strip copies stood; `a` / `at` then only
    // carry the dimensions and the entry count, fa / fat are composites over the chunks' copies
    std::vector<slp_matrix *> chunks;
    std::vector<slp::i64> chunk_row0;      // first row of every chunk
    slp::DevBuf<double> rowsq;             // a chunk: [2 * rows] the two sums of squares behind the ADMM row scaling (tools.py:272-290)
    slp::i64 expect_chunks = 0;            // slp_matrix_chunked_expect: chunks to come in all (0: unknown, nothing is reserved ahead)
    slp::i64 tall_block_multiple = 0;      // a chunk about to join a chunked matrix of K chunks: row blocks in multiples of CUs / gcd(CUs, K)
    slp::i64 tall_rows_before = 0, tall_rows_total = 0;   // ... or, when the chunked matrix knows its row count: the chunk's share of its row blocks
    slp::i64 expect_rows = 0;              // slp_matrix_chunked_expect_rows: rows of the whole chunked matrix (0: unknown)
    ~slp_matrix();
};

namespace slp {
void finish_stats(CsrDev &a);          // max row length of a freshly built CSR
void build_transpose(slp_matrix *m);   // stable: rows increasing inside every column
void ensure_transposed(slp_matrix *m); // a strip copy of A^T if the matrix qualifies (tall cells need no transposed CSR), else build_transpose
// per row i of the CSR: sq[2 i] = sum a_ij^2, sq[2 i + 1] = sum (a_ij / ||a_i||)^2 in the lane order of the ADMM row scaling
// (slp_admm_cg.hip) -- what a chunked matrix keeps of a chunk's CSR for a later ADMM set-up
void matrix_row_squares(const CsrDev &a, double *sq);
int lanes_for(const CsrDev &a, int order);
void launch_spmv(const CsrDev &a, const double *x, double *y, int order);
// The strip copy of one orientation (built lazily), or NULL when the matrix does not qualify.
const StripJds *fast_format(slp_matrix *m, bool transposed);
// y = A x (transposed: y = A^T x) with the best kernel for the matrix.
void matrix_spmv(slp_matrix *m, bool transposed, const double *x, double *y, int order);
// device-side setup transforms of lp_admm (slp_spgemm.hip); b / b2: device vectors scaled in place (may be NULL)
slp_matrix *matrix_precondition_rows(slp_matrix *a, double *b, double *b2);
slp_matrix *matrix_standard_form(slp_matrix *a_eq, slp_matrix *a_ineq);
void invalidate_derived(slp_matrix *m);  // after the CSR values were modified in place
void require_csr(const slp_matrix *m, const char *what);  // throws once slp_matrix_release_csr has dropped the CSR entries
bool matrix_dictionary(slp_matrix *m);   // value_dictionary() of the matrix unless its format policy rules the dictionary out
// rows r0 .. r1 of a plain matrix (with its CSR) as a matrix of its own: a device copy of that part of the CSR
slp_matrix *matrix_row_slice(slp_matrix *m, i64 r0, i64 r1);
// a matrix the caller owns keeps only the product copies built so far: the CSR arrays of both orientations go
void matrix_drop_csr(slp_matrix *m);
// chunked matrix: the composite of this orientation's copies of chunks k0 .. k1 (a view: the copies stay the chunks')
void composite_of_chunks(const slp_matrix *g, bool transposed, size_t k0, size_t k1, StripJds &f);
// two-stage deterministic reductions; result lands in out[0..k) (device), see slp_reduce.hip
}  // namespace slp

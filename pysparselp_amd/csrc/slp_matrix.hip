// slp_matrix.hip -- library context, device CSR matrices (both orientations),
// CSR SpMV / SpMV^T.  Replaces the scipy.sparse operands and
// _sparsetools.csr_matvec / csc_matvec calls of the reference's solver loops
// (ChambollePockPPD.py:206,216,235,240 ; ADMM.py:95,148,220,262).
#include <chrono>
#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <cstring>
#include <cstdlib>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

bool trace_enabled() {
    static const int e = [] { const char *v = getenv("SLP_TRACE"); return (v && v[0] == '1') ? 1 : 0; }();
    return e != 0;
}
double trace_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void trace_slow(const char *what, size_t bytes, double t0) {
    const double ms = (trace_now() - t0) * 1e3;
    if (ms > 2.0) fprintf(stderr, "[slp trace]     %-10s %8.1f MB %9.3f ms\n", what, (double)bytes / 1e6, ms);
}

// ---- caching device allocator (slp_common.h) ---------------------------------------------------------------
static std::multimap<size_t, void *> g_free_blocks;  // capacity -> block
static size_t g_cached_bytes = 0;
static std::mutex g_alloc_mutex;
// bookkeeping for slp_alloc_stats: what the driver calls cost and how much memory the library ever held at once
static double g_driver_seconds = 0.0;
static size_t g_held_bytes = 0, g_peak_bytes = 0;
static long long g_driver_calls = 0;
// Reservations: hipMalloc of blocks that will be asked for soon, issued by a helper thread while the GPU (and this thread) work on
// something else -- on this platform a hipMalloc costs 24-28 ms per GB on many boxes (the driver clears the memory), 0.36 s
// for the 13 GB packet stream of one row chunk of config 4, 5.8 s of the 12 s that LP took to set up.  The blocks land in the
// cache; a request that fits a reservation still in flight waits for it instead of going to the driver a second time.
static std::multimap<size_t, int> g_reserving;   // capacity -> (unused): blocks the helper is allocating
static std::condition_variable g_reserve_cv;
static std::thread g_reserve_thread;
static double g_background_seconds = 0.0;

static size_t size_class(size_t bytes) {
    // (+ 16: a kernel may fetch the 16-byte piece that holds a vector's last element -- slp_tall_spmv.hip's x-tile pieces)
    size_t want = (bytes + 16 + 255) & ~(size_t)255;
    if (want > ((size_t)64 << 20)) {
        // large blocks in size classes of 1/32 .. 1/64 of their size: the CSR arrays, sort buffers and packet streams of
        // successive row chunks (sizes equal to a fraction of a percent) then reuse each other's blocks instead of going to the
        // driver for a new multi-GB allocation each time
        size_t g = 1;
        while ((g << 6) <= want) g <<= 1;
        want = (want + g - 1) & ~(g - 1);
    }
    return want;
}

static void reserve_join() {
    if (g_reserve_thread.joinable()) g_reserve_thread.join();
}

void dev_reserve_async(const std::vector<size_t> &sizes) {
    // OPT-IN (SLP_RESERVE=1).  Measured at config 4: on a box whose hipMalloc runs at 27 ms per GB the set-up went 10.4 -> 7.0 s;
    // on a quick-malloc box it went 5.3 -> 8.8 s, and either way the blocks taken ahead sit beside the conversion's cached
    // temporaries: the peak held rose from 279.8 to 302.9 GB of the 309 GB device.  Not worth that by default.
    static const bool on = [] { const char *e = getenv("SLP_RESERVE"); return e && e[0] == '1'; }();
    if (!on || sizes.empty()) return;
    static const bool at_exit = (atexit([] { reserve_join(); }), true);  // a joinable std::thread must not reach its destructor
    (void)at_exit;
    reserve_join();  // (one helper at a time: requests are few and large)
    std::vector<size_t> caps;
    {
        std::lock_guard<std::mutex> lock(g_alloc_mutex);
        for (size_t b : sizes) {
            const size_t c = size_class(b);
            // already parked in the cache?  then nothing to do for this one
            auto it = g_free_blocks.find(c);
            if (it != g_free_blocks.end()) continue;
            caps.push_back(c);
            g_reserving.emplace(c, 0);
        }
    }
    if (caps.empty()) return;
    const int device = ctx().device;
    g_reserve_thread = std::thread([caps, device] {
        (void)hipSetDevice(device);
        for (size_t c : caps) {
            void *p = nullptr;
            const double t0 = trace_now();
            const hipError_t e = hipMalloc(&p, c);
            if (e != hipSuccess) (void)hipGetLastError();  // no memory for a reservation: the request itself will deal with it
            std::lock_guard<std::mutex> lock(g_alloc_mutex);
            g_background_seconds += trace_now() - t0;
            ++g_driver_calls;
            auto it = g_reserving.find(c);
            if (it != g_reserving.end()) g_reserving.erase(it);
            if (e == hipSuccess) {
                g_free_blocks.emplace(c, p);
                g_cached_bytes += c;
                g_held_bytes += c;
                if (g_held_bytes > g_peak_bytes) g_peak_bytes = g_held_bytes;
            }
            g_reserve_cv.notify_all();
        }
    });
}

void comm_sync_side();  // slp_comm.hip: drains the second stream (asynchronous all-reduces may still write cached blocks)

static void trim_cache() {   // (called with g_alloc_mutex held)
    comm_sync_side();
    const double t0 = trace_now();
    for (auto &kv : g_free_blocks) { (void)hipFree(kv.second); g_held_bytes -= kv.first; ++g_driver_calls; }
    g_driver_seconds += trace_now() - t0;
    g_free_blocks.clear();
    g_cached_bytes = 0;
}

void *dev_alloc(size_t bytes, size_t *capacity) {
    static const bool off = [] { const char *e = getenv("SLP_NO_ALLOC_CACHE"); return e && e[0] == '1'; }();
    const size_t want = size_class(bytes);
    std::unique_lock<std::mutex> lock(g_alloc_mutex);
    if (!off) {
        // a cached block is taken when it is not much larger than the request: up to 2 x for small ones, up to 12.5 % above
        // 64 MB (a 10 GB request must not sit on a 20 GB block for its whole lifetime: the out-of-memory retry cannot get
        // that slack back)
        const size_t limit = want > ((size_t)64 << 20) ? want + want / 8 : 2 * want + (1u << 20);
        for (;;) {
            auto it = g_free_blocks.lower_bound(want);
            if (it != g_free_blocks.end() && it->first <= limit) {
                void *p = it->second;
                *capacity = it->first;
                g_cached_bytes -= it->first;
                g_free_blocks.erase(it);
                return p;
            }
            auto rv = g_reserving.lower_bound(want);   // a reservation of this size is on its way: wait for it
            if (rv == g_reserving.end() || rv->first > limit) break;
            const double t0 = trace_now();
            g_reserve_cv.wait(lock);
            g_driver_seconds += trace_now() - t0;      // (time this thread lost to the driver all the same)
        }
    }
    void *p = nullptr;
    const double t0 = trace_now();
    hipError_t e = hipMalloc(&p, want);
    ++g_driver_calls;
    if (e != hipSuccess && !g_free_blocks.empty()) {  // out of memory with blocks parked in the cache: give them back, retry
        (void)hipGetLastError();
        SLP_HIP(hipStreamSynchronize(ctx().stream));
        trim_cache();
        e = hipMalloc(&p, want);
        ++g_driver_calls;
    }
    g_driver_seconds += trace_now() - t0;
    if (e != hipSuccess) throw Error(std::string("hipMalloc of ") + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
    if (trace_enabled()) trace_slow("hipMalloc", want, t0);
    g_held_bytes += want;
    if (g_held_bytes > g_peak_bytes) g_peak_bytes = g_held_bytes;
    *capacity = want;
    return p;
}

void dev_free(void *p, size_t capacity) {
    static const bool off = [] { const char *e = getenv("SLP_NO_ALLOC_CACHE"); return e && e[0] == '1'; }();
    if (!p) return;
    if (off || capacity == 0) {
        const double t0 = trace_now();
        (void)hipFree(p);
        if (trace_enabled()) trace_slow("hipFree", capacity, t0);
        std::lock_guard<std::mutex> lock(g_alloc_mutex);
        g_driver_seconds += trace_now() - t0;
        g_held_bytes -= capacity;
        ++g_driver_calls;
        return;
    }
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    g_free_blocks.emplace(capacity, p);
    g_cached_bytes += capacity;
}

static Context g_ctx;
Context &ctx_unchecked() { return g_ctx; }
Context &ctx() {
    if (!g_ctx.ready)
        throw Error("libslp_hip: slp_init() has not succeeded in this process (no HIP device bound; there is no CPU fallback)");
    return g_ctx;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// y = A x.  L lanes per row; a workgroup of 256 threads covers 256/L rows per
// step and grid-strides over the rows, so consecutive groups read consecutive
// rows (coalesced across groups for short rows, inside a group for long ones).
template <int L>
__global__ __launch_bounds__(kBlock) void k_spmv(i64 nrow, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                 const double *__restrict__ val, const double *__restrict__ x,
                                                 double *__restrict__ y) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    // all lanes of a group run the same trip count, so the shuffles inside row_dot stay convergent per group
    for (i64 row = group; row < nrow; row += ngroups) {
        const double s = row_dot<L>(ptr, idx, val, x, row, sub);
        if (sub == 0) y[row] = s;
    }
}

// ---- device transposition (build_transpose) ----------------------------------------------------------------
// rowid[k] = the row of stored entry k: one wavefront per row, coalesced stores
__global__ __launch_bounds__(kBlock) void k_expand_rows(i64 nrow, const i64 *__restrict__ ptr, i32 *__restrict__ rowid) {
    const int lane = threadIdx.x & (kWave - 1);
    const i64 wave = ((i64)blockIdx.x * kBlock + threadIdx.x) / kWave, nwaves = (i64)gridDim.x * kBlock / kWave;
    for (i64 r = wave; r < nrow; r += nwaves) {
        const i64 s = ptr[r], e = ptr[r + 1];
        for (i64 k = s + lane; k < e; k += kWave) rowid[k] = (i32)r;
    }
}

// The row pointer of the transposed matrix from the SORTED column keys: tptr[c] = first position whose key is >= c.
// Position p writes the pointer of every column in (key[p-1], key[p]] (a run of empty columns is written by the entry
// that ends it); the last position also closes (key[nnz-1], ncol].  Replaces a histogram of 2e9 global atomics.
__global__ void k_ptr_from_sorted(i64 nnz, i64 ncol, const unsigned int *__restrict__ key, i64 *__restrict__ tptr) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x) {
        const i64 c = key[p], prev = p > 0 ? (i64)key[p - 1] : -1;
        for (i64 j = prev + 1; j <= c; ++j) tptr[j] = p;
        if (p == nnz - 1)
            for (i64 j = c + 1; j <= ncol; ++j) tptr[j] = nnz;
    }
}

__global__ void k_max_row_len(i64 nrow, const i64 *__restrict__ ptr, unsigned long long *out) {
    unsigned long long m = 0;
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        const unsigned long long l = (unsigned long long)(ptr[r + 1] - ptr[r]);
        m = l > m ? l : m;
    }
    atomicMax(out, m);
}

// ---------------------------------------------------------------------------
int lanes_for(const CsrDev &a, int order) {
    if (order == SLP_ORDER_SEQUENTIAL) return 1;
    const double mean = a.mean_row_len();
    if (order == SLP_ORDER_AUTO && mean <= 16.0) return 1;
    int l = 2;
    while (l < 64 && (double)l * 4.0 < mean) l <<= 1;  // about >= 4 entries per lane
    return l;
}

void launch_spmv(const CsrDev &a, const double *x, double *y, int order) {
    if (a.nrow == 0) return;
    const int lanes = lanes_for(a, order);
    const int grid = grid_for(a.nrow * lanes, kBlock);
    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_spmv<L>), dim3(grid), dim3(kBlock), 0, ctx().stream, a.nrow,
                                                 a.ptr.p, a.idx.p, a.val.p, x, y));
    SLP_HIP(hipGetLastError());
}

bool comm_active();  // slp_comm.hip

// Shape of A^T without its arrays (the transposed CSR is only formed when something walks it)
static void transposed_shape(slp_matrix *m) {
    if (m->have_at) return;
    m->at.nrow = m->a.ncol;
    m->at.ncol = m->a.nrow;
    m->at.nnz = m->a.nnz;
}

const StripJds *fast_format(slp_matrix *m, bool transposed) {
    if (!m->chunks.empty()) return transposed ? &m->fat : &m->fa;  // chunked matrix: the composites of the chunks' copies
    StripJds &f = transposed ? m->fat : m->fa;
    bool &tried = transposed ? m->tried_fat : m->tried_fa;
    if (!tried) {
        require_csr(m, "building a strip copy");
        tried = true;
        transposed_shape(m);
        if (m->format_policy == 2) return nullptr;  // CSR kernels only
        CsrDev shape;  // this orientation's dimensions (what the format choices look at)
        shape.nrow = transposed ? m->a.ncol : m->a.nrow;
        shape.ncol = transposed ? m->a.nrow : m->a.ncol;
        shape.nnz = m->a.nnz;
        // few distinct stored values (rounded coefficients, +-1 patterns): 4-byte entries, values looked up in LDS
        // quads (4096-row blocks, 3-byte entries) when that still leaves enough row blocks to fill the chip without
        // splitting strips -- a split changes the association of the row sums, and on one GPU every product is the
        // sequential CSR sum bit for bit -- else pairs.  Measured at config 3 for A^T (245 quad blocks): quads with the
        // strips split over two workgroups 1.92 ms, pairs unsplit 2.18 ms, quads unsplit (half the chip) 3.02 ms.  With the
        // rows partitioned over several GPUs the partial sums are re-associated by the all-reduce anyway: quads always
        // (10-13 % faster per iteration on the row blocks of 2 / 4 / 8 ranks, tools/variant_rule.py).
        // SLP_DICT_VARIANT=1|2 forces a geometry.
        const char *ev = getenv("SLP_DICT_VARIANT");
        int variant = ((shape.nrow + 4095) / 4096 >= 384 || comm_active()) ? 2 : 1;
        if (ev && (ev[0] == '1' || ev[0] == '2')) variant = ev[0] - '0';
        else if (variant == 2 && !strip_wanted(shape, 2) && strip_wanted(shape, 1)) variant = 1;  // too sparse for the narrower quad strips
        const bool dict = strip_wanted(shape, variant) && matrix_dictionary(m);
        // long rows that are sparse inside every LDS-sized window (the slice of a 1e7-variable LP): tall cells -- for A^T taken
        // straight from the CSR of A, no transposed CSR is formed
        const bool tall = tall_wanted(shape.nrow, shape.ncol, shape.nnz);
        if (!dict && tall && matrix_dictionary(m) && tall_build(m->a, transposed, f, &m->vdict, m->tall_block_multiple, m->tall_rows_before, m->tall_rows_total)) return &f;
        if (!dict && !strip_wanted(shape, 0) && tall && tall_build(m->a, transposed, f, nullptr, m->tall_block_multiple, m->tall_rows_before, m->tall_rows_total)) return &f;  // arbitrary values: fp64 entries
        if (transposed) build_transpose(m);  // the other copies are converted from the orientation's own CSR
        const CsrDev &a = transposed ? m->at : m->a;
        if (dict) strip_build(a, f, &m->vdict, variant);
        else if (strip_wanted(a, 0)) strip_build(a, f, nullptr, 0);
        else if (strip_wanted(a, 3))  // long rows over a width far beyond an L2: wide strips, x gathered from L2
            strip_build(a, f, matrix_dictionary(m) ? &m->vdict : nullptr, 3);
    }
    return f.ok ? &f : nullptr;
}

// Makes products with A^T possible: a strip copy of A^T when the matrix qualifies for one (tall cells come straight from the
// CSR of A), else the transposed CSR.
void ensure_transposed(slp_matrix *m) {
    if (!m->chunks.empty() || m->have_at) return;
    if (m->csr_released) return;  // (the copies were settled before the release)
    if (fast_format(m, true)) return;
    build_transpose(m);
}

void require_csr(const slp_matrix *m, const char *what) {
    if (!m->chunks.empty())
        throw Error(std::string(what) + ": a chunked matrix never holds the CSR of the whole problem (slp_matrix_chunked_append keeps "
                                        "only the strip copies of every row chunk)");
    if (m->csr_released)
        throw Error(std::string(what) + ": the CSR entries of this matrix were released (slp_matrix_release_csr); only the products "
                                        "over its strip copies remain");
}

bool matrix_dictionary(slp_matrix *m) { return m->format_policy == 0 && value_dictionary(m->a, m->vdict); }

void matrix_spmv(slp_matrix *m, bool transposed, const double *x, double *y, int order) {
    // the strip kernels sum every row with one accumulator in storage order: valid for every `order` -- unless the copy was built
    // with a strip-range split (few row blocks: a 1/8 row partition; SLP_TALL_SPLIT), where a row's sum is S partial sums added
    // in range order.  SLP_ORDER_SEQUENTIAL asks for the single chain of the CSR walk whatever the partition (the generator's
    // b_upper = ceil((A x_f + ..) 1000) / 1000 shows the order of a row's additions in a tenth of the rows, slp_random.hip): LDS
    // strips then run one workgroup per row block over all strips; split tall cells hand over to the CSR kernel while the CSR exists.
    if (const StripJds *f = fast_format(m, transposed)) {
        if (order == SLP_ORDER_SEQUENTIAL && strip_has_tall_split(*f) && m->chunks.empty() && !m->csr_released) {
            if (transposed) build_transpose(m);
            launch_spmv(transposed ? m->at : m->a, x, y, order);
            return;
        }
        if (order == SLP_ORDER_SEQUENTIAL) ++g_strip_single_chain;
        try {
            strip_spmv(*f, x, y);
        } catch (...) {
            if (order == SLP_ORDER_SEQUENTIAL) --g_strip_single_chain;
            throw;
        }
        if (order == SLP_ORDER_SEQUENTIAL) --g_strip_single_chain;
        return;
    }
    require_csr(m, "CSR product");
    if (transposed) build_transpose(m);
    launch_spmv(transposed ? m->at : m->a, x, y, order);
}

void invalidate_derived(slp_matrix *m) {
    m->have_at = false;
    m->at = CsrDev();
    m->fa = StripJds();
    m->fat = StripJds();
    m->tried_fa = m->tried_fat = false;
    m->vdict = ValueDict();
}

void finish_stats(CsrDev &a) {  // fills a.max_row_len (kernel choices depend on it: every constructor must call this)
    DevBuf<unsigned long long> mx(1);
    mx.zero();
    if (a.nrow) {
        hipLaunchKernelGGL(k_max_row_len, dim3(grid_for(a.nrow, kBlock)), dim3(kBlock), 0, ctx().stream, a.nrow, a.ptr.p, mx.p);
        SLP_HIP(hipGetLastError());
    }
    unsigned long long h = 0;
    mx.download(&h, 1);
    a.max_row_len = (i64)h;
}

// Stable device transposition: a radix sort of the entries keyed by column
// keeps, inside every column, the entries in storage order of A, i.e. by
// increasing row (and storage order inside a row) -- the order in which
// scipy's csc_matvec accumulates `y * A`.
void build_transpose(slp_matrix *m) {
    if (m->have_at) return;
    require_csr(m, "build_transpose");
    Phase ph("build_transpose");
    const CsrDev &a = m->a;
    CsrDev &t = m->at;
    hipStream_t st = ctx().stream;
    t.nrow = a.ncol;
    t.ncol = a.nrow;
    t.nnz = a.nnz;
    t.ptr.alloc((size_t)t.nrow + 1);
    t.idx.alloc((size_t)a.nnz);
    t.val.alloc((size_t)a.nnz);
    if (a.nnz) {
        // Stable LSD radix sort of the entries keyed by column, the (row, value) pair of every entry travelling with its key:
        // whole records stream through every pass (no positions, so no 2^32 limit and no random gather afterwards -- the
        // gather by sorted position moved 0.3-0.5 TB at config 3, profiles/r02_c3_pmc_hbm.json).  The row of every source
        // entry comes from a coalesced expansion of the row pointer.
        DevBuf<i32> rowid((size_t)a.nnz);
        hipLaunchKernelGGL(k_expand_rows, dim3(grid_for(a.nrow * kWave, kBlock)), dim3(kBlock), 0, st, a.nrow, a.ptr.p, rowid.p);
        SLP_HIP(hipGetLastError());
        DevBuf<unsigned int> key_out((size_t)a.nnz);
        unsigned int bits = 1;
        while (bits < 32 && ((i64)1 << bits) < a.ncol) ++bits;
        const unsigned int *keys_in = reinterpret_cast<const unsigned int *>(a.idx.p);
        auto vals_in = rocprim::make_zip_iterator(rocprim::make_tuple((const i32 *)rowid.p, (const double *)a.val.p));
        auto vals_out = rocprim::make_zip_iterator(rocprim::make_tuple(t.idx.p, t.val.p));
        size_t bytes = 0;
        SLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, key_out.p, vals_in, vals_out, (size_t)a.nnz, 0u, bits, st));
        {
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys_in, key_out.p, vals_in, vals_out, (size_t)a.nnz, 0u, bits, st));
            SLP_HIP(hipStreamSynchronize(st));
        }
        rowid.release();
        // column pointer straight from the sorted keys
        hipLaunchKernelGGL(k_ptr_from_sorted, dim3(grid_for(a.nnz, kBlock)), dim3(kBlock), 0, st, a.nnz, t.nrow, key_out.p, t.ptr.p);
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(st));
    } else {
        t.ptr.zero();
    }
    finish_stats(t);
    m->have_at = true;
}

// One pass over the uploaded arrays: indptr non-decreasing, 0 <= index < ncol.  Every entry point that takes host CSR
// arrays goes through matrix_from_host, so a bad index is an slp_last_error instead of an out-of-bounds device access.
__global__ void k_validate_csr(i64 nrow, i64 ncol, i64 nnz, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, int *__restrict__ bad) {
    const i64 t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x, stride = (i64)gridDim.x * blockDim.x;
    int b = 0;
    for (i64 r = t0; r < nrow; r += stride)
        if (ptr[r + 1] < ptr[r]) b |= 1;
    for (i64 k = t0; k < nnz; k += stride)
        if (idx[k] < 0 || (i64)idx[k] >= ncol) b |= 2;
    if (b) atomicOr(bad, b);
}

static slp_matrix *matrix_from_host(i64 nrow, i64 ncol, const i64 *indptr, const i32 *indices, const double *data) {
    SLP_REQUIRE(nrow >= 0 && ncol >= 0 && indptr != nullptr, "slp_matrix_create: bad arguments");
    SLP_REQUIRE(ncol < (i64)1 << 31, "column count must fit int32");
    const i64 nnz = indptr[nrow];
    SLP_REQUIRE(indptr[0] == 0 && nnz >= 0, "slp_matrix_create: indptr must start at 0");
    SLP_REQUIRE(nnz == 0 || (indices != nullptr && data != nullptr), "slp_matrix_create: NULL indices / data");
    ctx();
    auto *m = new slp_matrix();
    try {
        m->a.nrow = nrow;
        m->a.ncol = ncol;
        m->a.nnz = nnz;
        m->a.ptr.upload(indptr, (size_t)nrow + 1);
        m->a.idx.upload(indices, (size_t)nnz);
        m->a.val.upload(data, (size_t)nnz);
        DevBuf<int> bad(1);
        bad.zero();
        hipLaunchKernelGGL(k_validate_csr, dim3(grid_for(std::max(nrow, nnz), kBlock)), dim3(kBlock), 0, ctx().stream, nrow, ncol, nnz,
                           m->a.ptr.p, m->a.idx.p, bad.p);
        SLP_HIP(hipGetLastError());
        int hbad = 0;
        bad.download(&hbad, 1);
        SLP_REQUIRE(!(hbad & 1), "slp_matrix_create: indptr must be non-decreasing");
        SLP_REQUIRE(!(hbad & 2), "slp_matrix_create: column index out of range [0, ncol)");
        finish_stats(m->a);
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

// ---- row gather: out row r = scale[r] * (row src[r] of a); used for the device-side one-sided stacking
// [A[up]; -A[lo]] of ChambollePockPPD.py:74-88 (the CSR never returns to the host)
__global__ void k_gather_len(i64 count, const i64 *__restrict__ src, const i64 *__restrict__ ptr, unsigned long long *__restrict__ len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < count; r += (i64)gridDim.x * blockDim.x)
        len[r] = (unsigned long long)(ptr[src[r] + 1] - ptr[src[r]]);
}

__global__ __launch_bounds__(kBlock) void k_gather_rows(i64 count, const i64 *__restrict__ src, const double *__restrict__ scale,
                                                        const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                        const double *__restrict__ val, const i64 *__restrict__ optr,
                                                        i32 *__restrict__ oidx, double *__restrict__ oval) {
    const int lane = threadIdx.x & (kWave - 1);
    const i64 wave = ((i64)blockIdx.x * kBlock + threadIdx.x) / kWave, nwaves = (i64)gridDim.x * kBlock / kWave;
    for (i64 r = wave; r < count; r += nwaves) {  // one wavefront per row: coalesced copies
        const i64 s = ptr[src[r]], e = ptr[src[r] + 1], o = optr[r];
        const double f = scale[r];
        for (i64 k = s + lane; k < e; k += kWave) {
            oidx[o + (k - s)] = idx[k];
            oval[o + (k - s)] = (f == 1.0) ? val[k] : f * val[k];  // -1: an exact negation, like scipy's `-A`
        }
    }
}

static slp_matrix *matrix_gather_rows(slp_matrix *a, i64 count, const i64 *rows, const double *scale) {
    SLP_REQUIRE(a && count >= 0 && (count == 0 || (rows && scale)), "slp_matrix_gather_rows: bad arguments");
    require_csr(a, "slp_matrix_gather_rows");
    for (i64 r = 0; r < count; ++r) SLP_REQUIRE(rows[r] >= 0 && rows[r] < a->a.nrow, "slp_matrix_gather_rows: row out of range");
    hipStream_t st = ctx().stream;
    auto *m = new slp_matrix();
    try {
        DevBuf<i64> src((size_t)count);
        DevBuf<double> sc((size_t)count);
        src.upload(rows, (size_t)count);
        sc.upload(scale, (size_t)count);
        DevBuf<unsigned long long> len((size_t)count + 1);
        len.zero();
        if (count) hipLaunchKernelGGL(k_gather_len, dim3(grid_for(count, kBlock)), dim3(kBlock), 0, st, count, src.p, a->a.ptr.p, len.p);
        SLP_HIP(hipGetLastError());
        m->a.nrow = count;
        m->a.ncol = a->a.ncol;
        m->a.ptr.alloc((size_t)count + 1);
        size_t bytes = 0;
        SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, len.p, (unsigned long long *)m->a.ptr.p, 0ull, (size_t)count + 1,
                                        rocprim::plus<unsigned long long>(), st));
        DevBuf<char> tmp(bytes);
        SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, len.p, (unsigned long long *)m->a.ptr.p, 0ull, (size_t)count + 1,
                                        rocprim::plus<unsigned long long>(), st));
        i64 nnz = 0;
        SLP_HIP(hipMemcpyAsync(&nnz, m->a.ptr.p + count, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        m->a.nnz = nnz;
        m->a.idx.alloc((size_t)nnz);
        m->a.val.alloc((size_t)nnz);
        if (count)
            hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(count * kWave, kBlock)), dim3(kBlock), 0, st, count, src.p, sc.p, a->a.ptr.p,
                               a->a.idx.p, a->a.val.p, m->a.ptr.p, m->a.idx.p, m->a.val.p);
        SLP_HIP(hipGetLastError());
        finish_stats(m->a);
        SLP_HIP(hipStreamSynchronize(st));
    } catch (...) {
        delete m;
        throw;
    }
    return m;
}

slp_matrix *matrix_row_slice(slp_matrix *m, i64 r0, i64 r1) {
    SLP_REQUIRE(m && 0 <= r0 && r0 <= r1 && r1 <= m->a.nrow, "matrix_row_slice: bad row range");
    std::vector<i64> rows((size_t)(r1 - r0));
    for (i64 r = r0; r < r1; ++r) rows[(size_t)(r - r0)] = r;
    std::vector<double> one(rows.size(), 1.0);   // (v * 1.0 = v)
    return matrix_gather_rows(m, r1 - r0, rows.data(), one.data());
}

void matrix_drop_csr(slp_matrix *m) {
    SLP_HIP(hipStreamSynchronize(ctx().stream));
    m->a.ptr.release(); m->a.idx.release(); m->a.val.release();
    m->at.ptr.release(); m->at.idx.release(); m->at.val.release();
    m->tried_fa = m->tried_fat = true;
    m->csr_released = true;
}

}  // namespace slp

using namespace slp;

extern "C" {

int slp_version(void) { return 100; }

// 0 for the shipped library.  Bit 0: built with -DSLP_ABLATION (kernels with parts removed: wrong results); bit 1: a kernel-lab
// variant (`make variant`, -DSLP_LAB_VARIANT).  pysparselp_amd/_lib.py refuses a non-zero library unless SLP_LIB_VARIANT asked for it.
int slp_build_flags(void) {
    int f = 0;
#ifdef SLP_ABLATION
    f |= 1;
#endif
#ifdef SLP_LAB_VARIANT
    f |= 2;
#endif
    return f;
}

slp_matrix *slp_matrix_gather_rows(slp_matrix *a, int64_t count, const int64_t *rows, const double *scale) {
    SLP_API_PTR({ return matrix_gather_rows(a, count, rows, scale); })
}

const char *slp_last_error(void) { return g_err.c_str(); }

int slp_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return -1;
    }
    return n;
}

int slp_init(int device) {
    SLP_API_INT({
        Context &c = ctx_unchecked();
        int n = 0;
        SLP_HIP(hipGetDeviceCount(&n));
        SLP_REQUIRE(n > 0, "no HIP device visible");
        SLP_REQUIRE(device >= 0 && device < n, "slp_init: device index out of range");
        if (c.ready && c.device == device) return 0;
        SLP_REQUIRE(!c.ready, "slp_init: this process is already bound to another device");
        SLP_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        SLP_HIP(hipGetDeviceProperties(&prop, device));
        c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        SLP_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        SLP_HIP(hipEventCreate(&c.ev0));
        SLP_HIP(hipEventCreate(&c.ev1));
        c.device = device;
        c.ready = true;
    })
}

int slp_trim(void) {
    SLP_API_INT({
        SLP_HIP(hipStreamSynchronize(ctx().stream));
        reserve_join();
        std::lock_guard<std::mutex> lock(g_alloc_mutex);
        trim_cache();
    })
}

int64_t slp_cached_bytes(void) { return (int64_t)g_cached_bytes; }

int slp_alloc_stats(double out[5], int reset) {
    SLP_API_INT({
        std::lock_guard<std::mutex> lock(g_alloc_mutex);
        if (out) {
            out[0] = g_driver_seconds;
            out[1] = (double)g_peak_bytes;
            out[2] = (double)g_held_bytes;
            out[3] = (double)g_driver_calls;
            out[4] = g_background_seconds;
        }
        if (reset) { g_driver_seconds = 0.0; g_background_seconds = 0.0; g_peak_bytes = g_held_bytes; g_driver_calls = 0; }
    })
}

int slp_synchronize(void) { SLP_API_INT({ SLP_HIP(hipStreamSynchronize(ctx().stream)); }) }

int slp_timer_start(void) { SLP_API_INT({ SLP_HIP(hipEventRecord(ctx().ev0, ctx().stream)); }) }

int slp_timer_stop(double *ms) {
    SLP_API_INT({
        SLP_HIP(hipEventRecord(ctx().ev1, ctx().stream));
        SLP_HIP(hipEventSynchronize(ctx().ev1));
        float f = 0.f;
        SLP_HIP(hipEventElapsedTime(&f, ctx().ev0, ctx().ev1));
        if (ms) *ms = (double)f;
    })
}

slp_matrix *slp_matrix_create(int64_t nrow, int64_t ncol, const int64_t *indptr, const int32_t *indices,
                              const double *data) {
    SLP_API_PTR({ return matrix_from_host(nrow, ncol, indptr, indices, data); })
}

void slp_matrix_destroy(slp_matrix *a) { delete a; }

int64_t slp_matrix_nnz(const slp_matrix *a) { return a ? a->a.nnz : -1; }

int slp_matrix_spmv(slp_matrix *m, const double *x, double *y, int order) {
    SLP_API_INT({
        SLP_REQUIRE(m && x && y, "slp_matrix_spmv: NULL argument");
        m->vx.upload(x, (size_t)m->a.ncol);
        if (m->vy.n < (size_t)m->a.nrow) m->vy.alloc((size_t)m->a.nrow);
        matrix_spmv(m, false, m->vx.p, m->vy.p, order);
        m->vy.download(y, (size_t)m->a.nrow);
    })
}

int slp_matrix_spmv_t(slp_matrix *m, const double *y, double *out, int order) {
    SLP_API_INT({
        SLP_REQUIRE(m && y && out, "slp_matrix_spmv_t: NULL argument");
        DevBuf<double> vy((size_t)m->a.nrow), vo((size_t)m->a.ncol);
        vy.upload(y, (size_t)m->a.nrow);
        matrix_spmv(m, true, vy.p, vo.p, order);
        vo.download(out, (size_t)m->a.ncol);
    })
}

int slp_matrix_spmv_abs_pow(slp_matrix *m, int transposed, double p, const double *x, double *y) {
    SLP_API_INT({
        SLP_REQUIRE(m && x && y, "slp_matrix_spmv_abs_pow: NULL argument");
        const StripJds *f = fast_format(m, transposed != 0);
        SLP_REQUIRE(f && strip_abs_pow_supported(*f), "slp_matrix_spmv_abs_pow: needs a strip / tall-cell copy of this orientation that can "
                                                     "raise its entries to a power (slp_matrix_spmv_kernel 1-4, 6, 7)");
        const size_t nin = (size_t)(transposed ? m->a.nrow : m->a.ncol), nout = (size_t)(transposed ? m->a.ncol : m->a.nrow);
        DevBuf<double> vx(nin), vo(nout);
        vx.upload(x, nin);
        strip_spmv_abs_pow(*f, p, vx.p, vo.p);
        SLP_HIP(hipStreamSynchronize(ctx().stream));
        vo.download(y, nout);
    })
}

int slp_matrix_download(slp_matrix *m, int transposed, int64_t *indptr, int32_t *indices, double *data) {
    SLP_API_INT({
        SLP_REQUIRE(m, "slp_matrix_download: NULL matrix");
        require_csr(m, "slp_matrix_download");
        if (transposed) build_transpose(m);
        const CsrDev &a = transposed ? m->at : m->a;
        if (indptr) a.ptr.download(indptr, (size_t)a.nrow + 1);
        if (indices) a.idx.download(indices, (size_t)a.nnz);
        if (data) a.val.download(data, (size_t)a.nnz);
    })
}

int slp_matrix_download_rows(slp_matrix *m, int transposed, int64_t row0, int64_t count, int64_t *indptr, int32_t *indices,
                             double *data) {
    SLP_API_INT({
        SLP_REQUIRE(m, "slp_matrix_download_rows: NULL matrix");
        require_csr(m, "slp_matrix_download_rows");
        if (transposed) build_transpose(m);
        const CsrDev &a = transposed ? m->at : m->a;
        SLP_REQUIRE(row0 >= 0 && count >= 0 && row0 + count <= a.nrow, "slp_matrix_download_rows: rows out of range");
        hipStream_t st = ctx().stream;
        i64 ends[2] = {0, 0};
        SLP_HIP(hipMemcpyAsync(&ends[0], a.ptr.p + row0, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipMemcpyAsync(&ends[1], a.ptr.p + row0 + count, sizeof(i64), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        const size_t k0 = (size_t)ends[0], k = (size_t)(ends[1] - ends[0]);
        if (indptr) SLP_HIP(hipMemcpyAsync(indptr, a.ptr.p + row0, (size_t)(count + 1) * sizeof(i64), hipMemcpyDeviceToHost, st));
        if (indices && k) SLP_HIP(hipMemcpyAsync(indices, a.idx.p + k0, k * sizeof(i32), hipMemcpyDeviceToHost, st));
        if (data && k) SLP_HIP(hipMemcpyAsync(data, a.val.p + k0, k * sizeof(double), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
    })
}

int slp_matrix_release_csr(slp_matrix *m) {
    SLP_API_INT({
        SLP_REQUIRE(m, "slp_matrix_release_csr: NULL matrix");
        if (m->csr_released || !m->chunks.empty()) return 0;
        const StripJds *f0 = fast_format(m, false), *f1 = fast_format(m, true);
        SLP_REQUIRE(f0 && f1, "slp_matrix_release_csr: the matrix does not run on strip copies in both orientations; its CSR arrays "
                              "are the only copy of the entries");
        SLP_REQUIRE(m->csr_bound == 0, "slp_matrix_release_csr: a solver created on this matrix iterates on its CSR arrays "
                                       "(one of its iteration halves has no strip copy to run on); destroy it first");
        SLP_HIP(hipStreamSynchronize(ctx().stream));
        m->a.idx.release(); m->a.val.release();
        m->at.idx.release(); m->at.val.release();
        m->csr_released = true;
    })
}

int slp_device_memory(int64_t *free_bytes, int64_t *total_bytes) {
    SLP_API_INT({
        ctx();
        size_t f = 0, t = 0;
        SLP_HIP(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = (int64_t)f;
        if (total_bytes) *total_bytes = (int64_t)t;
    })
}

int slp_matrix_set_format(slp_matrix *m, int policy) {
    SLP_API_INT({
        SLP_REQUIRE(m && policy >= 0 && policy <= 2, "slp_matrix_set_format: bad arguments");
        require_csr(m, "slp_matrix_set_format");
        SLP_REQUIRE(m->borrowers == 0, "slp_matrix_set_format: a solver created on this matrix is still alive (it holds pointers "
                                       "into the copies this call would free)");
        if (policy == m->format_policy) return 0;
        SLP_HIP(hipStreamSynchronize(ctx().stream));
        m->fa = StripJds();
        m->fat = StripJds();
        m->tried_fa = m->tried_fat = false;
        m->format_policy = policy;
    })
}

int slp_matrix_spmv_kernel(slp_matrix *m, int transposed) {
    try {
        SLP_REQUIRE(m, "slp_matrix_spmv_kernel: NULL matrix");
        const StripJds *f = fast_format(m, transposed != 0);
        if (!f) return 0;
        if (!f->parts.empty()) f = f->parts[0];  // chunked matrix: the kernel of its first chunk (all chunks of one shape share it)
        if (f->tall) return f->D > 0 ? 6 : 7;
        if (f->wide) return f->D > 0 ? 4 : 5;
        return f->D > 0 ? (f->rpl == 4 ? 3 : 2) : 1;
    } catch (const std::exception &e) {
        set_error(e.what());
        return -1;
    }
}

int64_t slp_matrix_format_bytes(slp_matrix *m, int transposed) {
    try {
        SLP_REQUIRE(m, "slp_matrix_format_bytes: NULL matrix");
        const StripJds *f = fast_format(m, transposed != 0);
        if (!f) return (int64_t)(12 * m->a.nnz + 8 * ((transposed ? m->a.ncol : m->a.nrow) + 1));
        return (int64_t)strip_format_bytes(*f);
    } catch (const std::exception &e) {
        set_error(e.what());
        return -1;
    }
}

int slp_product_timing(int on) {
    SLP_API_INT({
        (void)ctx();
        product_timing(on != 0);
    })
}

int slp_product_timing_read(double out[3]) {
    SLP_API_INT({
        SLP_REQUIRE(out, "slp_product_timing_read: NULL argument");
        product_timing_read(out);
    })
}

int slp_matrix_bench_spmv(slp_matrix *m, int transposed, int order, int reps, double *ms) {
    SLP_API_INT({
        SLP_REQUIRE(m && reps > 0 && ms, "slp_matrix_bench_spmv: bad arguments");
        if (transposed) ensure_transposed(m);
        CsrDev a;  // dimensions of the orientation
        a.nrow = transposed ? m->a.ncol : m->a.nrow;
        a.ncol = transposed ? m->a.nrow : m->a.ncol;
        DevBuf<double> vx((size_t)a.ncol), vy((size_t)a.nrow);
        std::vector<double> h((size_t)a.ncol);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 1e-3 * (double)(i % 1000);
        vx.upload(h.data(), h.size());
        // warm-up: at least one product, then products until ~100 ms have gone by (at most 64).  The vector above was filled on the
        // host while the GPU sat idle: the first products after that run below the clocks of a loaded chip -- a 5-product
        // measurement of a 3 ms product read 3.28 ms where 100 products read 2.80 (config 5's shape; tools/lab, round 6).
        {
            matrix_spmv(m, transposed != 0, vx.p, vy.p, order);
            SLP_HIP(hipStreamSynchronize(ctx().stream));
            const double t0 = trace_now();
            for (int w = 0; w < 64 && trace_now() - t0 < 0.1; ++w) {
                matrix_spmv(m, transposed != 0, vx.p, vy.p, order);
                SLP_HIP(hipStreamSynchronize(ctx().stream));
            }
        }
        const char *two = getenv("SLP_BENCH_TWO_VECTORS");  // time the two-vector pass (strip format only)
        const StripJds *f2 = (two && two[0] == '1') ? fast_format(m, transposed != 0) : nullptr;
        DevBuf<double> vx2, vy2;
        if (f2) {
            vx2.alloc((size_t)a.ncol);
            vy2.alloc((size_t)a.nrow);
            vx2.copy_from(vx);
            strip_spmv2(*f2, vx.p, vx2.p, vy.p, vy2.p);
        }
        SLP_HIP(hipEventRecord(ctx().ev0, ctx().stream));
        for (int r = 0; r < reps; ++r) {
            if (f2) strip_spmv2(*f2, vx.p, vx2.p, vy.p, vy2.p);
            else matrix_spmv(m, transposed != 0, vx.p, vy.p, order);
        }
        SLP_HIP(hipEventRecord(ctx().ev1, ctx().stream));
        SLP_HIP(hipEventSynchronize(ctx().ev1));
        float f = 0.f;
        SLP_HIP(hipEventElapsedTime(&f, ctx().ev0, ctx().ev1));
        *ms = (double)f / reps;
    })
}

}  // extern "C"

// slp_chunked.hip -- a constraint matrix assembled from ROW CHUNKS whose CSR never coexists: the way an LP larger than
// one CSR copy of itself gets resident on one GPU (BASELINE config 4's 1e7 x 2e7 matrix at density 1e-4: 2e10 stored
// entries = 240 GB of CSR per orientation, but 104 GB per orientation as tall cells).
//
//   g = slp_matrix_chunked_create(ncol)
//   for every chunk of rows, in order:  c = slp_matrix_create(...) or slp_matrix_random(...);  slp_matrix_chunked_append(g, c)
//
// append converts the chunk into its product copies for BOTH orientations (tall cells of A_k and, straight from the same
// CSR, of A_k^T; LDS strips where those serve the shape), keeps the two per-row sums of squares the ADMM row scaling needs
// (tools.py:272-290), and releases the chunk's CSR: what stays is 5-10 bytes per entry instead of 24.  The group then
// behaves like any slp_matrix in the products and the at-scale solvers (slp_cp_create_on, slp_admm_cg_create_on*,
// slp_random_lp_vectors chunk by chunk):
//   A x    = the chunks' products, one row range each
//   A^T y  = chunk k multiplies its slice of y and CONTINUES the column sums chunk k - 1 left in the output (the kernels'
//            `accum` entry): every column is still one chain of additions in row order -- bit for bit the unchunked
//            product and scipy's csc_matvec (ChambollePockPPD.py:206,216 ; ADMM.py:148), for ANY chunking.
// It is the 8-rank row partition of DESIGN.md section 5 minus the wire: rank k's block = chunk k.
#include "slp_common.h"
#include "slp_kernels.h"

slp_matrix::~slp_matrix() {
    for (slp_matrix *c : chunks) delete c;
}

namespace slp {

void composite_of_chunks(const slp_matrix *g, bool transposed, size_t k0, size_t k1, StripJds &f) {
    f.parts.clear();
    f.part_off.clear();
    f.parts_cols = transposed;
    f.ok = k1 > k0;
    f.nrow = transposed ? g->a.ncol : g->a.nrow;
    f.ncol = transposed ? g->a.nrow : g->a.ncol;
    f.nnz = 0;
    f.D = 0;
    f.tall = true;
    bool first = true;
    for (size_t k = k0; k < k1; ++k) {
        const StripJds *p = transposed ? &g->chunks[k]->fat : &g->chunks[k]->fa;
        f.parts.push_back(p);
        f.part_off.push_back(g->chunk_row0[k]);
        f.nnz += g->chunks[k]->a.nnz;
        f.D = first ? p->D : (p->D > 0 && f.D > 0 ? std::max(f.D, p->D) : 0);  // > 0: every chunk runs on a value dictionary
        f.tall = f.tall && p->tall;
        first = false;
    }
    f.fused = f.tall && tall_fuse(f);   // one launch per product when every chunk runs on tall cells of one kind
}

static void refresh_composites(slp_matrix *g) {
    for (int t = 0; t < 2; ++t) composite_of_chunks(g, t != 0, 0, g->chunks.size(), t ? g->fat : g->fa);
    g->at.nrow = g->a.ncol;
    g->at.ncol = g->a.nrow;
    g->at.nnz = g->a.nnz;
}

}  // namespace slp

using namespace slp;

extern "C" {

slp_matrix *slp_matrix_chunked_create(int64_t ncol) {
    SLP_API_PTR({
        SLP_REQUIRE(ncol > 0 && ncol < ((i64)1 << 31), "slp_matrix_chunked_create: bad column count");
        ctx();
        auto *g = new slp_matrix();
        g->a.ncol = ncol;
        g->tried_fa = g->tried_fat = true;
        g->have_at = true;  // nothing may ask for the transposed CSR of a chunked matrix
        g->csr_released = true;
        return g;
    })
}

int slp_matrix_chunked_append(slp_matrix *g, slp_matrix *c) {
    SLP_API_INT({
        SLP_REQUIRE(g && c && g != c, "slp_matrix_chunked_append: NULL argument");
        SLP_REQUIRE(g->csr_released && g->a.ptr.p == nullptr && (g->a.nrow == 0 || !g->chunks.empty()),
                    "slp_matrix_chunked_append: the first argument is not a chunked matrix (slp_matrix_chunked_create)");
        SLP_REQUIRE(c->chunks.empty() && !c->csr_released, "slp_matrix_chunked_append: the chunk must be a plain matrix with its CSR");
        SLP_REQUIRE(c->a.ncol == g->a.ncol, "slp_matrix_chunked_append: the chunk has another column count");
        SLP_REQUIRE(c->a.nrow > 0 && c->a.nnz > 0, "slp_matrix_chunked_append: empty chunk");
        SLP_REQUIRE(g->borrowers == 0, "slp_matrix_chunked_append: a solver created on the chunked matrix is alive");
        SLP_REQUIRE(c->borrowers == 0 && !c->scaled, "slp_matrix_chunked_append: the chunk is in use by a solver (or was scaled in place)");
        // sub-vectors of y start at the chunk's first row: the strip kernels stage x with 16-byte loads
        SLP_REQUIRE(g->a.nrow % 2 == 0, "slp_matrix_chunked_append: every chunk but the last must have an even number of rows");
        Phase ph("slp_matrix_chunked_append");
        // K chunks announced (slp_matrix_chunked_expect): the chunks' tall cells will run in ONE grid (tall_fuse), so this chunk
        // brings 1 / K of a multiple of the CU count of row blocks instead of a whole multiple -- its row blocks stay as tall
        // as the LDS allows however finely the rows are chunked (config 4 in 16 chunks: 128 blocks of 9766 rows each, 2048 in
        // the grid; a whole multiple per chunk would halve the blocks' height and double the per-cell work)
        if (g->expect_rows > 0 && g->a.nrow + c->a.nrow <= g->expect_rows && !c->tried_fa) {
            // the whole matrix's row count is known (slp_matrix_chunked_expect_rows): the chunk brings its share of the row blocks
            const char *e = getenv("SLP_TALL_FUSE");
            if (!(e && e[0] == '0')) {
                c->tall_rows_before = g->a.nrow;
                c->tall_rows_total = g->expect_rows;
            }
        } else if (g->expect_chunks > 1 && !c->tried_fa) {
            const char *e = getenv("SLP_TALL_FUSE");
            if (!(e && e[0] == '0')) {
                i64 a = ctx().num_cu, b = g->expect_chunks;
                while (b) { const i64 t = a % b; a = b; b = t; }
                c->tall_block_multiple = ctx().num_cu / a;
            }
        }
        // both product copies straight from the chunk's CSR
        const StripJds *f0 = fast_format(c, false), *f1 = fast_format(c, true);
        SLP_REQUIRE(f0 && f1, "slp_matrix_chunked_append: the chunk does not qualify for strip copies in both orientations (too small or "
                              "unsorted rows): use an ordinary matrix for problems of that size");
        if (f0->D > 0 && f1->D > 0) {  // what a later ADMM set-up needs of the entries (deferred row scaling)
            c->rowsq.alloc(2 * (size_t)c->a.nrow);
            matrix_row_squares(c->a, c->rowsq.p);
        }
        SLP_REQUIRE(slp_matrix_release_csr(c) == 0, slp_last_error());
        c->a.ptr.release();   // (a plain release keeps the row pointers for diagnostics; a chunk needs nothing of them)
        c->at.ptr.release();
        g->chunks.push_back(c);
        g->chunk_row0.push_back(g->a.nrow);
        g->a.nrow += c->a.nrow;
        g->a.nnz += c->a.nnz;
        g->a.max_row_len = std::max(g->a.max_row_len, c->a.max_row_len);
        refresh_composites(g);
        // more chunks to come: their packet streams will be as large as this one's -- have a helper thread take those blocks
        // from the driver while the next chunk is generated and converted
        if ((i64)g->chunks.size() < g->expect_chunks) {
            std::vector<size_t> sizes;
            for (const StripJds *f : {f0, f1})
                if (f->tall)
                    for (const auto &b : f->tall_pay) sizes.push_back(b.cap);
            dev_reserve_async(sizes);
        }
    })
}

int slp_matrix_chunked_expect(slp_matrix *g, int64_t chunks) {
    SLP_API_INT({
        SLP_REQUIRE(g && g->csr_released && g->a.ptr.p == nullptr && chunks >= 0, "slp_matrix_chunked_expect: bad arguments");
        g->expect_chunks = chunks;
    })
}

int slp_matrix_chunked_expect_rows(slp_matrix *g, int64_t chunks, int64_t rows) {
    SLP_API_INT({
        SLP_REQUIRE(g && g->csr_released && g->a.ptr.p == nullptr && chunks >= 0 && rows >= 0, "slp_matrix_chunked_expect_rows: bad arguments");
        g->expect_chunks = chunks;
        g->expect_rows = rows;
    })
}

int64_t slp_matrix_chunks(const slp_matrix *g) { return g ? (int64_t)g->chunks.size() : -1; }

int64_t slp_matrix_strip_width(slp_matrix *m, int transposed) {
    if (!m) return -1;
    int64_t c = 0;
    const int rc = [&]() -> int {
        SLP_API_INT({
            const StripJds *f = fast_format(m, transposed != 0);
            if (f && !m->chunks.empty() && !f->parts.empty()) f = f->parts[0];
            c = (f && f->ok) ? (int64_t)f->C : 0;
        })
    }();
    return rc == 0 ? c : -1;
}

int64_t slp_matrix_product_launches(const slp_matrix *g, int transposed) {
    if (!g) return -1;
    if (g->chunks.empty()) return 1;
    return (transposed ? g->fat : g->fa).fused ? 1 : (int64_t)g->chunks.size();
}

}  // extern "C"

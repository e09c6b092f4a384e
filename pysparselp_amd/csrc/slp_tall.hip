// slp_tall.hip -- SpMV for rows that are LONG over a width far beyond the caches but SPARSE inside any LDS-sized
// window: the per-rank slice of a 10^7-variable LP (2.5e6 x 1e7 at density 1e-4: 1000 entries per row, 0.4 per 4096
// columns) and its transpose.  Reference products: `a * x` / `y * a` of ChambollePockPPD.py:206,216,235,240 and
// ADMM.py:148,220,262 (scipy csr_matvec / csc_matvec).
//
// Why another format: the LDS strips of slp_strip.hip pay 3 bytes of metadata per (row, strip) and a strip-JDS cell per
// 2048 rows -- fine at >= 3 entries per (row, strip), hopeless at 0.4; the wide strips that replaced them there gather
// every x from L2 (one 64-byte request per entry: 2e11 requests/s, 0.10 of the HBM peak).  Here the cell is TALL: a
// workgroup owns R ~ 10^4 rows (their R running sums live in LDS, 78 KB) and walks strips of 4096 columns (the x-tile,
// double-buffered, 64 KB of LDS), so that a cell holds ~4000 entries although a row has < 1.
//
//   item (5 bytes)  = value id (11 bits) | column inside the strip (12 bits) << 11 | (local row + 1) (14 bits) << 23
//                     -- self-describing: no per-row metadata, rows without entries in a cell cost nothing; row field 0 is
//                     the product kernel's scratch cell: a skip item, and the zero word a lane outside a slot loads, need
//                     no predicate anywhere
//   cell            = the items of its non-empty rows, every row handed WHOLE to one of the 1024 lanes (rows sorted by
//                     their entry count, dealt round by round: lane p takes sorted positions p, p + 1024, ...), so a
//                     lane's list is its rows' entries in storage order and list lengths never increase with p
//   packet          = 8 consecutive list positions ("slots") of all lanes: slot k holds item k of every lane whose
//                     list is longer than k -- a prefix of the lanes -- so lane p reads word p of the slot: coalesced,
//                     no lengths stored.  Low 4 bytes of the items slot by slot, then the fifth bytes, four slots to
//                     a word.  32-byte header: payload offset, the 8 slot widths, the x-tile it carries.
//
// A lane adds its items into the running sums in order: `acc[row] += value * x[col]`, one row after the other, so every
// row is accumulated by ONE lane per cell, cells in strip order: the result is the sequential single-accumulator CSR
// row sum, bit for bit, for ANY number of row blocks -- R is chosen so that the row blocks are a multiple of the CU
// count (no strips split over workgroups, no partial sums re-associated).
//
// The kernel (slp_tall_spmv.hip) is a software pipeline over packets (typically one per cell): the payload of packet j + 4 and the header
// of packet j + 8 are being loaded while packet j is consumed; every global load is unconditional (lanes without work
// address past the end of the buffer descriptor: no memory request), so the compiler counts what is in flight
// (s_waitcnt vmcnt(N), never 0), and the x-tile of the NEXT cell rides on the first packet of the current one.
// One barrier per cell.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"
#include "slp_tall.h"

namespace slp {


// ---- build, step 1: one 64-bit key per stored entry of the pass, in CSR order ------------------------------------------
// The copy is built in PASSES over ranges of its row blocks, so that the temporaries (keys, sorted keys, sort scratch) are a
// fraction of the matrix (SLP_TALL_PASS_NNZ entries per pass).  TR = false: the copy of A -- a pass is a range of rows of the
// CSR, its entries are contiguous.  TR = true: the copy of A^T taken STRAIGHT from the CSR of A (no transposed CSR is formed):
// row of the copy = column of A, column of the copy = row of A; a pass is a range [c0, c1) of A's columns, and because the
// rows of A are sorted by column its entries are one segment per row (k_tall_range), written at the scanned offsets `opos`.
//   key = cell ((row block - b0) * T + strip) << 37 | local row << 23 | column inside the strip << 11 | value id.
// TR = false: a stable sort by the cell bits leaves every cell's entries in (row, storage) order.  TR = true: the keys come in
// A's order (rows of A = columns of the copy, increasing), so a stable sort by (cell, local row) leaves the entries of a row
// of the copy in increasing column = the storage order of A^T as a stable transposition builds it (csc_matvec's order).
__global__ void k_tall_range(i64 nrow, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, i64 c0, i64 c1,
                             i64 *__restrict__ lo, i64 *__restrict__ len) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        const i64 s = ptr[r], e = ptr[r + 1];
        auto lower = [&](i64 c) {
            i64 a = s, b = e;
            while (a < b) {
                const i64 mid = (a + b) >> 1;
                if ((i64)idx[mid] < c) a = mid + 1;
                else b = mid;
            }
            return a;
        };
        const i64 l = lower(c0), h = lower(c1);
        lo[r] = l;
        len[r] = h - l;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) len[nrow] = 0;
}

// rows strictly increasing in column?  (the binary searches of k_tall_range rely on it)
__global__ __launch_bounds__(kBlock) void k_tall_sorted(i64 nnz, const i32 *__restrict__ idx, i64 nrow, const i64 *__restrict__ ptr,
                                                        int *__restrict__ bad) {
    const int lane = threadIdx.x & (kWave - 1);
    const i64 wave = ((i64)blockIdx.x * kBlock + threadIdx.x) / kWave, nwaves = (i64)gridDim.x * kBlock / kWave;
    for (i64 r = wave; r < nrow; r += nwaves) {
        const i64 s = ptr[r], e = ptr[r + 1];
        for (i64 k = s + 1 + lane; k < e; k += kWave)
            if (idx[k - 1] >= idx[k]) atomicOr(bad, 1);
    }
    (void)nnz;
}

template <bool TR>
__global__ __launch_bounds__(kBlock) void k_tall_keys(i64 r0, i64 nrow, int R, i64 T, int cshift, i64 b0, int D, const unsigned long long *__restrict__ dkeys,
                                                      const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                      const double *__restrict__ val, const i64 *__restrict__ lo,
                                                      const i64 *__restrict__ opos, i64 obase, unsigned long long *__restrict__ keys,
                                                      double *__restrict__ kvals, int *__restrict__ bad) {
    __shared__ unsigned long long skey[kTallDictMax];
    for (int q = threadIdx.x; q < D; q += kBlock) skey[q] = dkeys[q];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const i64 wave = ((i64)blockIdx.x * kBlock + threadIdx.x) / kWave, nwaves = (i64)gridDim.x * kBlock / kWave;
    for (i64 rr = wave; rr < nrow; rr += nwaves) {
        const i64 r = r0 + rr;                                   // row of the CSR
        const i64 s = TR ? lo[rr] : ptr[r];
        const i64 e = TR ? s + (opos[rr + 1] - opos[rr]) : ptr[r + 1];
        const i64 o = TR ? opos[rr] : s - obase;                 // where the row's keys go
        for (i64 k = s + lane; k < e; k += kWave) {
            const i32 j = idx[k];
            if (!TR && k > s && idx[k - 1] >= j) atomicOr(bad, 1);  // rows must be strictly increasing in column
            int id = 0;
            if (D > 0) {  // (D == 0: fp64 entries, the value travels beside the key)
                const unsigned long long key = tall_value_key((unsigned long long)__double_as_longlong(val[k]));
                int hi = D - 1;
                while (id < hi) {
                    const int mid = (id + hi) >> 1;
                    if (skey[mid] < key) id = mid + 1;
                    else hi = mid;
                }
                if (skey[id] != key) atomicOr(bad, 2);
            }
            const i64 trow = TR ? (i64)j : r, tcol = TR ? r : (i64)j;  // row / column of the copy
            const unsigned long long b = (unsigned long long)(trow / R - b0), rl = (unsigned long long)(trow % R);
            const unsigned long long t = (unsigned long long)(tcol >> cshift), cl = (unsigned long long)(tcol & (((i64)1 << cshift) - 1));
            keys[o + (k - s)] = ((b * (unsigned long long)T + t) << kTallCellShift) | (rl << (kTallIdBits + kTallColBits)) | (cl << kTallIdBits) |
                                (unsigned long long)id;
            if (kvals) kvals[o + (k - s)] = val[k];
        }
    }
}

// cellptr[c] = first sorted position whose cell is >= c (as k_ptr_from_sorted of slp_matrix.hip, 64-bit keys)
__global__ void k_tall_cellptr(i64 nnz, i64 ncell, const unsigned long long *__restrict__ key, i64 *__restrict__ cptr) {
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x) {
        const i64 c = (i64)(key[p] >> kTallCellShift), prev = p > 0 ? (i64)(key[p - 1] >> kTallCellShift) : -1;
        for (i64 j = prev + 1; j <= c; ++j) cptr[j] = p;
        if (p == nnz - 1)
            for (i64 j = c + 1; j <= ncell; ++j) cptr[j] = nnz;
    }
}

// ---- build, step 2: packets -------------------------------------------------------------------------------------------
// REGISTERS (round 6).  Everything derived from the thread index -- LDS addresses of ten arrays, the masks of ~80 comparisons
// `i < p / 32`, `i < wave` of the unrolled scans -- is invariant over the cells a workgroup takes; the compiler computed all of it
// once in front of the cell loop and then could not keep it: 42 vector registers spilled to scratch (164 bytes per lane, stored in
// the prologue, re-read in every cell: 160 KB per cell against the cell's 32 KB of keys -- more than the L2 share of a workgroup)
// and 130 scalar ones.  SLP_OPAQUE(v) is an empty asm statement that claims to rewrite v: what is derived from v
// after it cannot be hoisted in front of it.  The thread index is made opaque once per cell, the row of the 32 x 32 scans once per
// scan, and the scans are unrolled by 8 (not 32: 32 64-bit LDS reads in flight were 64 registers by themselves): 125 / 128 vector
// registers, NO scratch, 42 scalar spills left (kernel arguments parked in lanes of one vector register -- no memory behind them).
#define SLP_OPAQUE(v) asm volatile("" : "+v"(v))
// Sums along the lanes of a wave by DPP moves (vector ALU; `__shfl_up` is a ds_bpermute: the LDS pipe and its latency, six times
// per scan).  dpp<CTRL, ROWS>(x): x of the lane CTRL names, 0 where there is none; lanes of the 16-lane rows not in ROWS get 0.
// row_shr:n = 0x110 + n (n lanes to the left inside the row), row_bcast:15 = 0x142 (lane 15 of a row to every lane of the next row),
// row_bcast:31 = 0x143 (lane 31 to the upper half).
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ unsigned int tall_dpp(unsigned int x) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xf, true);
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ unsigned long long tall_dpp(unsigned long long x) {
    return (unsigned long long)tall_dpp<CTRL, ROWS>((unsigned int)(x >> 32)) << 32 | tall_dpp<CTRL, ROWS>((unsigned int)x);
}
// inclusive sums over the lanes of a wave (SEG = 64) or of each half of it (SEG = 32)
template <int SEG, typename U>
__device__ __forceinline__ U tall_lane_sums(U x) {
    x += tall_dpp<0x111>(x);
    x += tall_dpp<0x112>(x);
    x += tall_dpp<0x114>(x);
    x += tall_dpp<0x118>(x);
    x += tall_dpp<0x142, 0xa>(x);
    if (SEG == 64) x += tall_dpp<0x143, 0xc>(x);
    return x;
}
// block-wide exclusive scan of one word per thread; total in *tot
__device__ __forceinline__ unsigned long long tall_block_scan(unsigned long long v, unsigned long long *wtot, unsigned long long *tot) {
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    SLP_OPAQUE(w);  // (the sixteen `i < w` below are compared here, not once per kernel and kept in 32 scalar registers)
    const unsigned long long inc = tall_lane_sums<kWave>(v);
    if (lane == kWave - 1) wtot[w] = inc;
    __syncthreads();
    unsigned long long base = 0, all = 0;
    for (int i = 0; i < kTallT / kWave; ++i) {
        const unsigned long long t = wtot[i];
        if (i < w) base += t;
        all += t;
    }
    __syncthreads();
    *tot = all;
    return base + inc - v;
}

// ONE pass over the cells, any cell by any workgroup.  Where a cell's packets go depends on the sizes of all cells before it, and
// a cell's size is only known once its rows have been dealt -- most of the work.  Round 1-4 ran the kernel twice (sizes, host
// scan, fill), one workgroup per ROW BLOCK (a chunk's copy of A has 128 of them: half the chip).  Now the cells are handed out
// by a ticket counter in cell order, a workgroup deals its cell, PUBLISHES the cell's sizes and sums the sizes of the cells
// before it by a decoupled look-back (as a single-pass device scan does: every cell publishes first its own sizes, then the
// sums up to and including itself; a later cell adds own sizes backwards until it meets such sums -- tickets are taken in order
// by running workgroups, so whatever a cell waits for is being worked on), and writes its packets behind them: the payload comes
// out compact in cell order = stream order, into a buffer sized by an estimate (`cap_w` words, `cap_p` packets; a cell that would
// not fit writes nothing and the host repeats the pass with the exact totals).  Packet headers are written with offsets relative
// to the cell; k_tall_dir moves them into the streams' directories.
// The scan's state is two 64-bit words per cell, each stamped with its own level in bits 62-63 (0 nothing yet, 1 the cell's own
// sizes, 2 the sums up to and including the cell) and written / polled by relaxed agent-scope atomics: nothing else has to be
// ordered against them, so no release / acquire -- on this chip those are write-backs and invalidations of a whole L2, twice
// per cell and once per poll (measured: 132 ms per 2.5e9 entries with them).  The writer stamps `w` first, then `p`; a reader takes `w`,
// then `p`, and retries until both carry the same level.
#define SLP_TB_PROF(k) do {} while (0)
struct TallScan {
    unsigned long long *w;       // level << 62 | payload words (own, or up to and including the cell)
    unsigned long long *p;       // level << 62 | packets
    unsigned long long *before_w, *before_p;  // (plain, read after the pass) payload words / packets of the cells before this one
    unsigned long long *ticket;  // next cell to hand out
};
constexpr unsigned long long kScanMask = (1ull << 62) - 1;
__device__ __forceinline__ void tall_scan_put(const TallScan &sc, i64 c, unsigned long long level, unsigned long long w, unsigned long long k) {
    __hip_atomic_store(&sc.w[c], (level << 62) | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sc.p[c], (level << 62) | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int tall_scan_get(const TallScan &sc, i64 c, unsigned long long *w, unsigned long long *k) {
    for (;;) {   // both words in flight together (each level of a cell is written once: equal levels = one writer's pair)
        const unsigned long long a = __hip_atomic_load(&sc.w[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(&sc.p[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(a >> 62) || (a >> 62) != (b >> 62)) continue;
        *w = a & kScanMask;
        *k = b & kScanMask;
        return (unsigned int)(a >> 62);
    }
}

__device__ __forceinline__ i64 tall_range_of(i64 t, i64 T, int S) {  // the strip range (of S) that holds strip t: T s / S <= t < T (s + 1) / S
    i64 s = t * S / T;
    while (T * (s + 1) / S <= t) ++s;
    while (T * s / S > t) --s;
    return s;
}

// DICT: 5-byte items (value id inside); !DICT: 4-byte items (column | row << 12) + the fp64 value in a parallel array at the same offsets.
template <bool DICT>
__global__ __launch_bounds__(kTallT) void k_tall_build(int R, i64 T, int S, int cshift, i64 ncell, const unsigned long long *__restrict__ keys,
                                                       const double *__restrict__ svals, const i64 *__restrict__ cellptr, TallScan sc,
                                                       i64 cap_w, i64 cap_p, TallPkt *__restrict__ sdir,
                                                       unsigned int *__restrict__ spay, double *__restrict__ spayv) {
    __shared__ unsigned int cnt[kTallRmax];       // entries of the row inside the cell
    __shared__ unsigned int rstart[kTallRmax];    // position of the row's first entry inside the cell
    __shared__ unsigned short posrow[kTallRmax];  // sorted position -> local row
    __shared__ unsigned int nlane[kTallT];        // list length of every lane (non-increasing)
    __shared__ unsigned long long wtot[kTallT / kWave];
    __shared__ unsigned int width[kTallSlots];
    __shared__ unsigned int bcnt[kTallBuckets * 32], bstart[kTallBuckets * 32 + 1];  // (count, bank class) buckets
    __shared__ unsigned long long cbuf[2 * 32 * 33], ctot[2 * 32];  // scans over the 32 threads of a bank class (two words per thread)
    __shared__ unsigned int ccls[32 * 33], call[32];   // demands of the 32 lanes of every bank class
    // one-entry rows of a bank class that no lane of the class takes, as sums: spare1x[r] = those of the classes before r, [32] = all
    __shared__ unsigned int spare1x[33];
    // rows of two or more entries, per count class k (counts 6 .. 2) and bank class r: lanes of the class without a row of their own
    // in the classes before r (lackx), rows without a lane in the classes before r (sparex; [33 k + 32] = all)
    __shared__ unsigned int lackx[(kTallBuckets - 1) * 32], sparex[(kTallBuckets - 1) * 33];
    __shared__ i64 s_cell, s_c0, s_tn;   // the workgroup's next cell, where its keys start, the strip after it (take_next)
    __shared__ int s_n;
    __shared__ unsigned int s_dump[kTallT];   // where the touches of the next cell's keys land (never read)
    __shared__ unsigned long long s_before_w, s_before_p;
    auto tile_of = [&](i64 t) -> unsigned int { return t >= 0 ? (unsigned int)(t << cshift) : kNoTile; };
    // The workgroup's NEXT cell, by one thread: the ticket (a cell without entries takes part in the scan with sizes of zero and
    // the next ticket is drawn), where the cell's keys lie, and the next strip of its stream with entries in the row block (its
    // x-tile rides on this cell's first packet).  cell = (row block b, strip t); the stream it belongs to: strip range sr of S
    // (S > 1: few, tall row blocks whose strips are shared by S workgroups of the product kernel -- see tall_geometry).
    // Rounds 3-6a: drawn by thread 0 at the top of a cell with 1023 threads waiting, then every thread read cellptr[] itself
    // (3 dependent round trips to L2 in front of a cell's first key: 2 of its 19 us).  Now thread 64 draws it while wave 0 sums the
    // sizes of the cells before the current one, and the workgroup touches the next cell's keys (one dword per 128 bytes) while it
    // stores the current cell's packets: the first real read of a key finds it in L2.
    // (The ticket itself is drawn two barriers earlier -- `drawn` -- so that its round trip is over when take_next runs; the
    // three cellptr[] entries a cell usually needs are fetched together.)
    auto take_next = [&](unsigned long long drawn) {
        i64 c = (i64)drawn, lo = 0, hi = 0, hi2 = 0;
        for (;;) {
            if (c >= ncell) break;
            lo = cellptr[c];
            hi = cellptr[c + 1];
            hi2 = cellptr[c + 1 < ncell ? c + 2 : c + 1];
            if (hi > lo) break;
            tall_scan_put(sc, c, 1ull, 0ull, 0ull);
            c = (i64)atomicAdd(sc.ticket, 1ull);
        }
        s_cell = c;
        if (c < ncell) {
            const i64 b = c / T, t = c % T;
            const i64 t_end = S > 1 ? T * (tall_range_of(t, T, S) + 1) / S : T;
            i64 tn = -1;
            if (t + 1 < t_end && hi2 > hi) tn = t + 1;   // (cell c + 1 is strip t + 1 of the same row block)
            else
                for (i64 u = t + 2; u < t_end; ++u)
                    if (cellptr[b * T + u + 1] > cellptr[b * T + u]) { tn = u; break; }
            s_tn = tn;
            s_c0 = lo;
            s_n = (int)(hi - lo);
        }
    };
    static_assert(kTallT >= 2 * kWave, "take_next runs on the first lane of the second wave, beside wave 0's look-back");
    if (threadIdx.x == kWave) take_next(atomicAdd(sc.ticket, 1ull));
    for (;;) {
        int p = threadIdx.x;
        SLP_OPAQUE(p);   // per cell: see REGISTERS above
        __syncthreads();
        SLP_TB_PROF(0);
        const i64 cell = s_cell;
        if (cell >= ncell) break;
        const i64 t = cell % T, tn = s_tn, c0 = s_c0;
        const int n = s_n;
        unsigned long long drawn = 0;
        for (int r = p; r < R; r += kTallT) cnt[r] = 0;
        __syncthreads();
        SLP_TB_PROF(1);
        for (int i0 = p; i0 < n; i0 += 4 * kTallT) {   // four rounds' keys in flight together (index clamped, use predicated)
            unsigned long long ka[4], kb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * kTallT < n ? i0 + j * kTallT : n - 1;
                ka[j] = keys[c0 + i];
                kb[j] = keys[c0 + (i > 0 ? i - 1 : 0)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * kTallT;
                if (i < n) {
                    const unsigned int r = (unsigned int)(ka[j] >> (kTallIdBits + kTallColBits)) & ((1u << kTallRowBits) - 1);
                    const unsigned int rp = i > 0 ? (unsigned int)(kb[j] >> (kTallIdBits + kTallColBits)) & ((1u << kTallRowBits) - 1) : ~0u;
                    if (r != rp) rstart[r] = (unsigned int)i;
                    atomicAdd(&cnt[r], 1u);
                }
            }
        }
        __syncthreads();
        SLP_TB_PROF(2);
        // Sorted positions: rows with entries, by min(count, 6) descending and, inside a count, by their BANK CLASS
        // rho = row % 32 (the LDS bank pair of the row's running sum in the product kernel): 192 buckets.
        // Thread (i, rho) = p walks the rows rho + 32 m of its bank class for a range of m; an exclusive scan over the 32
        // threads of the class (through LDS) gives every row its place inside its (count, class) bucket -- rows in increasing
        // order: the same layout in the sizing pass and in the writing pass.
        const int crows = R > (int)(p & 31) ? (R - (int)(p & 31) + 31) / 32 : 0;  // rows of the class: rho + 32 m, m < crows
        const int mpl = ((R + 31) / 32 + 31) / 32;                                // consecutive m per thread
        const int m0 = (p >> 5) * mpl < crows ? (p >> 5) * mpl : crows, m1 = m0 + mpl < crows ? m0 + mpl : crows;
        // both words of a thread in one scan (v[0], v[1] -> exclusive sums; tot[0], tot[1] the class's totals)
        // Through LDS TRANSPOSED (pairs at (class, thread of the class), rows of 33 pairs): thread p picks up the pair of thread
        // p % 32 of class p / 32, so the 32 threads of a class sit in the 32 lanes of half a wave and the scan is lane sums; the
        // exclusive sums go back the same way.  (Rounds 3-6a: every thread read all 32 pairs of its class -- 512 KB through the LDS
        // pipe per scan, 1.7 of a cell's 24 us.)
        auto class_scan2 = [&](unsigned long long *v, unsigned long long *tot) {
            const int mine = 2 * ((p & 31) * 33 + (p >> 5)), theirs = 2 * ((p >> 5) * 33 + (p & 31));
            cbuf[mine] = v[0];
            cbuf[mine + 1] = v[1];
            __syncthreads();
            const unsigned long long u0 = cbuf[theirs], u1 = cbuf[theirs + 1];
            const unsigned long long s0 = tall_lane_sums<32>(u0), s1 = tall_lane_sums<32>(u1);
            cbuf[theirs] = s0 - u0;
            cbuf[theirs + 1] = s1 - u1;
            if ((p & 31) == 31) { ctot[2 * (p >> 5)] = s0; ctot[2 * (p >> 5) + 1] = s1; }
            __syncthreads();
            v[0] = cbuf[mine];
            v[1] = cbuf[mine + 1];
            tot[0] = ctot[2 * (p & 31)];
            tot[1] = ctot[2 * (p & 31) + 1];
        };
        unsigned long long ha = 0, hb = 0;  // counters of the counts 6,5,4 (21 bits each) / 3,2,1
        for (int m = m0; m < m1; ++m) {
            const unsigned int c = cnt[(p & 31) + 32 * m];
            if (c) {
                const unsigned int cl = c < (unsigned)kTallBuckets ? c : kTallBuckets;
                if (cl > 3) ha += 1ull << (21 * (kTallBuckets - cl));
                else hb += 1ull << (21 * (3 - cl));
            }
        }
        unsigned long long hv[2] = {ha, hb}, tv[2];
        class_scan2(hv, tv);
        unsigned long long ea = hv[0], eb = hv[1];
        const unsigned long long ta = tv[0], tb = tv[1];
        if (p < 32)
            for (int cl = kTallBuckets; cl >= 1; --cl)
                bcnt[(kTallBuckets - cl) * 32 + p] = (unsigned int)(((cl > 3 ? ta : tb) >> (21 * ((cl > 3 ? kTallBuckets : 3) - cl))) & 0x1fffffu);
        __syncthreads();
        SLP_TB_PROF(3);
        if (p < kWave) {   // exclusive scan of the 192 bucket counts by one wave: three per lane (one thread walking them was
            const unsigned int c0 = bcnt[3 * p], c1 = bcnt[3 * p + 1], c2 = bcnt[3 * p + 2];  // 8 of the ~30 us a cell takes)
            unsigned int inc = c0 + c1 + c2;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const unsigned int o = __shfl_up(inc, off, kWave);
                if (p >= off) inc += o;
            }
            const unsigned int ex = inc - (c0 + c1 + c2);
            bstart[3 * p] = ex;
            bstart[3 * p + 1] = ex + c0;
            bstart[3 * p + 2] = ex + c0 + c1;
            if (p == kWave - 1) bstart[kTallBuckets * 32] = inc;
        }
        __syncthreads();
        SLP_TB_PROF(4);
        // Lanes [S, E) of a count class take its rows: a lane one of its own bank class as far as those last.  Which rows the lanes
        // without one get is settled by two sums over the bank classes (lane sums inside the 32 lanes of a count class) instead of
        // every such lane walking the 32 classes twice (64 dependent LDS reads a wave: 4 of the 31 us a cell took).
        if (p < (kTallBuckets - 1) * 32) {
            const unsigned int k = (unsigned)p >> 5, r = (unsigned)p & 31u;
            const unsigned int S = bstart[k * 32], E = bstart[k * 32 + 32];
            const unsigned int first = S + ((r + 32u - (S & 31u)) & 31u), nl = first < E ? (E - first + 31u) / 32u : 0u;
            const unsigned int rr = bstart[k * 32 + r + 1] - bstart[k * 32 + r];
            const unsigned int lack = nl > rr ? nl - rr : 0u, spare = rr > nl ? rr - nl : 0u;
            const unsigned int il = tall_lane_sums<32>(lack), is = tall_lane_sums<32>(spare);
            lackx[k * 32 + r] = il - lack;
            sparex[k * 33 + r] = is - spare;
            if (r == 31u) sparex[k * 33 + 32] = is;
        }
        for (int m = m0; m < m1; ++m) {
            const unsigned int r = (unsigned int)((p & 31) + 32 * m), c = cnt[r];
            if (c) {
                const unsigned int cl = c < (unsigned)kTallBuckets ? c : kTallBuckets;
                unsigned int within;
                if (cl > 3) { within = (unsigned int)((ea >> (21 * (kTallBuckets - cl))) & 0x1fffffu); ea += 1ull << (21 * (kTallBuckets - cl)); }
                else { within = (unsigned int)((eb >> (21 * (3 - cl))) & 0x1fffffu); eb += 1ull << (21 * (3 - cl)); }
                posrow[bstart[(kTallBuckets - cl) * 32 + (p & 31)] + within] = (unsigned short)r;
            }
        }
        __syncthreads();
        SLP_TB_PROF(5);
        const unsigned int npos = bstart[kTallBuckets * 32];
        const unsigned int n2 = bstart[(kTallBuckets - 1) * 32];  // rows with two or more entries come first
        // Dealing rows to lanes.  Sparse cells (all rows with two or more entries fit the 1024 lanes -- the regime this
        // format is for): lane p < n2 takes one of those rows, then every lane takes as many one-entry rows as bring its
        // list to its target length (items / 1024, one more for the first items % 1024 lanes): all lists are that long except those
        // of the few rows longer than it, so the 16 waves of the workgroup reach the cell's barrier together.  WHICH row of a count a lane takes is
        // free: lanes take rows of their own bank class (p % 32 == row % 32) as far as those last -- the 32 lanes of a
        // half-wave then read and write the running sums without bank conflicts -- and the rest in order.
        // Otherwise (dense cells) round by round: lane p takes positions p, p + 1024, ...
        const bool fill = n2 <= (unsigned)kTallT;  // uniform
        const unsigned int rho = (unsigned)p & 31u;
        unsigned int pos0 = npos, c0p = 0;  // the lane's row of two or more entries (fill mode), or its first position
        if (fill) {
            if ((unsigned)p < n2) {
                int cl = kTallBuckets;  // the count class whose lanes [S, E) hold p
                while (cl > 2 && (unsigned)p >= bstart[(kTallBuckets - cl + 1) * 32]) --cl;
                const unsigned int *bs = bstart + (kTallBuckets - cl) * 32;  // bs[r] .. bs[r + 1]: rows of bank class r
                const unsigned int S = bs[0], E = bs[32];
                auto lanes_of = [&](unsigned int r) -> unsigned int {  // lanes of [S, E) in bank class r
                    const unsigned int first = S + ((r + 32u - (S & 31u)) & 31u);
                    return first < E ? (E - first + 31u) / 32u : 0u;
                };
                const unsigned int first = S + ((rho + 32u - (S & 31u)) & 31u), i = ((unsigned)p - first) / 32u;
                const unsigned int nr = bs[rho + 1] - bs[rho];
                if (i < nr) {
                    pos0 = bs[rho] + i;  // a row of the lane's own bank class
                } else {                 // the u-th lane without one takes the u-th row without a lane
                    const unsigned int k = (unsigned)(kTallBuckets - cl);
                    const unsigned int u = i - nr + lackx[k * 32 + rho];
                    const unsigned int *sx = sparex + k * 33;
                    if (u < sx[32]) {
                        unsigned int lo = 0, hi = 32;   // the class r with sx[r] <= u < sx[r + 1]
#pragma unroll
                        for (int it = 0; it < 5; ++it) { const unsigned int mid = (lo + hi) >> 1; if (sx[mid] <= u) lo = mid; else hi = mid; }
                        pos0 = bs[lo] + lanes_of(lo) + (u - sx[lo]);
                    }
                }
                c0p = cnt[posrow[pos0]];
            }
        } else {
            pos0 = (unsigned)p < npos ? (unsigned)p : npos;
            c0p = pos0 < npos ? cnt[posrow[pos0]] : 0u;
        }
        unsigned int mine = c0p, own0 = 0, nown = 0, left0 = 0, nleft = 0;
        {
            // one-entry rows (fill mode): first those of the lane's own bank class, in lane order inside the class ...
            const unsigned int *b1 = bstart + (kTallBuckets - 1) * 32;
            // The lane's target length: items / 1024, one more for the first items % 1024 lanes -- the targets add up to the
            // cell's items, so the one-entry rows go round (up to the rows longer than their lane's target).  With ceil(items /
            // 1024) for every lane (rounds 3-5a, -DSLP_TALL_DEAL_CEIL) the demand exceeded the supply by 1024 - items % 1024: the bank
            // classes ran dry at different lanes, the lists of the last ~150 lanes came out between 0 and the target at random, and
            // the non-increasing envelope over them was paid in skip items: 2 % of the words at the metric's density (lists of 4
            // from 3983 items), 20 % where lists are 5 long.
            const unsigned int tau = (unsigned)n / kTallT + ((unsigned)p < (unsigned)n % kTallT ? 1u : 0u);
            const unsigned int want = (fill && c0p < tau) ? tau - c0p : 0u;
            ccls[rho * 33 + (p >> 5)] = want;   // (transposed as in class_scan2: the class's 32 demands in the lanes of half a wave)
            __syncthreads();
            {
                const int theirs = (p >> 5) * 33 + (int)rho;
                const unsigned int u = ccls[theirs], su = tall_lane_sums<32>(u);
                ccls[theirs] = su - u;
                if (rho == 31u) call[p >> 5] = su;
            }
            __syncthreads();
            const unsigned int before = ccls[rho * 33 + (p >> 5)], all = call[rho];
            const unsigned int n1 = b1[rho + 1] - b1[rho];
            own0 = b1[rho] + before;
            nown = before >= n1 ? 0u : (want < n1 - before ? want : n1 - before);
            if (p < 32) {   // rows of the class nobody of the class takes, summed over the classes before it
                const unsigned int sp = n1 > all ? n1 - all : 0u;
                const unsigned int is = tall_lane_sums<32>(sp);
                spare1x[p] = is - sp;
                if (p == 31) spare1x[32] = is;
            }
            __syncthreads();
            SLP_TB_PROF(6);
            if (p == kWave) drawn = atomicAdd(sc.ticket, 1ull);   // the workgroup's next ticket: used by take_next below
            // ... then, for what is still missing, the rows left over in other classes: in lane order
            unsigned long long total;
            left0 = (unsigned int)tall_block_scan(want - nown, wtot, &total);
            const unsigned int nspare = spare1x[32];
            nleft = left0 >= nspare ? 0u : (want - nown < nspare - left0 ? want - nown : nspare - left0);
            if (fill) mine += nown + nleft;
            else for (unsigned int pos = p + kTallT; pos < npos; pos += kTallT) mine += cnt[posrow[pos]];
        }
        // sorted position of the lane's q-th row, or npos when its list has ended
        auto rowpos = [&](unsigned int q) -> unsigned int {
            if (!fill) { const unsigned int pos = q * kTallT + p; return pos < npos ? pos : npos; }
            if (pos0 < npos) { if (q == 0) return pos0; --q; }
            if (q < nown) return own0 + q;
            q -= nown;
            if (q >= nleft) return npos;
            const unsigned int *b1 = bstart + (kTallBuckets - 1) * 32;
            const unsigned int u = left0 + q;   // (< spare1x[32]: q < nleft)
            unsigned int lo = 0, hi = 32;       // the class r with spare1x[r] <= u < spare1x[r + 1]
#pragma unroll
            for (int it = 0; it < 5; ++it) { const unsigned int mid = (lo + hi) >> 1; if (spare1x[mid] <= u) lo = mid; else hi = mid; }
            return b1[lo + 1] - (spare1x[lo + 1] - spare1x[lo]) + (u - spare1x[lo]);
        };
        // List lengths are non-increasing in p up to a few exceptions (rows with >= 6 entries in bucket order; the lanes
        // that find no one-entry row left).  Slot widths therefore come from the non-increasing envelope cover[p] = max over
        // lanes >= p of their list lengths: slot k is as wide as the lanes whose ENVELOPE exceeds k, and a lane inside it
        // whose own list has ended stores a skip item there.
        unsigned int cover = mine;
        {
            // suffix maximum over the 1024 lanes: inside the wave by shuffles, across waves through LDS
            const int lane = p & (kWave - 1), w = p / kWave;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const unsigned int o = __shfl_down(cover, off, kWave);
                if (lane + off < kWave) cover = o > cover ? o : cover;
            }
            if (lane == 0) wtot[w] = cover;
            __syncthreads();
            unsigned int right = 0;
            for (int i = w + 1; i < kTallT / kWave; ++i) right = (unsigned int)wtot[i] > right ? (unsigned int)wtot[i] : right;
            cover = right > cover ? right : cover;
            __syncthreads();
        }
        nlane[p] = cover;
        // The cell's sizes from the envelope alone: slot k is as wide as the lanes whose envelope exceeds k, so the slots of all
        // packets hold sum_p cover[p] words; the fifth bytes take a word per lane of slot 8 g and of slot 8 g + 4 of every packet g.
        // A cell's payload is rounded up to 16 bytes (every stream then starts 16-byte aligned).  (The waves' sums cross the same
        // barrier as the envelope itself.)
        unsigned long long cw = 0;
        {
            unsigned long long w = cover;
            if (DICT) w += (cover + 7u) / 8u + (cover > 4u ? (cover - 4u + 7u) / 8u : 0u);
            w = tall_lane_sums<kWave>(w);
            if ((p & (kWave - 1)) == kWave - 1) wtot[p / kWave] = w;
        }
        __syncthreads();
        SLP_TB_PROF(7);
        const unsigned int longest = nlane[0];
        const unsigned int nxt = p + 1 < kTallT ? nlane[p + 1] : 0u;   // the envelope of the lane to the right
        for (int i = 0; i < kTallT / kWave; ++i) cw += wtot[i];
        const unsigned long long cell_w = (cw + 3ull) & ~3ull, cell_p = longest ? (longest + kTallSlots - 1) / kTallSlots : 1u;
        if (p == 0) tall_scan_put(sc, cell, 1ull, cell_w, cell_p);  // the cell's sizes stand: later cells need not wait for more
        bool fits = true;
        unsigned int *pay = nullptr;
        double *payv = nullptr;
        TallPkt *pk = nullptr;
        unsigned int woff = 0, npk = 0;  // payload offset inside the cell / packets of the cell so far (uniform)
        // the lane's cursor over its own list: its q-th row (sorted position mypos), entry s of that row
        unsigned int q = 0, s = 0, mypos = rowpos(0);
        unsigned int myrow = mypos < npos ? posrow[mypos] : 0u, mycnt = mypos < npos ? cnt[myrow] : 0u;
        for (unsigned int g = 0; g * kTallSlots < longest || g == 0; ++g) {
            // slot widths of this packet: width[j] = first lane whose envelope is <= 8 g + j = the lanes whose envelope exceeds it
            // (non-increasing).  Written by the lane at the step: the last one above 8 g + j -- or by lane 0 when there is none.
            // (Rounds 3-5: a binary search over nlane[] by lanes 0..7, ten dependent LDS reads with 1016 lanes waiting.)
#pragma unroll
            for (int j = 0; j < kTallSlots; ++j) {
                const unsigned int k = g * kTallSlots + j;
                if (nxt <= k && k < cover) width[j] = (unsigned int)p + 1u;
                if (p == 0 && cover <= k) width[j] = 0u;
            }
            __syncthreads();
            SLP_TB_PROF(8);
            unsigned int wd[kTallSlots], words = 0;
            for (int j = 0; j < kTallSlots; ++j) { wd[j] = width[j]; words += wd[j]; }
            const unsigned int hi0 = words, hi1 = words + wd[0];
            if (DICT) words += wd[0] + wd[4];
            // The lane's items of this packet in three steps, so that their keys are fetched TOGETHER (one after the other, each
            // waited for before the cursor moved on, they were 5-8 dependent L2 round trips per cell): where the keys lie (the
            // cursor's walk: LDS only), the loads, the items.  A lane inside a slot whose list has ended stores a skip item: row
            // field 0 = the product kernel's scratch cell, like the zero a load past the buffer descriptor returns for a lane
            // outside the slot.
            unsigned int kofs[kTallSlots], have = 0;
#pragma unroll
            for (int j = 0; j < kTallSlots; ++j) {
                kofs[j] = 0;
                if ((unsigned)p < wd[j] && mypos < npos) {
                    kofs[j] = rstart[myrow] + s;
                    have |= 1u << j;
                    if (++s == mycnt) {
                        ++q; s = 0;
                        mypos = rowpos(q);
                        myrow = mypos < npos ? posrow[mypos] : 0u;
                        mycnt = mypos < npos ? cnt[myrow] : 0u;
                    }
                }
            }
            unsigned long long key[kTallSlots];
            double value[kTallSlots];
#pragma unroll
            for (int j = 0; j < kTallSlots; ++j) {
                key[j] = (have >> j & 1u) ? keys[c0 + kofs[j]] : 0ull;
                value[j] = (!DICT && (have >> j & 1u)) ? svals[c0 + kofs[j]] : 0.0;
            }
            if (g == 0) {
                // Where the cell's packets go: the sizes of all cells before it, summed backwards by one wave from the cells' own
                // sizes until a cell with its sums is met (behind the loads above: their latency and the look-back's overlap).
                if (p == kWave) take_next(drawn);   // (beside wave 0's look-back: after the barrier below everybody has the next cell)
                if (p < kWave) {
                    unsigned long long bw = 0, bp = 0;
                    for (i64 j = cell - 1; j >= 0; j -= kWave) {
                        const i64 qc = j - p;   // lane p looks at the p-th cell back
                        unsigned int st = 2u;   // (before the first cell: sums of zero)
                        unsigned long long w = 0, k = 0;
                        if (qc >= 0) st = tall_scan_get(sc, qc, &w, &k);
                        const unsigned long long summed = __ballot(st == 2u);
                        const int f = summed ? __ffsll((long long)summed) - 1 : kWave;  // the nearest cell that has its sums
                        if (p > f) { w = 0; k = 0; }
#pragma unroll
                        for (int off = kWave / 2; off; off >>= 1) { w += __shfl_xor(w, off, kWave); k += __shfl_xor(k, off, kWave); }
                        bw += w; bp += k;
                        if (summed) break;
                    }
                    if (p == 0) {
                        tall_scan_put(sc, cell, 2ull, bw + cell_w, bp + cell_p);
                        sc.before_w[cell] = bw;
                        sc.before_p[cell] = bp;
                        s_before_w = bw;
                        s_before_p = bp;
                    }
                }
                __syncthreads();
                SLP_TB_PROF(9);
                // touch the next cell's keys: one dword per 128 bytes, loaded straight into an LDS area nobody reads (no register
                // waits for it; as a load into a register it had to be waited for at the top of the next cell -- together with
                // every packet store before it, which nothing waits for otherwise)
                if (s_cell < ncell && p * 16 < s_n) {
                    const unsigned int *g = reinterpret_cast<const unsigned int *>(keys + s_c0) + p * 32;
                    const unsigned int lds = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)(s_dump + (p & ~(kWave - 1))));
                    unsigned int m0_saved;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(m0_saved) : "s"(lds), "v"(g) : "memory");
                }
                fits = (i64)(s_before_w + cell_w) <= cap_w && (i64)(s_before_p + cell_p) <= cap_p;  // uniform
                pay = spay + s_before_w;
                payv = DICT ? nullptr : spayv + s_before_w;
                pk = sdir + s_before_p;
            }
            if (!fits) break;   // (the host repeats the pass with the exact sizes; the scan itself is complete)
            {
                unsigned int *base = pay + woff;
                unsigned int hb0 = 0, hb1 = 0, so = 0;
#pragma unroll
                for (int j = 0; j < kTallSlots; ++j) {
                    if ((unsigned)p < wd[j]) {
                        unsigned int item = 0u, hib = 0u;
                        if (have >> j & 1u) {
                            const unsigned int r1 = ((unsigned int)(key[j] >> (kTallIdBits + kTallColBits)) & ((1u << kTallRowBits) - 1)) + 1u;  // local row + 1
                            if (DICT) {
                                item = ((unsigned int)key[j] & ((1u << (kTallIdBits + kTallColBits)) - 1)) | ((r1 & 0x1ffu) << 23);
                                hib = r1 >> 9;
                            } else {
                                item = ((unsigned int)(key[j] >> kTallIdBits) & (kTallC - 1)) | (r1 << kTallColBits);
                            }
                        }
                        base[so + p] = item;
                        if (!DICT) payv[woff + so + p] = value[j];
                        if (j < 4) hb0 |= hib << (8 * j);
                        else hb1 |= hib << (8 * (j - 4));
                    }
                    so += wd[j];
                }
                if (DICT && (unsigned)p < wd[0]) base[hi0 + p] = hb0;
                if (DICT && (unsigned)p < wd[4]) base[hi1 + p] = hb1;
                if (p == 0) {
                    TallPkt h;
                    h.off = woff;   // (inside the cell: k_tall_dir adds where the cell lies in its stream)
                    h.xsrc = (g == 0) ? (kPktNewCell | tile_of(tn)) : kNoTile;
                    for (int i = 0; i < 4; ++i) h.c[i] = wd[2 * i] | (wd[2 * i + 1] << 16);
                    h.strip = (unsigned int)t;
                    h.pad = 0;
                    pk[npk] = h;
                }
            }
            woff += words;
            ++npk;
            __syncthreads();  // width[] is rewritten by the next packet
            SLP_TB_PROF(10);
        }
    }
}

// What the single pass is given room for: per cell of n entries its n items and the fifth bytes' words of min(n, 1024) lists of
// ceil(n / 1024) items (what the dealing aims at; rows longer than that and the envelope's skip items -- 2-3 % on the benchmark's
// LPs -- come out of the margin the host adds).
template <bool DICT>
__global__ __launch_bounds__(kBlock) void k_tall_estimate(i64 ncell, const i64 *__restrict__ cellptr, unsigned long long *__restrict__ est) {
    unsigned long long w = 0, k = 0;
    for (i64 c = (i64)blockIdx.x * kBlock + threadIdx.x; c < ncell; c += (i64)gridDim.x * kBlock) {
        const unsigned long long n = (unsigned long long)(cellptr[c + 1] - cellptr[c]);
        if (!n) continue;
        const unsigned long long L = n < (unsigned)kTallT ? n : (unsigned)kTallT, tau = (n + kTallT - 1) / kTallT;
        unsigned long long cw = n;
        if (DICT) cw += L * ((tau + 7) / 8 + (tau > 4 ? (tau - 4 + 7) / 8 : 0));
        w += (cw + 3ull) & ~3ull;
        k += (tau + kTallSlots - 1) / kTallSlots;
    }
#pragma unroll
    for (int off = kWave / 2; off; off >>= 1) { w += __shfl_xor(w, off, kWave); k += __shfl_xor(k, off, kWave); }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicAdd(&est[0], w); atomicAdd(&est[1], k); }
}

// The streams' extents from the scan: stream v = (row block, strip range) = the cells [cs, ce): payload words before it
// (its base: the payload is compact in cell order), its words, packets before it, its packets.  Only cells with entries carry
// sums: the sums in front of a cell are those of the nearest such cell before it (all streams together walk every cell at most
// twice).
__global__ void k_tall_streams(i64 V, i64 T, int S, const i64 *__restrict__ cellptr, TallScan sc, i64 *__restrict__ ext) {
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (i64)gridDim.x * blockDim.x) {
        const i64 b = v / S, cs = b * T + T * (v % S) / S, ce = b * T + T * (v % S + 1) / S;
        auto before = [&](i64 c, i64 *w, i64 *k) {  // payload words / packets of the cells [0, c)
            for (--c; c >= 0 && cellptr[c + 1] == cellptr[c];) --c;
            *w = c >= 0 ? (i64)(sc.w[c] & kScanMask) : 0;
            *k = c >= 0 ? (i64)(sc.p[c] & kScanMask) : 0;
        };
        i64 w0, p0, w1, p1;
        before(cs, &w0, &p0);
        before(ce, &w1, &p1);
        ext[4 * v] = w0;
        ext[4 * v + 1] = w1 - w0;
        ext[4 * v + 2] = p0;
        ext[4 * v + 3] = p1 - p0;
    }
}

// The streams' packet directories: [leading packet without items: the x-tile of the stream's first cell] [the cells' packets,
// payload offsets now relative to the stream] [packets without items up to a whole group of 2 x depth, then 2 x depth more that are
// only ever prefetched].  A thread per cell, then a thread per stream for its ends.
__global__ void k_tall_dir(i64 ncell, i64 V, i64 T, int S, int cshift, const i64 *__restrict__ cellptr, TallScan sc, const i64 *__restrict__ ext,
                           const i64 *__restrict__ pkt_ptr, const TallPkt *__restrict__ sdir, TallPkt *__restrict__ dir) {
    const i64 i0 = (i64)blockIdx.x * blockDim.x + threadIdx.x, step = (i64)gridDim.x * blockDim.x;
    for (i64 c = i0; c < ncell; c += step) {
        if (cellptr[c + 1] == cellptr[c]) continue;
        const i64 pk0 = (i64)sc.before_p[c], np = (i64)(sc.p[c] & kScanMask) - pk0, w0 = (i64)sc.before_w[c];
        const i64 b = c / T, t = c % T, v = b * S + (S > 1 ? tall_range_of(t, T, S) : 0);
        TallPkt *out = dir + pkt_ptr[v] + 1 + (pk0 - ext[4 * v + 2]);
        for (i64 j = 0; j < np; ++j) {
            TallPkt h = sdir[pk0 + j];
            h.off += (unsigned int)(w0 - ext[4 * v]);
            out[j] = h;
        }
    }
    for (i64 v = i0; v < V; v += step) {
        const i64 b = v / S, t_begin = T * (v % S) / S, t_end = T * (v % S + 1) / S;
        i64 first = -1;
        for (i64 u = t_begin; u < t_end; ++u)
            if (cellptr[b * T + u + 1] > cellptr[b * T + u]) { first = u; break; }
        TallPkt h;
        h.off = 0; h.xsrc = first >= 0 ? (unsigned int)(first << cshift) : kNoTile;
        h.c[0] = h.c[1] = h.c[2] = h.c[3] = 0; h.strip = 0; h.pad = 0;
        dir[pkt_ptr[v]] = h;
        h.off = (unsigned int)ext[4 * v + 1];
        h.xsrc = kNoTile;
        for (i64 k = pkt_ptr[v] + 1 + ext[4 * v + 3]; k < pkt_ptr[v + 1]; ++k) dir[k] = h;
    }
}
// ---- host side ---------------------------------------------------------------------------------------------------------
// Rows per block: as tall as the LDS allows, and such that the blocks are (nearly) a multiple of the CU count -- every
// CU then walks the same number of row blocks and no strip is ever split over workgroups.
// Strip-range split (SLP_TALL_SPLIT = S, or -1 = automatic; default 0 = never): short, very wide matrices -- a row block of the
// block-splitting ADMM with 5e5 rows x 5e7 columns -- have too few rows for 256 tall row blocks: with R = rows / 256 every
// workgroup re-stages the whole x (400 MB) for a few hundred items per cell.  With the split the row blocks stay as tall as the
// LDS allows and S workgroups share the strips of one; the partial sums of a row are added in range order (k_strip_combine):
// deterministic, but no longer the single chain of the CSR row sum -- for solvers with a tolerance bar (block ADMM's
// conjugate gradients), hence opt-in.
// `per_cell` = expected entries of a row inside one strip: the dealing of a cell's rows to the 1024 lanes balances the lists (and
// keeps the running sums free of bank conflicts) only while the rows with TWO OR MORE entries in the cell fit one per lane --
// R * P(Poisson(per_cell) >= 2) <= ~900.  Denser matrices get shorter row blocks for that (measured on 2.5e6 x 1e7: density
// 2e-4 at R = 9766 ran at 1.5 TB/s, 17.5 ms, against 4.2 ms for half the entries at 1e-4).
// `block_multiple` (0 = the CU count): the row blocks come in multiples of this -- a row chunk that will run in ONE grid with
// the other chunks of its chunked matrix (tall_fuse) only has to bring its share of a multiple of the CU count.
// `rows_total` > 0 (round 6): the rows are a chunk of a chunked matrix of `rows_total` rows whose chunks run in ONE grid, `rows_before`
// of them in the chunks in front -- the chunk brings its SHARE of a multiple of the CU count of row blocks: the blocks of all
// chunks add up to that multiple exactly whatever the chunks' sizes (the cumulative shares are rounded, not each chunk's), and
// every chunk's blocks are as tall as the whole matrix's would be.  (The per-chunk multiple of round 5 assumes equal chunks: an LP
// whose equality rows are cut off into chunks of their own came out with 2112 row blocks for 256 CUs -- a ninth round of the
// grid a quarter full, A x 14 % slower -- and blocks of 8334 rows.)
static void tall_geometry(i64 nrow, i64 T, double per_cell, i64 block_multiple, int *R_out, int *S_out, i64 rows_before = 0,
                          i64 rows_total = 0) {
    const i64 cus = block_multiple > 0 ? std::min<i64>(block_multiple, ctx().num_cu) : ctx().num_cu;
    const double p2 = 1.0 - exp(-per_cell) * (1.0 + per_cell);
    const i64 rcap = std::max<i64>(1024, std::min<i64>(kTallRmax, p2 > 0.0 ? (i64)(900.0 / p2) : kTallRmax));
    const char *es = getenv("SLP_TALL_SPLIT");
    const int want = es ? atoi(es) : 0;
    const char *e = getenv("SLP_TALL_R");
    if (rows_total > 0 && rows_before >= 0 && rows_before + nrow <= rows_total && want == 0 && !(e && atoi(e) > 0)) {
        const i64 all = ctx().num_cu;
        // (a margin of 1.5 % on the height: a chunk's share is a whole number of blocks)
        const i64 total = std::max<i64>(1, (i64)ceil((double)rows_total / ((double)all * (double)rcap * 0.985))) * all;
        auto upto = [&](i64 r) { return (i64)llround((double)total * (double)r / (double)rows_total); };
        const i64 blocks = upto(rows_before + nrow) - upto(rows_before);
        if (blocks > 0) {
            const i64 R = (nrow + blocks - 1) / blocks;
            if (R >= 1024 && R <= rcap && (nrow + R - 1) / R == blocks) { *S_out = 1; *R_out = (int)R; return; }
        }
    }
    if (want != 0 && !(e && atoi(e) > 0)) {
        const i64 R = std::min<i64>(rcap, std::max<i64>(nrow, 1)), B = (nrow + R - 1) / R;
        i64 S = want > 0 ? want : std::max<i64>(1, cus / B);
        S = std::max<i64>(1, std::min<i64>(S, std::max<i64>(1, T / 64)));  // at least 64 strips per range
        if (S > 1) { *R_out = (int)R; *S_out = (int)S; return; }
    }
    *S_out = 1;
    if (e && atoi(e) > 0) { *R_out = std::min(atoi(e), kTallRmax); return; }
    i64 k = (nrow + cus * rcap - 1) / (cus * rcap);
    if (k < 1) k = 1;
    i64 R = (nrow + k * cus - 1) / (k * cus);
    if (R < 1024) R = std::min<i64>(1024, std::max<i64>(nrow, 1));  // small matrices: fewer, still tall blocks
    *R_out = (int)std::min<i64>(R, rcap);
}

// Entries per sort pass (SLP_TALL_PASS_NNZ): the keys of a pass and the sort's scratch are ~ 16 bytes per entry (32 with fp64
// entries), so a pass of 5e8 entries needs ~ 8 GB whatever the size of the matrix; the sorted keys of the whole copy (8 bytes
// per entry) stay until the packets are written -- all row blocks in ONE launch (a launch per pass leaves most of the chip
// idle: measured 5 x 142 ms against 142 ms on the 2.5e6 x 1e7 slice).
static i64 tall_pass_nnz() {
    const char *e = getenv("SLP_TALL_PASS_NNZ");
    const i64 v = e ? atoll(e) : 500000000ll;
    return v > 0 ? v : 500000000ll;
}

// The tall-cell copy of `a` (transposed: of a^T) straight from the CSR of `a`.  Keys are drawn and sorted in passes over ranges
// of the copy's row blocks (a pass's cells all precede the next pass's: the sorted ranges line up into the sorted whole).
bool tall_build(const CsrDev &a, bool transposed, StripJds &f, const ValueDict *dict, i64 block_multiple, i64 rows_before, i64 rows_total) {
    Phase ph(transposed ? "tall_build (A^T from the CSR of A)" : "tall_build");
    hipStream_t st = ctx().stream;
    f = StripJds();
    if ((dict && (dict->D <= 0 || dict->D > kTallDictMax)) || a.nrow == 0 || a.ncol == 0 || a.nnz == 0) return false;
    const i64 nrowF = transposed ? a.ncol : a.nrow, ncolF = transposed ? a.nrow : a.ncol;
    if (ncolF * 8 >= ((i64)1 << 31)) return false;  // x is addressed through a buffer descriptor with 32-bit byte offsets
    // Strip width (4096, 2048 or 1024 columns; items keep their 12-bit column field): the width whose cells the product kernel is
    // predicted to walk fastest.  Per stored entry the kernel pays for the bytes of its item (4 + the fifth bytes' words: one per
    // lane and four list positions) and for its share of the cell: a fixed part (barrier, pipeline, list ends) and the x-tile
    // (8 C bytes) -- fitted on the 2.5e6-row slices of tools/tall_density_sweep.sh (ms per 2.49e9 entries: 0.393 per byte,
    // 0.95 + 0.34 C / 4096 per cell of 4000 entries; predictions within 3 % of 10 of the 12 measured cases, 10 % low where a cell's lists
    // are five items long).  What decides is the cell's size n = R x (entries per row and strip): R is what tall_geometry
    // leaves of the LDS's 9984 rows once the row blocks are a multiple of the CU count and the rows with two or more entries fit one
    // per lane, so a denser matrix keeps tall blocks only on narrower strips (density 2e-4 on the slice: 4096 columns -> R = 3256,
    // 4.25 / 3.76 ms; 2048 -> R = 9766, 3.09 / 3.07 ms), while the metric's density stays on 4096 (0.41 entries per row and strip,
    // n = 4000; 2048 would halve n: predicted 4.6 ms against 3.26).  SLP_TALL_C forces a width.
    int cshift = kTallColBits;
    {
        double best = 0.0;
        for (int cs = kTallColBits; cs >= kTallColBits - 2; --cs) {
            const i64 Tc = (ncolF + ((i64)1 << cs) - 1) >> cs;
            const double pc = (double)a.nnz / (double)nrowF / (double)Tc;
            int Rc = 0, Sc = 1;
            tall_geometry(nrowF, Tc, pc, transposed ? 0 : block_multiple, &Rc, &Sc, transposed ? 0 : rows_before, transposed ? 0 : rows_total);
            const double n = std::max(1.0, (double)std::min<i64>(Rc, nrowF) * pc), tau = std::ceil(n / kTallT);
            const double bytes = dict ? 4.0 + 4.0 * std::min(n, (double)kTallT) * std::ceil(tau / 4.0) / n : 12.0;
            const double cost = 0.393 * bytes + (0.95 + 0.34 * (double)((i64)1 << cs) / kTallC) * 4000.0 / n;
            if (cs == kTallColBits || cost < best) { best = cost; cshift = cs; }
        }
        if (const char *e = getenv("SLP_TALL_C")) { const int c = atoi(e); if (c == 4096 || c == 2048 || c == 1024) cshift = c == 4096 ? 12 : (c == 2048 ? 11 : 10); }
    }
    const i64 Cw = (i64)1 << cshift;
    const i64 T = (ncolF + Cw - 1) / Cw;
    int R = 0, S = 1;
    tall_geometry(nrowF, T, (double)a.nnz / (double)nrowF / (double)T, transposed ? 0 : block_multiple, &R, &S, transposed ? 0 : rows_before, transposed ? 0 : rows_total);
    const i64 B = (nrowF + R - 1) / R, ncell = B * T, V = B * S;  // V workgroups: (row block, strip range)
    unsigned int cellbits = 1;
    while (((i64)1 << cellbits) < ncell) ++cellbits;
    if (kTallCellShift + cellbits > 64) return false;
    i64 P = (a.nnz + tall_pass_nnz() - 1) / tall_pass_nnz();
    P = std::max<i64>(1, std::min<i64>(P, B));
    DevBuf<int> bad(1);
    bad.zero();
    if (transposed) {  // the per-row column ranges of the passes are binary searches: the rows must be sorted
        hipLaunchKernelGGL(k_tall_sorted, dim3(grid_for(a.nrow * kWave, kBlock)), dim3(kBlock), 0, st, a.nnz, a.idx.p, a.nrow, a.ptr.p, bad.p);
        SLP_HIP(hipGetLastError());
        int hbad = 0;
        bad.download(&hbad, 1);
        if (hbad) return false;
    }
    DevBuf<unsigned long long> sorted((size_t)a.nnz);
    DevBuf<double> svals;  // fp64 entries: the values in the sorted order
    if (!dict) svals.alloc((size_t)a.nnz);
    {
        Phase p1("  tall: keys + sort");
        DevBuf<i64> lo, len, opos;
        if (transposed) { lo.alloc((size_t)a.nrow); len.alloc((size_t)a.nrow + 1); opos.alloc((size_t)a.nrow + 1); }
        const int D = dict ? dict->D : 0;
        const unsigned long long *dk = dict ? dict->keys.p : (const unsigned long long *)nullptr;
        // TR: by (cell, local row) -- see k_tall_keys
        const unsigned b0 = transposed ? (unsigned)(kTallIdBits + kTallColBits) : (unsigned)kTallCellShift, b1 = (unsigned)kTallCellShift + cellbits;
        i64 done = 0;  // entries of the passes so far = where this pass's sorted keys go
        for (i64 pass = 0; pass < P; ++pass) {
            const i64 bb0 = B * pass / P, bb1 = B * (pass + 1) / P;
            if (bb1 == bb0) continue;
            const i64 f0 = bb0 * (i64)R, f1 = std::min<i64>(bb1 * (i64)R, nrowF);  // rows of the copy in this pass
            i64 n_p = 0, obase = 0;
            if (!transposed) {
                i64 ends[2];
                SLP_HIP(hipMemcpyAsync(&ends[0], a.ptr.p + f0, sizeof(i64), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipMemcpyAsync(&ends[1], a.ptr.p + f1, sizeof(i64), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipStreamSynchronize(st));
                obase = ends[0];
                n_p = ends[1] - ends[0];
            } else {
                hipLaunchKernelGGL(k_tall_range, dim3(grid_for(a.nrow, kBlock)), dim3(kBlock), 0, st, a.nrow, a.ptr.p, a.idx.p, f0, f1, lo.p, len.p);
                SLP_HIP(hipGetLastError());
                size_t bytes = 0;
                SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, len.p, opos.p, (i64)0, (size_t)a.nrow + 1, rocprim::plus<i64>(), st));
                DevBuf<char> tmp(bytes);
                SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, len.p, opos.p, (i64)0, (size_t)a.nrow + 1, rocprim::plus<i64>(), st));
                SLP_HIP(hipMemcpyAsync(&n_p, opos.p + a.nrow, sizeof(i64), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipStreamSynchronize(st));
            }
            if (n_p == 0) continue;
            DevBuf<unsigned long long> keys((size_t)n_p);
            DevBuf<double> kvals;
            if (!dict && transposed) kvals.alloc((size_t)n_p);
            const i64 scan_rows = transposed ? a.nrow : f1 - f0, r0 = transposed ? 0 : f0;
            if (transposed)
                hipLaunchKernelGGL((k_tall_keys<true>), dim3(grid_for(scan_rows * kWave, kBlock)), dim3(kBlock), 0, st, r0, scan_rows, R, T, cshift, (i64)0, D, dk,
                                   a.ptr.p, a.idx.p, a.val.p, lo.p, opos.p, obase, keys.p, kvals.p, bad.p);
            else
                hipLaunchKernelGGL((k_tall_keys<false>), dim3(grid_for(scan_rows * kWave, kBlock)), dim3(kBlock), 0, st, r0, scan_rows, R, T, cshift, (i64)0, D, dk,
                                   a.ptr.p, a.idx.p, a.val.p, (const i64 *)nullptr, (const i64 *)nullptr, obase, keys.p, (double *)nullptr, bad.p);
            SLP_HIP(hipGetLastError());
            size_t bytes = 0;
            if (dict) {
                SLP_HIP(rocprim::radix_sort_keys(nullptr, bytes, keys.p, sorted.p + done, (size_t)n_p, b0, b1, st));
                DevBuf<char> tmp(bytes);
                SLP_HIP(rocprim::radix_sort_keys(tmp.p, bytes, keys.p, sorted.p + done, (size_t)n_p, b0, b1, st));
                SLP_HIP(hipStreamSynchronize(st));
            } else {
                const double *vin = transposed ? kvals.p : a.val.p + obase;
                SLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.p, sorted.p + done, vin, svals.p + done, (size_t)n_p, b0, b1, st));
                DevBuf<char> tmp(bytes);
                SLP_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys.p, sorted.p + done, vin, svals.p + done, (size_t)n_p, b0, b1, st));
                SLP_HIP(hipStreamSynchronize(st));
            }
            done += n_p;
        }
        int hbad = 0;
        bad.download(&hbad, 1);
        if (hbad || done != a.nnz) return false;  // unsorted rows (or a value outside the dictionary)
    }
    DevBuf<i64> cellptr((size_t)ncell + 1);
    hipLaunchKernelGGL(k_tall_cellptr, dim3(grid_for(a.nnz, kBlock)), dim3(kBlock), 0, st, a.nnz, ncell, sorted.p, cellptr.p);
    SLP_HIP(hipGetLastError());
    Phase p2("  tall: packets (one pass)");
    DevBuf<unsigned long long> scan(4 * (size_t)ncell + 1), est(2);
    TallScan sc;
    sc.w = scan.p; sc.p = scan.p + ncell; sc.ticket = scan.p + 2 * ncell; sc.before_w = sc.ticket + 1; sc.before_p = sc.before_w + ncell;
    est.zero();
    if (dict) hipLaunchKernelGGL((k_tall_estimate<true>), dim3(grid_for(ncell, kBlock)), dim3(kBlock), 0, st, ncell, cellptr.p, est.p);
    else hipLaunchKernelGGL((k_tall_estimate<false>), dim3(grid_for(ncell, kBlock)), dim3(kBlock), 0, st, ncell, cellptr.p, est.p);
    SLP_HIP(hipGetLastError());
    unsigned long long hest[2];
    est.download(hest, 2);
    // Room for the pass.  It writes straight into the copy's own buffer (a pass into scratch + a copy into an exact-size buffer
    // held both at once: config 5's peak 276 -> 293 of 309 GB), so the room has to be right: the estimate times what the LAST copy
    // built by this process needed of its estimate (the chunks and both orientations of one LP are statistically alike: 1.03 on the
    // benchmark's LPs; kept per orientation and item form) + 0.3 %; the first copy gets + 10 %.  A pass that did not fit is repeated into buffers of the exact size; a buffer
    // more than 1 % too large (the first copy; an LP unlike the last) is exchanged for one of the exact size.  The directory is 32 bytes
    // per ~20 KB cell.
    static std::atomic<double> last_needs[2][2];   // by orientation and item form: the two copies of a chunk differ by ~1 % (zero: none yet)
    std::atomic<double> &last_need_slot = last_needs[transposed ? 1 : 0][dict ? 1 : 0];
    const double last_need = last_need_slot.load();
    const double room = (last_need > 0.5 && last_need < 2.0) ? last_need * 1.003 : 1.10;
    i64 cap_w = (i64)((double)hest[0] * room) + 65536, cap_p = 2 * (i64)hest[1] + 4096;
    if (const char *e = getenv("SLP_TALL_BUILD_ROOM")) cap_w = std::max<i64>(4, (i64)((double)hest[0] * atof(e)));  // (tests: force the second attempt)
    i64 tot_w = 0, tot_p = 0;
    DevBuf<i64> ext(4 * (size_t)V);
    std::vector<i64> hext(4 * (size_t)V);
    DevBuf<unsigned int> spay, sdir;
    DevBuf<double> spayv;
    for (int attempt = 0;; ++attempt) {
        spay.release();   // (a second attempt: the first one's buffers go back before the exact ones are taken)
        spayv.release();
        spay.alloc((size_t)cap_w + 64);
        sdir.alloc((size_t)cap_p * 8);
        if (!dict) spayv.alloc((size_t)cap_w + 64);
        SLP_HIP(hipMemsetAsync(scan.p, 0, (2 * (size_t)ncell + 1) * sizeof(unsigned long long), st));  // levels and the ticket
        // as many workgroups as the chip holds at once (the kernel's LDS: one per compute unit); each takes cells until none is left
        const unsigned grid = (unsigned)std::min<i64>(ncell, (i64)ctx().num_cu);
        if (dict) hipLaunchKernelGGL((k_tall_build<true>), dim3(grid), dim3(kTallT), 0, st, R, T, S, cshift, ncell, sorted.p, (const double *)nullptr, cellptr.p, sc,
                                     cap_w, cap_p, reinterpret_cast<TallPkt *>(sdir.p), spay.p, (double *)nullptr);
        else hipLaunchKernelGGL((k_tall_build<false>), dim3(grid), dim3(kTallT), 0, st, R, T, S, cshift, ncell, sorted.p, svals.p, cellptr.p, sc,
                                cap_w, cap_p, reinterpret_cast<TallPkt *>(sdir.p), spay.p, spayv.p);
        hipLaunchKernelGGL(k_tall_streams, dim3(grid_for(V, kBlock)), dim3(kBlock), 0, st, V, T, S, cellptr.p, sc, ext.p);
        SLP_HIP(hipGetLastError());
        ext.download(hext.data(), hext.size());
        tot_w = hext[4 * (V - 1)] + hext[4 * (V - 1) + 1];
        tot_p = hext[4 * (V - 1) + 2] + hext[4 * (V - 1) + 3];
        if (tot_w <= cap_w && tot_p <= cap_p) break;
        SLP_REQUIRE(attempt == 0, "tall_build: the pass with the exact sizes did not fit");
        cap_w = tot_w; cap_p = tot_p;   // some cell did not fit and wrote nothing: once more, with the sizes the pass has found
    }
    sorted.release();  // (the keys are spent: their memory serves the copy itself)
    svals.release();
    std::vector<i64> hbase((size_t)V + 1), hpkt((size_t)V + 1);
    const i64 limit = dict ? ((i64)1 << 28) : ((i64)1 << 27);  // payload words of one workgroup (32-bit byte offsets; fp64: 2 x)
    hpkt[0] = 0;
    for (i64 v = 0; v < V; ++v) {
        if (hext[4 * v + 1] >= limit) return false;  // a workgroup's payload outgrows its 32-bit byte offsets: another format serves
        hbase[v] = hext[4 * v];
        const i64 np = 1 + hext[4 * v + 3];  // whole groups of 2 x depth packets, then 2 x depth more that are only ever prefetched
        hpkt[v + 1] = hpkt[v] + (np + 2 * kTallDepth - 1) / (2 * kTallDepth) * (2 * kTallDepth) + 2 * kTallDepth;
    }
    hbase[V] = tot_w;
    DevBuf<i64> dpkt;
    dpkt.upload(hpkt.data(), hpkt.size());
    if (hest[0]) last_need_slot.store((double)tot_w / (double)hest[0]);
    if (cap_w > tot_w + tot_w / 100 + 65536) {   // too much room: the copy moves into buffers of its size
        DevBuf<unsigned int> exact((size_t)tot_w + 64);
        SLP_HIP(hipMemcpyAsync(exact.p, spay.p, (size_t)tot_w * sizeof(unsigned int), hipMemcpyDeviceToDevice, st));
        DevBuf<double> exactv;
        if (!dict) {
            exactv.alloc((size_t)tot_w + 64);
            SLP_HIP(hipMemcpyAsync(exactv.p, spayv.p, (size_t)tot_w * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        SLP_HIP(hipStreamSynchronize(st));
        spay = std::move(exact);
        spayv = std::move(exactv);
    }
    f.tall_dir.emplace_back((size_t)hpkt[V] * 8);
    f.tall_pay.emplace_back(std::move(spay));
    DevBuf<unsigned int> &dir = f.tall_dir.back(), &pay = f.tall_pay.back();
    double *vals = nullptr;
    if (!dict) { f.tall_val.emplace_back(std::move(spayv)); vals = f.tall_val.back().p; }
    hipLaunchKernelGGL(k_tall_dir, dim3(grid_for(std::max<i64>(ncell, V), kBlock)), dim3(kBlock), 0, st, ncell, V, T, S, cshift, cellptr.p, sc, ext.p, dpkt.p,
                       reinterpret_cast<const TallPkt *>(sdir.p), reinterpret_cast<TallPkt *>(dir.p));
    SLP_HIP(hipGetLastError());
    std::vector<TallWg> wg((size_t)V);
    for (i64 v = 0; v < V; ++v) {
        const i64 b = v / S;
        wg[(size_t)v].dir = dir.p + hpkt[v] * 8;
        wg[(size_t)v].pay = pay.p + hbase[v];
        wg[(size_t)v].val = vals ? vals + hbase[v] : nullptr;
        wg[(size_t)v].dict = nullptr;   // (the launch's table: strip_spmv_with_dict swaps it; a fused composite names each chunk's)
        wg[(size_t)v].npk = hpkt[v + 1] - hpkt[v];
        wg[(size_t)v].row0 = (v % S) * nrowF + b * (i64)R;   // S > 1: into the partial-sum array, one slice per strip range
        wg[(size_t)v].x0 = 0;
        wg[(size_t)v].ncol = ncolF;
        wg[(size_t)v].nrows = (int)std::min<i64>(R, nrowF - b * (i64)R);
        wg[(size_t)v].D = dict ? dict->D : 0;
    }
    if (S > 1 && V > 8) {
        // Strip-range split (a block of BASELINE config 5: 51 row blocks x 5 ranges): the workgroups of ONE range read the same fifth
        // of x (80 MB of 400), so they belong on one XCD's L2 -- workgroups whose index is equal mod 8 share an XCD
        // (cdna_hip_programming.md T1; a placement hint, never a correctness matter: the ranges' partial sums are combined in range
        // order by k_tall_combine whatever ran where).  The table is laid out so that XCD label l walks a contiguous stretch of the
        // range-major order (bijective for any V): every XCD then pulls at most two ranges' x instead of all five (round 5's PMC:
        // 16.8 GB per product for a 12.9 GB copy, the 400 MB x fetched by every XCD).
        const i64 q = V / 8, r = V % 8;
        std::vector<TallWg> placed((size_t)V);
        for (i64 orig = 0; orig < V; ++orig) {
            const i64 xcd = orig % 8;
            const i64 pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;   // position in the range-major order
            const i64 sr = pos / B, b = pos % B;
            placed[(size_t)orig] = wg[(size_t)(b * S + sr)];
        }
        wg.swap(placed);
    }
    f.tall_wg.upload(wg.data(), wg.size());
    SLP_HIP(hipStreamSynchronize(st));
    // (what a product streams: the words written, not the buffer's margin)
    f.tall_bytes = (dir.n + (size_t)tot_w) * sizeof(unsigned int) + (vals ? (size_t)tot_w * sizeof(double) : 0) + wg.size() * sizeof(TallWg) +
                   (dict ? (size_t)dict->D * sizeof(double) : 0);
    f.nrow = nrowF; f.ncol = ncolF; f.nnz = a.nnz; f.T = T; f.B = B; f.C = (int)Cw; f.rpl = 1;
    f.D = dict ? dict->D : 0;
    f.dict = dict ? dict->values.p : nullptr;
    f.tall = true;
    f.tall_R = R;
    f.S = S;
    if (S > 1) f.part.alloc((size_t)S * (size_t)nrowF);
    f.ok = true;
    return true;
}

// Long rows over a width far beyond an L2, too sparse for the LDS strips: 0.05 .. 2.5 entries per (row, 4096 columns).
bool tall_wanted(i64 nrow, i64 ncol, i64 nnz) {
    const char *e = getenv("SLP_TALL");
    if (e && e[0] == '0') return false;
    const char *m = getenv("SLP_STRIP_MIN_NNZ");
    const i64 min_nnz = m ? atoll(m) : 30000000ll;
    if (nnz < min_nnz || nrow <= 0 || ncol <= 0) return false;
    const double per_cell = ((double)nnz / (double)nrow) / (double)((ncol + kTallC - 1) / kTallC);
    return per_cell >= 0.05 && per_cell < 2.5;
}

}  // namespace slp

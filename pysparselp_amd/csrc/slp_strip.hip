// slp_strip.hip -- the bandwidth-bound SpMV for long rows: column strips with the
// x-tile staged in LDS, rows stored as jagged diagonals inside every
// (row block, strip) cell ("strip-JDS").
//
// Why: with ~1000 stored entries per row and uniformly random columns, the plain
// CSR kernel gathers x[j] from an 8-16 MB vector 2e9 times per product; every
// gather is a separate cache-line request to L2 / Infinity Cache and the kernel
// runs at the fabric's line rate (measured: 1.1 TB/s algorithmic, 14 % of HBM
// peak) instead of streaming the matrix.  Here the matrix is cut into vertical
// strips of C columns; a workgroup owns R rows, keeps their R running sums in
// LDS, and walks the strips: it stages x[strip] (C doubles) in LDS once and then
// streams the cell's entries with perfectly coalesced loads and gathers x from LDS.
//
// Layout of one cell (row block b, strip t), all cells back to back, b-major:
//   rows of the cell are sorted by their entry count in the strip (descending),
//   `perm[p]` = local row at sorted position p, `len[p]` = its count;
//   jagged diagonal s holds the s-th entry of every row with len > s, in sorted
//   order, padded to an even number of entries, at offset soff[s] inside the cell.
//   Entry = (fp64 value, uint16 column inside the strip): 10 B instead of 12 B.
// Thread p of the 1024 owns sorted positions 2p and 2p+1: one 16-byte load brings
// the s-th value of both rows, one 4-byte load both columns.
// A row's entries keep their column order across and inside strips, and each
// row is accumulated by one thread at a time starting from its running sum, so
// the result equals the SEQUENTIAL single-accumulator sum of the CSR row -- bit
// for bit (rows must be sorted by column, which holds for generated matrices
// and for every device-built transpose).
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

bool comm_active();  // slp_comm.hip

constexpr int kStripC = 7680;    // columns per strip: 60 KB of x in LDS
constexpr int kDictC = 5888;     // value-dictionary variant: 46 KB of x + 16 KB of distinct values in LDS
constexpr int kDictMax = 2048;   // most distinct stored values the dictionary variant takes
constexpr int kDictHash = 16384; // slots of the detection hash set
constexpr unsigned long long kDictEmpty = ~0ull;
constexpr int kQuadC = 3968;     // quad variant: 4 rows per lane (4096-row blocks), 31 KB of x, 12-bit column + 12-bit value id
constexpr int kQuadR = 4 * 1024;
#ifndef SLP_QUAD_G
#define SLP_QUAD_G 2
#endif
#ifndef SLP_QUAD_U1
#define SLP_QUAD_U1 3  // 12-byte quad loads in flight per lane: 3 measured best with the branch-free slot body (1.92 ms; 4: 1.98, 2: 2.38)
#endif
constexpr int kQuadG = SLP_QUAD_G;       // rows of a lane's quad whose LDS gathers are issued together
constexpr int kQuadU1 = SLP_QUAD_U1, kQuadU2 = 6;  // 12-byte quad loads in flight per lane (64 VGPRs at two workgroups per CU: 6 would spill)
constexpr int kWideC = 131072;   // wide strips: 1 MB of x per strip, gathered from L2 (no LDS tile): rows too sparse for the LDS strips
constexpr int kWideColShift = 11; // wide dictionary entry: value id in the low 11 bits, column inside the strip above
#ifndef SLP_DICT_U1
#define SLP_DICT_U1 8
#endif
constexpr int kDictU1 = SLP_DICT_U1, kDictU2 = 8;  // entry-pair loads in flight per lane (one / two right-hand sides)
constexpr int kStripT = 1024;    // threads per workgroup
constexpr int kStripR = 2048;    // rows per block (two per thread)
constexpr int kStripSL = 256;
#ifndef SLP_NT_DEFAULT
#define SLP_NT_DEFAULT 2  // fp64 strips: non-temporal 16-byte value loads (3.52 vs 3.88 ms at config 3), plain 4-byte column loads
#endif    // slots (max entries of one row inside one strip)
static_assert(kStripC % 2 == 0 && kDictC % 2 == 0 && kStripR == 2 * kStripT, "strip geometry");

// ---- distinct stored values ------------------------------------------------------
// order-preserving map of fp64 bit patterns to unsigned integers (-0.0 sorts right below +0.0)
__host__ __device__ inline unsigned long long value_key(unsigned long long bits) {
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}

// Insert every stored value's bit pattern into an open-addressing hash set; `count` = distinct values so
// far.  Gives up (count > limit) as soon as the matrix turns out to have too many.
// A workgroup remembers in LDS (direct-mapped, 4096 slots) which bit patterns it has already found in the global set: with the few
// hundred distinct values a dictionary matrix has, all but the first look-ups of a pattern end there instead of in an atomic load
// from L2 per stored entry (10.8 -> 6.4 ms per 1.25e9 entries).  A slot only ever holds a pattern its writer has seen IN the
// global set, so a stale or overwritten slot costs a global look-up, never a missed insertion.
__global__ __launch_bounds__(kBlock) void k_value_set(i64 nnz, const double *__restrict__ val, unsigned long long *table,
                                                      unsigned int *count, unsigned int limit) {
    constexpr int kSeen = 4096;
    __shared__ unsigned long long seen[kSeen];
    for (int i = threadIdx.x; i < kSeen; i += kBlock) seen[i] = kDictEmpty;
    __syncthreads();
    // one pattern: false = no dictionary for this matrix
    auto take = [&](double v) -> bool {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        if (bits == kDictEmpty || v != v) {  // NaN entries (or the sentinel): no dictionary
            atomicAdd(count, limit + 1);
            return false;
        }
        const unsigned long long mixed = bits * 0x9E3779B97F4A7C15ull;
        const unsigned int slot = (unsigned int)(mixed >> 52) & (kSeen - 1);
        if (seen[slot] == bits) return true;
        unsigned int h = (unsigned int)(mixed >> 40) & (kDictHash - 1);
        for (int probes = 0;; ++probes) {
            if (probes >= 64) {  // a set this crowded holds far more than `limit` values: give up (never spins on a full table)
                atomicAdd(count, limit + 1);
                return false;
            }
            unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == bits) break;
            if (cur == kDictEmpty) {
                cur = atomicCAS(&table[h], kDictEmpty, bits);
                if (cur == kDictEmpty) { atomicAdd(count, 1u); break; }
                if (cur == bits) break;
            }
            h = (h + 1) & (kDictHash - 1);
        }
        seen[slot] = bits;   // (in the global set now)
        return true;
    };
    // four entries of a thread in flight together (one at a time the kernel ran at the latency of its loads: 0.8 TB/s)
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 it = 0;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += 4 * stride, ++it) {
        if ((it & 15) == 0 && __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > limit) return;
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = val[k + j * stride < nnz ? k + j * stride : k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!take(v[j])) return;
    }
}

// ---- conversion -------------------------------------------------------------
// A thread's view of its row's entries: one aligned chunk of N of them in registers, refilled (16-byte loads) when the row's
// cursor leaves it; an entry is picked with a compare-select chain (static register indices).  The conversion walks 2048-4096
// rows per workgroup strip by strip, a handful of entries per row and strip: read straight from memory, every 64-byte sector
// of a row's stream came in again for each of its pieces -- the strips in between push it out of the L2 (PMC: 14-15 x the
// matrix fetched, the kernels running at the speed of that traffic).
template <class V, int N>
struct EntryWindow {
    static_assert((N & (N - 1)) == 0 && N * sizeof(V) >= 16, "a power of two, at least one 16-byte load");
    V v[N];
    i64 w;
    __device__ __forceinline__ V get(const V *__restrict__ base, i64 k, i64 total, bool aligned) {
        const i64 c = k / N;
        if (c != w) {
            w = c;
            const i64 k0 = c * N;
            if (aligned && k0 + N <= total) {
                constexpr int PER = 16 / (int)sizeof(V);
                const uint4 *p4 = reinterpret_cast<const uint4 *>(base + k0);
#pragma unroll
                for (int q = 0; q < N / PER; ++q) {
                    const uint4 u = p4[q];
                    if (sizeof(V) == 4) {
                        v[q * PER + 0] = (V)__builtin_bit_cast(int, u.x);
                        v[q * PER + 1] = (V)__builtin_bit_cast(int, u.y);
                        v[q * PER + 2] = (V)__builtin_bit_cast(int, u.z);
                        v[q * PER + 3] = (V)__builtin_bit_cast(int, u.w);
                    } else {
                        v[q * PER + 0] = (V)__hiloint2double((int)u.y, (int)u.x);
                        v[q * PER + 1] = (V)__hiloint2double((int)u.w, (int)u.z);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < N; ++q) v[q] = k0 + q < total ? base[k0 + q] : V(0);
            }
        }
        const int i = (int)(k & (N - 1));
        V r = v[0];
#pragma unroll
        for (int q = 1; q < N; ++q) r = (i == q) ? v[q] : r;
        return r;
    }
};

// From the rows-per-count histogram of a cell: start[l] = rows with more than l entries (= first sorted position of count l,
// = entries of slot l), offs[s] = where slot s begins (slots padded to a multiple of RPL entries); returns the padded cell
// size.  Run by the first wave, four counts per lane and two scans over the wave (one thread walked the 256 counts twice).
template <int RPL>
__device__ __forceinline__ unsigned int strip_slot_scan(const unsigned int *hist, unsigned int *start, unsigned int *offs) {
    static_assert(kStripSL == 256, "four counts per lane of one wave");
    const int lane = threadIdx.x;  // (called with threadIdx.x < 64)
    const uint4 h = reinterpret_cast<const uint4 *>(hist)[lane];
    const unsigned int mine = h.x + h.y + h.z + h.w;
    unsigned int suf = mine;       // counts of this lane and all higher ones
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int o = __shfl_down(suf, d);
        if (lane + d < 64) suf += o;
    }
    uint4 st;
    st.w = suf - mine;
    st.z = st.w + h.w;
    st.y = st.z + h.z;
    st.x = st.y + h.y;
    const uint4 pd = make_uint4((st.x + (RPL - 1u)) & ~(RPL - 1u), (st.y + (RPL - 1u)) & ~(RPL - 1u), (st.z + (RPL - 1u)) & ~(RPL - 1u),
                                (st.w + (RPL - 1u)) & ~(RPL - 1u));
    const unsigned int padded = pd.x + pd.y + pd.z + pd.w;
    unsigned int pre = padded;     // padded slots of this lane and all lower ones
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int o = __shfl_up(pre, d);
        if (lane >= d) pre += o;
    }
    if (start) reinterpret_cast<uint4 *>(start)[lane] = st;
    if (offs) {
        uint4 of;
        of.x = pre - padded;
        of.y = of.x + pd.x;
        of.z = of.y + pd.y;
        of.w = of.z + pd.z;
        reinterpret_cast<uint4 *>(offs)[lane] = of;
    }
    return __shfl(pre, 63);
}

// pass 1: per (block, strip): every row's entry count (uint8) and the padded cell size
//         sum_s even(cnt[s]),  cnt[s] = rows of the cell with more than s entries
template <int C, int RPL>
__global__ __launch_bounds__(kStripT) void k_strip_count(i64 nrow, i64 T, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                         unsigned char *__restrict__ len, unsigned long long *__restrict__ total,
                                                         int *__restrict__ bad) {
    __shared__ __attribute__((aligned(16))) unsigned int hist[kStripSL];
    constexpr int R = RPL * kStripT;  // rows per block, RPL per thread
    const i64 b = blockIdx.x;
    i64 k[RPL], e[RPL];
    i32 prev[RPL];
    EntryWindow<i32, 16> wi[RPL];
    const i64 nnz = ptr[nrow];
    const bool aligned = (reinterpret_cast<uintptr_t>(idx) & 15) == 0;
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        const i64 row = b * R + h * kStripT + threadIdx.x;
        k[h] = e[h] = 0;
        prev[h] = -1;
        wi[h].w = -1;
        if (row < nrow) { k[h] = ptr[row]; e[h] = ptr[row + 1]; }
    }
    for (i64 t = 0; t < T; ++t) {
        if (threadIdx.x < kStripSL) hist[threadIdx.x] = 0;
        __syncthreads();
        const i64 hi = (t + 1) * (i64)C;
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
            i64 c = 0;
            while (k[h] < e[h]) {
                const i32 j = wi[h].get(idx, k[h], nnz, aligned);
                if (j >= hi) break;
                if (j <= prev[h]) atomicOr(bad, 1);  // rows must be strictly increasing in column
                prev[h] = j;
                ++k[h]; ++c;
            }
            if (c >= kStripSL) { atomicOr(bad, 2); c = kStripSL - 1; }
            len[(b * T + t) * R + h * kStripT + threadIdx.x] = (unsigned char)c;
            atomicAdd(&hist[c], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // slot l holds (rows with a count > l) entries, padded to a multiple of RPL
            const unsigned int sum = strip_slot_scan<RPL>(hist, nullptr, nullptr);
            if (threadIdx.x == 0) total[b * T + t] = sum;
        }
        __syncthreads();
    }
}

// pass 2: sort the rows of every cell by count, write perm / sorted len / slot offsets and the entries
// (the output arrays are zero-filled beforehand: the pad entries stay (0.0, column 0))
// D > 0: value-dictionary output -- `keys` are the D sorted value keys, entries go to `oent` as packed pairs
// {id(2p), id(2p+1), col(2p), col(2p+1)} of uint16 (pair index = entry index / 2) instead of oval / ocol
// RPL == 4 (always with a dictionary): entries are 24-bit (id | column << 12), four of them -- sorted positions
// 4p .. 4p+3 of one slot -- form a 12-byte quad; offsets and `base` still count entries.
// WIDE (strips of kWideC columns, RPL == 2): 32-bit columns -- dictionary entries are one uint32 (id | column << 11) in
// `oent`, fp64 entries keep `oval` and a uint32 column array in `ocol`.
template <int C, int RPL, bool WIDE>
__global__ __launch_bounds__(kStripT) void k_strip_fill(i64 nrow, i64 T, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                        const double *__restrict__ val, const unsigned char *__restrict__ len,
                                                        const i64 *__restrict__ base, unsigned short *__restrict__ perm,
                                                        unsigned char *__restrict__ slen, unsigned int *__restrict__ soff,
                                                        double *__restrict__ oval, unsigned short *__restrict__ ocol, int D,
                                                        const unsigned long long *__restrict__ keys,
                                                        unsigned short *__restrict__ oent) {
    __shared__ __attribute__((aligned(16))) unsigned int hist[kStripSL];   // rows with exactly this count, then the per-count cursor
    __shared__ __attribute__((aligned(16))) unsigned int start[kStripSL];  // rows with a larger count  (= first sorted position of this count)
    __shared__ __attribute__((aligned(16))) unsigned int offs[kStripSL];   // slot offsets
    __shared__ unsigned long long skey[kDictMax];
    for (int q = threadIdx.x; q < D; q += kStripT) skey[q] = keys[q];
    constexpr int R = RPL * kStripT;
    const i64 b = blockIdx.x;
    i64 k[RPL];
    // (64 registers of windows: 64-byte chunks of both streams with two rows per thread, 32-byte chunks with four)
    EntryWindow<i32, RPL == 2 ? 16 : 8> wi[RPL];
    EntryWindow<double, RPL == 2 ? 8 : 4> wv[RPL];
    const i64 nnz = ptr[nrow];
    const bool aligned = ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(val)) & 15) == 0;
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        const i64 row = b * R + h * kStripT + threadIdx.x;
        k[h] = (row < nrow) ? ptr[row] : 0;
        wi[h].w = -1;
        wv[h].w = -1;
    }
    for (i64 t = 0; t < T; ++t) {
        const i64 cell = b * T + t;
        if (threadIdx.x < kStripSL) hist[threadIdx.x] = 0;
        __syncthreads();
        unsigned int c[RPL];
        for (int h = 0; h < RPL; ++h) {
            c[h] = len[cell * R + h * kStripT + threadIdx.x];
            atomicAdd(&hist[c[h]], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) strip_slot_scan<RPL>(hist, start, offs);  // cnt[s] = start[s]; slot s occupies cnt[s] padded to RPL
        __syncthreads();
        if (threadIdx.x < kStripSL) {
            soff[cell * kStripSL + threadIdx.x] = offs[threadIdx.x];
            hist[threadIdx.x] = 0;
        }
        __syncthreads();
        const i64 bs = base[cell];
        const i32 col0 = (i32)(t * (i64)C);
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
            const unsigned int pos = start[c[h]] + atomicAdd(&hist[c[h]], 1u);
            perm[cell * R + pos] = (unsigned short)(h * kStripT + threadIdx.x);
            slen[cell * R + pos] = (unsigned char)c[h];
            for (unsigned int s = 0; s < c[h]; ++s) {
                const i64 o = bs + offs[s] + pos;
                const unsigned int jw = (unsigned int)(wi[h].get(idx, k[h] + s, nnz, aligned) - col0);
                const unsigned short jc = (unsigned short)jw;
                const double vv = wv[h].get(val, k[h] + s, nnz, aligned);
                if (D > 0) {
                    const unsigned long long key = value_key((unsigned long long)__double_as_longlong(vv));
                    int lo = 0, hi = D - 1;  // the value is in the dictionary: plain binary search
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (skey[mid] < key) lo = mid + 1;
                        else hi = mid;
                    }
                    if (WIDE) {
                        reinterpret_cast<unsigned int *>(oent)[o] = (unsigned int)lo | (jw << kWideColShift);
                    } else if (RPL == 4) {
                        const unsigned int e24 = (unsigned int)lo | ((unsigned int)jc << 12);
                        unsigned char *o8 = reinterpret_cast<unsigned char *>(oent) + o * 3;
                        o8[0] = (unsigned char)e24;
                        o8[1] = (unsigned char)(e24 >> 8);
                        o8[2] = (unsigned char)(e24 >> 16);
                    } else {
                        oent[(o >> 1) * 4 + (o & 1)] = (unsigned short)lo;
                        oent[(o >> 1) * 4 + 2 + (o & 1)] = jc;
                    }
                } else if (WIDE) {
                    oval[o] = vv;
                    reinterpret_cast<unsigned int *>(ocol)[o] = jw;
                } else {
                    oval[o] = vv;
                    ocol[o] = jc;
                }
            }
            k[h] += c[h];
        }
        // Quad variant: the up to three positions that pad a slot to whole quads become PAD ENTRIES (value id D, column C): the
        // kernel keeps -0.0 at dv[D] and 1.0 at xt[C], so a pad adds -0.0 * 1.0 = -0.0 to a running sum -- v + -0.0 == v bit
        // for bit for every v -- and the slot body needs no predicate (k_qstrip_spmv).
        if (RPL == 4 && D > 0 && threadIdx.x < kStripSL) {
            const unsigned int cnt = start[threadIdx.x], padded = (cnt + 3u) & ~3u;   // rows with more than `slot` entries
            const unsigned int e24 = (unsigned int)D | ((unsigned int)C << 12);
            for (unsigned int q = cnt; q < padded; ++q) {
                unsigned char *o8 = reinterpret_cast<unsigned char *>(oent) + (bs + offs[threadIdx.x] + q) * 3;
                o8[0] = (unsigned char)e24;
                o8[1] = (unsigned char)(e24 >> 8);
                o8[2] = (unsigned char)(e24 >> 16);
            }
        }
        // Pair variant: the one position that pads an odd slot to whole pairs, the same way (k_dstrip_spmv)
        if (RPL == 2 && D > 0 && !WIDE && threadIdx.x < kStripSL && (start[threadIdx.x] & 1u)) {
            const i64 o = bs + offs[threadIdx.x] + start[threadIdx.x];
            oent[(o >> 1) * 4 + (o & 1)] = (unsigned short)D;
            oent[(o >> 1) * 4 + 2 + (o & 1)] = (unsigned short)C;
        }
        __syncthreads();
    }
    if (RPL == 2 && D > 0 && !WIDE && blockIdx.x == gridDim.x - 1 && threadIdx.x < 2) {   // the all-pad pair behind the last cell
        const i64 o = ((base[(i64)gridDim.x * T] + 1) & ~(i64)1) + threadIdx.x;
        oent[(o >> 1) * 4 + (o & 1)] = (unsigned short)D;
        oent[(o >> 1) * 4 + 2 + (o & 1)] = (unsigned short)C;
    }
    // ... and one quad of pad entries right behind the last cell: where a lane whose rows have all ended points its load
    if (RPL == 4 && D > 0 && blockIdx.x == gridDim.x - 1 && threadIdx.x < 4) {
        const i64 endq = (base[(i64)gridDim.x * T] + 3) >> 2;   // (every cell is a whole number of quads)
        const unsigned int e24 = (unsigned int)D | ((unsigned int)C << 12);
        unsigned char *o8 = reinterpret_cast<unsigned char *>(oent) + (endq * 4 + threadIdx.x) * 3;
        o8[0] = (unsigned char)e24;
        o8[1] = (unsigned char)(e24 >> 8);
        o8[2] = (unsigned char)(e24 >> 16);
    }
}

// Entry streams are read exactly once per product: non-temporal loads keep them from displacing the x-tiles (which
// every workgroup re-reads) in the caches.  NT is a template switch so that both forms can be timed (SLP_NT_LOADS=0/1).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
    if (!NT) return *p;
    const f64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const f64x2_t *>(p));
    return make_double2(v.x, v.y);
}
template <bool NT>
__device__ __forceinline__ ushort2 ld_stream(const ushort2 *p) {
    if (!NT) return *p;
    const unsigned int v = __builtin_nontemporal_load(reinterpret_cast<const unsigned int *>(p));
    return make_ushort2((unsigned short)(v & 0xffffu), (unsigned short)(v >> 16));
}
template <bool NT>
__device__ __forceinline__ uint2 ld_stream(const uint2 *p) {
    if (!NT) return *p;
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t *>(p));
    return make_uint2(v.x, v.y);
}

// ---- the product ---------------------------------------------------------------
// One workgroup per row block; 60 KB x-tile + 16 KB running sums + 1 KB slot offsets of LDS
// (two workgroups per CU).  out[row] = sum over the row (single accumulator, storage order).
// ABLATE != 0 is instantiated only in the -DSLP_ABLATION build (timing experiments, wrong results): 1 = no x-tile
// staging, 2 = no entry streaming
// POW: every stored value v enters as |v|^pw * 1.0 (the matrix of the Chambolle-Pock preconditioner sums, slp_cp.hip).
template <int ABLATE, bool NT, bool NT4 = true, bool POW = false, bool ACC = false>
__global__ __launch_bounds__(kStripT, 8) void k_strip_spmv(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                           const unsigned short *__restrict__ perm,
                                                           const unsigned char *__restrict__ slen,
                                                           const unsigned int *__restrict__ soff, const double *__restrict__ val,
                                                           const unsigned short *__restrict__ col, const double *__restrict__ x,
                                                           double *__restrict__ out, double pw) {
    __shared__ double xt[kStripC];
    __shared__ double acc[kStripR];
    __shared__ unsigned int offs[kStripSL];
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    // ACC: the running sums start from what `out` holds (a row chunk of a chunked matrix continuing the column sums of
    // the chunks before it: the chain of additions of the unchunked product); partial-sum mode takes it in the combine.
    // (A template parameter: as a run-time flag it cost the dictionary kernels, which sit exactly at their 64 registers,
    // 7-10 spilled registers and 7-12 % of their speed.)
    const bool cont = ACC && gridDim.y == 1;
    acc[p] = (cont && b * kStripR + p < nrow) ? out[b * kStripR + p] : 0.0;
    acc[p + kStripT] = (cont && b * kStripR + kStripT + p < nrow) ? out[b * kStripR + kStripT + p] : 0.0;
    // gridDim.y > 1: this workgroup covers only its share of the strips and writes partial sums
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
        // stage the strip of x with 16-byte loads
        const i64 c0 = t * (i64)kStripC;
#pragma unroll
        for (int q = 0; q < (ABLATE == 1 ? 0 : (kStripC / 2 + kStripT - 1) / kStripT); ++q) {
            const int j = (q * kStripT + p) * 2;
            if (j < kStripC) {
                double2 v = make_double2(0.0, 0.0);
                if (c0 + j + 1 < ncol) v = *reinterpret_cast<const double2 *>(x + c0 + j);
                else if (c0 + j < ncol) v.x = x[c0 + j];
                *reinterpret_cast<double2 *>(&xt[j]) = v;
            }
        }
        if (p < kStripSL) offs[p] = soff[cell * kStripSL + p];
        const ushort2 r = reinterpret_cast<const ushort2 *>(perm + cell * kStripR)[p];
        const uchar2 nn = reinterpret_cast<const uchar2 *>(slen + cell * kStripR)[p];
        const unsigned int n0 = (ABLATE == 2) ? 0u : nn.x, n1 = (ABLATE == 2) ? 0u : nn.y;  // n0 >= n1 (sorted)
        const double2 *__restrict__ v2 = reinterpret_cast<const double2 *>(val + base[cell]);
        const ushort2 *__restrict__ c2 = reinterpret_cast<const ushort2 *>(col + base[cell]);
        __syncthreads();
        double a0 = acc[r.x], a1 = acc[r.y];
        unsigned int s = 0;
        for (; s + 4 <= n0; s += 4) {  // four independent 16-byte + 4-byte loads in flight per lane
            const unsigned int o0 = (offs[s] >> 1) + p, o1 = (offs[s + 1] >> 1) + p, o2 = (offs[s + 2] >> 1) + p,
                               o3 = (offs[s + 3] >> 1) + p;
            double2 w0 = ld_stream<NT>(v2 + o0), w1 = ld_stream<NT>(v2 + o1), w2 = ld_stream<NT>(v2 + o2), w3 = ld_stream<NT>(v2 + o3);
            if (POW) {
                w0.x = abs_pow(w0.x, pw) * 1.0; w0.y = abs_pow(w0.y, pw) * 1.0; w1.x = abs_pow(w1.x, pw) * 1.0; w1.y = abs_pow(w1.y, pw) * 1.0;
                w2.x = abs_pow(w2.x, pw) * 1.0; w2.y = abs_pow(w2.y, pw) * 1.0; w3.x = abs_pow(w3.x, pw) * 1.0; w3.y = abs_pow(w3.y, pw) * 1.0;
            }
            const ushort2 j0 = ld_stream<NT && NT4>(c2 + o0), j1 = ld_stream<NT && NT4>(c2 + o1), j2 = ld_stream<NT && NT4>(c2 + o2),
                          j3 = ld_stream<NT && NT4>(c2 + o3);
            a0 += w0.x * xt[j0.x];
            a0 += w1.x * xt[j1.x];
            a0 += w2.x * xt[j2.x];
            a0 += w3.x * xt[j3.x];
            if (s < n1) a1 += w0.y * xt[j0.y];
            if (s + 1 < n1) a1 += w1.y * xt[j1.y];
            if (s + 2 < n1) a1 += w2.y * xt[j2.y];
            if (s + 3 < n1) a1 += w3.y * xt[j3.y];
        }
        for (; s < n0; ++s) {
            const unsigned int o = (offs[s] >> 1) + p;
            double2 w = ld_stream<NT>(v2 + o);
            if (POW) { w.x = abs_pow(w.x, pw) * 1.0; w.y = abs_pow(w.y, pw) * 1.0; }
            const ushort2 j = ld_stream<NT && NT4>(c2 + o);
            a0 += w.x * xt[j.x];
            if (s < n1) a1 += w.y * xt[j.y];
        }
        acc[r.x] = a0;
        acc[r.y] = a1;
        __syncthreads();
    }
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        if (row < nrow) out[(i64)blockIdx.y * nrow + row] = acc[h * kStripT + p];
    }
}


// Two right-hand sides in one pass over the matrix: out0 = A x0, out1 = A x1.  The matrix (the only
// HBM-sized operand) is streamed once; both x-tiles and both sets of running sums live in LDS (153 KB,
// one workgroup per CU).  Per row and per vector the additions are the same chain as in k_strip_spmv,
// so each output is bit-identical to the single-vector product.
template <bool ACC>
__global__ __launch_bounds__(kStripT, 4) void k_strip_spmv2(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                            const unsigned short *__restrict__ perm,
                                                            const unsigned char *__restrict__ slen,
                                                            const unsigned int *__restrict__ soff, const double *__restrict__ val,
                                                            const unsigned short *__restrict__ col, const double *__restrict__ x0,
                                                            const double *__restrict__ x1, double *__restrict__ out0,
                                                            double *__restrict__ out1) {
    __shared__ double xt0[kStripC];
    __shared__ double xt1[kStripC];
    __shared__ double acc0[kStripR];
    __shared__ double acc1[kStripR];
    __shared__ unsigned int offs[kStripSL];
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    const bool cont = ACC && gridDim.y == 1;  // as in k_strip_spmv
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        acc0[h * kStripT + p] = (cont && row < nrow) ? out0[row] : 0.0;
        acc1[h * kStripT + p] = (cont && row < nrow) ? out1[row] : 0.0;
    }
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
        const i64 c0 = t * (i64)kStripC;
#pragma unroll
        for (int q = 0; q < (kStripC / 2 + kStripT - 1) / kStripT; ++q) {
            const int j = (q * kStripT + p) * 2;
            if (j < kStripC) {
                double2 v = make_double2(0.0, 0.0), u = make_double2(0.0, 0.0);
                if (c0 + j + 1 < ncol) {
                    v = *reinterpret_cast<const double2 *>(x0 + c0 + j);
                    u = *reinterpret_cast<const double2 *>(x1 + c0 + j);
                } else if (c0 + j < ncol) {
                    v.x = x0[c0 + j];
                    u.x = x1[c0 + j];
                }
                *reinterpret_cast<double2 *>(&xt0[j]) = v;
                *reinterpret_cast<double2 *>(&xt1[j]) = u;
            }
        }
        if (p < kStripSL) offs[p] = soff[cell * kStripSL + p];
        const ushort2 r = reinterpret_cast<const ushort2 *>(perm + cell * kStripR)[p];
        const uchar2 nn = reinterpret_cast<const uchar2 *>(slen + cell * kStripR)[p];
        const unsigned int n0 = nn.x, n1 = nn.y;
        const double2 *__restrict__ v2 = reinterpret_cast<const double2 *>(val + base[cell]);
        const ushort2 *__restrict__ c2 = reinterpret_cast<const ushort2 *>(col + base[cell]);
        __syncthreads();
        double a0 = acc0[r.x], a1 = acc0[r.y], b0 = acc1[r.x], b1 = acc1[r.y];
        unsigned int s = 0;
        for (; s + 4 <= n0; s += 4) {
            const unsigned int o0 = (offs[s] >> 1) + p, o1 = (offs[s + 1] >> 1) + p, o2 = (offs[s + 2] >> 1) + p,
                               o3 = (offs[s + 3] >> 1) + p;
            const double2 w0 = v2[o0], w1 = v2[o1], w2 = v2[o2], w3 = v2[o3];
            const ushort2 j0 = c2[o0], j1 = c2[o1], j2 = c2[o2], j3 = c2[o3];
            a0 += w0.x * xt0[j0.x]; b0 += w0.x * xt1[j0.x];
            a0 += w1.x * xt0[j1.x]; b0 += w1.x * xt1[j1.x];
            a0 += w2.x * xt0[j2.x]; b0 += w2.x * xt1[j2.x];
            a0 += w3.x * xt0[j3.x]; b0 += w3.x * xt1[j3.x];
            if (s < n1) { a1 += w0.y * xt0[j0.y]; b1 += w0.y * xt1[j0.y]; }
            if (s + 1 < n1) { a1 += w1.y * xt0[j1.y]; b1 += w1.y * xt1[j1.y]; }
            if (s + 2 < n1) { a1 += w2.y * xt0[j2.y]; b1 += w2.y * xt1[j2.y]; }
            if (s + 3 < n1) { a1 += w3.y * xt0[j3.y]; b1 += w3.y * xt1[j3.y]; }
        }
        for (; s < n0; ++s) {
            const unsigned int o = (offs[s] >> 1) + p;
            const double2 w = v2[o];
            const ushort2 j = c2[o];
            a0 += w.x * xt0[j.x]; b0 += w.x * xt1[j.x];
            if (s < n1) { a1 += w.y * xt0[j.y]; b1 += w.y * xt1[j.y]; }
        }
        acc0[r.x] = a0; acc0[r.y] = a1;
        acc1[r.x] = b0; acc1[r.y] = b1;
        __syncthreads();
    }
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        if (row < nrow) {
            out0[(i64)blockIdx.y * nrow + row] = acc0[h * kStripT + p];
            out1[(i64)blockIdx.y * nrow + row] = acc1[h * kStripT + p];
        }
    }
}

// Value-dictionary variant, NV = 1 or 2 right-hand sides per pass.  An entry pair is one 8-byte load
// {id0, id1, col0, col1}; the value is dict[id] read from LDS -- the same fp64 number the CSR holds, so every
// row sum is still the sequential single-accumulator sum, bit for bit.  NV = 1: 46 KB x-tile + 16 KB sums
// + 16 KB dictionary + 1 KB offsets = 79 KB (two workgroups per CU); NV = 2: 141 KB (one per CU).
template <int NV, bool NT, bool ACC = false>
__global__ __launch_bounds__(kStripT, NV == 1 ? 8 : 4) void k_dstrip_spmv(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                                         const unsigned short *__restrict__ perm,
                                                                         const unsigned char *__restrict__ slen,
                                                                         const unsigned int *__restrict__ soff,
                                                                         const unsigned short *__restrict__ ent,
                                                                         const double *__restrict__ dict, int D,
                                                                         const double *__restrict__ x0, const double *__restrict__ x1,
                                                                         double *__restrict__ out0, double *__restrict__ out1) {
#ifdef SLP_DSTRIP_PADS   // lab: pad entries (value id D, column kDictC: -0.0 at dv[D], 1.0 at xt[kDictC]) as in k_qstrip_spmv
    __shared__ double xt[NV][kDictC + 2];
    __shared__ double acc[NV][kStripR];
    __shared__ double dv[kDictMax + 1];
#else
    __shared__ double xt[NV][kDictC];
    __shared__ double acc[NV][kStripR];
    __shared__ double dv[kDictMax];
#endif
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    const bool cont = ACC && gridDim.y == 1;  // as in k_strip_spmv
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        acc[0][h * kStripT + p] = (cont && row < nrow) ? out0[row] : 0.0;
        if (NV == 2) acc[NV - 1][h * kStripT + p] = (cont && row < nrow) ? out1[row] : 0.0;
    }
    for (int q = p; q < D; q += kStripT) dv[q] = dict[q];
#ifdef SLP_DSTRIP_PADS
    if (p == 0) {
        dv[D] = -0.0;
        xt[0][kDictC] = 1.0;
        if (NV == 2) xt[NV - 1][kDictC] = 1.0;
    }
    const unsigned long long padp_abs = (unsigned long long)((base[(i64)gridDim.x * T] + 1) >> 1);   // the all-pad pair behind the last cell
#endif
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    // Software pipeline: the x-tile and the per-strip row metadata of strip t + 1 are loaded into registers
    // while strip t's entries stream, so the tile's L2 / Infinity-Cache latency is off the critical path.
    constexpr int kTileQ = (kDictC / 2 + kStripT - 1) / kStripT;
    double2 tv[NV][kTileQ];
    ushort2 r_next = make_ushort2(0, 0);
    uchar2 nn_next = make_uchar2(0, 0);
    i64 base_next = 0;
    auto prefetch = [&](i64 t) {
        const i64 cell = b * T + t;
        const i64 c0 = t * (i64)kDictC;
        const double *__restrict__ xa = x0 + c0, *__restrict__ xb = x1 + c0;
        const int live = (int)((ncol - c0 < (i64)kDictC) ? ncol - c0 : (i64)kDictC);  // columns of this strip (uniform)
#pragma unroll
        for (int q = 0; q < kTileQ; ++q) {
            const int j = (q * kStripT + p) * 2;
            double2 v = make_double2(0.0, 0.0), u = make_double2(0.0, 0.0);
            if (j + 1 < live) {
                v = *reinterpret_cast<const double2 *>(xa + j);
                if (NV == 2) u = *reinterpret_cast<const double2 *>(xb + j);
            } else if (j < live) {
                v.x = xa[j];
                if (NV == 2) u.x = xb[j];
            }
            tv[0][q] = v;
            if (NV == 2) tv[NV - 1][q] = u;
        }
        r_next = reinterpret_cast<const ushort2 *>(perm + cell * kStripR)[p];
        nn_next = reinterpret_cast<const uchar2 *>(slen + cell * kStripR)[p];
        base_next = base[cell];
    };
    if (t_begin < t_end) prefetch(t_begin);
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
#pragma unroll
        for (int q = 0; q < kTileQ; ++q) {
            const int j = (q * kStripT + p) * 2;
            if (j < kDictC) {
                *reinterpret_cast<double2 *>(&xt[0][j]) = tv[0][q];
                if (NV == 2) *reinterpret_cast<double2 *>(&xt[NV - 1][j]) = tv[NV - 1][q];
            }
        }
        const ushort2 r = r_next;
        const unsigned int n0 = nn_next.x, n1 = nn_next.y;  // n0 >= n1 (sorted)
        const uint2 *__restrict__ e2 = reinterpret_cast<const uint2 *>(ent) + (base_next >> 1);
#ifdef SLP_DSTRIP_PADS
        const unsigned int padp = (unsigned int)(padp_abs - (unsigned long long)(base_next >> 1));
#endif
        __syncthreads();
        if (t + 1 < t_end) prefetch(t + 1);
        double a0 = acc[0][r.x], a1 = acc[0][r.y], b0 = 0.0, b1 = 0.0;
        if (NV == 2) { b0 = acc[NV - 1][r.x]; b1 = acc[NV - 1][r.y]; }
        // (Measured SLOWER here, A^T y at config 3, 2.16-2.23 ms as written: a branch-free form of this step -- all four
        // gathers issued together, selects instead of branches, as in k_qstrip_spmv -- 2.43 ms; row 0 by select under a
        // wave-uniform guard, row 1 by branch 2.28; 6 / 10 / 12 pair loads in flight instead of 8: 2.38 / 2.53 / 3.04.)
#define SLP_DSTRIP_STEP(q, live1)                                                   \
    {                                                                               \
        const double w0 = dv[(q).x & 0xffffu];                                      \
        const unsigned int ja = (q).y & 0xffffu;                                    \
        a0 += w0 * xt[0][ja];                                                       \
        if (NV == 2) b0 += w0 * xt[NV - 1][ja];                                     \
        if (live1) {                                                                \
            const double w1 = dv[(q).x >> 16];                                      \
            const unsigned int jb = (q).y >> 16;                                    \
            a1 += w1 * xt[0][jb];                                                   \
            if (NV == 2) b1 += w1 * xt[NV - 1][jb];                                 \
        }                                                                           \
    }
        // kDictU independent 8-byte loads in flight per lane; slots past the lane's count read the pair at
        // offset p (always inside the array, see strip_build) and are skipped in the sums
        // Rows are sorted by count, so a wave's first lane holds the wave's largest count: the loop is wave-uniform
        // and the slot offsets come through the scalar cache.
        constexpr int kDictU = NV == 1 ? kDictU1 : kDictU2;
        const unsigned int n0w = (unsigned int)__builtin_amdgcn_readfirstlane((int)n0);
        const unsigned int *__restrict__ so = soff + cell * kStripSL;
        for (unsigned int s = 0; s < n0w; s += kDictU) {
            uint2 q[kDictU];
            unsigned int of[kDictU];  // s + i <= 255: counts are < 256 and s is a multiple of kDictU
#pragma unroll
            for (int i = 0; i < kDictU; ++i) of[i] = so[s + i];
#ifndef SLP_DSTRIP_PADS   // dead terms skipped by branches (a lane whose rows have ended reads the pair at offset p)
#pragma unroll
            for (int i = 0; i < kDictU; ++i) q[i] = ld_stream<NT>(e2 + ((s + i < n0) ? (of[i] >> 1) + p : p));
#pragma unroll
            for (int i = 0; i < kDictU; ++i)
                if (s + i < n0) SLP_DSTRIP_STEP(q[i], s + i < n1)
#else
            // lab (round 5, -DSLP_DSTRIP_PADS): no per-lane predicate -- dead terms are pad entries worth -0.0 (the slot's padding
            // position, or the all-pad pair for a lane whose rows have ended), as in k_qstrip_spmv.  Measured on config 3, same
            // box: A^T y 2.51-2.53 ms against 2.47-2.50 for the branches (profiles/r05_c3_pairs_predicate_free_ab.log): not taken.
#pragma unroll
            for (int i = 0; i < kDictU; ++i) q[i] = ld_stream<NT>(e2 + ((s + i < n0) ? (of[i] >> 1) + p : padp));
#pragma unroll
            for (int i = 0; i < kDictU; ++i)
                if (s + i < n0w) SLP_DSTRIP_STEP(q[i], true)
#endif
        }
#undef SLP_DSTRIP_STEP
        acc[0][r.x] = a0;
        acc[0][r.y] = a1;
        if (NV == 2) { acc[NV - 1][r.x] = b0; acc[NV - 1][r.y] = b1; }
        __syncthreads();
    }
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        if (row < nrow) {
            out0[(i64)blockIdx.y * nrow + row] = acc[0][h * kStripT + p];
            if (NV == 2) out1[(i64)blockIdx.y * nrow + row] = acc[NV - 1][h * kStripT + p];
        }
    }
}

// Quad variant of the value-dictionary kernel: 4096-row blocks, lane p owns sorted positions 4p .. 4p+3,
// one 12-byte load brings slot s of all four rows (24-bit entries: 12-bit value id, 12-bit column).
// Against the pair variant: 3 instead of 4 bytes per stored entry, and half the x-tile staging per entry
// (the tile is shared by twice as many rows).  NV = 1: 31 KB x-tile + 32 KB sums + 16 KB dictionary
// (two workgroups per CU); NV = 2: 142 KB.  Same single-accumulator, storage-order row sums.
struct alignas(4) Quad12 { unsigned int x, y, z; };

typedef unsigned int u32x3_t __attribute__((ext_vector_type(3)));

// ABL != 0 exists only in the -DSLP_ABLATION build (tools/ablate_quads.py; WRONG results): 1 = no value-table lookup,
// 2 = no LDS gathers at all, 3 = no entry loads (synthetic entries from registers)
#ifndef SLP_QUAD_WAVES
#define SLP_QUAD_WAVES 8  // waves per SIMD the single-vector quad kernel is compiled for (8: two workgroups per CU, 64 VGPRs)
#endif
template <int NV, bool NT = false, int ABL = 0, bool ACC = false>
__global__ __launch_bounds__(kStripT, NV == 1 ? SLP_QUAD_WAVES : 4) void k_qstrip_spmv(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                                         const unsigned short *__restrict__ perm,
                                                                         const unsigned char *__restrict__ slen,
                                                                         const unsigned int *__restrict__ soff,
                                                                         const unsigned short *__restrict__ ent,
                                                                         const double *__restrict__ dict, int D,
                                                                         const double *__restrict__ x0, const double *__restrict__ x1,
                                                                         double *__restrict__ out0, double *__restrict__ out1) {
    // Round 5: NO predicate in the slot body.  A position that pads a slot to whole quads is a PAD ENTRY (value id D, column
    // kQuadC; k_strip_fill), and a lane whose rows have all ended loads the all-pad quad behind the last cell: with -0.0 at dv[D]
    // and 1.0 at xt[kQuadC] such an entry adds -0.0 * 1.0 = -0.0 -- and v + -0.0 == v bit for bit for every v, signed zeros
    // included -- so every sum is still the sequential storage-order sum, without the compare and the two selects per term that
    // kept the dead terms out (rounds 2-4).
    __shared__ double xt[NV][kQuadC + 2];
    __shared__ double acc[NV][kQuadR];
    __shared__ double dv[kDictMax + 1];
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    const bool cont = ACC && gridDim.y == 1;  // as in k_strip_spmv
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const i64 row = b * kQuadR + h * kStripT + p;
        acc[0][p + h * kStripT] = (cont && row < nrow) ? out0[row] : 0.0;
        if (NV == 2) acc[NV - 1][p + h * kStripT] = (cont && row < nrow) ? out1[row] : 0.0;
    }
    for (int q = p; q < D; q += kStripT) dv[q] = dict[q];
    if (p == 0) {
        dv[D] = -0.0;   // (whatever table the launch brought -- strip_spmv_with_dict swaps it -- the pad value is the kernel's own)
        xt[0][kQuadC] = 1.0;
        if (NV == 2) xt[NV - 1][kQuadC] = 1.0;
    }
    const unsigned long long padq_abs = (unsigned long long)((base[(i64)gridDim.x * T] + 3) >> 2);   // the all-pad quad behind the last cell
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    constexpr int kTileQ = (kQuadC / 2 + kStripT - 1) / kStripT;
    double2 tv[NV][kTileQ];
    ushort4 r_next = make_ushort4(0, 0, 0, 0);
    uchar4 nn_next = make_uchar4(0, 0, 0, 0);
    i64 base_next = 0;
    auto prefetch = [&](i64 t) {
        const i64 cell = b * T + t;
        const i64 c0 = t * (i64)kQuadC;
        const double *__restrict__ xa = x0 + c0, *__restrict__ xb = x1 + c0;
        const int live = (int)((ncol - c0 < (i64)kQuadC) ? ncol - c0 : (i64)kQuadC);
#pragma unroll
        for (int q = 0; q < kTileQ; ++q) {
            const int j = (q * kStripT + p) * 2;
            double2 v = make_double2(0.0, 0.0), u = make_double2(0.0, 0.0);
            if (j + 1 < live) {
                v = *reinterpret_cast<const double2 *>(xa + j);
                if (NV == 2) u = *reinterpret_cast<const double2 *>(xb + j);
            } else if (j < live) {
                v.x = xa[j];
                if (NV == 2) u.x = xb[j];
            }
            tv[0][q] = v;
            if (NV == 2) tv[NV - 1][q] = u;
        }
        r_next = reinterpret_cast<const ushort4 *>(perm + cell * kQuadR)[p];
        nn_next = reinterpret_cast<const uchar4 *>(slen + cell * kQuadR)[p];
        base_next = base[cell];
    };
    if (t_begin < t_end) prefetch(t_begin);
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
#pragma unroll
        for (int q = 0; q < kTileQ; ++q) {
            const int j = (q * kStripT + p) * 2;
            if (j < kQuadC) {
                *reinterpret_cast<double2 *>(&xt[0][j]) = tv[0][q];
                if (NV == 2) *reinterpret_cast<double2 *>(&xt[NV - 1][j]) = tv[NV - 1][q];
            }
        }
        const ushort4 r = r_next;
        const unsigned int n0 = nn_next.x;  // the longest of the lane's four rows (sorted: n0 >= n1 >= n2 >= n3)
        const Quad12 *__restrict__ e4 = reinterpret_cast<const Quad12 *>(ent) + (base_next >> 2);
        const unsigned int padq = (unsigned int)(padq_abs - (unsigned long long)(base_next >> 2));   // (relative to this cell: < 2^32 quads per copy)
        // NT: the cell's entries through a wave-uniform buffer descriptor, 12-byte loads with the non-temporal policy
        const unsigned long long e4u = (unsigned long long)e4;
        const void *e4s = (const void *)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(e4u >> 32)) << 32) |
                                         (unsigned int)__builtin_amdgcn_readfirstlane((int)e4u));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(e4s), 0, 0x7fffffff, 0x00020000);
        __syncthreads();
        if (t + 1 < t_end) prefetch(t + 1);
        double a[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v) { a[v][0] = acc[v][r.x]; a[v][1] = acc[v][r.y]; a[v][2] = acc[v][r.z]; a[v][3] = acc[v][r.w]; }
        constexpr int kU = NV == 1 ? kQuadU1 : kQuadU2;
        const unsigned int n0w = (unsigned int)__builtin_amdgcn_readfirstlane((int)n0);  // sorted: the wave's largest count
        const unsigned int *__restrict__ so = soff + cell * kStripSL;
        for (unsigned int s = 0; s < n0w; s += kU) {
            Quad12 q[kU];
            unsigned int of[kU];
#pragma unroll
            for (int i = 0; i < kU; ++i) of[i] = so[(s + i) & (kStripSL - 1)];
#pragma unroll
            for (int i = 0; i < kU; ++i) {
                const unsigned int qi = (s + i < n0) ? (of[i] >> 2) + p : padq;
                if (ABL == 3) {
                    q[i].x = qi * 2654435761u; q[i].y = q[i].x ^ (qi << 7); q[i].z = q[i].y + 0x9e3779b9u;
                    q[i].x &= 0xff7ff7ffu; q[i].y &= 0xf7ff7ff7u; q[i].z &= 0x7ff7ff7fu;  // ids < 2048, columns < 3968
                } else if (NT) {
                    // (lab, SLP_NT_QUADS=1: 12-byte loads with the non-temporal policy through a wave-uniform descriptor; a lane
                    // whose rows have ended addresses the all-pad quad through the same descriptor: 32-bit byte offsets, so only
                    // for copies below 2 GB (the descriptor's range) -- strip_spmv_one checks)
                    const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(rs, qi * 12u, 0, 2);
                    q[i].x = v.x; q[i].y = v.y; q[i].z = v.z;
                } else {
                    q[i] = e4[qi];
                }
            }
            // Branch-free, predicate-free slot body: all eight LDS gathers of a slot (value table and x-tile for the lane's four
            // rows) are issued back to back -- every address is valid -- and every term is simply added: dead terms are pad
            // entries worth -0.0.  (With a branch per term the compiler waits for each term's two gathers before it issues the
            // next pair, s_waitcnt lgkmcnt(0) after every term; rounds 2-4 predicated the accumulation with a select instead.)
#pragma unroll
            for (int i = 0; i < kU; ++i) {
                if (s + i < n0w) {  // wave-uniform: slots beyond the wave's longest row are skipped as a whole
                    const unsigned int e[4] = {q[i].x, (q[i].x >> 24) | (q[i].y << 8), (q[i].y >> 16) | (q[i].z << 16), q[i].z >> 8};
                    // two rows at a time: four gathers in flight (eight spill registers at the 64-VGPR budget)
#pragma unroll
                    for (int g = 0; g < 4; g += kQuadG) {
                        double w[kQuadG], xv[NV][kQuadG];
#pragma unroll
                        for (int h = 0; h < kQuadG; ++h) {
                            w[h] = (ABL == 1 || ABL == 2) ? (double)(e[g + h] & 0xfffu) : dv[e[g + h] & 0xfffu];
                            const unsigned int j = (e[g + h] >> 12) & 0xfffu;
                            xv[0][h] = ABL == 2 ? (double)j : xt[0][j];
                            if (NV == 2) xv[NV - 1][h] = xt[NV - 1][j];
                        }
#pragma unroll
                        for (int h = 0; h < kQuadG; ++h) {
                            a[0][g + h] = a[0][g + h] + w[h] * xv[0][h];
                            if (NV == 2) a[NV - 1][g + h] = a[NV - 1][g + h] + w[h] * xv[NV - 1][h];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) { acc[v][r.x] = a[v][0]; acc[v][r.y] = a[v][1]; acc[v][r.z] = a[v][2]; acc[v][r.w] = a[v][3]; }
        __syncthreads();
    }
    for (int h = 0; h < 4; ++h) {
        const i64 row = b * kQuadR + h * kStripT + p;
        if (row < nrow) {
            out0[(i64)blockIdx.y * nrow + row] = acc[0][h * kStripT + p];
            if (NV == 2) out1[(i64)blockIdx.y * nrow + row] = acc[NV - 1][h * kStripT + p];
        }
    }
}

// Wide strips: rows too sparse for the LDS tile (fewer than ~3 entries per 6 K columns) but long over the whole
// width -- e.g. 250-1000 entries over 10^7 columns.  The plain CSR kernel then gathers x from a vector far larger
// than an L2 (80 MB) and runs at the fabric's line rate (8-9 % of the HBM peak).  Here the matrix is cut into
// strips of kWideC columns (1 MB of x): all workgroups walk the strips in the same order, so the strip of x they
// gather from stays L2-resident, and inside a (row block, strip) cell the entries are stored as jagged diagonals
// exactly like in the LDS strips (coalesced entry stream).  Row sums keep the sequential storage order.
// DICT: 4-byte entries (value id | column << 11) + value table in LDS; else fp64 value + uint32 column.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <bool DICT, bool ACC = false>
__global__ __launch_bounds__(kStripT, 8) void k_wstrip_spmv(i64 nrow, i64 ncol, i64 T, const i64 *__restrict__ base,
                                                            const unsigned short *__restrict__ perm,
                                                            const unsigned char *__restrict__ slen,
                                                            const unsigned int *__restrict__ soff, const unsigned int *__restrict__ ent,
                                                            const double *__restrict__ val, const unsigned int *__restrict__ col,
                                                            const double *__restrict__ dict, int D, const double *__restrict__ x0,
                                                            const double *__restrict__ x1, double *__restrict__ out0,
                                                            double *__restrict__ out1) {
    // one right-hand side per pass: two strips of x would compete for the L2 (a two-vector version measured 50 ms for the
    // pair against 2 x 12.5 ms on the 2.5e6 x 1e7 slice)
    constexpr int NV = 1;
    __shared__ double acc[NV][kStripR];
    __shared__ double dv[DICT ? kDictMax : 1];
    const i64 b = blockIdx.x;
    const int p = threadIdx.x;
    const bool cont = ACC && gridDim.y == 1;  // as in k_strip_spmv
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        acc[0][h * kStripT + p] = (cont && row < nrow) ? out0[row] : 0.0;
    }
    if (DICT)
        for (int q = p; q < D; q += kStripT) dv[q] = dict[q];
    constexpr int kU = DICT ? 6 : 4;  // entry pairs in flight per lane; each brings two gathers per right-hand side
    const i64 t_begin = (T * (i64)blockIdx.y) / gridDim.y, t_end = (T * (i64)(blockIdx.y + 1)) / gridDim.y;
    for (i64 t = t_begin; t < t_end; ++t) {
        const i64 cell = b * T + t;
        const double *__restrict__ xa = x0 + t * (i64)kWideC, *__restrict__ xb = x1 + t * (i64)kWideC;
        const ushort2 r = reinterpret_cast<const ushort2 *>(perm + cell * kStripR)[p];
        const uchar2 nn = reinterpret_cast<const uchar2 *>(slen + cell * kStripR)[p];
        const unsigned int n0 = nn.x, n1 = nn.y;  // n0 >= n1 (sorted)
        const i64 e0 = base[cell] >> 1;            // pair index of the cell's first entry
        const uint2 *__restrict__ e2 = reinterpret_cast<const uint2 *>(DICT ? ent : col) + e0;
        const double2 *__restrict__ v2 = reinterpret_cast<const double2 *>(val) + e0;
        __syncthreads();  // the sums written in the previous strip (by other lanes: rows are re-sorted per cell)
        double a0 = acc[0][r.x], a1 = acc[0][r.y], b0 = 0.0, b1 = 0.0;
        if (NV == 2) { b0 = acc[NV - 1][r.x]; b1 = acc[NV - 1][r.y]; }
        const unsigned int n0w = (unsigned int)__builtin_amdgcn_readfirstlane((int)n0);  // sorted: the wave's largest count
        const unsigned int *__restrict__ so = soff + cell * kStripSL;
        for (unsigned int s = 0; s < n0w; s += kU) {
            uint2 q[kU];
            double2 w[kU];
            unsigned int of[kU];
#pragma unroll
            for (int i = 0; i < kU; ++i) of[i] = so[(s + i) & (kStripSL - 1)];
#pragma unroll
            for (int i = 0; i < kU; ++i) {
                const unsigned int o = (s + i < n0) ? (of[i] >> 1) + p : p;  // masked slots read a harmless pair
                // streamed once: non-temporal, so that the L2 keeps the strip of x
                const u32x2 qq = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(e2) + o);
                q[i] = make_uint2(qq.x, qq.y);
                if (!DICT) {
                    const f64x2 ww = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(v2) + o);
                    w[i] = make_double2(ww.x, ww.y);
                }
            }
            double g0[kU], g1[kU], h0[kU], h1[kU];
#pragma unroll
            for (int i = 0; i < kU; ++i) {  // the x gathers: L2 hits while the workgroups walk the strips together
                const unsigned int ja = (s + i < n0) ? (DICT ? q[i].x >> kWideColShift : q[i].x) : 0u;
                const unsigned int jb = (s + i < n1) ? (DICT ? q[i].y >> kWideColShift : q[i].y) : 0u;
                g0[i] = xa[ja];
                g1[i] = xa[jb];
                if (NV == 2) { h0[i] = xb[ja]; h1[i] = xb[jb]; }
            }
#pragma unroll
            for (int i = 0; i < kU; ++i) {
                if (s + i < n0) {
                    const double wa = DICT ? dv[q[i].x & ((1u << kWideColShift) - 1)] : w[i].x;
                    a0 += wa * g0[i];
                    if (NV == 2) b0 += wa * h0[i];
                    if (s + i < n1) {
                        const double wb = DICT ? dv[q[i].y & ((1u << kWideColShift) - 1)] : w[i].y;
                        a1 += wb * g1[i];
                        if (NV == 2) b1 += wb * h1[i];
                    }
                }
            }
        }
        acc[0][r.x] = a0;
        acc[0][r.y] = a1;
        if (NV == 2) { acc[NV - 1][r.x] = b0; acc[NV - 1][r.y] = b1; }
    }
    __syncthreads();
    for (int h = 0; h < 2; ++h) {
        const i64 row = b * kStripR + h * kStripT + p;
        if (row < nrow) {
            out0[(i64)blockIdx.y * nrow + row] = acc[0][h * kStripT + p];
            if (NV == 2) out1[(i64)blockIdx.y * nrow + row] = acc[NV - 1][h * kStripT + p];
        }
    }
}

// out[row] = ((part[0][row] + part[1][row]) + ...) in strip order (deterministic); accum: continuing from out[row]
__global__ void k_strip_combine(i64 nrow, int S, const double *__restrict__ part, double *__restrict__ out, int accum) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        double a = accum ? out[r] + part[r] : part[r];
        for (int s = 1; s < S; ++s) a += part[(i64)s * nrow + r];
        out[r] = a;
    }
}

__global__ void k_dict_pow(int D, const double *__restrict__ dict, double p, double *__restrict__ out) {
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < D; q += gridDim.x * blockDim.x) out[q] = abs_pow(dict[q], p) * 1.0;
}

// ---- host side ------------------------------------------------------------------
// The sorted distinct stored values of `a`, if there are at most kDictMax (SLP_VALUE_DICT=0 turns the
// variant off).  One pass over the values; gives up early on matrices with many distinct values.
bool value_dictionary(const CsrDev &a, ValueDict &d) {
    if (d.state >= 0) return d.state == 1;
    d.state = 0;
    const char *e = getenv("SLP_VALUE_DICT");
    if ((e && e[0] == '0') || a.nnz == 0) return false;
    Phase ph("value_dictionary");
    hipStream_t st = ctx().stream;
    DevBuf<unsigned long long> table((size_t)kDictHash);
    DevBuf<unsigned int> count(1);
    SLP_HIP(hipMemsetAsync(table.p, 0xff, (size_t)kDictHash * sizeof(unsigned long long), st));
    count.zero();
    hipLaunchKernelGGL(k_value_set, dim3(grid_for(a.nnz, kBlock)), dim3(kBlock), 0, st, a.nnz, a.val.p, table.p, count.p,
                       (unsigned int)kDictMax);
    SLP_HIP(hipGetLastError());
    unsigned int n = 0;
    count.download(&n, 1);
    if (n == 0 || n > (unsigned int)kDictMax) return false;
    std::vector<unsigned long long> h((size_t)kDictHash), keys;
    table.download(h.data(), h.size());
    for (unsigned long long bits : h)
        if (bits != kDictEmpty) keys.push_back(value_key(bits));
    if (keys.size() != n) return false;
    std::sort(keys.begin(), keys.end());
    std::vector<double> vals(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) {
        const unsigned long long k = keys[i], bits = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        std::memcpy(&vals[i], &bits, sizeof(double));
    }
    d.keys.upload(keys.data(), keys.size());
    d.values.upload(vals.data(), vals.size());
    d.D = (int)n;
    d.state = 1;
    return true;
}

// Builds the strip format of `a` (rows sorted by column).  Returns false (and leaves f.ok == false)
// when the matrix does not qualify: unsorted rows, or a row with >= 256 entries inside one strip.
// dict != NULL: the value-dictionary variant (narrower strips, 4-byte entries).
template <int C, int RPL, bool WIDE = false>
static bool strip_build_c(const CsrDev &a, StripJds &f, const ValueDict *dict) {
    constexpr int kStripR = RPL * kStripT;  // rows per block of this variant
    Phase ph(WIDE ? "strip_build (wide)" : (RPL == 4 ? "strip_build (quads)" : (dict ? "strip_build (pairs)" : "strip_build (fp64)")));
    hipStream_t st = ctx().stream;
    f = StripJds();
    if (a.nrow == 0 || a.nnz == 0) return false;
    const i64 T = (a.ncol + C - 1) / C, B = (a.nrow + kStripR - 1) / kStripR;
    const size_t cells = (size_t)(B * T);
    DevBuf<unsigned char> len(cells * kStripR);
    DevBuf<unsigned long long> total(cells + 1);
    DevBuf<int> bad(1);
    total.zero();
    bad.zero();
    hipLaunchKernelGGL((k_strip_count<C, RPL>), dim3((unsigned)B), dim3(kStripT), 0, st, a.nrow, T, a.ptr.p, a.idx.p, len.p, total.p, bad.p);
    SLP_HIP(hipGetLastError());
    int hbad = 0;
    bad.download(&hbad, 1);
    if (hbad) return false;
    f.base.alloc(cells + 1);
    {
        size_t bytes = 0;
        SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, total.p, (unsigned long long *)f.base.p, 0ull, cells + 1,
                                        rocprim::plus<unsigned long long>(), st));
        DevBuf<char> tmp(bytes);
        SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, total.p, (unsigned long long *)f.base.p, 0ull, cells + 1,
                                        rocprim::plus<unsigned long long>(), st));
        SLP_HIP(hipStreamSynchronize(st));
    }
    i64 padded = 0;
    SLP_HIP(hipMemcpyAsync(&padded, f.base.p + cells, sizeof(i64), hipMemcpyDeviceToHost, st));
    SLP_HIP(hipStreamSynchronize(st));
    f.perm.alloc(cells * kStripR);
    f.slen.alloc(cells * kStripR);
    f.soff.alloc(cells * kStripSL);
    if (dict) {
        // pair variant: 2 uint16 per entry; quad variant: 3 bytes per entry; + one row of pairs / quads for the
        // kernel's masked-off loads
        f.ent.alloc(RPL == 4 ? (3 * (size_t)padded + 1) / 2 + 6 * (size_t)kStripT : 2 * (size_t)padded + 4 * (size_t)kStripT);
        f.ent.zero();
        f.D = dict->D;
        f.dict = dict->values.p;
    } else {
        f.val.alloc((size_t)padded + 2 * (size_t)kStripT);              // + one row of pairs for masked-off loads (wide kernel)
        f.col.alloc((WIDE ? 2 : 1) * ((size_t)padded + 2 * (size_t)kStripT));  // wide: uint32 columns
        f.val.zero();
        f.col.zero();
    }
    f.wide = WIDE;
    hipLaunchKernelGGL((k_strip_fill<C, RPL, WIDE>), dim3((unsigned)B), dim3(kStripT), 0, st, a.nrow, T, a.ptr.p, a.idx.p, a.val.p, len.p,
                       f.base.p, f.perm.p, f.slen.p, f.soff.p, f.val.p, f.col.p, f.D, dict ? dict->keys.p : nullptr, f.ent.p);
    SLP_HIP(hipGetLastError());
    SLP_HIP(hipStreamSynchronize(st));
    f.nrow = a.nrow; f.ncol = a.ncol; f.nnz = a.nnz; f.T = T; f.B = B; f.C = C; f.rpl = RPL;
    // Few row blocks (a 1/4 or 1/8 row partition of the constraints): split every block's strips over S
    // workgroups so that the launch still fills the 256 CUs x 2 resident workgroups -- under a communicator only: the split
    // re-associates a row's sum (S partial sums added in range order), which the all-reduce of a row partition does anyway;
    // on ONE GPU every product stays the sequential CSR sum bit for bit, also for a short matrix or a short row chunk of a
    // chunked one (round 6: a 2e5-row chunk of config 3 came out with S = 8 and Chambolle-Pock lost its last bits).
    // SLP_STRIP_SPLIT=S forces a split (tests, lab).
    const char *es = getenv("SLP_STRIP_SPLIT");
    int S = es ? atoi(es) : 1;
    if (!es && a.nnz >= 30000000 && comm_active()) while (S < 8 && B * S < 384 && 2 * S <= T) S *= 2;
    if (S < 1) S = 1;
    if (S > T) S = (int)T;
    f.S = S;
    if (S > 1) f.part.alloc((size_t)S * (size_t)a.nrow);
    f.ok = true;
    return true;
}

// variant: 0 = fp64 entries, 1 = value dictionary with pairs of rows per lane, 2 = value dictionary, quads,
// 3 = wide strips (x gathered from L2; with or without a dictionary)
bool strip_build(const CsrDev &a, StripJds &f, const ValueDict *dict, int variant) {
    if (variant == 3) return strip_build_c<kWideC, 2, true>(a, f, dict);
    if (!dict) return strip_build_c<kStripC, 2>(a, f, nullptr);
    return variant == 2 ? strip_build_c<kQuadC, 4>(a, f, dict) : strip_build_c<kDictC, 2>(a, f, dict);
}

static int nt_level() {
    static const int v = [] { const char *e = getenv("SLP_NT_LOADS"); return e ? atoi(e) : SLP_NT_DEFAULT; }();
    return v;
}
static bool nt_quads() {
    static const int v = [] { const char *e = getenv("SLP_NT_QUADS"); return e ? atoi(e) : 0; }();
    return v != 0;
}
static bool nt_loads() {  // SLP_NT_LOADS=0 / 1: plain / non-temporal entry loads in the LDS-strip kernels
    static const int v = [] { const char *e = getenv("SLP_NT_LOADS"); return e ? atoi(e) : SLP_NT_DEFAULT; }();
    return v != 0;
}

// the kernels take `accum` as a template parameter (see k_strip_spmv): CALL is written in terms of a constexpr bool ACC
#define SLP_WITH_ACC(accum, CALL)                          \
    do {                                                   \
        if (accum) { constexpr bool ACC = true; CALL; }    \
        else { constexpr bool ACC = false; CALL; }         \
    } while (0)

static void wide_launch(const StripJds &f, int nv, const double *x0, const double *x1, double *o0, double *o1, int accum, int S = 0) {
    const dim3 grid((unsigned)f.B, (unsigned)(S > 0 ? S : f.S)), block(kStripT);
    hipStream_t st = ctx().stream;
    const unsigned int *ent = reinterpret_cast<const unsigned int *>(f.ent.p), *col = reinterpret_cast<const unsigned int *>(f.col.p);
#define SLP_WIDE(DICT)                                                                                                                \
    hipLaunchKernelGGL((k_wstrip_spmv<DICT, ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p, f.slen.p, f.soff.p, ent, \
                       f.val.p, col, f.dict, f.D, x0, x1, o0, o1)
    (void)nv;
    if (f.D > 0) SLP_WITH_ACC(accum, SLP_WIDE(true));
    else SLP_WITH_ACC(accum, SLP_WIDE(false));
#undef SLP_WIDE
}

int g_strip_single_chain = 0;

static void strip_combine(const StripJds &f, const double *part, double *out, int accum) {
    hipLaunchKernelGGL(k_strip_combine, dim3(grid_for(f.nrow, kBlock)), dim3(kBlock), 0, ctx().stream, f.nrow, f.S, part, out, accum);
}

// One copy (not a composite).  accum: the sums continue from what `out` holds (see k_strip_spmv).
static void strip_spmv_one(const StripJds &f, const double *x, double *out, int accum) {
    if (f.tall) {
        tall_spmv(f, x, out, accum);
        return;
    }
    // g_strip_single_chain (matrix_spmv in SLP_ORDER_SEQUENTIAL): one workgroup per row block walks ALL strips, whatever
    // strip-range split the copy was built with -- every row sum is the single chain of the CSR walk
    const int S = g_strip_single_chain > 0 ? 1 : f.S;
    if (f.wide) {
        wide_launch(f, 1, x, x, S > 1 ? f.part.p : out, nullptr, accum, S);
        if (S > 1) strip_combine(f, f.part.p, out, accum);
        SLP_HIP(hipGetLastError());
        return;
    }
    const dim3 grid((unsigned)f.B, (unsigned)S), block(kStripT);
    hipStream_t st = ctx().stream;
    double *dst = S > 1 ? f.part.p : out;
    if (f.D > 0) {
#define SLP_QLAUNCH(NT, ABL)                                                                                                      \
    hipLaunchKernelGGL((k_qstrip_spmv<1, NT, ABL, ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p, f.slen.p,    \
                       f.soff.p, f.ent.p, f.dict, f.D, x, x, dst, (double *)nullptr)
#define SLP_DLAUNCH(NT)                                                                                                           \
    hipLaunchKernelGGL((k_dstrip_spmv<1, NT, ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p, f.slen.p,         \
                       f.soff.p, f.ent.p, f.dict, f.D, x, x, dst, (double *)nullptr)
#ifdef SLP_ABLATION
        if (f.rpl == 4 && getenv("SLP_QSTRIP_ABLATE") && atoi(getenv("SLP_QSTRIP_ABLATE")) > 0) {
            const int ab = atoi(getenv("SLP_QSTRIP_ABLATE"));
            constexpr bool ACC = false;
            if (ab == 1) SLP_QLAUNCH(false, 1);
            else if (ab == 2) SLP_QLAUNCH(false, 2);
            else SLP_QLAUNCH(false, 3);
        } else
#endif
        if (f.rpl == 4 && nt_quads() && f.ent.n * sizeof(unsigned short) < ((size_t)1 << 31)) SLP_WITH_ACC(accum, SLP_QLAUNCH(true, 0));
        else if (f.rpl == 4) SLP_WITH_ACC(accum, SLP_QLAUNCH(false, 0));
        else if (nt_level() == 1) SLP_WITH_ACC(accum, SLP_DLAUNCH(true));  // 8-byte non-temporal loads measured SLOWER than plain ones (2.33 vs 2.23 ms): explicit only
        else SLP_WITH_ACC(accum, SLP_DLAUNCH(false));
#undef SLP_QLAUNCH
#undef SLP_DLAUNCH
        if (S > 1) strip_combine(f, f.part.p, out, accum);
        SLP_HIP(hipGetLastError());
        return;
    }
#define SLP_STRIP_LAUNCH(A, NTF, NT4F)                                                                                             \
    hipLaunchKernelGGL((k_strip_spmv<A, NTF, NT4F, false, ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p,       \
                       f.slen.p, f.soff.p, f.val.p, f.col.p, x, dst, 0.0)
    const bool nt = nt_loads();
#ifdef SLP_ABLATION  // `make ablation` (tools/ablate_strip.py) only: the ablated kernels return WRONG sums; not in libslp_hip.so
    const char *e = getenv("SLP_STRIP_ABLATE");
    const int ab = e ? atoi(e) : 0;
    if (ab == 1) { constexpr bool ACC = false; SLP_STRIP_LAUNCH(1, false, true); }
    else if (ab == 2) { constexpr bool ACC = false; SLP_STRIP_LAUNCH(2, false, true); }
    else
#endif
    if (nt && nt_level() == 2) SLP_WITH_ACC(accum, SLP_STRIP_LAUNCH(0, true, false));
    else if (nt) SLP_WITH_ACC(accum, SLP_STRIP_LAUNCH(0, true, true));
    else SLP_WITH_ACC(accum, SLP_STRIP_LAUNCH(0, false, true));
#undef SLP_STRIP_LAUNCH
    if (S > 1) strip_combine(f, f.part.p, out, accum);
    SLP_HIP(hipGetLastError());
}

// Composite copies (a chunked matrix, slp_chunked.hip): the row chunks' copies one after the other.  Rows orientation: chunk k
// writes its own rows of `out`.  Columns orientation (the copy of A^T: chunk k holds columns part_off[k].. of it): chunk k reads
// its slice of x and CONTINUES the sums chunk k - 1 left in `out` -- every sum is the single chain of the unchunked product.
template <class F>
static void for_parts(const StripJds &f, F call) {
    for (size_t k = 0; k < f.parts.size(); ++k) call(*f.parts[k], f.parts_cols ? f.part_off[k] : 0, f.parts_cols ? 0 : f.part_off[k],
                                                     (f.parts_cols && k > 0) ? 1 : 0);
}

// ---- products timed where they run (slp_product_timing; bench.py's roofline.timed_region) --------------------------------------
// A pair of HIP events around every single-vector product that goes through strip_spmv -- the products of the matrix-free ADMM, of
// Chambolle-Pock on strip copies and of the block projections -- on the stream it is enqueued on, while the switch is on: the
// durations of the products INSIDE the timed iterations (all launches of a product: a chunk-by-chunk composite, the combine of a
// strip-range split), read afterwards.  Nothing is recorded while it is off.
namespace {
struct ProductTimer {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs: 2 k opens product k, 2 k + 1 closes it
    size_t used = 0;
};
ProductTimer g_product_timer;
constexpr size_t kProductEventsMax = 1u << 20;
struct ProductScope {
    size_t at = (size_t)-1;
    ProductScope() {
        ProductTimer &t = g_product_timer;
        if (!t.on || t.used + 2 > kProductEventsMax) return;
        // not inside a stream capture (launch-bound LPs replay captured iterations: an event recorded there would be a graph node,
        // not a time stamp -- such products go uncounted)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx().stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
        while (t.ev.size() < t.used + 2) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            t.ev.push_back(e);
        }
        at = t.used;
        t.used += 2;
        (void)hipEventRecord(t.ev[at], ctx().stream);
    }
    ~ProductScope() {
        if (at != (size_t)-1) (void)hipEventRecord(g_product_timer.ev[at + 1], ctx().stream);
    }
};
}  // namespace
void product_timing(bool on) {
    g_product_timer.on = on;
    if (on) g_product_timer.used = 0;
}
// out[0] = products recorded since the switch went on, out[1] = the sum of their durations (ms), out[2] = the longest one (ms)
void product_timing_read(double out[3]) {
    ProductTimer &t = g_product_timer;
    out[0] = out[1] = out[2] = 0.0;
    for (size_t k = 0; k + 1 < t.used; k += 2) {
        SLP_HIP(hipEventSynchronize(t.ev[k + 1]));
        float ms = 0.f;
        SLP_HIP(hipEventElapsedTime(&ms, t.ev[k], t.ev[k + 1]));
        out[0] += 1.0;
        out[1] += (double)ms;
        if ((double)ms > out[2]) out[2] = (double)ms;
    }
}

void strip_spmv(const StripJds &f, const double *x, double *out) {
    ProductScope timed;
    if (f.parts.empty()) { strip_spmv_one(f, x, out, 0); return; }
    if (f.fused) { tall_spmv_fused(f, x, out); return; }   // all chunks' tall cells in one launch
    for_parts(f, [&](const StripJds &g, i64 xo, i64 oo, int accum) { strip_spmv_one(g, x + xo, out + oo, accum); });
}

static void strip_spmv_pow_one(const StripJds &f, double pw, const double *x, double *out, int accum) {
    if (f.ok && f.tall && f.D == 0) { tall_spmv_pow(f, pw, x, out, accum); return; }
    SLP_REQUIRE(f.ok && !f.wide && !f.tall && f.D == 0, "strip_spmv_pow: not an fp64 strip copy");
    SLP_WITH_ACC(accum, hipLaunchKernelGGL((k_strip_spmv<0, true, false, true, ACC>), dim3((unsigned)f.B, (unsigned)f.S), dim3(kStripT), 0,
                                           ctx().stream, f.nrow, f.ncol, f.T, f.base.p, f.perm.p, f.slen.p, f.soff.p, f.val.p, f.col.p, x,
                                           f.S > 1 ? f.part.p : out, pw));
    if (f.S > 1) strip_combine(f, f.part.p, out, accum);
    SLP_HIP(hipGetLastError());
}

// y = |A|^pw x over the fp64 strip copy: every row the same chain of additions as the CSR walk with |v|^pw * 1.0 terms.
void strip_spmv_pow(const StripJds &f, double pw, const double *x, double *out) {
    if (f.parts.empty()) { strip_spmv_pow_one(f, pw, x, out, 0); return; }
    for_parts(f, [&](const StripJds &g, i64 xo, i64 oo, int accum) { strip_spmv_pow_one(g, pw, x + xo, out + oo, accum); });
}

// The same product with another value table behind the same ids: the copy of the matrix whose stored values are
// table[id] instead of dict[id] (e.g. |value|^p for the Chambolle-Pock preconditioners) -- valid for dictionary copies.
static void strip_spmv_with_dict_one(const StripJds &f, const double *table, const double *x, double *out, int accum) {
    SLP_REQUIRE(f.D > 0 && table, "strip_spmv_with_dict: the copy has no value dictionary");
    StripJds &g = const_cast<StripJds &>(f);
    const double *saved = g.dict;
    g.dict = table;  // (kernel arguments are taken at launch; the launch order on the stream does the rest)
    try {
        strip_spmv_one(f, x, out, accum);
    } catch (...) {
        g.dict = saved;
        throw;
    }
    g.dict = saved;
}

void strip_spmv_with_dict(const StripJds &f, const double *table, const double *x, double *out) {
    SLP_REQUIRE(f.parts.empty(), "strip_spmv_with_dict: composite copies carry one dictionary per chunk (use strip_spmv_abs_pow)");
    strip_spmv_with_dict_one(f, table, x, out, 0);
}

// Can strip_spmv_abs_pow run on this copy?  Dictionary copies of every kind, fp64 LDS strips and fp64 tall cells (not the
// fp64 wide strips, the fallback of the fallback).
bool strip_abs_pow_supported(const StripJds &f) {
    if (!f.parts.empty()) {
        for (const StripJds *g : f.parts)
            if (!strip_abs_pow_supported(*g)) return false;
        return true;
    }
    return f.ok && (f.D > 0 || !f.wide);
}

// out = |A|^pw x: every stored value v enters as |v|^pw * 1.0 (the sums behind the Chambolle-Pock preconditioners,
// ChambollePockPPD.py:134,144,161,172) -- dictionary copies through a |v|^pw table, fp64 LDS strips on the fly; the
// same chain of additions per row as the CSR walk.
void strip_spmv_abs_pow(const StripJds &f, double pw, const double *x, double *out) {
    auto one = [&](const StripJds &g, const double *gx, double *gout, int accum) {
        if (g.D > 0) {
            DevBuf<double> table((size_t)g.D);  // (stream-ordered release: the launch below is enqueued before the block is reused)
            hipLaunchKernelGGL(k_dict_pow, dim3(8), dim3(kBlock), 0, ctx().stream, g.D, g.dict, pw, table.p);
            strip_spmv_with_dict_one(g, table.p, gx, gout, accum);
        } else {
            strip_spmv_pow_one(g, pw, gx, gout, accum);
        }
    };
    if (f.parts.empty()) { one(f, x, out, 0); return; }
    for_parts(f, [&](const StripJds &g, i64 xo, i64 oo, int accum) { one(g, x + xo, out + oo, accum); });
}

static void strip_spmv2_one(const StripJds &f, const double *x0, const double *x1, double *out0, double *out1, int accum) {
    if (f.wide || f.tall) {
        // two strips of x would compete for the L2: two passes
        strip_spmv_one(f, x0, out0, accum);
        strip_spmv_one(f, x1, out1, accum);
        return;
    }
    hipStream_t st = ctx().stream;
    double *o0 = out0, *o1 = out1;
    if (f.S > 1) {
        if (f.part2.n < 2 * (size_t)f.S * (size_t)f.nrow) f.part2.alloc(2 * (size_t)f.S * (size_t)f.nrow);
        o0 = f.part2.p;
        o1 = f.part2.p + (size_t)f.S * (size_t)f.nrow;
    }
    const dim3 grid((unsigned)f.B, (unsigned)f.S), block(kStripT);
    if (f.D > 0 && f.rpl == 4)
        SLP_WITH_ACC(accum, hipLaunchKernelGGL((k_qstrip_spmv<2, false, 0, ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p,
                                               f.slen.p, f.soff.p, f.ent.p, f.dict, f.D, x0, x1, o0, o1));
    else if (f.D > 0)
        SLP_WITH_ACC(accum, hipLaunchKernelGGL((k_dstrip_spmv<2, false, ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p,
                                               f.slen.p, f.soff.p, f.ent.p, f.dict, f.D, x0, x1, o0, o1));
    else
        SLP_WITH_ACC(accum, hipLaunchKernelGGL((k_strip_spmv2<ACC>), grid, block, 0, st, f.nrow, f.ncol, f.T, f.base.p, f.perm.p, f.slen.p,
                                               f.soff.p, f.val.p, f.col.p, x0, x1, o0, o1));
    if (f.S > 1) {
        strip_combine(f, o0, out0, accum);
        strip_combine(f, o1, out1, accum);
    }
    SLP_HIP(hipGetLastError());
}

void strip_spmv2(const StripJds &f, const double *x0, const double *x1, double *out0, double *out1) {
    if (f.parts.empty()) { strip_spmv2_one(f, x0, x1, out0, out1, 0); return; }
    for_parts(f, [&](const StripJds &g, i64 xo, i64 oo, int accum) { strip_spmv2_one(g, x0 + xo, x1 + xo, out0 + oo, out1 + oo, accum); });
}

bool strip_has_tall_split(const StripJds &f) {
    if (!f.parts.empty()) {
        for (const StripJds *g : f.parts)
            if (strip_has_tall_split(*g)) return true;
        return false;
    }
    return f.ok && f.tall && f.S > 1;
}

size_t strip_format_bytes(const StripJds &f) {
    if (!f.parts.empty()) {
        size_t b = 0;
        for (const StripJds *g : f.parts) b += strip_format_bytes(*g);
        return b;
    }
    if (f.tall) return f.tall_bytes;
    const size_t entries = f.D > 0 ? f.ent.n * sizeof(unsigned short) + (size_t)f.D * sizeof(double)
                                   : f.val.n * sizeof(double) + f.col.n * sizeof(unsigned short);
    return entries + f.perm.n * sizeof(unsigned short) + f.slen.n + f.soff.n * sizeof(unsigned int) + f.base.n * sizeof(i64);
}

// Does the format pay?  Long rows (the gather-bound regime) and enough entries per (row, strip) to amortise
// the 3 bytes of per-(row, strip) metadata and the x-tile staging.
bool strip_wanted(const CsrDev &a, int variant) {
    const char *e = getenv("SLP_STRIP_MIN_NNZ");  // below this size launch latency, not the gathers, dominates
    const i64 min_nnz = e ? atoll(e) : 30000000ll;  // measured cross-over vs the CSR kernel (tools/strip_threshold.py)
    if (a.nnz < min_nnz) return false;
    const int C = variant == 3 ? kWideC : (variant == 2 ? kQuadC : (variant == 1 ? kDictC : kStripC));
    const double per_cell = a.mean_row_len() / (double)((a.ncol + C - 1) / C);
    if (variant == 3) return a.ncol > 2 * (i64)kWideC && per_cell >= 2.0 && per_cell <= 160.0;  // x far larger than an L2
    return per_cell >= 3.0 && per_cell <= 64.0;
}

}  // namespace slp

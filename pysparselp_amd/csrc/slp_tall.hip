// slp_tall.hip -- SpMV for rows that are LONG over a width far beyond the caches but SPARSE inside any LDS-sized
// window: the per-rank slice of a 10^7-variable LP (2.5e6 x 1e7 at density 1e-4: 1000 entries per row, 0.4 per 4096
// columns) and its transpose.  Reference products: `a * x` / `y * a` of ChambollePockPPD.py:206,216,235,240 and
// ADMM.py:148,220,262 (scipy csr_matvec / csc_matvec).
//
// Why another format: the LDS strips of slp_strip.hip pay 3 bytes of metadata per (row, strip) and a strip-JDS cell per
// 2048 rows -- fine at >= 3 entries per (row, strip), hopeless at 0.4; the wide strips that replaced them there gather
// every x from L2 (one 64-byte request per entry: 2e11 requests/s, 0.10 of the HBM peak).  Here the cell is TALL: a
// workgroup owns R ~ 10^4 rows (their R running sums live in LDS, 78 KB) and walks strips of 4096 columns (the x-tile,
// double-buffered, 64 KB of LDS), so that a cell holds ~4000 entries although a row has < 1.  Only rows that HAVE
// entries in a cell appear in it:
//
//   cell (row block b, strip t)  =  "packets" of <= 1024 positions; a position is one row with 1..6 of its entries
//   packet payload (uint32 words) = [ perm: local row of every position, uint16 x cnt0 ]
//                                   [ slot 0: first entry of every position, cnt0 words ] [ slot 1: cnt1 words ] ...
//   entry word                    = value id (11 bits) | column inside the strip (12 bits) << 11
//   positions are sorted by their entry count (descending), so slot s is a prefix of the positions: lane p of the
//   workgroup reads word p of every slot it takes part in -- coalesced, no per-row lengths stored.
//
// Per stored entry: 4 bytes + 2 bytes per non-empty (row, cell) (~5.6 B at 0.4 entries per (row, strip)); the 32-byte
// packet headers are < 1 %.  A row's entries keep their column order (strips ascending, storage order inside a cell)
// and every row is accumulated by ONE thread at a time starting from its running sum: the result is the sequential
// single-accumulator CSR row sum, bit for bit, for ANY number of row blocks -- R is chosen so that the row blocks are
// a multiple of the CU count (no strips split over workgroups, no partial sums re-associated).
//
// The kernel is a software pipeline over packets: the payload of packet j + 4 and the header of packet j + 8 are
// being loaded while packet j is consumed; every global load is unconditional (lanes without work re-read word 0),
// so the compiler counts what is in flight (s_waitcnt vmcnt(N), never 0), and the x-tile of the NEXT cell arrives as
// 16 KB chunks riding on the packets of the current one.  One barrier per cell.
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

constexpr int kTallC = 4096;       // columns per strip (12-bit column inside the strip); 32 KB of x
constexpr int kTallT = 1024;       // threads per workgroup = positions per packet
constexpr int kTallSlots = 6;      // entries of one row inside one packet (longer runs continue in later packets)
constexpr int kTallDepth = 4;      // packets of payload in flight per lane; headers run 2 x this ahead
constexpr int kTallDictMax = 2048;
constexpr int kTallIdBits = 11, kTallColBits = 12, kTallRowBits = 14;
constexpr int kTallCellShift = kTallIdBits + kTallColBits + kTallRowBits;  // sort key: cell | row | column | id
static_assert(kTallRmax < (1 << kTallRowBits) && kTallC == (1 << kTallColBits) && kTallDictMax == (1 << kTallIdBits), "tall geometry");
// LDS of the product kernel: sums + value table + two x-tiles
static_assert(kTallRmax * 8 + kTallDictMax * 8 + 2 * kTallC * 8 <= 160 * 1024, "tall cells: LDS budget");

constexpr unsigned int kPktNewCell = 0x80000000u;  // first packet of a cell: barrier, then the other x-tile
constexpr unsigned int kPktBarrier = 0x40000000u;  // a row of this packet continues from an earlier packet of the cell
constexpr unsigned int kNoChunk = 0xffffffffu;

// 32-byte packet header (8 dwords; lane l & 7 of a wave loads dword l & 7)
struct TallPkt {
    unsigned int off;     // payload offset of the packet inside its row block (words)
    unsigned int flags;   // kPktNewCell | kPktBarrier
    unsigned int c01, c23, c45;  // cnt[s] = positions of the packet with more than s entries, 16 bits each (cnt[0] = positions)
    unsigned int xsrc;    // first column of the 2048-column chunk of x this packet carries for the NEXT cell, or kNoChunk
    unsigned int strip;   // strip index (diagnostics)
    unsigned int pad;
};
static_assert(sizeof(TallPkt) == 32, "packet header");

__host__ __device__ inline unsigned long long tall_value_key(unsigned long long bits) {  // as value_key() of slp_strip.hip
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}

// ---- build, step 1: one 64-bit key per stored entry, in CSR order ---------------------------------------------------
// key = cell (row block * T + strip) << 37 | local row << 23 | column inside the strip << 11 | value id.
// A stable sort by the cell bits alone then leaves every cell's entries in (row, storage) order.
__global__ __launch_bounds__(kBlock) void k_tall_keys(i64 nrow, int R, i64 T, int D, const unsigned long long *__restrict__ dkeys,
                                                      const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                      const double *__restrict__ val, unsigned long long *__restrict__ keys,
                                                      int *__restrict__ bad) {
    __shared__ unsigned long long skey[kTallDictMax];
    for (int q = threadIdx.x; q < D; q += kBlock) skey[q] = dkeys[q];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const i64 wave = ((i64)blockIdx.x * kBlock + threadIdx.x) / kWave, nwaves = (i64)gridDim.x * kBlock / kWave;
    for (i64 r = wave; r < nrow; r += nwaves) {
        const i64 s = ptr[r], e = ptr[r + 1];
        const unsigned long long b = (unsigned long long)(r / R), rl = (unsigned long long)(r % R);
        for (i64 k = s + lane; k < e; k += kWave) {
            const i32 j = idx[k];
            if (k > s && idx[k - 1] >= j) atomicOr(bad, 1);  // rows must be strictly increasing in column
            const unsigned long long key = tall_value_key((unsigned long long)__double_as_longlong(val[k]));
            int lo = 0, hi = D - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (skey[mid] < key) lo = mid + 1;
                else hi = mid;
            }
            if (skey[lo] != key) atomicOr(bad, 2);
            const unsigned long long t = (unsigned long long)(j / kTallC), cl = (unsigned long long)(j % kTallC);
            keys[k] = ((b * (unsigned long long)T + t) << kTallCellShift) | (rl << (kTallIdBits + kTallColBits)) | (cl << kTallIdBits) |
                      (unsigned long long)lo;
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
// block-wide exclusive scan of one 64-bit word per thread (three 21-bit counters packed); total in *tot
__device__ __forceinline__ unsigned long long tall_block_scan(unsigned long long v, unsigned long long *wtot, unsigned long long *tot) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += o;
    }
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

// One workgroup per row block walks its cells in strip order.  WRITE = false: sizes only (payload words, packets);
// WRITE = true: headers and payload at the offsets the host derived from the sizes.
template <bool WRITE>
__global__ __launch_bounds__(kTallT) void k_tall_build(int R, i64 T, i64 ncol, const unsigned long long *__restrict__ keys,
                                                       const i64 *__restrict__ cellptr, i64 *__restrict__ sizes,
                                                       const i64 *__restrict__ blk_base, const i64 *__restrict__ pkt_ptr,
                                                       TallPkt *__restrict__ dir, unsigned int *__restrict__ payload) {
    __shared__ unsigned int cnt[kTallRmax];       // entries of the row inside the cell
    __shared__ unsigned int rstart[kTallRmax];    // position of the row's first entry inside the cell
    __shared__ unsigned short posrow[kTallRmax];  // sorted position -> local row (one level at a time)
    __shared__ unsigned long long wtot[kTallT / kWave];
    __shared__ unsigned int smax;
    const int p = threadIdx.x;
    const i64 b = blockIdx.x;
    const int rpt = (R + kTallT - 1) / kTallT;  // consecutive rows per thread in the sorting passes
    unsigned int *pay = WRITE ? payload + blk_base[b] : nullptr;
    TallPkt *pk = WRITE ? dir + pkt_ptr[b] : nullptr;
    i64 woff = 0, npk = 0;  // running payload offset / packet count of the row block (uniform)

    auto empty_packet = [&](unsigned int xsrc, unsigned int strip) {
        if (WRITE && p == 0) {
            TallPkt h;
            h.off = (unsigned int)woff; h.flags = 0; h.c01 = h.c23 = h.c45 = 0; h.xsrc = xsrc; h.strip = strip; h.pad = 0;
            pk[npk] = h;
        }
        ++npk;
    };
    auto chunk_of = [&](i64 t, int half) -> unsigned int {  // the half-strip `half` of strip t, if it has any column
        const i64 c0 = t * (i64)kTallC + (i64)half * (kTallC / 2);
        return (t >= 0 && c0 < ncol) ? (unsigned int)c0 : kNoChunk;
    };
    auto next_cell = [&](i64 t) -> i64 {  // first strip > t with entries in this row block, or -1
        for (i64 u = t + 1; u < T; ++u)
            if (cellptr[b * T + u + 1] > cellptr[b * T + u]) return u;
        return -1;
    };

    i64 t = next_cell(-1);
    // the x-tile of the first cell rides on two leading packets without positions
    empty_packet(chunk_of(t, 0), 0);
    empty_packet(chunk_of(t, 1), 0);
    while (t >= 0) {
        const i64 tn = next_cell(t);
        const i64 c0 = cellptr[b * T + t];
        const int n = (int)(cellptr[b * T + t + 1] - c0);
        for (int r = p; r < R; r += kTallT) cnt[r] = 0;
        if (p == 0) smax = 0;
        __syncthreads();
        for (int i = p; i < n; i += kTallT) {
            const unsigned int r = (unsigned int)(keys[c0 + i] >> (kTallIdBits + kTallColBits)) & ((1u << kTallRowBits) - 1);
            const unsigned int rp = i > 0 ? (unsigned int)(keys[c0 + i - 1] >> (kTallIdBits + kTallColBits)) & ((1u << kTallRowBits) - 1) : ~0u;
            if (r != rp) rstart[r] = (unsigned int)i;
            atomicAdd(&cnt[r], 1u);
        }
        __syncthreads();
        {
            unsigned int m = 0;
            for (int r = p; r < R; r += kTallT) m = cnt[r] > m ? cnt[r] : m;
            for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_down(m, off, kWave); m = o > m ? o : m; }
            if ((p & (kWave - 1)) == 0) atomicMax(&smax, m);
        }
        __syncthreads();
        const unsigned int maxcnt = smax;
        for (unsigned int lev = 0; lev * kTallSlots < maxcnt; ++lev) {
            // positions of this level = rows with more than 6 * lev entries, sorted by the length of their piece (6 .. 1),
            // rows in increasing order inside a length
            unsigned long long ha = 0, hb = 0;  // counters of lengths 6,5,4 (21 bits each) / 3,2,1
            const int r0 = p * rpt, r1 = (r0 + rpt < R) ? r0 + rpt : R;
            for (int r = r0; r < r1; ++r) {
                const unsigned int c = cnt[r];
                if (c > lev * kTallSlots) {
                    const unsigned int len = (c - lev * kTallSlots < (unsigned)kTallSlots) ? c - lev * kTallSlots : kTallSlots;
                    if (len > 3) ha += 1ull << (21 * (kTallSlots - len));
                    else hb += 1ull << (21 * (3 - len));
                }
            }
            unsigned long long ta, tb;
            unsigned long long ea = tall_block_scan(ha, wtot, &ta), eb = tall_block_scan(hb, wtot, &tb);
            unsigned int tot[kTallSlots + 1], start[kTallSlots + 1], g[kTallSlots];  // by length
            for (int len = kTallSlots; len >= 1; --len)
                tot[len] = (unsigned int)(((len > 3 ? ta : tb) >> (21 * ((len > 3 ? kTallSlots : 3) - len))) & 0x1fffffu);
            unsigned int run = 0;
            for (int len = kTallSlots; len >= 1; --len) { start[len] = run; run += tot[len]; }
            for (int s = 0; s < kTallSlots; ++s) {  // g[s] = positions with more than s entries
                unsigned int c = 0;
                for (int len = s + 1; len <= kTallSlots; ++len) c += tot[len];
                g[s] = c;
            }
            const unsigned int npos = g[0];
            if (WRITE) {
                for (int r = r0; r < r1; ++r) {
                    const unsigned int c = cnt[r];
                    if (c > lev * kTallSlots) {
                        const unsigned int len = (c - lev * kTallSlots < (unsigned)kTallSlots) ? c - lev * kTallSlots : kTallSlots;
                        unsigned int within;
                        if (len > 3) { within = (unsigned int)((ea >> (21 * (kTallSlots - len))) & 0x1fffffu); ea += 1ull << (21 * (kTallSlots - len)); }
                        else { within = (unsigned int)((eb >> (21 * (3 - len))) & 0x1fffffu); eb += 1ull << (21 * (3 - len)); }
                        posrow[start[len] + within] = (unsigned short)r;
                    }
                }
                __syncthreads();
            }
            unsigned int npkt = (npos + kTallT - 1) / kTallT;
            if (lev == 0 && npkt < 2) npkt = 2;  // two packets at least: they carry the two halves of the next cell's x-tile
            for (unsigned int q = 0; q < npkt; ++q) {
                unsigned int c[kTallSlots];
                for (int s = 0; s < kTallSlots; ++s) {
                    const unsigned int lo = q * kTallT;
                    c[s] = g[s] > lo ? (g[s] - lo < (unsigned)kTallT ? g[s] - lo : kTallT) : 0u;
                }
                unsigned int words = (c[0] + 1) >> 1;
                if (WRITE) {
                    const unsigned int pos = q * kTallT + p;
                    if (pos < npos) {
                        const unsigned int r = posrow[pos];
                        const unsigned int len = (cnt[r] - lev * kTallSlots < (unsigned)kTallSlots) ? cnt[r] - lev * kTallSlots : kTallSlots;
                        unsigned int *base = pay + woff;
                        reinterpret_cast<unsigned short *>(base)[p] = (unsigned short)r;
                        unsigned int so = words;
                        const unsigned long long *src = keys + c0 + rstart[r] + lev * kTallSlots;
                        for (unsigned int s = 0; s < len; ++s) {
                            base[so + p] = (unsigned int)src[s] & ((1u << (kTallIdBits + kTallColBits)) - 1);
                            so += c[s];
                        }
                    }
                    if (p == 0) {
                        TallPkt h;
                        h.off = (unsigned int)woff;
                        h.flags = (q == 0) ? (lev == 0 ? kPktNewCell : kPktBarrier) : 0u;
                        h.c01 = c[0] | (c[1] << 16); h.c23 = c[2] | (c[3] << 16); h.c45 = c[4] | (c[5] << 16);
                        h.xsrc = (lev == 0 && q < 2) ? chunk_of(tn, (int)q) : kNoChunk;
                        h.strip = (unsigned int)t;
                        h.pad = 0;
                        pk[npk] = h;
                    }
                }
                for (int s = 0; s < kTallSlots; ++s) words += c[s];
                woff += words;
                ++npk;
            }
            __syncthreads();  // posrow is rewritten by the next level
        }
        t = tn;
    }
    // whole groups of 2 x depth packets, then 2 x depth more that are only ever prefetched
    while (npk % (2 * kTallDepth)) empty_packet(kNoChunk, 0);
    for (int i = 0; i < 2 * kTallDepth; ++i) empty_packet(kNoChunk, 0);
    if (!WRITE && p == 0) { sizes[2 * b] = woff; sizes[2 * b + 1] = npk; }
}

// ---- the product -------------------------------------------------------------------------------------------------------
struct TallRegs {
    unsigned int perm;
    unsigned int e[kTallSlots];
    double x0, x1;
};

__global__ __launch_bounds__(kTallT) void k_tall_spmv(i64 nrow, i64 ncol, int R, const i64 *__restrict__ pkt_ptr,
                                                      const i64 *__restrict__ blk_base, const unsigned int *__restrict__ dirw,
                                                      const unsigned int *__restrict__ payload, const double *__restrict__ dict, int D,
                                                      const double *__restrict__ x, double *__restrict__ out) {
    __shared__ double acc[kTallRmax];
    __shared__ double dv[kTallDictMax];
    __shared__ double xt[2][kTallC];
    const int p = threadIdx.x;
    const unsigned int wbase = (unsigned int)(p & ~(kWave - 1));
    const i64 b = blockIdx.x;
    for (int r = p; r < R; r += kTallT) acc[r] = 0.0;
    for (int q = p; q < D; q += kTallT) dv[q] = dict[q];
    const unsigned int *__restrict__ hd = dirw + pkt_ptr[b] * 8 + (p & 7);  // this lane's dword of every header
    const unsigned int *__restrict__ pay = payload + blk_base[b];
    const int npk = (int)(pkt_ptr[b + 1] - pkt_ptr[b]) - 2 * kTallDepth;      // the last 2 x depth packets are prefetch targets only
    const i64 xmax = ncol - 1;
    int cur = 0;

    TallRegs regs[kTallDepth];
    unsigned int hw[2 * kTallDepth];

    auto issue = [&](TallRegs &g, unsigned int h) {
        const unsigned int off = (unsigned int)__builtin_amdgcn_readlane((int)h, 0);
        const unsigned int c01 = (unsigned int)__builtin_amdgcn_readlane((int)h, 2), c23 = (unsigned int)__builtin_amdgcn_readlane((int)h, 3),
                           c45 = (unsigned int)__builtin_amdgcn_readlane((int)h, 4);
        const unsigned int xsrc = (unsigned int)__builtin_amdgcn_readlane((int)h, 5);
        const unsigned int c[kTallSlots] = {c01 & 0xffffu, c01 >> 16, c23 & 0xffffu, c23 >> 16, c45 & 0xffffu, c45 >> 16};
        const unsigned int *__restrict__ w = pay + off;
        g.perm = reinterpret_cast<const unsigned short *>(w)[(unsigned)p < c[0] ? p : 0];
        unsigned int so = (c[0] + 1) >> 1;
#pragma unroll
        for (int s = 0; s < kTallSlots; ++s) {
            g.e[s] = __builtin_nontemporal_load(w + so + ((unsigned)p < c[s] ? p : 0));  // streamed once
            so += c[s];
        }
        i64 j = (xsrc == kNoChunk) ? 0 : (i64)xsrc + 2 * p;
        const i64 j0 = j < xmax ? j : xmax, j1 = j + 1 < xmax ? j + 1 : xmax;
        g.x0 = x[j0];
        g.x1 = x[j1];
    };

    auto consume = [&](const TallRegs &g, unsigned int h) {
        const unsigned int flags = (unsigned int)__builtin_amdgcn_readlane((int)h, 1);
        const unsigned int c01 = (unsigned int)__builtin_amdgcn_readlane((int)h, 2), c23 = (unsigned int)__builtin_amdgcn_readlane((int)h, 3),
                           c45 = (unsigned int)__builtin_amdgcn_readlane((int)h, 4);
        const unsigned int xsrc = (unsigned int)__builtin_amdgcn_readlane((int)h, 5);
        const unsigned int c[kTallSlots] = {c01 & 0xffffu, c01 >> 16, c23 & 0xffffu, c23 >> 16, c45 & 0xffffu, c45 >> 16};
        if (flags & (kPktNewCell | kPktBarrier)) {
            // sums of the previous cell (written by other lanes) and the x-tile chunks: LDS only, the loads stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (flags & kPktNewCell) cur ^= 1;
        }
        if (xsrc != kNoChunk) {
            double2 v = make_double2(g.x0, g.x1);
            *reinterpret_cast<double2 *>(&xt[cur ^ 1][(xsrc & (kTallC - 1)) + 2 * p]) = v;
        }
        if (wbase < c[0]) {
            const bool live = (unsigned)p < c[0];
            const unsigned int row = live ? g.perm : 0u;
            double a = acc[row];
            const double *__restrict__ tile = xt[cur];
#define SLP_TALL_SLOT(S)                                                                 \
    {                                                                                    \
        const unsigned int e = g.e[S];                                                   \
        const double t = a + dv[e & ((1u << kTallIdBits) - 1)] * tile[(e >> kTallIdBits) & (kTallC - 1)]; \
        a = ((unsigned)p < c[S]) ? t : a;                                                \
    }
            SLP_TALL_SLOT(0)
            if (wbase < c[1]) {
                SLP_TALL_SLOT(1)
                if (wbase < c[2]) {
                    SLP_TALL_SLOT(2)
                    if (wbase < c[3]) {
                        SLP_TALL_SLOT(3)
                        if (wbase < c[4]) {
                            SLP_TALL_SLOT(4)
                            if (wbase < c[5]) SLP_TALL_SLOT(5)
                        }
                    }
                }
            }
#undef SLP_TALL_SLOT
            if (live) acc[row] = a;
        }
    };

    // prologue: headers of the first 2 x depth packets, payload of the first depth
#pragma unroll
    for (int u = 0; u < 2 * kTallDepth; ++u) hw[u] = hd[(i64)u * 8];
#pragma unroll
    for (int u = 0; u < kTallDepth; ++u) issue(regs[u], hw[u]);
    __syncthreads();
    for (int jj = 0; jj < npk; jj += 2 * kTallDepth) {
#pragma unroll
        for (int u = 0; u < 2 * kTallDepth; ++u) {
            consume(regs[u % kTallDepth], hw[u]);                                   // packet jj + u
            issue(regs[u % kTallDepth], hw[(u + kTallDepth) % (2 * kTallDepth)]);   // payload of packet jj + u + depth
            hw[u] = hd[(i64)(jj + u + 2 * kTallDepth) * 8];                         // header of packet jj + u + 2 depth
        }
    }
    __syncthreads();
    for (int r = p; r < R; r += kTallT) {
        const i64 row = b * (i64)R + r;
        if (row < nrow) out[row] = acc[r];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Rows per block: as tall as the LDS allows, and such that the blocks are (nearly) a multiple of the CU count -- every
// CU then walks the same number of row blocks and no strip is ever split over workgroups.
static int tall_rows_per_block(i64 nrow) {
    const char *e = getenv("SLP_TALL_R");
    if (e && atoi(e) > 0) return std::min(atoi(e), kTallRmax);
    const i64 cus = ctx().num_cu;
    i64 k = (nrow + cus * (i64)kTallRmax - 1) / (cus * (i64)kTallRmax);
    if (k < 1) k = 1;
    i64 R = (nrow + k * cus - 1) / (k * cus);
    if (R < 1024) R = std::min<i64>(1024, std::max<i64>(nrow, 1));  // small matrices: fewer, still tall blocks
    return (int)std::min<i64>(R, kTallRmax);
}

bool tall_build(const CsrDev &a, StripJds &f, const ValueDict *dict) {
    Phase ph("tall_build");
    hipStream_t st = ctx().stream;
    f = StripJds();
    if (!dict || dict->D <= 0 || dict->D > kTallDictMax || a.nrow == 0 || a.nnz == 0) return false;
    const int R = tall_rows_per_block(a.nrow);
    const i64 T = (a.ncol + kTallC - 1) / kTallC, B = (a.nrow + R - 1) / R, ncell = B * T;
    unsigned int cellbits = 1;
    while (((i64)1 << cellbits) < ncell) ++cellbits;
    if (kTallCellShift + cellbits > 64) return false;
    DevBuf<i64> cellptr((size_t)ncell + 1);
    DevBuf<unsigned long long> sorted((size_t)a.nnz);
    {
        DevBuf<unsigned long long> keys((size_t)a.nnz);
        DevBuf<int> bad(1);
        bad.zero();
        hipLaunchKernelGGL(k_tall_keys, dim3(grid_for(a.nrow * kWave, kBlock)), dim3(kBlock), 0, st, a.nrow, R, T, dict->D, dict->keys.p,
                           a.ptr.p, a.idx.p, a.val.p, keys.p, bad.p);
        SLP_HIP(hipGetLastError());
        int hbad = 0;
        bad.download(&hbad, 1);
        if (hbad) return false;  // unsorted rows (or a value outside the dictionary)
        size_t bytes = 0;
        SLP_HIP(rocprim::radix_sort_keys(nullptr, bytes, keys.p, sorted.p, (size_t)a.nnz, (unsigned)kTallCellShift,
                                         (unsigned)kTallCellShift + cellbits, st));
        DevBuf<char> tmp(bytes);
        SLP_HIP(rocprim::radix_sort_keys(tmp.p, bytes, keys.p, sorted.p, (size_t)a.nnz, (unsigned)kTallCellShift,
                                         (unsigned)kTallCellShift + cellbits, st));
        SLP_HIP(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_tall_cellptr, dim3(grid_for(a.nnz, kBlock)), dim3(kBlock), 0, st, a.nnz, ncell, sorted.p, cellptr.p);
    SLP_HIP(hipGetLastError());
    DevBuf<i64> sizes(2 * (size_t)B);
    hipLaunchKernelGGL((k_tall_build<false>), dim3((unsigned)B), dim3(kTallT), 0, st, R, T, a.ncol, sorted.p, cellptr.p, sizes.p,
                       (const i64 *)nullptr, (const i64 *)nullptr, (TallPkt *)nullptr, (unsigned int *)nullptr);
    SLP_HIP(hipGetLastError());
    std::vector<i64> hs(2 * (size_t)B), hbase((size_t)B + 1), hpkt((size_t)B + 1);
    sizes.download(hs.data(), hs.size());
    hbase[0] = hpkt[0] = 0;
    for (i64 b = 0; b < B; ++b) {
        SLP_REQUIRE(hs[2 * b] < (i64)0xffffffffll, "tall cells: a row block's payload exceeds 2^32 words");
        hbase[b + 1] = hbase[b] + ((hs[2 * b] + 3) & ~(i64)3);  // 16-byte aligned row blocks
        hpkt[b + 1] = hpkt[b] + hs[2 * b + 1];
    }
    f.tall_base.upload(hbase.data(), hbase.size());
    f.tall_pkt.upload(hpkt.data(), hpkt.size());
    f.tall_dir.alloc((size_t)hpkt[B] * 8);
    f.tall_pay.alloc((size_t)hbase[B] + 2 * (size_t)kTallT * (kTallSlots + 1));  // + room for the clamped loads of empty packets
    f.tall_pay.zero();
    hipLaunchKernelGGL((k_tall_build<true>), dim3((unsigned)B), dim3(kTallT), 0, st, R, T, a.ncol, sorted.p, cellptr.p, (i64 *)nullptr,
                       f.tall_base.p, f.tall_pkt.p, reinterpret_cast<TallPkt *>(f.tall_dir.p), f.tall_pay.p);
    SLP_HIP(hipGetLastError());
    SLP_HIP(hipStreamSynchronize(st));
    f.nrow = a.nrow; f.ncol = a.ncol; f.nnz = a.nnz; f.T = T; f.B = B; f.C = kTallC; f.rpl = 1;
    f.D = dict->D;
    f.dict = dict->values.p;
    f.tall = true;
    f.tall_R = R;
    f.S = 1;
    f.ok = true;
    return true;
}

void tall_spmv(const StripJds &f, const double *x, double *out) {
    hipLaunchKernelGGL(k_tall_spmv, dim3((unsigned)f.B), dim3(kTallT), 0, ctx().stream, f.nrow, f.ncol, f.tall_R, f.tall_pkt.p,
                       f.tall_base.p, f.tall_dir.p, f.tall_pay.p, f.dict, f.D, x, out);
    SLP_HIP(hipGetLastError());
}

// Long rows over a width far beyond an L2, too sparse for the LDS strips: 0.05 .. 2.5 entries per (row, 4096 columns).
bool tall_wanted(const CsrDev &a) {
    const char *e = getenv("SLP_TALL");
    if (e && e[0] == '0') return false;
    const char *m = getenv("SLP_STRIP_MIN_NNZ");
    const i64 min_nnz = m ? atoll(m) : 30000000ll;
    if (a.nnz < min_nnz || a.nrow <= 0) return false;
    const double per_cell = a.mean_row_len() / (double)((a.ncol + kTallC - 1) / kTallC);
    return per_cell >= 0.05 && per_cell < 2.5;
}

}  // namespace slp

// slp_admm.hip -- projected Gauss-Seidel sweep and the ADMM loop on the device.
// Replaces gaussSiedel.pyx:83-153 (boundedGaussSeidelClass) -- the reference's
// only native code on the hot path -- and the loop of lp_admm (ADMM.py:143-268).
//
// The reference sweep is sequential: row i uses x[j] already updated for
// j < i and not yet updated for j > i.  That order is kept exactly by a level
// schedule: level(i) = 1 + max level(j) over j < i coupled to i through M or
// M^T; rows of one level touch disjoint unknowns and run in parallel, levels
// run one kernel after the other on the stream.  Every row is summed by one
// thread in storage order, so x is bit-identical to the sequential sweep.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "slp_common.h"
#include "slp_kernels.h"

namespace slp {

// Row-dot of one Gauss-Seidel row, gaussSiedel.pyx:139-141: v = sum_k x[idx[k]] * val[k] in storage order with
// one accumulator.  The loads of 16 entries are issued together (a level is a latency chain: pointer -> index ->
// x gather -> add), the adds stay sequential.
__device__ __forceinline__ double gs_row_dot(i64 s, i64 e, const i32 *__restrict__ idx, const double *__restrict__ val,
                                             const double *x) {
    double v = 0.0;
    for (i64 k = s; k < e; k += 16) {
        i32 j[16];
        double a[16], xv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const i64 kk = (k + q < e) ? k + q : e - 1;
            j[q] = idx[kk];
            a[q] = val[kk];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) xv[q] = x[j[q]];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (k + q < e) v += xv[q] * a[q];
    }
    return v;
}

// One level of one sweep.  The matrix is stored in LEVEL ORDER: position t (first..first+count) holds row rows[t];
// its entries are ptr[t]..ptr[t+1], so pointer and entry loads do not wait for the row id.
// BOUNDED: boundedGaussSeidelClass.solve (:131-152).  !BOUNDED: GaussSeidel (:58-69), the plain SOR update
// nv = (b - v + D x_i) / D ; x_i = w nv + (1 - w) x_i, no clamp.
template <bool BOUNDED>
__global__ __launch_bounds__(kBlock) void k_gs_level(i64 first, i64 count, const i32 *__restrict__ rows, const i64 *__restrict__ ptr,
                                                     const i32 *__restrict__ idx, const double *__restrict__ val,
                                                     const double *__restrict__ invd, const double *__restrict__ b,
                                                     const double *__restrict__ lo, const double *__restrict__ hi,
                                                     double *__restrict__ x, double w) {
    const i64 t = (i64)blockIdx.x * kBlock + threadIdx.x;
    if (t >= count) return;
    const i64 pos = first + t;
    const i32 i = rows[pos];
    const i64 s = ptr[pos], e = ptr[pos + 1];
    const double bi = b[i], inv = invd[pos], xi = x[i];  // independent of the row walk
    double v = gs_row_dot(s, e, idx, val, x);
    if (BOUNDED) {
        const double l = lo[i], u = hi[i];
        v = w * (bi - v) * inv + xi;                                     // :145
        if (v < l) v = l;                                                // :148-151
        else if (v > u) v = u;
    } else {
        const double d = lo[pos];                                        // diagonal, level-ordered (passed in `lo`)
        const double nv = (bi - v + d * xi) * inv;                       // :67
        v = w * nv + (1 - w) * xi;                                       // :68
    }
    x[i] = v;
}

// Small systems: the whole sweep inside ONE workgroup, levels separated by a
// workgroup barrier instead of a kernel boundary (launch latency dominates
// otherwise: SC105 has 41 levels of ~4 rows).
template <bool BOUNDED>
__global__ __launch_bounds__(1024) void k_gs_sweep_one_block(i64 nlevels, const i64 *__restrict__ lptr,
                                                             const i32 *__restrict__ rows, const i64 *__restrict__ ptr,
                                                             const i32 *__restrict__ idx, const double *__restrict__ val,
                                                             const double *__restrict__ invd, const double *__restrict__ b,
                                                             const double *__restrict__ lo, const double *__restrict__ hi,
                                                             double *__restrict__ x, double w, int sweeps) {
    for (int s = 0; s < sweeps; ++s) {
        for (i64 l = 0; l < nlevels; ++l) {
            const i64 beg = lptr[l], end = lptr[l + 1];
            for (i64 t = beg + threadIdx.x; t < end; t += blockDim.x) {
                const i32 i = rows[t];
                double v = 0.0;
                for (i64 k = ptr[t]; k < ptr[t + 1]; ++k) v += __hip_atomic_load(&x[idx[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * val[k];
                if (BOUNDED) {
                    v = w * (b[i] - v) * invd[t] + x[i];
                    const double lw = lo[i], u = hi[i];
                    if (v < lw) v = lw;
                    else if (v > u) v = u;
                } else {
                    const double xi = x[i];
                    const double nv = (b[i] - v + lo[t] * xi) * invd[t];  // lo = level-ordered diagonal
                    v = w * nv + (1 - w) * xi;
                }
                x[i] = v;
            }
            __syncthreads();  // same CU: stores of this level are visible to the next one
        }
    }
}

// Single-workgroup sweep for systems with many narrow dependency levels (image grids: Potts 256^2 has 512
// levels of ~900 rows).  One launch per level costs a kernel boundary plus three dependent memory hops
// (row pointer -> entries -> x) per level; here a level costs one workgroup barrier and ONE hop on the
// critical path: the x gather.  The other hops are taken ahead of time, one per step, into a register
// ring: step s issues (1) the x gathers of step s, (2) the entry / right-hand side / bound loads of step
// s + 1 from the lane-slot record that arrived a step earlier, (3) the slot record of step s + 2.  (A wave's
// loads return in order, so a deeper ring would not hide more: the wait for step s + 1's gathers also waits
// for everything issued before them.)  All loads are unconditional (idle lanes and short rows read harmless
// addresses) so the loop is straight-line code.
// A step is up to 1024 lane slots of one level.  A row takes ceil(len / kGsEntries) neighbouring lanes of
// one wave; lane j gathers x for entries [j E, (j+1) E), then the row's single accumulator travels from
// lane to lane (__shfl_up) in entry order: the same sequential, storage-order sum as in k_gs_level, bit for
// bit, while no lane holds more than E entries.
constexpr int kGsEntries = 4;  // entries of a row per lane
constexpr int kGsMaxSeg = 16;  // lanes per row at most (longer rows: the per-level kernels are used instead)
constexpr int kGsRing = 3;

// lane slots [first, first + count), chain rounds, barrier: bit 0 = barrier after the step, bit 1 = the step has entries of class
// "far" (windowed kernel)
struct GsStep { int first, count, maxseg, barrier; };
// lane slot: row id, row position in level order, seg | lanes << 8 | entries of this lane << 16 | idle << 24, and where the
// windowed kernel keeps the row's new x in its LDS ring (-1: nowhere)
struct alignas(16) GsSlot { i32 row; i32 t; int info; int pad; };
// the lane's entries, stored per lane slot so that a step streams them with three 16-byte loads per lane
struct alignas(16) GsEnt { i32 idx[kGsEntries]; double val[kGsEntries]; };
// per row position, packed before every sweep (k_gs_pack): right-hand side, bounds and the row's own x
struct alignas(16) GsRow { double b, lo, hi, xi; };

template <bool BOUNDED>
__global__ void k_gs_pack(i64 n, const i32 *__restrict__ rows, const double *__restrict__ b, const double *__restrict__ lo,
                          const double *__restrict__ hi, const double *__restrict__ x, GsRow *__restrict__ out) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i32 i = rows[t];
        GsRow r;
        r.b = b[i];
        r.lo = BOUNDED ? lo[i] : lo[t];  // unbounded: the level-ordered diagonal
        r.hi = BOUNDED ? hi[i] : 0.0;
        r.xi = x[i];                      // only the row's own update writes x[i]
        out[t] = r;
    }
}

template <bool BOUNDED>
__global__ __launch_bounds__(1024) void k_gs_sweep_pipelined(int nsteps, const GsStep *__restrict__ steps, const GsSlot *__restrict__ slots,
                                                             const GsEnt *__restrict__ ents, const double *__restrict__ invd,
                                                             const GsRow *__restrict__ packed, double *__restrict__ x, double w) {
    struct Stage {
        GsSlot sl;
        int live;  // kept apart from the loaded record: touching `sl` right after its load would wait for it
        GsEnt en;
        GsRow rw;
        double invd;
    };
    Stage st[kGsRing];
    const int tid = threadIdx.x;
    auto load_slot = [&](Stage &g, int s) {  // hop 1: slot record and (addressed by the slot index) the lane's entries
        const int sc = s < nsteps ? s : nsteps - 1;  // past the end: a harmless reload, marked dead
        const GsStep sd = steps[sc];
        g.live = (s < nsteps && tid < sd.count) ? 1 : 0;
        const i64 i = (i64)sd.first + (tid < sd.count ? tid : 0);
        g.sl = slots[i];
        g.en = ents[i];
    };
    auto load_data = [&](Stage &g) {  // hop 2: addressed by the row position
        g.rw = packed[g.sl.t];
        g.invd = invd[g.sl.t];
    };
    auto process = [&](const Stage &g, int s, Stage &g2, Stage &g3) {
        const GsStep sd = steps[s];
        double xg[kGsEntries];
#pragma unroll
        for (int e = 0; e < kGsEntries; ++e) xg[e] = __hip_atomic_load(&x[g.en.idx[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        load_data(g2);            // step s + 1
        load_slot(g3, s + 2);     // step s + 2
        const int seg = (!g.live || (g.sl.info >> 24)) ? -1 : (g.sl.info & 0xff);
        const int nlane = (g.sl.info >> 8) & 0xff, len = (g.sl.info >> 16) & 0xff;
        double v = 0.0, carry = 0.0;
        for (int r = 0; r < sd.maxseg; ++r) {
            if (seg == r) {
                v = carry;
#pragma unroll
                for (int e = 0; e < kGsEntries; ++e)
                    if (e < len) v += xg[e] * g.en.val[e];
            }
            const double up = __shfl_up(v, 1);
            if (seg == r + 1) carry = up;
        }
        if (seg >= 0 && seg == nlane - 1) {
            if (BOUNDED) {
                v = w * (g.rw.b - v) * g.invd + g.rw.xi;
                if (v < g.rw.lo) v = g.rw.lo;
                else if (v > g.rw.hi) v = g.rw.hi;
            } else {
                const double nv = (g.rw.b - v + g.rw.lo * g.rw.xi) * g.invd;
                v = w * nv + (1 - w) * g.rw.xi;
            }
            x[g.sl.row] = v;
        }
        if (sd.barrier & 1) __syncthreads();  // last step of a level: the next level reads these x
    };
    load_slot(st[0], 0);
    load_slot(st[1], 1);
    load_data(st[0]);
    for (int s = 0; s < nsteps; s += kGsRing) {
#pragma unroll
        for (int j = 0; j < kGsRing; ++j)
            if (s + j < nsteps) process(st[j], s + j, st[(j + 1) % kGsRing], st[(j + 2) % kGsRing]);
    }
}

// ---- the windowed form of the single-workgroup sweep ---------------------------------------------------------------------
// k_gs_sweep_pipelined is bound by the instruction issue of ONE compute unit (PMC on the Potts 256^2 system: the four SIMDs
// issue in 85 % of the cycles): every wave runs every step whether it has lanes in it or not, every entry costs a 64-byte
// gather request plus predication, and the chain runs the step's longest row in every wave.  Here:
//  * the unit of work is a WAVE SLOT (64 lane slots).  A level's wave slots are dealt to the 16 waves; each wave walks its own
//    list of headers (GsStepW: where its 64 lane slots are, chain rounds, barrier after it, has "far" entries) and a wave
//    with nothing to do in a level only joins the level's barrier;
//  * each entry of a lane slot is classed when the plan is built, by where the value it reads comes from:
//      static  the source row is final before this kernel starts (an earlier segment) or is not touched before the reader's
//              own step (the row itself and rows of later levels): x[source] * value is formed by k_gs_pack_terms, chip-wide,
//              just before the kernel -- the product is rounded once wherever it is formed, and the sum keeps its order;
//      near    the source row is updated by this kernel in one of the last kGsWinLevels - 1 levels: its new x is read from an
//              LDS ring of kGsWinLevels levels (each row writes its result to x and to the ring);
//      far     updated by this kernel longer ago: a gather from x (none on grid problems; wave slots without any skip it);
//    the body has no class test: a term is ring[code] * dyn, where static entries and the padding of short lanes point at a
//    ring cell that holds 1.0 (y * 1.0 == y bit for bit) and padding terms are -0.0 (v + -0.0 == v bit for bit);
//  * nothing that is waited for was issued less than two wave slots earlier: headers and lane records are fetched 4 ahead,
//    terms 2 ahead, the row record (last lane of a row only) 2 ahead.  Headers are read with vector loads (an s_load would
//    share lgkmcnt with the LDS reads); the accumulator goes from lane to lane by DPP (wave_shr:1), not the LDS crossbar.
// Streams: 8 B per wave slot, 16 B + 32 B per lane slot (GsLane + position, GsDyn), 40 B per row (GsRowW).
// One compute unit reads 60 (HBM) / 110 (MALL) / 150 (L2) GB/s (tools/lab/one_cu_stream.cpp): with the loads out of the way
// of the arithmetic (a build without them runs the Potts 256^2 sweep 9 x faster) the bytes ARE the time.
constexpr int kGsWinLevels = 4;
constexpr int kGsWide = 4096;        // rows of a level that fit one ring slot
constexpr int kGsWaves = 16;
constexpr int kGsOne = kGsWinLevels * kGsWide;  // the ring cell that holds 1.0
static_assert(kGsOne < 0xffff, "ring indices are 16 bits");
// code: ring index of the value to multiply with (kGsOne: none).  info: lane of the row | last lane of a row << 4 |
// entries of this lane << 5 | mask of far entries << 8.  ldsw: where the row's new x goes in the ring (0xffff: nowhere).
// The row position of a lane slot (both the result it writes and its GsRowW record are addressed by it) is kept in an array
// of its own, read 4 wave slots ahead like the lane record: as a member of the record, the compiler moves it out of the loaded
// tuple right behind the load, which waits for it.  Results go to `xpos` in position order (coalesced; k_gs_unpack scatters
// them to x after the kernel), row records are read in position order (a stream: in row order every 40-byte record would
// cost a 128-byte line).
struct alignas(4) GsLane { unsigned short code[kGsEntries]; unsigned short info, ldsw; };
struct alignas(16) GsDyn { double v[kGsEntries]; };     // static: x[source] * value; near / far: value; padding: -0.0
struct alignas(8) GsRowW { double b, lo, hi, xi, invd; };  // per row position
struct alignas(8) GsStepW { int first; unsigned meta; };  // meta: active | chain rounds << 1 | barrier << 6 | has far << 7 | (bands) levels stored << 8

// ---- bands: a run of narrow levels on several compute units -------------------------------------------------------------------
// The dependency chain of a run cannot be shortened, but its bytes (lane, term and row records: 44 MB per Potts 256^2 sweep through
// one CU's load path) can be split: the rows of the run are cut into P index ranges ("bands"), one workgroup each, every band with
// level numbers of its own (1 + the deepest lower neighbour INSIDE the band).  A row's lower neighbours in lower bands are
// "external": a fetch wave (wave 0 of the workgroup, no rows of its own) reads them from the results of the lower band's
// workgroup kGsFetchD levels before they are needed and puts them into LDS cells next to the ring, where the compute waves pick
// them up like any near value -- so the compute waves never wait for another compute unit.  A band publishes how many of its
// levels are stored (`prog`, one counter per band); the fetch wave of a higher band checks the counters it needs (per level, one
// number per lower band, cumulative) before it issues the reads, and spins only when the lower band is not far enough ahead.
// Lower bands never wait for higher ones (higher neighbours are read as the OLD x, which k_gs_pack_terms has already folded into
// the terms), so the workgroups cannot deadlock whatever order they are dispatched in.  On a grid-like matrix band p starts
// (its rows' depth in band p - 1) + kGsFetchD levels behind band p - 1 and then keeps pace.  Every row's arithmetic is the
// single-workgroup kernel's: the results are bit for bit the sequential sweep's.
//
// "Stored" without stalling the pipeline: a wave's vector-memory operations complete in issue order, so once the lane record of
// wave slot k (issued in wave slot k - 4, behind a scheduling barrier) has arrived, the result store of wave slot k - 6 and all
// before it have left the wave and reached the L2 (they are device-scope stores: written through).  The plan puts into header k
// the number of levels whose stores that covers; the wave copies it to LDS, the fetch wave publishes the minimum over the waves.
constexpr int kGsFetchK = 2;                       // external values per level of a band: at most kGsFetchK * 64 (else no bands)
#ifndef SLP_GS_FETCH_D
#define SLP_GS_FETCH_D 4
#endif
constexpr int kGsFetchD = SLP_GS_FETCH_D;          // levels between the read of an external value and its use
constexpr int kGsMaxBands = 16;
constexpr int kGsBandWaves = 15;                   // compute waves of a band's workgroup (+ the fetch wave: 1024 threads)
constexpr int kGsExt = kGsOne + 1 + 64;            // first LDS cell of the external values (two generations: level parity)
constexpr int kGsWinCells = kGsExt + 2 * kGsFetchK * 64;
constexpr int kGsProgStride = 32;                  // ints between two bands' counters (a 128-byte line each)
constexpr int kGsDone = 0x7fffffff;
#ifndef SLP_GS_XSCOPE
#define SLP_GS_XSCOPE __HIP_MEMORY_SCOPE_AGENT   // results another band reads, and the reads
#endif
#ifndef SLP_GS_FSCOPE
#define SLP_GS_FSCOPE __HIP_MEMORY_SCOPE_AGENT   // the bands' counters
#endif
static_assert(kGsWinCells < 0xffff, "ring indices are 16 bits");
// per band: first of its waves + 1 header offsets, compute waves, levels, first fetch position (per level kGsFetchK * 64, padded
// by 3 kGsFetchD levels), first requirement row (row 0: before the first level; row l + 1: checked in level l for the reads of
// level l + kGsFetchD; 16 ints each, padded by 2 kGsFetchD rows), first scratch cell
struct GsBand { int hoff, waves, nlev, fsrc, freq, scratch, pad0, pad1; };

// right-hand side, bounds and own x of every row, by position (x[row] only changes in the row's own step)
template <bool BOUNDED>
__device__ __forceinline__ void gs_pack_rows(i64 n, const i32 *__restrict__ rows, const double *__restrict__ b, const double *__restrict__ lo,
                                             const double *__restrict__ hi, const double *__restrict__ x, const double *__restrict__ invd,
                                             GsRowW *__restrict__ out) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i32 i = rows[t];
        GsRowW r;
        r.b = b[i];
        r.lo = BOUNDED ? lo[i] : lo[t];  // unbounded: the level-ordered diagonal
        r.hi = BOUNDED ? hi[i] : 0.0;
        r.xi = x[i];
        r.invd = invd[t];
        out[t] = r;
    }
}

// ROWS != 0: the sweep's first windowed segment also packs the row records of ALL rows (a launch of its own before; a row's
// record holds the row's own old x, which no earlier segment of the sweep touches): 1 bounded, 2 plain
template <int ROWS>
__global__ void k_gs_pack_terms(i64 first, i64 count, const GsEnt *__restrict__ ents, const GsLane *__restrict__ lanes,
                                const double *__restrict__ x, GsDyn *__restrict__ dyn, int *__restrict__ prog, int nprog, i64 n,
                                const i32 *__restrict__ rows, const double *__restrict__ b, const double *__restrict__ lo,
                                const double *__restrict__ hi, const double *__restrict__ invd, GsRowW *__restrict__ rowsw) {
    if (blockIdx.x == 0 && (int)threadIdx.x < nprog) prog[threadIdx.x * kGsProgStride] = 0;  // (bands) nothing stored yet
    if (ROWS) gs_pack_rows<ROWS == 1>(n, rows, b, lo, hi, x, invd, rowsw);
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (i64)gridDim.x * blockDim.x) {
        const GsEnt en = ents[first + k];
        const GsLane la = lanes[first + k];
        const int cnt = (la.info >> 5) & 7, far = la.info >> 8;
        GsDyn d;
#pragma unroll
        for (int e = 0; e < kGsEntries; ++e) {
            const bool fixed = la.code[e] == kGsOne && !((far >> e) & 1);
            d.v[e] = e >= cnt ? -0.0 : (fixed ? x[en.idx[e]] * en.val[e] : en.val[e]);
        }
        dyn[first + k] = d;
    }
}

// x[row] = the result the windowed kernel left at the row's position
__global__ void k_gs_unpack(i64 first, i64 count, const i32 *__restrict__ rows, const double *__restrict__ xpos, double *__restrict__ x) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (i64)gridDim.x * blockDim.x) x[rows[first + k]] = xpos[first + k];
}

// a result of another band's workgroup
__device__ __forceinline__ double gs_read_result(double *p) {
#ifdef SLP_GS_XRMW
    return __longlong_as_double((long long)__hip_atomic_fetch_or((unsigned long long *)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, SLP_GS_XSCOPE);
#endif
}

// lane i receives lane i - 1's v inside its group of 16 lanes (the first lane of a group keeps its own): the plan keeps a row's
// lanes inside one group.  row_shr is the cross-lane path inside a DPP row; the shift over the whole wave (wave_shr) takes
// ~20 ns per dependent step against 3.4 for an add (tools/lab/chain_latency.cpp) -- once per chain round of every level.
__device__ __forceinline__ double gs_lane_up(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);  // row_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// The loop body has no memory instruction under a branch: with loads or stores on some paths only, the compiler can no longer
// count what is in flight (s_waitcnt vmcnt) and falls back to waiting for everything, which undoes the prefetch.  So every
// lane loads its row record, every lane stores (lanes that do not finish a row: to a scratch cell), and a wave that has no
// wave slot in a level repeats the level's first one (same inputs, same results, same addresses: harmless).  FAR = false is
// the instantiation for plans without far entries.
// SAFE (bands only; SLP_GS_BANDS_SAFE=1 when the solver is created): a wave waits for its result stores themselves
// (s_waitcnt vmcnt(0)) before it publishes a level as stored, instead of relying on the in-order completion of its
// vector-memory operations -- see "stored" below and DESIGN.md section 3 (precondition of the default form).
template <bool BOUNDED, bool FAR, bool BANDS, bool SAFE = false>
__global__ __launch_bounds__(64 * kGsWaves) void k_gs_sweep_windowed(const int *__restrict__ hoff, const GsStepW *__restrict__ hdr,
                                                                    const GsLane *__restrict__ lanes, const i32 *__restrict__ lane_row,
                                                                    const GsDyn *__restrict__ dyn, const GsEnt *__restrict__ ents,
                                                                    const GsRowW *__restrict__ rowsw, double *__restrict__ x,
                                                                    double *__restrict__ scratch, double w,
                                                                    const GsBand *__restrict__ bands, const i32 *__restrict__ fsrc,
                                                                    const int *__restrict__ freq, int *__restrict__ prog) {
    // (lane_row holds row POSITIONS, x is the position-ordered result buffer, far entries name positions)
    // the ring, the cell that holds 1.0, one scratch cell per lane, (bands) two generations of external values
    __shared__ double win[BANDS ? kGsWinCells : kGsOne + 1 + 64];
    __shared__ __attribute__((aligned(16))) int progw[16];  // (bands) levels stored, per compute wave
    constexpr int RA = 6, RB = 3, RC = 2;  // ring sizes: headers + lane records (4 ahead), terms (2 ahead), row records (2 ahead)
    GsStepW rec[RA];
    GsLane la[RA];
    i32 lrow[RA];
    GsDyn dv[RB];
    GsRowW rw[RC];
    const int lane = threadIdx.x & 63;
    int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (BANDS) {
        const GsBand bd = bands[blockIdx.x];
        if (wave > bd.waves) return;         // (before any barrier: a barrier waits for the waves that are still running only)
        hoff += bd.hoff;
        scratch += bd.scratch;
        if (threadIdx.x < 16) progw[threadIdx.x] = (int)threadIdx.x < bd.waves ? 0 : kGsDone;
        if (threadIdx.x == 0) win[kGsOne] = 1.0;
        if (wave == 0) {                     // the fetch wave
            constexpr int K = kGsFetchK, D = kGsFetchD;
            const int band = blockIdx.x, nlev = bd.nlev;
            const int *rq = freq + bd.freq + (lane & 15);
            const i32 *fs = fsrc + bd.fsrc + lane;
            int *const mine = prog + band * kGsProgStride;
            const int *const theirs = prog + (lane < band ? lane : band) * kGsProgStride;
            int known = 0;                   // lane q < band: levels of band q stored, as last seen
            auto wait_for = [&](int need) {
                while (__any(lane < band && need > known)) {
                    known = __hip_atomic_load(theirs, __ATOMIC_ACQUIRE, SLP_GS_FSCOPE);  // the reads below stay below
                    if (__any(lane < band && need > known)) __builtin_amdgcn_s_sleep(4);
                }
            };
            int rqv[D];
            i32 src[D][K];
            double val[D][K];
            wait_for(rq[0]);                 // levels 0 .. D - 1
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int k = 0; k < K; ++k) val[d][k] = gs_read_result(x + fs[(d * K + k) * 64]);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                rqv[d] = rq[(d + 1) * 16];
#pragma unroll
                for (int k = 0; k < K; ++k) src[d][k] = fs[((d + D) * K + k) * 64];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) win[kGsExt + k * 64 + lane] = val[0][k];
            __syncthreads();
            for (int l0 = 0; l0 < nlev; l0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int l = l0 + d;
                    wait_for(rqv[d]);        // the lower bands have stored what level l + D reads
                    rqv[d] = rq[(l + D + 1) * 16];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        val[d][k] = gs_read_result(x + src[d][k]);
                        src[d][k] = fs[((l + 2 * D) * K + k) * 64];
                    }
                    // level l + 1's values (read in level l + 1 - D) into the generation the compute waves are not reading
                    // (behind the last level the compute waves may still repeat its wave slots: nothing more goes into the cells then)
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        win[l + 1 < nlev ? kGsExt + ((l + 1) & 1) * (K * 64) + k * 64 + lane : kGsOne + 1 + lane] = val[(d + 1) % D][k];
                    // the least of the waves' counts: every lane reads all 16 (broadcast reads) -- a shuffle tree costs four dependent
                    // trips through the LDS crossbar on the one wave every level waits for
                    int m;
                    {
                        const int4 *pw = reinterpret_cast<const int4 *>(progw);
                        const int4 a = pw[0], b = pw[1], c = pw[2], e = pw[3];
                        m = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
                        m = min(m, min(min(min(c.x, c.y), min(c.z, c.w)), min(min(e.x, e.y), min(e.z, e.w))));
                    }
                    __hip_atomic_store(mine, m, __ATOMIC_RELAXED, SLP_GS_FSCOPE);
                    if (l < nlev) __syncthreads();
                }
            }
            __syncthreads();                 // every compute wave has waited for its stores
            __hip_atomic_store(mine, kGsDone, __ATOMIC_RELAXED, SLP_GS_FSCOPE);
            return;
        }
        wave -= 1;
    } else if (threadIdx.x == 0) {
        win[kGsOne] = 1.0;
    }
    const int h0 = hoff[wave], nsteps = hoff[wave + 1] - h0;  // this wave's headers
    const GsStepW *steps = hdr + h0;
    double *const spill = scratch + threadIdx.x;
    __syncthreads();
    int vzero;  // a zero the compiler cannot see through: keeps the header loads on the vector path
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    auto load_rec = [&](int j, int s) { rec[j] = steps[(s < nsteps ? s : nsteps - 1) + vzero]; };
    auto slot_of = [&](int j) { return (size_t)(unsigned)__builtin_amdgcn_readfirstlane(rec[j].first) + (unsigned)lane; };
    auto issue_a = [&](int j) { la[j] = lanes[slot_of(j)]; lrow[j] = lane_row[slot_of(j)]; };
    auto issue_b = [&](int jb, int j) { dv[jb] = dyn[slot_of(j)]; };
    auto issue_c = [&](int jc, int j) { rw[jc] = rowsw[lrow[j]]; };
    auto process = [&](int s, int j, int jb, int jc) {
        const unsigned meta = __builtin_amdgcn_readfirstlane(rec[j].meta);
        const size_t slot = slot_of(j);
        const size_t slot_a = slot_of((j + 4) % RA), slot_b = slot_of((j + 2) % RA);
        // First what the level's chain waits for: the ring reads of this wave slot (their addresses come from the lane record that
        // arrived four wave slots ago).  The prefetches of later wave slots are issued behind them, into the LDS latency.
        __builtin_amdgcn_sched_barrier(0);
        const GsLane me = la[j];
        double xv[kGsEntries];
#pragma unroll
        for (int e = 0; e < kGsEntries; ++e) xv[e] = win[me.code[e]];
        // (the scheduler must not issue the load into rec[j] above the reads of its old value: the two would then be live
        // together, in two registers, and the copy between them lands right behind the load and waits for it)
        __builtin_amdgcn_sched_barrier(0);
        load_rec(j, s + RA);                                  // the header of wave slot s has been read out
        la[(j + 4) % RA] = lanes[slot_a];
        lrow[(j + 4) % RA] = lane_row[slot_a];
        dv[(jb + 2) % RB] = dyn[slot_b];
        __builtin_amdgcn_sched_barrier(0);
        if (FAR) {
            if (meta & 128u) {  // uniform over the wave
                const GsEnt en = ents[slot];
#pragma unroll
                for (int e = 0; e < kGsEntries; ++e)
                    if ((me.info >> (8 + e)) & 1) xv[e] = __hip_atomic_load(&x[en.idx[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        double term[kGsEntries];
#pragma unroll
        for (int e = 0; e < kGsEntries; ++e) term[e] = xv[e] * dv[jb].v[e];
        const int seg = me.info & 15, rounds = (int)((meta >> 1) & 31u);
        double v = 0.0, carry = 0.0;
        for (int r = 0; r < rounds; ++r) {
            if (seg == r) {
                v = carry;
#pragma unroll
                for (int e = 0; e < kGsEntries; ++e) v += term[e];
            }
            const double up = gs_lane_up(v);
            if (seg == r + 1) carry = up;
        }
        {
            const GsRowW r = rw[jc];
            if (BOUNDED) {
                v = w * (r.b - v) * r.invd + r.xi;
                v = v < r.lo ? r.lo : (v > r.hi ? r.hi : v);
            } else {
                const double nv = (r.b - v + r.lo * r.xi) * r.invd;
                v = w * nv + (1 - w) * r.xi;
            }
            const bool last = me.info & 16;
            if (BANDS) __hip_atomic_store(last ? x + lrow[j] : spill, v, __ATOMIC_RELAXED, SLP_GS_XSCOPE);  // written through: other CUs read it
            else *(last ? x + lrow[j] : spill) = v;
            win[last && me.ldsw != 0xffff ? (int)me.ldsw : kGsOne + 1 + lane] = v;
            if (BANDS) {
                // this wave slot's lane record has arrived, so the stores of wave slot s - 6 and before are in the L2: `stored`
                // (from the plan) counts the levels that completes; the empty asm keeps the write behind the record's arrival
                int stored = (int)(meta >> 8), dep = me.info;
                // PRECONDITION of this form: a wave's vector-memory operations complete in issue order (loads, stores and atomics
                // count together in vmcnt: MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)").  SAFE does not rely on it: the plan then
                // counts the levels up to THIS wave slot (lag 0) and the wave drains its stores before saying so.
                if (SAFE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(stored) : "v"(dep));
                if (lane == 0) progw[wave] = stored;
            }
        }
        issue_c(jc, (j + 2) % RA);                            // rw[jc] is free again
        if (meta & 64u) __syncthreads();                      // the wave's last wave slot of a level: the next level reads these x
    };
#pragma unroll
    for (int j = 0; j < RA; ++j) load_rec(j, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_a(j);
    issue_b(0, 0);
    issue_b(1, 1);
    issue_c(0, 0);
    issue_c(1, 1);
    // The first trip is peeled off the loop: the wait at the head of a loop is the one its most demanding way in needs, and the
    // way in from the loads above (the last of them feeds the first wave slot) needs vmcnt(0) -- as the loop's head, that wait
    // drained the whole prefetch ring on every trip (6 wave slots).  With the peeled trip both ways in look the same.
#pragma unroll
    for (int j = 0; j < RA; ++j) process(j, j, j % RB, j % RC);  // (the plan pads every wave's list to a multiple of RA, at least RA)
    for (int s = RA; s < nsteps; s += RA) {
#pragma unroll
        for (int j = 0; j < RA; ++j) process(s + j, j, j % RB, j % RC);
    }
    if (BANDS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) progw[wave] = kGsDone;
        __syncthreads();
    }
}

__global__ void k_invert_diag(i64 n, const i32 *__restrict__ rows, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                              const double *__restrict__ val, double *__restrict__ invd, double *__restrict__ diag) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i32 i = rows[t];
        double d = 0.0;  // A.diagonal(): 0 where no entry is stored
        for (i64 k = ptr[t]; k < ptr[t + 1]; ++k)
            if (idx[k] == i) d += val[k];
        invd[t] = 1.0 / d;  // gaussSiedel.pyx:91-92
        diag[t] = d;
    }
}

struct GsPlan {
    i64 n = 0, nnz = 0, nlevels = 0, max_width = 0;
    DevBuf<i64> ptr;         // level-ordered storage: position t holds row rows[t], entries ptr[t]..ptr[t+1]
    DevBuf<i32> idx;
    DevBuf<double> val, invd, diag; // invd[t] = 1 / M[rows[t], rows[t]], diag[t] = M[rows[t], rows[t]]
    DevBuf<i32> rows;        // rows sorted by level (stable: increasing row inside a level)
    DevBuf<i64> lptr_dev;    // level pointer on the device (single-workgroup path)
    std::vector<i64> lptr;   // level pointer on the host (launch sizes)
    bool one_block = false;
    bool pipelined = false;  // runs of narrow levels go through the single-workgroup kernel with the register ring
    bool has_far = false;    // (windowed) some entry reads a row updated by the same kernel more than kGsWinLevels - 1 levels before
    bool windowed = false;   // ... and those runs use k_gs_sweep_windowed (entry classes, LDS ring) instead of k_gs_sweep_pipelined
    // launch: level `first` with k_gs_level; else steps [first, first + count) = lane slots [slot_first, slot_first + slot_count)
    // bands > 0: the run is swept by `bands` workgroups (GsBand records band_first ...), `waves` = the most compute waves of one
    struct Segment { bool launch; i64 first, count, slot_first, slot_count, level_first, level_count, hoff; int waves; int bands = 0; i64 band_first = 0; };
    std::vector<Segment> segments;
    DevBuf<GsStep> steps;
    DevBuf<GsSlot> slots;
    DevBuf<GsEnt> ents;               // per lane slot
    DevBuf<GsLane> lanes;             // windowed kernel: per lane slot, the classes of its entries and the lane's place in its row
    DevBuf<i32> lane_row;             // ... and its row
    DevBuf<GsStepW> stepsw;           // ... the wave-slot headers, per segment and per wave
    DevBuf<int> hoffs;                // ... and (waves of the segment) + 1 offsets into them per segment
    mutable DevBuf<GsDyn> dyn;        // per lane slot: the terms, refreshed before every run of a segment (k_gs_pack_terms)
    mutable DevBuf<GsRowW> rowsw;     // per row position, refreshed before every sweep (k_gs_pack_rows)
    mutable DevBuf<double> scratch;   // where lanes that do not finish a row store
    mutable DevBuf<double> xpos;      // the windowed kernel's results, by row position
    mutable DevBuf<GsRow> packed;     // per row position, refreshed before every sweep
    bool bands_safe = false;          // SLP_GS_BANDS_SAFE=1 at plan time: headers count the levels up to their own wave slot, SAFE kernels
    int nbands = 0;                   // band records over all runs with bands (0: none)
    DevBuf<GsBand> bands;
    DevBuf<i32> fsrc;                 // the fetch waves' read positions
    DevBuf<int> freq;                 // ... and what the lower bands must have stored before
    mutable DevBuf<int> prog;         // levels stored, per band of the run being swept
};

// wave-slot headers of one workgroup: a level's wave slots dealt to the waves in turn; a wave without one in a level
// repeats the level's first wave slot (see k_gs_sweep_windowed); a wave's last header of a level carries the barrier.
// As many waves as the widest level has wave slots (at most max_waves): a wave with nothing of its own in a level repeats
// another's wave slot, and those repeats take issue cycles from the waves on its SIMD.  with_stored: header k also says
// how many levels are complete with the wave's wave slot k - 6 (bands).  `info(q0, &rounds, &far)`: chain rounds and far mask
// of the wave slot that starts at lane slot q0.  Appends to hd / hoffs; returns the group's first index into hoffs.
template <class Info>
static int gs_emit_headers(std::vector<GsStepW> &hd, std::vector<int> &hoffs, const std::vector<std::pair<i64, i64>> &lv, int max_waves,
                           bool with_stored, bool bands_safe, int *waves_out, Info info) {
    i64 widest = 1;
    for (const auto &r : lv) widest = std::max(widest, (r.second - r.first) / 64);
    const int waves = (int)std::min<i64>(max_waves, widest);
    *waves_out = waves;
    std::vector<std::vector<GsStepW>> per((size_t)waves);
    std::vector<std::vector<int>> done((size_t)waves);  // levels complete once the header's wave slot is stored
    for (size_t l = 0; l < lv.size(); ++l) {
        const i64 nw = (lv[l].second - lv[l].first) / 64;
        for (i64 q = 0; q < nw; ++q) {
            const size_t q0 = (size_t)(lv[l].first + 64 * q);
            unsigned rounds = 1, far = 0;
            info(q0, &rounds, &far);
            GsStepW h;
            h.first = (int)q0;
            h.meta = 1u | (rounds << 1) | (far ? 128u : 0u);
            per[(size_t)(q % waves)].push_back(h);
            done[(size_t)(q % waves)].push_back((int)l);
        }
        for (int wv = 0; wv < waves; ++wv) {
            if ((i64)wv >= nw) {  // nothing left for this wave: it repeats wave 0's
                per[(size_t)wv].push_back(per[0].back());
                done[(size_t)wv].push_back((int)l);
            }
            per[(size_t)wv].back().meta |= 64u;
            done[(size_t)wv].back() = (int)l + 1;
        }
    }
    const int off = (int)hoffs.size();
    for (int wv = 0; wv < waves; ++wv) {
        // the kernel's loop is unrolled by 6 without a remainder: pad with repeats of the wave's last wave slot
        // (after the last barrier; same inputs, same results), without the barrier flag
        while (per[(size_t)wv].size() % 6) {
            GsStepW h = per[(size_t)wv].back();
            h.meta &= ~64u;
            per[(size_t)wv].push_back(h);
            done[(size_t)wv].push_back((int)lv.size());
        }
        if (with_stored) {
            const char *el = getenv("SLP_GS_BANDS_LAG");  // lab only
            const size_t lag = bands_safe ? 0 : (el ? (size_t)atoi(el) : 6);
            for (size_t k = 0; k < per[(size_t)wv].size(); ++k)
                per[(size_t)wv][k].meta = (per[(size_t)wv][k].meta & 0xffu) | ((unsigned)(k >= lag ? done[(size_t)wv][k - lag] : 0) << 8);
        }
        hoffs.push_back((int)hd.size());
        hd.insert(hd.end(), per[(size_t)wv].begin(), per[(size_t)wv].end());
    }
    hoffs.push_back((int)hd.size());
    return off;
}

static void gs_plan(GsPlan &g, i64 n, const i64 *indptr, const i32 *indices, const double *data) {
    SLP_REQUIRE(n >= 0 && indptr, "gauss-seidel: bad arguments");
    SLP_REQUIRE(n < (i64)1 << 31, "gauss-seidel: dimension must fit int32");
    g.n = n;
    g.nnz = indptr[n];
    { const char *es = getenv("SLP_GS_BANDS_SAFE"); g.bands_safe = es && es[0] == '1'; }
    const auto plan_t0 = std::chrono::steady_clock::now();
    auto plan_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - plan_t0).count(); };
    double ms_levels = 0, ms_bands = 0, ms_permute = 0, ms_pack = 0;
    // dependency levels over the symmetrised pattern, lower part only
    std::vector<i32> level((size_t)n, 0);
    // need(j): the largest level among rows i < j that READ x[j] (anti-dependence: j must wait for i)
    std::vector<i32> need((size_t)n, 0);
    i32 maxlev = 0;
    for (i64 i = 0; i < n; ++i) {
        i32 lv = need[(size_t)i];
        for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
            const i32 j = indices[k];
            SLP_REQUIRE(j >= 0 && j < n, "gauss-seidel: column index out of range");
            if (j < i) lv = std::max(lv, (i32)(level[(size_t)j] + 1));
        }
        level[(size_t)i] = lv;
        for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
            const i32 j = indices[k];
            if (j > i) need[(size_t)j] = std::max(need[(size_t)j], (i32)(lv + 1));
        }
        maxlev = std::max(maxlev, lv);
    }
    // Rows that nothing waits for -- no coupled row of higher index, in either direction of the pattern -- need not sit at the
    // earliest level their inputs allow: when there are many (the slack unknowns of lp_admm's normal matrix: each is coupled
    // only with the variables of its own constraint, all of lower index) they are moved to one level of their own behind all
    // others, which is wide and therefore swept by the chip-wide per-level kernel instead of the single-workgroup one.  Every
    // row still sees exactly the values the sequential sweep gives it (its lower neighbours updated, no higher neighbour
    // exists), two such rows are never coupled, so x does not change by a bit.  SLP_GS_SINKS=0 keeps the earliest levels.
    {
        const char *es = getenv("SLP_GS_SINKS");
        std::vector<char> coupled_up((size_t)n, 0);
        for (i64 i = 0; i < n; ++i)
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
                const i32 j = indices[k];
                if (j != i) coupled_up[(size_t)std::min<i64>(i, j)] = 1;
            }
        i64 sinks = 0;
        for (i64 i = 0; i < n; ++i) sinks += (!coupled_up[(size_t)i] && level[(size_t)i] > 0) ? 1 : 0;
        if (sinks > 4096 && !(es && es[0] == '0')) {
            for (i64 i = 0; i < n; ++i)
                if (!coupled_up[(size_t)i] && level[(size_t)i] > 0) level[(size_t)i] = maxlev + 1;
            // renumber: levels that held nothing else are gone
            std::vector<i32> remap((size_t)maxlev + 2, 0);
            for (i64 i = 0; i < n; ++i) remap[(size_t)level[(size_t)i]] = 1;
            i32 next_level = 0;
            for (size_t l = 0; l < remap.size(); ++l) remap[l] = remap[l] ? next_level++ : -1;
            for (i64 i = 0; i < n; ++i) level[(size_t)i] = remap[(size_t)level[(size_t)i]];
            maxlev = next_level - 1;
        }
    }
    g.nlevels = n ? (i64)maxlev + 1 : 0;
    g.lptr.assign((size_t)g.nlevels + 1, 0);
    for (i64 i = 0; i < n; ++i) g.lptr[(size_t)level[(size_t)i] + 1]++;
    for (i64 l = 0; l < g.nlevels; ++l) {
        g.max_width = std::max(g.max_width, g.lptr[(size_t)l + 1]);
        g.lptr[(size_t)l + 1] += g.lptr[(size_t)l];
    }
    std::vector<i32> rows((size_t)n);
    {
        std::vector<i64> next(g.lptr.begin(), g.lptr.end() - (g.nlevels ? 1 : 0));
        for (i64 i = 0; i < n; ++i) rows[(size_t)next[(size_t)level[(size_t)i]]++] = (i32)i;
    }
    // a single workgroup wins while the per-level launch cost (>= ~1.5 us) exceeds the work of a level
    g.one_block = (g.max_width <= 2048) && (g.nnz <= 200000);
    // Runs of narrow levels: one CU with a barrier per level beats a launch per level (~5.6 us each).  Wide levels
    // (Potts: the first level holds a third of the unknowns) and levels with a very long row keep the per-level
    // kernel, which spreads over the chip.  The sweep is then a sequence of segments.
    // SLP_GS_PIPELINED=0/1 overrides the choice (tests, timing): 1 = every level through the single-workgroup kernel.
    const char *ep = getenv("SLP_GS_PIPELINED");
    const bool forced = ep && ep[0] == '1';
    const i64 wide = forced ? ((i64)1 << 40) : 4096;
    const char *ew = getenv("SLP_GS_WINDOW");
    const bool window = !(ew && ew[0] == '0') && n < ((i64)1 << 30);
    auto row_len = [&](i64 i) { return indptr[i + 1] - indptr[i]; };
    auto row_lanes = [&](i64 i) { return (int)std::max<i64>(1, (row_len(i) + kGsEntries - 1) / kGsEntries); };
    std::vector<char> is_launch((size_t)g.nlevels, 0);
    i64 narrow_levels = 0;
    for (i64 l = 0; l < g.nlevels; ++l) {
        const i64 beg = g.lptr[(size_t)l], end = g.lptr[(size_t)l + 1];
        bool launch = end - beg > wide;
        for (i64 t = beg; t < end && !launch; ++t)
            if (row_len(rows[(size_t)t]) > (i64)kGsMaxSeg * kGsEntries) launch = true;  // a very long row
        is_launch[(size_t)l] = launch ? 1 : 0;
        narrow_levels += launch ? 0 : 1;
    }
    const bool pipeline = n > 0 && g.nnz < ((i64)1 << 31) && !(ep && ep[0] == '0') && (forced || !g.one_block) &&
                          (forced || narrow_levels >= 16) && narrow_levels > 0;

    ms_levels = plan_ms();
    // ---- bands (see kGsFetchK): which runs of narrow levels are cut into row ranges, one workgroup each --------------------------
    // SLP_GS_BANDS=P forces P bands on every run they are possible on, 0 forbids them; otherwise a run gets the number of bands
    // (4, 8, 16 or none) a small timing model of the pipeline likes best.
    struct BandRun {
        i64 level_first = 0, level_count = 0;
        int P = 0;
        std::vector<std::vector<i64>> lptr;               // per band: row positions of its levels (nlev + 1, absolute)
        std::vector<std::vector<std::vector<i32>>> ext;   // per band and level: the external rows it reads (sorted, distinct)
        std::vector<std::vector<i32>> need;               // per band: nlev x 16, levels of band q that must be stored (cumulative)
    };
    std::vector<BandRun> band_runs;
    std::vector<i32> band_of, blev;  // per row (rows of runs with bands)
    if (pipeline && window && n < ((i64)1 << 25)) {
        const char *eb = getenv("SLP_GS_BANDS");
        const int want = eb ? atoi(eb) : -1;
        for (i64 l0 = 0; l0 < g.nlevels && want != 0;) {
            if (is_launch[(size_t)l0]) { ++l0; continue; }
            i64 l1 = l0;
            while (l1 < g.nlevels && !is_launch[(size_t)l1]) ++l1;
            const i64 t0 = g.lptr[(size_t)l0], t1 = g.lptr[(size_t)l1];
            std::vector<i32> rr(rows.begin() + t0, rows.begin() + t1);
            std::sort(rr.begin(), rr.end());
            i64 lanes_total = 0;
            for (i32 i : rr) lanes_total += row_lanes(i);
            // the single workgroup's time: a level costs its barrier-to-barrier chain and its bytes through one CU's load path
            // (calibrated on the Potts 256^2 run, 511 levels of 8 wave slots: one workgroup 530 us, of which 240 are the chain --
            // a build without the loads -- and 4 / 8 / 16 bands 287 / 265 / 327 us: a band's level costs 0.32 us + 0.09 per wave
            // slot, and a band's level is known to the next band ~6 us later: six wave slots, a counter store, a poll)
            const double tau1 = 0.47, tauP = 0.32, per_slot = 0.075, per_slot_band = 0.09, hop = 3.0, flag_latency = 6.0;  // us
            double t_single = 0;
            for (i64 l = l0; l < l1; ++l) {
                i64 ln = 0;
                for (i64 t = g.lptr[(size_t)l]; t < g.lptr[(size_t)l + 1]; ++t) ln += row_lanes(rows[(size_t)t]);
                t_single += tau1 + per_slot * (double)((ln + 63) / 64);
            }
            BandRun best;
            double t_best = want > 0 ? 1e300 : t_single / 1.1 - 5.0;
            std::vector<int> cand;
            // (a band's fetch lists and requirement rows are indexed with ints: runs of more than 2^20 levels keep one workgroup)
            if (l1 - l0 >= ((i64)1 << 20)) cand.clear();
            else if (want > 0) cand.push_back(std::min(want, kGsMaxBands));
            else if (lanes_total >= 32768 && l1 - l0 >= 64) cand = {4, 8, 16};
            if (band_of.empty() && !cand.empty()) { band_of.assign((size_t)n, -1); blev.assign((size_t)n, 0); }
            std::vector<i32> bo_best, bl_best;
            for (int P : cand) {
                if ((i64)rr.size() < P) continue;
                BandRun br;
                br.level_first = l0; br.level_count = l1 - l0; br.P = P;
                // index ranges with equal numbers of lane slots
                i64 acc = 0;
                for (i32 i : rr) {
                    band_of[(size_t)i] = (i32)std::min<i64>(P - 1, acc * P / std::max<i64>(1, lanes_total));
                    acc += row_lanes(i);
                }
                auto in_run = [&](i32 j) { return level[(size_t)j] >= l0 && level[(size_t)j] < l1; };
                // A band sweeps its rows in the order of the GLOBAL levels (its own level numbers: the global ones it has rows in,
                // counted up).  Every band then marches through the run at the pace of the global dependency chain, a higher band
                // the fetch distance and a hop behind the bands it reads from -- levels of a band's own (1 + the deepest lower
                // neighbour inside the band) would put rows whose external inputs come late into early levels, and the whole
                // band would wait for the latest of them.
                std::vector<i32> nlev((size_t)P, 0);
                {
                    std::vector<i32> ord((size_t)P * (size_t)(l1 - l0), 0);
                    for (i32 i : rr) ord[(size_t)band_of[(size_t)i] * (size_t)(l1 - l0) + (size_t)(level[(size_t)i] - l0)] = 1;
                    for (int p = 0; p < P; ++p) {
                        i32 cnt = 0;
                        for (i64 l = 0; l < l1 - l0; ++l) {
                            i32 &o = ord[(size_t)p * (size_t)(l1 - l0) + (size_t)l];
                            const i32 has = o;
                            o = cnt;
                            cnt += has;
                        }
                        nlev[(size_t)p] = cnt;
                    }
                    for (i32 i : rr) blev[(size_t)i] = ord[(size_t)band_of[(size_t)i] * (size_t)(l1 - l0) + (size_t)(level[(size_t)i] - l0)];
                }
                std::vector<unsigned long long> refs;  // band << 60 | level << 32 | external row
                for (i32 i : rr) {
                    const i32 p = band_of[(size_t)i], lv = blev[(size_t)i];
                    for (i64 k = indptr[i]; k < indptr[i + 1]; ++k) {
                        const i32 j = indices[k];
                        if (j < i && in_run(j) && band_of[(size_t)j] < p)
                            refs.push_back((unsigned long long)p << 60 | (unsigned long long)lv << 32 | (unsigned)j);
                    }
                }
                bool ok = true;
                for (int p = 0; p < P; ++p) ok = ok && nlev[(size_t)p] > 0 && nlev[(size_t)p] < (1 << 24);
                if (!ok) continue;
                std::sort(refs.begin(), refs.end());
                refs.erase(std::unique(refs.begin(), refs.end()), refs.end());
                br.ext.resize((size_t)P);
                br.need.resize((size_t)P);
                std::vector<std::vector<i64>> lanes_pl((size_t)P), rows_pl((size_t)P);
                for (int p = 0; p < P; ++p) {
                    br.ext[(size_t)p].resize((size_t)nlev[(size_t)p]);
                    br.need[(size_t)p].assign((size_t)nlev[(size_t)p] * 16, 0);
                    lanes_pl[(size_t)p].assign((size_t)nlev[(size_t)p], 0);
                    rows_pl[(size_t)p].assign((size_t)nlev[(size_t)p], 0);
                }
                for (i32 i : rr) {
                    lanes_pl[(size_t)band_of[(size_t)i]][(size_t)blev[(size_t)i]] += row_lanes(i);
                    rows_pl[(size_t)band_of[(size_t)i]][(size_t)blev[(size_t)i]] += 1;
                }
                for (unsigned long long r : refs) {
                    const int p = (int)(r >> 60);
                    const i32 lv = (i32)((r >> 32) & 0x0fffffffu), j = (i32)(r & 0xffffffffu);
                    std::vector<i32> &cell = br.ext[(size_t)p][(size_t)lv];
                    cell.push_back(j);
                    if (cell.size() > (size_t)kGsFetchK * 64) { ok = false; break; }
                    i32 &nd = br.need[(size_t)p][(size_t)lv * 16 + (size_t)band_of[(size_t)j]];
                    nd = std::max(nd, (i32)(blev[(size_t)j] + 1));
                }
                if (!ok) continue;
                for (int p = 0; p < P; ++p)
                    for (i32 lv = 1; lv < nlev[(size_t)p]; ++lv)
                        for (int q = 0; q < 16; ++q)
                            br.need[(size_t)p][(size_t)lv * 16 + q] = std::max(br.need[(size_t)p][(size_t)lv * 16 + q], br.need[(size_t)p][(size_t)(lv - 1) * 16 + q]);
                // the pipeline's finish time: a band's level starts when its own previous level is done and the lower bands have
                // stored what the reads issued in it (for kGsFetchD levels ahead) want, seen one flag latency later
                std::vector<std::vector<double>> fin((size_t)P);
                double t_all = 0;
                for (int p = 0; p < P; ++p) {
                    fin[(size_t)p].assign((size_t)nlev[(size_t)p], 0.0);
                    double t = 0;
                    for (i32 lv = 0; lv < nlev[(size_t)p]; ++lv) {
                        const i32 ahead = std::min<i32>(lv + kGsFetchD, nlev[(size_t)p] - 1);
                        for (int q = 0; q < p; ++q) {
                            const i32 nd = br.need[(size_t)p][(size_t)ahead * 16 + q];
                            if (nd > 0) t = std::max(t, fin[(size_t)q][(size_t)nd - 1] + flag_latency);
                        }
                        t += tauP + per_slot_band * (double)((lanes_pl[(size_t)p][(size_t)lv] + 63) / 64);
                        fin[(size_t)p][(size_t)lv] = t;
                    }
                    t_all = std::max(t_all, t + hop);
                }
                if (getenv("SLP_GS_VERBOSE"))
                    fprintf(stderr, "gauss-seidel bands: levels %lld..%lld, %lld lane slots: one workgroup %.0f us, %d bands %.0f us (levels of band 0: %d)\n",
                            (long long)l0, (long long)l1, (long long)lanes_total, t_single, P, t_all, (int)nlev[0]);
                if (t_all < t_best) {
                    t_best = t_all;
                    // positions: band after band, level after level, rows ascending inside a level
                    br.lptr.resize((size_t)P);
                    i64 at = t0;
                    for (int p = 0; p < P; ++p) {
                        br.lptr[(size_t)p].assign((size_t)nlev[(size_t)p] + 1, 0);
                        for (i32 lv = 0; lv < nlev[(size_t)p]; ++lv) {
                            br.lptr[(size_t)p][(size_t)lv] = at;
                            at += rows_pl[(size_t)p][(size_t)lv];
                        }
                        br.lptr[(size_t)p][(size_t)nlev[(size_t)p]] = at;
                    }
                    best = std::move(br);
                    bo_best.resize(rr.size());
                    bl_best.resize(rr.size());
                    for (size_t k = 0; k < rr.size(); ++k) { bo_best[k] = band_of[(size_t)rr[k]]; bl_best[k] = blev[(size_t)rr[k]]; }
                }
            }
            if (best.P > 0) {
                std::vector<std::vector<i64>> next = best.lptr;
                for (size_t k = 0; k < rr.size(); ++k) {
                    band_of[(size_t)rr[k]] = bo_best[k];
                    blev[(size_t)rr[k]] = bl_best[k];
                    rows[(size_t)next[(size_t)bo_best[k]][(size_t)bl_best[k]]++] = rr[k];
                }
                band_runs.push_back(std::move(best));
            } else if (!band_of.empty()) {
                for (i32 i : rr) band_of[(size_t)i] = -1;
            }
            l0 = l1;
        }
    }

    ms_bands = plan_ms() - ms_levels;
    {   // permute the matrix into position order on the host (one O(nnz) pass)
        std::vector<i64> p2((size_t)n + 1, 0);
        std::vector<i32> j2((size_t)g.nnz + kGsEntries, 0);   // padded: the pipelined sweep reads kGsEntries per lane unconditionally
        std::vector<double> v2((size_t)g.nnz + kGsEntries, 0.0);
        i64 o = 0;
        for (i64 t = 0; t < n; ++t) {
            const i64 i = rows[(size_t)t];
            p2[(size_t)t] = o;
            for (i64 k = indptr[i]; k < indptr[i + 1]; ++k, ++o) {
                j2[(size_t)o] = indices[k];
                v2[(size_t)o] = data[k];
            }
        }
        p2[(size_t)n] = o;
        g.ptr.upload(p2.data(), (size_t)n + 1);
        g.idx.upload(j2.data(), j2.size());
        g.val.upload(v2.data(), v2.size());
    }
    ms_permute = plan_ms() - ms_levels - ms_bands;
    g.rows.upload(rows.data(), (size_t)n);
    g.lptr_dev.upload(g.lptr.data(), g.lptr.size());
    g.invd.alloc((size_t)n);
    g.diag.alloc((size_t)n);
    if (n) {
        hipLaunchKernelGGL(k_invert_diag, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, ctx().stream, n, g.rows.p, g.ptr.p, g.idx.p,
                           g.val.p, g.invd.p, g.diag.p);
        SLP_HIP(hipGetLastError());
    }
    g.pipelined = false;
    if (pipeline) {
        std::vector<GsStep> st;
        std::vector<GsSlot> slots;
        std::vector<GsPlan::Segment> segs;
        std::vector<GsEnt> ents;
        std::vector<GsLane> lanes;
        std::vector<i32> lane_row;  // (windowed kernel) the row of every lane slot, 0 for idle ones
        const GsSlot idle = {0, 0, 1 << 24, -1};
        GsEnt noent;
        GsLane nolane;
        for (int e = 0; e < kGsEntries; ++e) { noent.idx[e] = 0; noent.val[e] = 0.0; nolane.code[e] = (unsigned short)kGsOne; }
        nolane.info = 0;  // no entries, not the last lane of a row
        nolane.ldsw = 0xffff;
        auto pad_wave = [&]() {
            while (slots.size() & 63) { slots.push_back(idle); ents.push_back(noent); lanes.push_back(nolane); lane_row.push_back(0); }
        };
        auto pad_group = [&]() {  // (gs_lane_up hands a row's sum on inside groups of 16 lanes)
            while (slots.size() & 15) { slots.push_back(idle); ents.push_back(noent); lanes.push_back(nolane); lane_row.push_back(0); }
        };

        std::vector<i64> lvs0((size_t)g.nlevels, 0), lvs1((size_t)g.nlevels, 0);  // lane slots of every narrow level
        // windowed kernel: position of every row, and the first level of the run of narrow levels being built
        std::vector<i32> pos;
        if (window) {
            pos.resize((size_t)n);
            for (i64 t = 0; t < n; ++t) pos[(size_t)rows[(size_t)t]] = (i32)t;
        }
        i64 seg_first_level = 0;
        std::vector<i64> kpos((size_t)n + 1, 0);  // position-ordered entry offset of every row position
        for (i64 t = 0; t < n; ++t) kpos[(size_t)t + 1] = kpos[(size_t)t] + row_len(rows[(size_t)t]);

        // wave-slot headers of one workgroup: a level's wave slots dealt to the waves in turn; a wave without one in a level
        // repeats the level's first wave slot (see k_gs_sweep_windowed); a wave's last header of a level carries the barrier.
        // As many waves as the widest level has wave slots (at most max_waves): a wave with nothing of its own in a level repeats
        // another's wave slot, and those repeats take issue cycles from the waves on its SIMD.  with_stored: header k also says
        // how many levels are complete with the wave's wave slot k - 6 (bands).
        std::vector<GsStepW> hd;
        std::vector<int> hoffs;
        auto emit_headers = [&](const std::vector<std::pair<i64, i64>> &lv, int max_waves, bool with_stored, int *waves_out) {
            return gs_emit_headers(hd, hoffs, lv, max_waves, with_stored, g.bands_safe, waves_out, [&](size_t q0, unsigned *rounds, unsigned *far) {
                unsigned r = 1, f = 0;
                for (size_t k = q0; k < q0 + 64; ++k) {
                    r = std::max(r, (unsigned)(lanes[k].info & 15) + 1u);
                    f |= (unsigned)(lanes[k].info >> 8);
                }
                *rounds = r;
                *far = f;
                if (f) g.has_far = true;
            });
        };
        // one lane of a row: entries [j E, (j + 1) E) of row position t; cls(row, e, lane record, entry record) sets the entry's class
        auto push_row = [&](i64 t, int ldsw, auto &&cls) {
            const i64 len = kpos[(size_t)t + 1] - kpos[(size_t)t];
            const int nl = (int)std::max<i64>(1, (len + kGsEntries - 1) / kGsEntries);
            for (int j = 0; j < nl; ++j) {
                const i64 left = len - (i64)j * kGsEntries;
                const int cnt = (int)std::max<i64>(0, std::min<i64>(kGsEntries, left));
                GsSlot sl;
                sl.row = rows[(size_t)t]; sl.t = (i32)t;
                sl.pad = ldsw;
                sl.info = j | (nl << 8) | (cnt << 16);
                slots.push_back(sl);
                GsEnt en = noent;
                GsLane cd = nolane;
                cd.info = (unsigned short)(j | (j == nl - 1 ? 16 : 0) | (cnt << 5));
                cd.ldsw = (unsigned short)(ldsw >= 0 ? ldsw : 0xffff);
                lane_row.push_back((i32)t);
                const i64 src = indptr[rows[(size_t)t]] + (i64)j * kGsEntries;  // the row's entries in storage order
                for (int e = 0; e < cnt; ++e) {
                    en.idx[e] = indices[src + e];
                    en.val[e] = data[src + e];
                    if (window) cls(indices[src + e], e, cd, en);
                }
                ents.push_back(en);
                lanes.push_back(cd);
            }
            return nl;
        };

        std::vector<GsBand> bands;
        std::vector<i32> fsrc;
        std::vector<int> freq;
        size_t next_band_run = 0;
        i64 scratch_cells = (i64)64 * kGsWaves;
        for (i64 l = 0; l < g.nlevels; ++l) {
            if (is_launch[(size_t)l]) {
                GsPlan::Segment sg;
                sg.launch = true; sg.first = l; sg.count = 1; sg.slot_first = 0; sg.slot_count = 0;
                sg.level_first = l; sg.level_count = 1; sg.hoff = 0; sg.waves = 0;
                segs.push_back(sg);
                continue;
            }
            if (next_band_run < band_runs.size() && band_runs[next_band_run].level_first == l) {
                // ---- a run with bands: one segment, one workgroup per band ----
                const BandRun &br = band_runs[next_band_run++];
                const i64 L0 = br.level_first, L1 = br.level_first + br.level_count;
                GsPlan::Segment sg;
                sg.launch = false; sg.first = (i64)st.size(); sg.count = 0; sg.slot_first = (i64)slots.size(); sg.slot_count = 0;
                sg.level_first = L0; sg.level_count = br.level_count; sg.hoff = 0; sg.waves = 0;
                sg.bands = br.P; sg.band_first = (i64)bands.size();
                for (int p = 0; p < br.P; ++p) {
                    const std::vector<i64> &bl = br.lptr[(size_t)p];
                    const i32 nlev = (i32)bl.size() - 1;
                    std::vector<std::pair<i64, i64>> lv((size_t)nlev);
                    for (i32 ll = 0; ll < nlev; ++ll) {
                        const i64 beg = bl[(size_t)ll], end = bl[(size_t)ll + 1];
                        const size_t slot0 = slots.size();
                        const bool fits_ring = end - beg <= kGsWide;
                        const std::vector<i32> &ex = br.ext[(size_t)p][(size_t)ll];
                        for (i64 t = beg; t < end; ++t) {
                            const i32 i = rows[(size_t)t];
                            if ((slots.size() & 15) + (size_t)row_lanes(i) > 16) pad_group();  // a row's lanes stay inside one group of 16
                            push_row(t, fits_ring ? (int)((ll % kGsWinLevels) * kGsWide + (t - beg)) : -1,
                                     [&](i32 js, int e, GsLane &cd, GsEnt &en) {
                                         if (js >= i || level[(size_t)js] < L0 || level[(size_t)js] >= L1) return;  // static
                                         if (band_of[(size_t)js] == p) {
                                             const i32 lj = blev[(size_t)js];
                                             if (lj > ll - kGsWinLevels && bl[(size_t)lj + 1] - bl[(size_t)lj] <= kGsWide) {
                                                 cd.code[e] = (unsigned short)((lj % kGsWinLevels) * kGsWide + (pos[(size_t)js] - bl[(size_t)lj]));
                                             } else {
                                                 cd.info |= (unsigned short)(1 << (8 + e));
                                                 en.idx[e] = pos[(size_t)js];
                                                 g.has_far = true;
                                             }
                                         } else {  // a lower band's row: the fetch wave has put it into this level's generation
                                             const size_t at = (size_t)(std::lower_bound(ex.begin(), ex.end(), js) - ex.begin());
                                             cd.code[e] = (unsigned short)(kGsExt + (ll & 1) * (kGsFetchK * 64) + (int)at);
                                         }
                                     });
                        }
                        pad_wave();
                        lv[(size_t)ll] = {(i64)slot0, (i64)slots.size()};
                    }
                    GsBand bd;
                    bd.nlev = nlev;
                    bd.fsrc = (int)fsrc.size();
                    for (i32 ll = 0; ll < nlev + 3 * kGsFetchD; ++ll)
                        for (int c = 0; c < kGsFetchK * 64; ++c) {
                            const bool real = ll < nlev && (size_t)c < br.ext[(size_t)p][(size_t)ll].size();
                            fsrc.push_back(real ? pos[(size_t)br.ext[(size_t)p][(size_t)ll][(size_t)c]] : (i32)bl[0]);
                        }
                    bd.freq = (int)freq.size();
                    for (i32 r = 0; r < nlev + 1 + 2 * kGsFetchD; ++r) {  // row 0: levels 0 .. D - 1; row l + 1: levels up to l + D
                        const i32 upto = std::min<i32>(nlev - 1, r == 0 ? kGsFetchD - 1 : r - 1 + kGsFetchD);
                        for (int q = 0; q < 16; ++q) freq.push_back(q < p ? br.need[(size_t)p][(size_t)upto * 16 + q] : 0);
                    }
                    int waves = 1;
                    bd.hoff = emit_headers(lv, kGsBandWaves, true, &waves);
                    bd.waves = waves;
                    bd.scratch = p * 1024;
                    bd.pad0 = bd.pad1 = 0;
                    bands.push_back(bd);
                    if (getenv("SLP_GS_VERBOSE") && atoi(getenv("SLP_GS_VERBOSE")) > 1) {
                        fprintf(stderr, "band %d: positions %lld..%lld, %d levels, %d waves, hoff %d, fsrc %d, freq %d; needs row 0:", p, (long long)bl[0],
                                (long long)bl[(size_t)nlev], nlev, waves, bd.hoff, bd.fsrc, bd.freq);
                        for (int q = 0; q < 4; ++q) fprintf(stderr, " %d", freq[(size_t)bd.freq + (size_t)q]);
                        fprintf(stderr, "; row 1:");
                        for (int q = 0; q < 4; ++q) fprintf(stderr, " %d", freq[(size_t)bd.freq + 16 + (size_t)q]);
                        fprintf(stderr, "; reads of level 0:");
                        for (int c = 0; c < 3; ++c) fprintf(stderr, " %d", fsrc[(size_t)bd.fsrc + (size_t)c]);
                        fprintf(stderr, "; headers of wave 0: %d, first metas:", hoffs[(size_t)bd.hoff + 1] - hoffs[(size_t)bd.hoff]);
                        for (int k = 0; k < 8 && hoffs[(size_t)bd.hoff] + k < (int)hd.size(); ++k) fprintf(stderr, " %x", hd[(size_t)hoffs[(size_t)bd.hoff] + (size_t)k].meta);
                        fprintf(stderr, "; last metas:");
                        for (int k = 8; k >= 1; --k) fprintf(stderr, " %x", hd[(size_t)hoffs[(size_t)bd.hoff + 1] - (size_t)k].meta);
                        fprintf(stderr, "\n");
                    }
                    sg.waves = std::max(sg.waves, waves);
                }
                scratch_cells = std::max<i64>(scratch_cells, (i64)br.P * 1024);
                sg.slot_count = (i64)slots.size() - sg.slot_first;
                segs.push_back(sg);
                l = L1 - 1;
                continue;
            }
            if (segs.empty() || segs.back().launch || segs.back().bands) seg_first_level = l;
            const i64 beg = g.lptr[(size_t)l], end = g.lptr[(size_t)l + 1];
            const size_t step0 = st.size();
            const size_t slot0 = slots.size();
            size_t first = slots.size();
            int maxseg = 1;
            bool has_far = false;
            auto close_step = [&](bool barrier) {
                while (window && ((slots.size() - first) & 63)) { slots.push_back(idle); ents.push_back(noent); lanes.push_back(nolane); lane_row.push_back(0); }
                GsStep sd;
                sd.first = (int)first; sd.count = (int)(slots.size() - first); sd.maxseg = maxseg;
                sd.barrier = (barrier ? 1 : 0) | (has_far ? 2 : 0);
                st.push_back(sd);
                first = slots.size();
                maxseg = 1;
                has_far = false;
            };
            const bool fits_ring = end - beg <= kGsWide;  // this level's results go to the LDS ring
            for (i64 t = beg; t < end; ++t) {
                const int nl = row_lanes(rows[(size_t)t]);
                size_t used = slots.size() - first;
                if ((used & 15) + (size_t)nl > 16) {  // a row's lanes stay inside one group of 16 lanes (of the step's lane slots)
                    while ((slots.size() - first) & 15) { slots.push_back(idle); ents.push_back(noent); lanes.push_back(nolane); lane_row.push_back(0); }
                    used = slots.size() - first;
                }
                if (used + (size_t)nl > 1024) close_step(false);  // next step of the same level: no barrier in between
                push_row(t, (window && fits_ring) ? (int)((l % kGsWinLevels) * kGsWide + (t - beg)) : -1,
                         [&](i32 js, int e, GsLane &cd, GsEnt &en) {
                             const i64 lj = level[(size_t)js];
                             if (lj < seg_first_level || lj >= l) {
                                 cd.code[e] = (unsigned short)kGsOne;  // static: final before this run starts, or not touched before this row's step
                             } else if (lj > l - kGsWinLevels && g.lptr[(size_t)lj + 1] - g.lptr[(size_t)lj] <= kGsWide) {
                                 cd.code[e] = (unsigned short)((lj % kGsWinLevels) * kGsWide + (pos[(size_t)js] - g.lptr[(size_t)lj]));  // near
                             } else {
                                 cd.code[e] = (unsigned short)kGsOne;  // far: gathered from the results; the position is read from the entry record
                                 cd.info |= (unsigned short)(1 << (8 + e));
                                 en.idx[e] = pos[(size_t)js];
                                 has_far = true;
                             }
                         });
                maxseg = std::max(maxseg, nl);
            }
            close_step(true);
            lvs0[(size_t)l] = (i64)slot0;
            lvs1[(size_t)l] = (i64)slots.size();
            if (!segs.empty() && !segs.back().launch && !segs.back().bands) {
                segs.back().count += (i64)(st.size() - step0);
                segs.back().slot_count += (i64)(slots.size() - slot0);
                segs.back().level_count += 1;
            } else {
                GsPlan::Segment sg;
                sg.launch = false; sg.first = (i64)step0; sg.count = (i64)(st.size() - step0);
                sg.slot_first = (i64)slot0; sg.slot_count = (i64)(slots.size() - slot0);
                sg.level_first = l; sg.level_count = 1; sg.hoff = 0; sg.waves = 0;
                segs.push_back(sg);
            }
        }
        ms_pack = plan_ms() - ms_levels - ms_bands - ms_permute;
        if (!slots.empty() && slots.size() < ((size_t)1 << 31)) {
            g.pipelined = true;
            g.one_block = false;
            g.steps.upload(st.data(), st.size());
            g.slots.upload(slots.data(), slots.size());
            g.ents.upload(ents.data(), ents.size());
            g.packed.alloc((size_t)n);
            g.segments = segs;
            g.windowed = window;
            if (window) {
                for (GsPlan::Segment &sg : segs) {
                    if (sg.launch || sg.bands) continue;
                    std::vector<std::pair<i64, i64>> lv;
                    for (i64 l = sg.level_first; l < sg.level_first + sg.level_count; ++l) lv.push_back({lvs0[(size_t)l], lvs1[(size_t)l]});
                    int waves = 1;
                    sg.hoff = emit_headers(lv, kGsWaves, false, &waves);
                    sg.waves = waves;
                }
                g.segments = segs;
                g.stepsw.upload(hd.data(), hd.size());
                g.hoffs.upload(hoffs.data(), hoffs.size());
                g.lanes.upload(lanes.data(), lanes.size());
                g.lane_row.upload(lane_row.data(), lane_row.size());
                g.dyn.alloc(lanes.size());
                g.rowsw.alloc((size_t)n);
                g.scratch.alloc((size_t)scratch_cells);
                g.xpos.alloc((size_t)n);
                if (!bands.empty()) {
                    g.nbands = (int)bands.size();
                    g.bands.upload(bands.data(), bands.size());
                    g.fsrc.upload(fsrc.data(), fsrc.size());
                    g.freq.upload(freq.data(), freq.size());
                    g.prog.alloc((size_t)kGsMaxBands * kGsProgStride);
                }
            }
        }
    }
    if (getenv("SLP_GS_VERBOSE"))
        fprintf(stderr, "gauss-seidel plan: n %lld, %lld entries, %lld levels: levels %.1f ms, bands %.1f ms, permute + upload %.1f ms, lane records %.1f ms, all %.1f ms (host)\n",
                (long long)n, (long long)g.nnz, (long long)g.nlevels, ms_levels, ms_bands, ms_permute, ms_pack, plan_ms());
}

}  // namespace slp
#include "slp_gs_plan_device.h"
namespace slp {

// SLP_GS_PLAN: "device" (default) -- the plan is built on the device from the matrix where it lies (gs_plan_device; the host plan only
// where that declines); "host" -- gs_plan from a host copy of the matrix (rounds 1-4); "check" -- both, and every array of the
// device plan must equal the host plan's (tests).
constexpr i64 kGsSmallPlan = 200000;   // stored entries up to which the host plan is the quicker one
static int gs_plan_mode() {
    const char *e = getenv("SLP_GS_PLAN");
    if (e && !strcmp(e, "host")) return 0;
    if (e && !strcmp(e, "check")) return 2;
    return 1;
}

template <class T>
static void gs_plan_same(const char *what, const DevBuf<T> &a, const DevBuf<T> &b, size_t count) {
    if (count == 0) return;
    SLP_REQUIRE(a.p && b.p && a.n >= count && b.n >= count, std::string("gs plan check: ") + what + ": missing or short array");
    std::vector<T> ha(count), hb(count);
    a.download(ha.data(), count);
    b.download(hb.data(), count);
    if (memcmp(ha.data(), hb.data(), count * sizeof(T)) != 0) {
        size_t k = 0;
        while (k < count && memcmp(&ha[k], &hb[k], sizeof(T)) == 0) ++k;
        throw Error(std::string("gs plan check: ") + what + " differs at element " + std::to_string(k) + " of " + std::to_string(count));
    }
}

// every array the sweep reads: device plan `d` against host plan `h`
static void gs_plan_compare(const GsPlan &d, const GsPlan &h) {
    auto same = [&](bool ok, const char *what) { SLP_REQUIRE(ok, std::string("gs plan check: ") + what + " differs"); };
    same(d.n == h.n && d.nnz == h.nnz && d.nlevels == h.nlevels && d.max_width == h.max_width, "sizes / levels");
    same(d.lptr == h.lptr, "level pointer");
    same(d.one_block == h.one_block && d.pipelined == h.pipelined && d.windowed == h.windowed && d.has_far == h.has_far && d.nbands == h.nbands,
         "plan kind (one_block / pipelined / windowed / has_far / bands)");
    const size_t n = (size_t)d.n;
    gs_plan_same("rows", d.rows, h.rows, n);
    gs_plan_same("ptr", d.ptr, h.ptr, n + 1);
    gs_plan_same("idx", d.idx, h.idx, (size_t)d.nnz + kGsEntries);
    gs_plan_same("val", d.val, h.val, (size_t)d.nnz + kGsEntries);
    gs_plan_same("invd", d.invd, h.invd, n);
    gs_plan_same("diag", d.diag, h.diag, n);
    gs_plan_same("lptr_dev", d.lptr_dev, h.lptr_dev, (size_t)d.nlevels + 1);
    same(d.segments.size() == h.segments.size(), "segment count");
    for (size_t k = 0; k < d.segments.size(); ++k) {
        const GsPlan::Segment &a = d.segments[k], &b = h.segments[k];
        same(a.launch == b.launch && a.level_first == b.level_first && a.level_count == b.level_count && a.bands == b.bands && a.waves == b.waves &&
                 (a.launch || (a.slot_first == b.slot_first && a.slot_count == b.slot_count && (a.bands || a.hoff == b.hoff) && a.band_first == b.band_first)) &&
                 (!a.launch || a.first == b.first),
             "a segment");
    }
    if (d.pipelined) {
        same(d.lanes.n == h.lanes.n && d.ents.n == h.ents.n && d.stepsw.n == h.stepsw.n && d.hoffs.n == h.hoffs.n, "lane slot / header counts");
        gs_plan_same("ents", d.ents, h.ents, d.ents.n);
        gs_plan_same("lanes", d.lanes, h.lanes, d.lanes.n);
        gs_plan_same("lane_row", d.lane_row, h.lane_row, d.lane_row.n);
        gs_plan_same("stepsw", d.stepsw, h.stepsw, d.stepsw.n);
        gs_plan_same("hoffs", d.hoffs, h.hoffs, d.hoffs.n);
        same(d.scratch.n == h.scratch.n && d.dyn.n == h.dyn.n, "scratch / term buffers");
        if (d.nbands) {
            same(d.fsrc.n == h.fsrc.n && d.freq.n == h.freq.n, "band table sizes");
            gs_plan_same("bands", d.bands, h.bands, (size_t)d.nbands);
            gs_plan_same("fsrc", d.fsrc, h.fsrc, d.fsrc.n);
            gs_plan_same("freq", d.freq, h.freq, d.freq.n);
        }
    }
}

// the plan for a matrix that lies on the device (dptr / didx / dval); host copies are made only where the host plan is needed
static void gs_plan_any(GsPlan &g, i64 n, i64 nnz, const i64 *dptr, const i32 *didx, const double *dval) {
    const int mode = gs_plan_mode();
    auto host_plan = [&](GsPlan &out) {
        std::vector<i64> mp((size_t)n + 1);
        std::vector<i32> mj((size_t)nnz);
        std::vector<double> mx((size_t)nnz);
        hipStream_t st = ctx().stream;
        SLP_HIP(hipMemcpyAsync(mp.data(), dptr, mp.size() * sizeof(i64), hipMemcpyDeviceToHost, st));
        if (nnz) SLP_HIP(hipMemcpyAsync(mj.data(), didx, mj.size() * sizeof(i32), hipMemcpyDeviceToHost, st));
        if (nnz) SLP_HIP(hipMemcpyAsync(mx.data(), dval, mx.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SLP_HIP(hipStreamSynchronize(st));
        Phase ph("gs_plan (host level schedule)");
        gs_plan(out, n, mp.data(), mj.data(), mx.data());
    };
    // (small systems -- the single-workgroup regime -- keep the host plan: a 2 MB download and microseconds of host work against
    // ~30 launches and a dozen synchronisations)
    if (mode == 0 || (mode == 1 && nnz <= kGsSmallPlan)) { host_plan(g); return; }
    bool ok;
    {
        Phase ph("gs_plan_device");
        ok = gs_plan_device(g, n, nnz, dptr, didx, dval);
    }
    if (!ok) { g = GsPlan(); host_plan(g); return; }
    if (mode == 2) {
        GsPlan h;
        host_plan(h);
        gs_plan_compare(g, h);
    }
}

// bounded = false: plain SOR sweep (no bounds); the kernels then read the diagonal through the `lo` argument
static void gs_sweep(const GsPlan &g, const double *b, const double *lo, const double *hi, double *x, double w, int sweeps,
                     bool bounded = true) {
    if (g.n == 0 || sweeps <= 0) return;
    hipStream_t st = ctx().stream;
    if (!bounded) { lo = g.diag.p; hi = g.diag.p; }
    if (g.one_block) {
        const int threads = g.max_width <= 64 ? 64 : (g.max_width <= 256 ? 256 : 1024);
        if (bounded)
            hipLaunchKernelGGL(k_gs_sweep_one_block<true>, dim3(1), dim3(threads), 0, st, g.nlevels, g.lptr_dev.p, g.rows.p, g.ptr.p,
                               g.idx.p, g.val.p, g.invd.p, b, lo, hi, x, w, sweeps);
        else
            hipLaunchKernelGGL(k_gs_sweep_one_block<false>, dim3(1), dim3(threads), 0, st, g.nlevels, g.lptr_dev.p, g.rows.p, g.ptr.p,
                               g.idx.p, g.val.p, g.invd.p, b, lo, hi, x, w, sweeps);
        SLP_HIP(hipGetLastError());
        return;
    }
    auto level_launch = [&](i64 l) {
        const i64 beg = g.lptr[(size_t)l], cnt = g.lptr[(size_t)l + 1] - beg;
        if (cnt <= 0) return;
        if (bounded)
            hipLaunchKernelGGL(k_gs_level<true>, dim3((unsigned)((cnt + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, beg, cnt,
                               g.rows.p, g.ptr.p, g.idx.p, g.val.p, g.invd.p, b, lo, hi, x, w);
        else
            hipLaunchKernelGGL(k_gs_level<false>, dim3((unsigned)((cnt + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, beg, cnt,
                               g.rows.p, g.ptr.p, g.idx.p, g.val.p, g.invd.p, b, lo, hi, x, w);
    };
    if (g.pipelined) {
        for (int s = 0; s < sweeps; ++s) {
            // (non-windowed kernel) right-hand side, bounds and own x of every row, in level order; the windowed kernel's row records
            // are packed by the first windowed segment's k_gs_pack_terms
            if (!g.windowed && bounded)
                hipLaunchKernelGGL(k_gs_pack<true>, dim3(grid_for(g.n, kBlock)), dim3(kBlock), 0, st, g.n, g.rows.p, b, lo, hi, x, g.packed.p);
            else if (!g.windowed)
                hipLaunchKernelGGL(k_gs_pack<false>, dim3(grid_for(g.n, kBlock)), dim3(kBlock), 0, st, g.n, g.rows.p, b, lo, hi, x, g.packed.p);
            bool rows_packed = false;
            for (const GsPlan::Segment &sg : g.segments) {
                if (sg.launch) {
                    level_launch(sg.first);
                } else if (g.windowed) {
                    {
                        const int pgrid = rows_packed ? grid_for(sg.slot_count, kBlock) : grid_for(std::max<i64>(sg.slot_count, g.n), kBlock);
                        auto pack = [&](auto kernel) {
                            hipLaunchKernelGGL(kernel, dim3(pgrid), dim3(kBlock), 0, st, sg.slot_first, sg.slot_count, g.ents.p, g.lanes.p, x,
                                               g.dyn.p, g.prog.p, sg.bands, g.n, g.rows.p, b, lo, hi, g.invd.p, g.rowsw.p);
                        };
                        if (rows_packed) pack(k_gs_pack_terms<0>);
                        else if (bounded) pack(k_gs_pack_terms<1>);
                        else pack(k_gs_pack_terms<2>);
                        rows_packed = true;
                    }
                    auto run = [&](auto kernel) {
                        // bands: one workgroup per band (its compute waves + the fetch wave); else one workgroup
                        hipLaunchKernelGGL(kernel, dim3(sg.bands ? sg.bands : 1), dim3(64 * (sg.waves + (sg.bands ? 1 : 0))), 0, st,
                                           g.hoffs.p + sg.hoff, g.stepsw.p, g.lanes.p, g.lane_row.p, g.dyn.p, g.ents.p, g.rowsw.p, g.xpos.p,
                                           g.scratch.p, w, g.bands.p + sg.band_first, g.fsrc.p, g.freq.p, g.prog.p);
                        const i64 t0 = g.lptr[(size_t)sg.level_first], t1 = g.lptr[(size_t)(sg.level_first + sg.level_count)];
                        hipLaunchKernelGGL(k_gs_unpack, dim3(grid_for(t1 - t0, kBlock)), dim3(kBlock), 0, st, t0, t1 - t0, g.rows.p, g.xpos.p, x);
                    };
                    if (sg.bands && g.bands_safe) {
                        if (bounded && g.has_far) run(k_gs_sweep_windowed<true, true, true, true>);
                        else if (bounded) run(k_gs_sweep_windowed<true, false, true, true>);
                        else if (g.has_far) run(k_gs_sweep_windowed<false, true, true, true>);
                        else run(k_gs_sweep_windowed<false, false, true, true>);
                    } else if (sg.bands) {
                        if (bounded && g.has_far) run(k_gs_sweep_windowed<true, true, true>);
                        else if (bounded) run(k_gs_sweep_windowed<true, false, true>);
                        else if (g.has_far) run(k_gs_sweep_windowed<false, true, true>);
                        else run(k_gs_sweep_windowed<false, false, true>);
                    } else if (bounded && g.has_far) run(k_gs_sweep_windowed<true, true, false>);
                    else if (bounded) run(k_gs_sweep_windowed<true, false, false>);
                    else if (g.has_far) run(k_gs_sweep_windowed<false, true, false>);
                    else run(k_gs_sweep_windowed<false, false, false>);
                } else if (bounded) {
                    hipLaunchKernelGGL(k_gs_sweep_pipelined<true>, dim3(1), dim3(1024), 0, st, (int)sg.count, g.steps.p + sg.first,
                                       g.slots.p, g.ents.p, g.invd.p, g.packed.p, x, w);
                } else {
                    hipLaunchKernelGGL(k_gs_sweep_pipelined<false>, dim3(1), dim3(1024), 0, st, (int)sg.count, g.steps.p + sg.first,
                                       g.slots.p, g.ents.p, g.invd.p, g.packed.p, x, w);
                }
            }
        }
        SLP_HIP(hipGetLastError());
        return;
    }
    for (int s = 0; s < sweeps; ++s)
        for (i64 l = 0; l < g.nlevels; ++l) level_launch(l);
    SLP_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// ADMM kernels
// q_j = (-c_j) + gamma_eq * (A^T b)_j    (constant part of :148)
__global__ void k_admm_q(i64 n, const double *__restrict__ c, const double *__restrict__ atb, double gamma_eq,
                         double *__restrict__ q) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
        q[j] = (-c[j]) + gamma_eq * atb[j];
}

// y_j = ((q_j + gamma_ineq * xp_j) - (A^T lambda)_j) - lambda_ineq_j, lambda_ineq == 0 on this branch (:148)
template <int L>
__global__ __launch_bounds__(kBlock) void k_admm_rhs(i64 n, const i64 *__restrict__ tptr, const i32 *__restrict__ tidx,
                                                     const double *__restrict__ tval, const double *__restrict__ lam,
                                                     const double *__restrict__ q, const double *__restrict__ xp,
                                                     double gamma_ineq, const double *__restrict__ lin, double *__restrict__ y) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 j = group; j < n; j += ngroups) {
        const double s = row_dot<L>(tptr, tidx, tval, lam, j, sub);
        if (sub == 0) y[j] = ((q[j] + gamma_ineq * xp[j]) - s) - lin[j];  // lambda_ineq stays 0 on the shipped branch
    }
}

// lambda_i += gamma_eq * ((A x)_i - b_i)   (:261-263)
template <int L>
__global__ __launch_bounds__(kBlock) void k_admm_multiplier(i64 m, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                            const double *__restrict__ val, const double *__restrict__ x,
                                                            const double *__restrict__ b, double gamma_eq,
                                                            double *__restrict__ lam) {
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    for (i64 i = group; i < m; i += ngroups) {
        const double ax = row_dot<L>(ptr, idx, val, x, i, sub);
        if (sub == 0) lam[i] = lam[i] + gamma_eq * (ax - b[i]);
    }
}

// report partials (:124-132,:220-222): per workgroup
//  part[0] sum r^2  part[1] sum lambda r  part[2] max |r|      (rows,  r = A x - b)
template <int L>
__global__ __launch_bounds__(kBlock) void k_admm_report_rows(i64 m, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                             const double *__restrict__ val, const double *__restrict__ x,
                                                             const double *__restrict__ b, const double *__restrict__ lam,
                                                             double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    const int sub = threadIdx.x & (L - 1);
    const i64 group = ((i64)blockIdx.x * kBlock + threadIdx.x) / L;
    const i64 ngroups = (i64)gridDim.x * kBlock / L;
    double s0 = 0.0, s1 = 0.0, mx = -__builtin_inf();
    const i64 rounds = (m + ngroups - 1) / ngroups;
    for (i64 it = 0; it < rounds; ++it) {
        const i64 i = group + it * ngroups;
        if (i < m) {
            const double ax = row_dot<L>(ptr, idx, val, x, i, sub);
            if (sub == 0) {
                const double r = ax - b[i];
                s0 += r * r;
                s1 += lam[i] * r;
                const double a = fabs(r);
                mx = a > mx ? a : mx;
            }
        }
    }
    const double r0 = block_reduce<false>(s0, lds), r1 = block_reduce<false>(s1, lds), r2 = block_reduce<true>(mx, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = r0;
        part[blockIdx.x * 3 + 1] = r1;
        part[blockIdx.x * 3 + 2] = r2;
    }
}

//  part[0] sum c x  part[1] sum (x-xp)^2  part[2] max(-x)      (columns)
__global__ __launch_bounds__(kBlock) void k_admm_report_cols(i64 n, const double *__restrict__ c, const double *__restrict__ x,
                                                             const double *__restrict__ xp, double *__restrict__ part) {
    __shared__ double lds[kBlock / kWave];
    double s0 = 0.0, s1 = 0.0, mx = -__builtin_inf();
    for (i64 j = (i64)blockIdx.x * kBlock + threadIdx.x; j < n; j += (i64)gridDim.x * kBlock) {
        const double xj = x[j], dx = xj - xp[j];
        s0 += c[j] * xj;
        s1 += dx * dx;
        mx = (-xj) > mx ? (-xj) : mx;
    }
    const double r0 = block_reduce<false>(s0, lds), r1 = block_reduce<false>(s1, lds), r2 = block_reduce<true>(mx, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = r0;
        part[blockIdx.x * 3 + 1] = r1;
        part[blockIdx.x * 3 + 2] = r2;
    }
}

__global__ __launch_bounds__(kBlock) void k_admm_report_final(int nr, const double *__restrict__ rp, int nc,
                                                              const double *__restrict__ cp, double *__restrict__ out) {
    __shared__ double lds[kBlock / kWave];
    double a0 = 0.0, a1 = 0.0, a2 = -__builtin_inf(), b0 = 0.0, b1 = 0.0, b2 = -__builtin_inf();
    for (int i = threadIdx.x; i < nr; i += kBlock) {
        a0 += rp[i * 3];
        a1 += rp[i * 3 + 1];
        a2 = rp[i * 3 + 2] > a2 ? rp[i * 3 + 2] : a2;
    }
    for (int i = threadIdx.x; i < nc; i += kBlock) {
        b0 += cp[i * 3];
        b1 += cp[i * 3 + 1];
        b2 = cp[i * 3 + 2] > b2 ? cp[i * 3 + 2] : b2;
    }
    const double r0 = block_reduce<false>(a0, lds), r1 = block_reduce<false>(a1, lds), r2 = block_reduce<true>(a2, lds);
    const double r3 = block_reduce<false>(b0, lds), r4 = block_reduce<false>(b1, lds), r5 = block_reduce<true>(b2, lds);
    if (threadIdx.x == 0) {
        out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; out[4] = r4; out[5] = r5;
    }
}

// unbounded-GS branch: x = alpha x + (1 - alpha) xp   (ADMM.py:181)
__global__ void k_admm_relax(i64 n, double alpha, double one_minus_alpha, const double *__restrict__ xp, double *__restrict__ x) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
        x[j] = alpha * x[j] + one_minus_alpha * xp[j];
}

// xp = clip(x + lambda_ineq / g_ineq, lb, ub) ; lambda_ineq += g_ineq (x - xp)   (ADMM.py:253-256)
__global__ void k_admm_project(i64 n, double gamma_ineq, const double *__restrict__ x, const double *__restrict__ lb,
                               const double *__restrict__ ub, double *__restrict__ xp, double *__restrict__ lin) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
        const double xj = x[j];
        double p = xj + lin[j] / gamma_ineq;
        const double l = lb[j], u = ub[j];
        p = (p < l) ? l : p;
        p = (p > u) ? u : p;
        xp[j] = p;
        lin[j] = lin[j] + gamma_ineq * (xj - p);
    }
}

// sum_j lambda_ineq_j (x_j - xp_j), one workgroup, fixed order (report term of the unbounded-GS branch)
__global__ __launch_bounds__(kBlock) void k_admm_lin_term(i64 n, const double *__restrict__ lin, const double *__restrict__ x,
                                                          const double *__restrict__ xp, double *__restrict__ out) {
    __shared__ double lds[kBlock / kWave];
    double s = 0.0;
    for (i64 j = threadIdx.x; j < n; j += kBlock) s += lin[j] * (x[j] - xp[j]);
    const double r = block_reduce<false>(s, lds);
    if (threadIdx.x == 0) out[0] = r;
}

__global__ void k_max0(i64 n, const double *__restrict__ x, double *__restrict__ xp) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x)
        xp[j] = (x[j] < 0.0) ? 0.0 : x[j];  // np.maximum(x, 0)  (ADMM.py:98)
}

}  // namespace slp

using namespace slp;

struct slp_gs {
    GsPlan plan;
    DevBuf<double> b, lo, hi, x;
};

struct slp_admm {
    slp_matrix *a = nullptr;  // standard-form constraint matrix (m x N)
    GsPlan plan;              // M
    i64 N = 0, m = 0;
    double gamma_eq = 2, gamma_ineq = 3;
    int order = SLP_ORDER_AUTO, lanes_rows = 1, lanes_cols = 1;
    bool xp_is_x = false;     // false only before the first multiplier step (:98 vs :259)
    int xstep = 0;            // 0: projected Gauss-Seidel (shipped flags); 1: plain Gauss-Seidel + over-relaxation (:164-181)
    double alpha = 1.4;       // :140
    DevBuf<double> lin;       // lambda_ineq (stays 0 for xstep 0)
    IterGraph graph;
    DevBuf<double> b, c, lb, ub, x, xp0, lam, q, y, rowparts, colparts, out;
};

namespace slp {
constexpr int kAdmmPartials = 4096;

static void admm_sweep(slp_admm *s) {
    hipStream_t st = ctx().stream;
    const CsrDev &at = s->a->at;
    const double *xp = s->xp_is_x ? s->x.p : s->xp0.p;
    const int lanes = s->lanes_cols;
    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_admm_rhs<L>), dim3(grid_for(s->N * lanes, kBlock)), dim3(kBlock), 0, st, s->N,
                                                 at.ptr.p, at.idx.p, at.val.p, s->lam.p, s->q.p, xp, s->gamma_ineq, s->lin.p, s->y.p));
    SLP_HIP(hipGetLastError());
    if (s->xstep == 0) {
        gs_sweep(s->plan, s->y.p, s->lb.p, s->ub.p, s->x.p, 1.0, 1);  // :162 maxiter=1, w=1
    } else {
        gs_sweep(s->plan, s->y.p, nullptr, nullptr, s->x.p, 1.0, 1, false);  // :179 GaussSeidel(m, y, x, maxiter=1, w=1.0)
        hipLaunchKernelGGL(k_admm_relax, dim3(grid_for(s->N, kBlock)), dim3(kBlock), 0, st, s->N, s->alpha, 1.0 - s->alpha, s->xp0.p,
                           s->x.p);  // :181
        SLP_HIP(hipGetLastError());
    }
}

static void admm_multiplier(slp_admm *s) {
    const CsrDev &a = s->a->a;
    if (s->xstep == 0) {
        s->xp_is_x = true;  // :259
    } else if (s->N) {       // :253-256: xp stays a vector of its own
        hipLaunchKernelGGL(k_admm_project, dim3(grid_for(s->N, kBlock)), dim3(kBlock), 0, ctx().stream, s->N, s->gamma_ineq, s->x.p,
                           s->lb.p, s->ub.p, s->xp0.p, s->lin.p);
        SLP_HIP(hipGetLastError());
    }
    if (s->m == 0) return;
    const int lanes = s->lanes_rows;
    SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_admm_multiplier<L>), dim3(grid_for(s->m * lanes, kBlock)), dim3(kBlock), 0,
                                                 ctx().stream, s->m, a.ptr.p, a.idx.p, a.val.p, s->x.p, s->b.p, s->gamma_eq,
                                                 s->lam.p));
    SLP_HIP(hipGetLastError());
}
}  // namespace slp

extern "C" {

slp_gs *slp_gs_create(int64_t n, const int64_t *indptr, const int32_t *indices, const double *data) {
    SLP_API_PTR({
        ctx();
        auto *g = new slp_gs();
        try {
            SLP_REQUIRE(n >= 0 && indptr, "gauss-seidel: bad arguments");
            SLP_REQUIRE(n < (i64)1 << 31, "gauss-seidel: dimension must fit int32");
            if (gs_plan_mode() == 0 || n == 0 || (gs_plan_mode() == 1 && indptr[n] <= kGsSmallPlan)) {
                gs_plan(g->plan, n, indptr, indices, data);
            } else {
                const i64 nnz = indptr[n];
                DevBuf<i64> dp;
                DevBuf<i32> dj;
                DevBuf<double> dx;
                dp.upload(indptr, (size_t)n + 1);
                dj.upload(indices, (size_t)nnz);
                dx.upload(data, (size_t)nnz);
                gs_plan_any(g->plan, n, nnz, dp.p, dj.p, dx.p);
            }
            g->b.alloc((size_t)n); g->lo.alloc((size_t)n); g->hi.alloc((size_t)n); g->x.alloc((size_t)n);
            SLP_HIP(hipStreamSynchronize(ctx().stream));
        } catch (...) { delete g; throw; }
        return g;
    })
}

void slp_gs_destroy(slp_gs *g) { delete g; }

int64_t slp_gs_num_levels(const slp_gs *g) { return g ? g->plan.nlevels : -1; }

int slp_gs_sweep_kind(const slp_gs *g) {
    if (!g) return -1;
    return g->plan.one_block ? 1 : (g->plan.pipelined ? (g->plan.windowed ? 3 : 2) : 0);
}

int slp_gs_num_bands(const slp_gs *g) { return g ? g->plan.nbands : -1; }

int slp_gs_solve(slp_gs *g, const double *b, const double *lower, const double *upper, double *x, int maxiter, double w) {
    SLP_API_INT({
        SLP_REQUIRE(g && b && lower && upper && x, "slp_gs_solve: NULL argument");
        const size_t n = (size_t)g->plan.n;
        g->b.upload(b, n); g->lo.upload(lower, n); g->hi.upload(upper, n); g->x.upload(x, n);
        gs_sweep(g->plan, g->b.p, g->lo.p, g->hi.p, g->x.p, w, maxiter);
        g->x.download(x, n);
    })
}

// everything after s->a, b, c, lb, ub, x are on the device: transposed copy, M's level schedule (M from the host arrays or,
// without them, formed on the device), the constant part of the right-hand side
static void admm_finish_create(slp_admm *s, const int64_t *m_indptr, const int32_t *m_indices, const double *m_data) {
    const i64 N = s->N, m = s->m;
    build_transpose(s->a);
    s->lanes_rows = lanes_for(s->a->a, s->order);
    s->lanes_cols = lanes_for(s->a->at, s->order);
    if (m_indptr && (gs_plan_mode() == 0 || N == 0 || (gs_plan_mode() == 1 && m_indptr[N] <= kGsSmallPlan))) {
        Phase ph("gs_plan (host level schedule)");
        gs_plan(s->plan, N, m_indptr, m_indices, m_data);
    } else if (m_indptr) {
        const i64 nnz = m_indptr[N];
        DevBuf<i64> dp;
        DevBuf<i32> dj;
        DevBuf<double> dx;
        dp.upload(m_indptr, (size_t)N + 1);
        dj.upload(m_indices, (size_t)nnz);
        dx.upload(m_data, (size_t)nnz);
        gs_plan_any(s->plan, N, nnz, dp.p, dj.p, dx.p);
    } else {
        // M = gamma_eq A^T A + gamma_ineq I formed on the device (slp_spgemm.hip, SMMP accumulation order) and planned where it
        // lies (gs_plan_device: levels, level order, lane records on the device; SLP_GS_PLAN=host: from one download of M)
        slp_matrix *mm = slp_matrix_normal(s->a, s->gamma_eq, s->gamma_ineq);
        if (!mm) throw Error(slp_last_error());
        try {
            gs_plan_any(s->plan, N, mm->a.nnz, mm->a.ptr.p, mm->a.idx.p, mm->a.val.p);
        } catch (...) { delete mm; throw; }
        delete mm;
    }
    s->xp0.alloc((size_t)N); s->lam.alloc((size_t)m); s->lam.zero();
    s->lin.alloc((size_t)N); s->lin.zero();
    s->q.alloc((size_t)N); s->y.alloc((size_t)N);
    s->rowparts.alloc((size_t)kAdmmPartials * 3); s->colparts.alloc((size_t)kAdmmPartials * 3); s->out.alloc(8);
    hipStream_t st = ctx().stream;
    if (N) {
        hipLaunchKernelGGL(k_max0, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, st, N, s->x.p, s->xp0.p);
        // A^T b in the reference's accumulation order (ADMM.py:95), then q
        launch_spmv(s->a->at, s->b.p, s->y.p, s->order == SLP_ORDER_AUTO ? SLP_ORDER_AUTO : s->order);
        hipLaunchKernelGGL(k_admm_q, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, st, N, s->c.p, s->y.p, s->gamma_eq, s->q.p);
        SLP_HIP(hipGetLastError());
    }
    SLP_HIP(hipStreamSynchronize(st));
}

slp_admm *slp_admm_create(int64_t N, int64_t m, const int64_t *a_indptr, const int32_t *a_indices, const double *a_data,
                          const double *b, const double *c, const double *lb, const double *ub, const double *x0,
                          const int64_t *m_indptr, const int32_t *m_indices, const double *m_data, double gamma_eq,
                          double gamma_ineq, int order) {
    SLP_API_PTR({
        SLP_REQUIRE(a_indptr && b && c && lb && ub && x0, "slp_admm_create: NULL argument");
        auto *s = new slp_admm();
        try {
            s->a = slp_matrix_create(m, N, a_indptr, a_indices, a_data);
            if (!s->a) throw Error(slp_last_error());
            s->N = N; s->m = m; s->gamma_eq = gamma_eq; s->gamma_ineq = gamma_ineq; s->order = order;
            s->b.upload(b, (size_t)m); s->c.upload(c, (size_t)N); s->lb.upload(lb, (size_t)N); s->ub.upload(ub, (size_t)N);
            s->x.upload(x0, (size_t)N);
            admm_finish_create(s, m_indptr, m_indices, m_data);
        } catch (...) { slp_admm_destroy(s); throw; }
        return s;
    })
}

__global__ void k_fill_const(i64 n, double v, double *__restrict__ p) {
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) p[j] = v;
}

// The whole setup chain of lp_admm on the device (ADMM.py:73-101): the two constraint blocks are uploaded as the caller
// holds them, then -- row normalisation of each block (tools.py:272-290), slack standard form (tools.py:88-127), row
// normalisation of the stacked system, M (slp_matrix_normal), A^T b -- run in HBM with the reference's entry orders and
// accumulation orders, so the state equals the host-prepared one bit for bit.
slp_admm *slp_admm_create_lp(int64_t n, int64_t m_eq, const int64_t *eq_indptr, const int32_t *eq_indices, const double *eq_data,
                             const double *b_eq, int64_t m_ineq, const int64_t *in_indptr, const int32_t *in_indices,
                             const double *in_data, const double *b_lower, const double *b_upper, const double *c, const double *lb,
                             const double *ub, const double *x0, double gamma_eq, double gamma_ineq, int use_preconditioning, int order) {
    SLP_API_PTR({
        SLP_REQUIRE(n >= 0 && m_eq >= 0 && m_ineq >= 0 && c && lb && ub, "slp_admm_create_lp: bad arguments");
        SLP_REQUIRE(in_indptr, "slp_admm_create_lp: the inequality block is required (the reference's standard form is undefined "
                               "without it, tools.py:92)");
        SLP_REQUIRE(m_eq == 0 || (eq_indptr && b_eq), "slp_admm_create_lp: NULL equality block");
        Phase ph("slp_admm_create_lp (total)");
        hipStream_t st = ctx().stream;
        auto *s = new slp_admm();
        slp_matrix *ae = nullptr, *ai = nullptr, *ae2 = nullptr, *ai2 = nullptr, *a2 = nullptr;
        auto drop = [&]() { delete ae; delete ai; delete ae2; delete ai2; delete a2; ae = ai = ae2 = ai2 = a2 = nullptr; };
        try {
            const i64 m = m_eq + m_ineq, N = n + m_ineq;
            s->N = N; s->m = m; s->gamma_eq = gamma_eq; s->gamma_ineq = gamma_ineq; s->order = order;
            if (eq_indptr) {  // a 0-row equality block stays a block, like `a_eq is not None` in the reference
                ae = slp_matrix_create(m_eq, n, eq_indptr, eq_indices, eq_data);
                if (!ae) throw Error(slp_last_error());
            }
            ai = slp_matrix_create(m_ineq, n, in_indptr, in_indices, in_data);
            if (!ai) throw Error(slp_last_error());
            // right-hand sides and bounds of the stacked system: b = [b_eq; 0], lb = [lb; b_lower], ub = [ub; b_upper], c = [c; 0]
            s->b.alloc((size_t)m); s->b.zero();
            s->c.alloc((size_t)N); s->c.zero();
            s->lb.alloc((size_t)N); s->ub.alloc((size_t)N); s->x.alloc((size_t)N); s->x.zero();
            if (m_eq) SLP_HIP(hipMemcpyAsync(s->b.p, b_eq, (size_t)m_eq * sizeof(double), hipMemcpyHostToDevice, st));
            if (n) {
                SLP_HIP(hipMemcpyAsync(s->c.p, c, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
                SLP_HIP(hipMemcpyAsync(s->lb.p, lb, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
                SLP_HIP(hipMemcpyAsync(s->ub.p, ub, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
                if (x0) SLP_HIP(hipMemcpyAsync(s->x.p, x0, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            }
            if (m_ineq) {
                if (b_lower) SLP_HIP(hipMemcpyAsync(s->lb.p + n, b_lower, (size_t)m_ineq * sizeof(double), hipMemcpyHostToDevice, st));
                else hipLaunchKernelGGL(k_fill_const, dim3(grid_for(m_ineq, kBlock)), dim3(kBlock), 0, st, m_ineq, -__builtin_inf(), s->lb.p + n);
                if (b_upper) SLP_HIP(hipMemcpyAsync(s->ub.p + n, b_upper, (size_t)m_ineq * sizeof(double), hipMemcpyHostToDevice, st));
                else hipLaunchKernelGGL(k_fill_const, dim3(grid_for(m_ineq, kBlock)), dim3(kBlock), 0, st, m_ineq, __builtin_inf(), s->ub.p + n);
            }
            SLP_HIP(hipStreamSynchronize(st));  // the host buffers may go away
            // ADMM.py:76-83: scale the rows of each block (the slack bounds are scaled with their rows; -inf / inf stay)
            if (ae) ae2 = matrix_precondition_rows(ae, s->b.p, nullptr);
            ai2 = matrix_precondition_rows(ai, s->lb.p + n, s->ub.p + n);
            // :84-86: [A_eq 0; A_ineq -I]; x0 = [x0; A_ineq x0] with the scaled block (csr_matvec over its stored order)
            a2 = matrix_standard_form(ae2, ai2);
            if (m_ineq) launch_spmv(ai2->a, s->x.p, s->x.p + n, SLP_ORDER_SEQUENTIAL);
            // :90-91: scale the rows of the stacked system
            if (use_preconditioning) {
                s->a = matrix_precondition_rows(a2, s->b.p, nullptr);
            } else {
                s->a = a2;
                a2 = nullptr;
            }
            SLP_HIP(hipStreamSynchronize(st));
            drop();
            admm_finish_create(s, nullptr, nullptr, nullptr);
        } catch (...) { drop(); slp_admm_destroy(s); throw; }
        return s;
    })
}

void slp_admm_destroy(slp_admm *s) {
    if (!s) return;
    delete s->a;
    delete s;
}

int slp_admm_iterate(slp_admm *s, int64_t k) {
    SLP_API_INT({
        SLP_REQUIRE(s && k >= 0, "slp_admm_iterate: bad arguments");
        auto one = [&]() { admm_sweep(s); admm_multiplier(s); };
        if (k > 0 && !s->xp_is_x && s->xstep == 0) {  // the first iteration reads xp0 instead of x: not part of the replayed graph
            one();
            --k;
        }
        // one launch per dependency level is launch-latency bound: replay the iteration as a captured graph
        if (s->a->a.nnz <= 20000000) s->graph.run(k, (s->plan.pipelined ? (i64)s->plan.segments.size() : s->plan.nlevels) > 64 || s->plan.one_block ? 1 : 8, one);
        else for (i64 it = 0; it < k; ++it) one();
    })
}

int slp_admm_set_xstep(slp_admm *s, int mode) {
    SLP_API_INT({
        SLP_REQUIRE(s && (mode == 0 || mode == 1), "slp_admm_set_xstep: mode must be 0 or 1");
        SLP_REQUIRE(!s->xp_is_x, "slp_admm_set_xstep: must be called before the first iteration");
        s->xstep = mode;
        s->graph.reset();
    })
}

int slp_admm_sweep_step(slp_admm *s) { SLP_API_INT({ SLP_REQUIRE(s, "NULL handle"); admm_sweep(s); }) }

int slp_admm_multiplier_step(slp_admm *s) { SLP_API_INT({ SLP_REQUIRE(s, "NULL handle"); admm_multiplier(s); }) }

int slp_admm_report(slp_admm *s, double out[3]) {
    SLP_API_INT({
        SLP_REQUIRE(s && out, "slp_admm_report: NULL argument");
        hipStream_t st = ctx().stream;
        const CsrDev &a = s->a->a;
        const double *xp = s->xp_is_x ? s->x.p : s->xp0.p;
        int gc = std::min(grid_for(s->N, kBlock), kAdmmPartials);
        hipLaunchKernelGGL(k_admm_report_cols, dim3(gc), dim3(kBlock), 0, st, s->N, s->c.p, s->x.p, xp, s->colparts.p);
        const int lanes = s->lanes_rows;
        int gr = std::min(grid_for(s->m * lanes, kBlock), kAdmmPartials);
        SLP_DISPATCH_LANES(lanes, hipLaunchKernelGGL((k_admm_report_rows<L>), dim3(gr), dim3(kBlock), 0, st, s->m, a.ptr.p, a.idx.p,
                                                     a.val.p, s->x.p, s->b.p, s->lam.p, s->rowparts.p));
        hipLaunchKernelGGL(k_admm_report_final, dim3(1), dim3(kBlock), 0, st, gr, s->rowparts.p, gc, s->colparts.p, s->out.p);
        SLP_HIP(hipGetLastError());
        double h[7];
        h[6] = 0.0;
        if (s->xstep != 0 && s->N) {
            hipLaunchKernelGGL(k_admm_lin_term, dim3(1), dim3(kBlock), 0, st, s->N, s->lin.p, s->x.p, xp, s->out.p + 6);
            SLP_HIP(hipGetLastError());
        }
        s->out.download(h, s->xstep != 0 ? 7 : 6);
        // c.x + 0.5*g_eq*sum r^2 + 0.5*g_ineq*sum (x-xp)^2 + lambda.r + lambda_ineq.(x-xp)   (:124-132)
        out[0] = h[3] + 0.5 * s->gamma_eq * h[0] + 0.5 * s->gamma_ineq * h[4] + h[1] + h[6];
        out[1] = h[2];                                // :221
        out[2] = (h[5] > 0.0) ? h[5] : 0.0;           // :222 max(0, -min x)
    })
}

int slp_admm_get_x(slp_admm *s, double *x, int64_t count) {
    SLP_API_INT({ SLP_REQUIRE(s && x && count >= 0 && count <= s->N, "slp_admm_get_x: bad arguments"); s->x.download(x, (size_t)count); })
}

int slp_admm_get_lambda(slp_admm *s, double *lam) {
    SLP_API_INT({ SLP_REQUIRE(s && lam, "NULL argument"); s->lam.download(lam, (size_t)s->m); })
}

int64_t slp_admm_num_levels(const slp_admm *s) { return s ? s->plan.nlevels : -1; }

int slp_admm_num_bands(const slp_admm *s) { return s ? s->plan.nbands : -1; }

int slp_admm_bench(slp_admm *s, int64_t k, double *ms) {
    SLP_API_INT({
        SLP_REQUIRE(s && k > 0 && ms, "slp_admm_bench: bad arguments");
        Context &c = ctx();
        SLP_HIP(hipEventRecord(c.ev0, c.stream));
        for (i64 it = 0; it < k; ++it) { admm_sweep(s); admm_multiplier(s); }
        SLP_HIP(hipEventRecord(c.ev1, c.stream));
        SLP_HIP(hipEventSynchronize(c.ev1));
        float f = 0.f;
        SLP_HIP(hipEventElapsedTime(&f, c.ev0, c.ev1));
        *ms = (double)f / (double)k;
    })
}

}  // extern "C"

// slp_gs_plan_device.h -- the Gauss-Seidel sweep's plan built ON THE DEVICE (included by slp_admm.hip behind gs_plan, whose host
// form it reproduces array for array).  Reference: gaussSiedel.pyx:87-92 -- the reference's constructor only inverts a diagonal;
// everything here exists because the sweep (gaussSiedel.pyx:131-152) is run level by level instead of row by row.
//
// What runs where.  On the device, from M as it lies in HBM (slp_matrix_normal's output -- no download of M): the dependency
// levels (Kahn's sweep over the symmetrised pattern: a launch per wide level, one workgroup with a barrier per level through
// runs of narrow ones), the rows that nothing waits for, the stable sort of the rows by level, the bands' row ranges, external
// references and requirement rows, the matrix permuted into level order, its inverted diagonal, the packing of every level's rows
// into lane slots, the per-lane-slot records (entries, entry classes, ring cells) and the fetch lists.  On the host, from tables
// with one entry per level, per (band, level) or per wave slot -- never per row or per entry (Potts 256^2: < 0.5 MB down, < 0.2 MB
// up, against 27 MB down and ~50 MB up for the host plan): which levels are swept chip-wide and which in one workgroup, the
// bands' timing model, and the per-wave header lists.  Measured (tools/gs_plan_timing.py): lp_admm(nb_iter=0) on Potts 256^2
// 50-60 -> 19.5 ms, on Potts 512^2 215 -> 48 ms; SLP_GS_PLAN=check requires every array of this plan to equal the host plan's.
#pragma once

namespace slp {

constexpr int kGspBlock = 256;

__device__ __forceinline__ unsigned long long wave_reduce_add_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---- levels ---------------------------------------------------------------------------------------------------------------
__global__ void k_gsp_validate(i64 n, i64 nnz, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, int *__restrict__ bad) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 s = ptr[i], e = ptr[i + 1];
        if (s < 0 || e < s || e > nnz || (i == 0 && s != 0) || (i == n - 1 && e != nnz)) { atomicOr(bad, 2); continue; }
        for (i64 k = s; k < e; ++k)
            if (idx[k] < 0 || idx[k] >= n) atomicOr(bad, 1);
    }
}

__global__ void k_gsp_rowid(i64 n, const i64 *__restrict__ ptr, i32 *__restrict__ rowid) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) rowid[k] = (i32)i;
}

// tptr[c] = first sorted position whose key is >= c (keys sorted ascending)
__global__ void k_gsp_ptr_from_sorted(i64 nnz, i64 ncol, const unsigned int *__restrict__ key, i64 *__restrict__ tptr) {
    if (nnz == 0) {
        for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j <= ncol; j += (i64)gridDim.x * blockDim.x) tptr[j] = 0;
        return;
    }
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (i64)gridDim.x * blockDim.x) {
        const i64 c = key[p], prev = p > 0 ? (i64)key[p - 1] : -1;
        for (i64 j = prev + 1; j <= c; ++j) tptr[j] = p;
        if (p == nnz - 1)
            for (i64 j = c + 1; j <= ncol; ++j) tptr[j] = nnz;
    }
}

// level(i) = 1 + the largest level among the rows j < i coupled to i through M or M^T (0 without one): the longest path in the
// dependency graph, level by level (Kahn): indeg(i) = lower neighbours of i (with multiplicity: an entry of row i and the mirror
// entry of column i count one each, and are taken off one each); the rows of level l take themselves off their higher
// neighbours' counts, and a row whose count reaches 0 is of level l + 1.  One launch per level over that level's rows only
// (a first version let every row's thread wait for its neighbours' levels in one launch: 133 ms on the Potts 256^2 matrix,
// 457 216 threads polling through 512 levels).  order[] collects the rows level after level (inside a level in no particular
// order: the stable sort by level below fixes the order the sweep uses).
__global__ void k_gsp_indeg(i64 n, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const i64 *__restrict__ tptr,
                            const i32 *__restrict__ trow, int *__restrict__ indeg, i32 *__restrict__ level, i32 *__restrict__ order,
                            int *__restrict__ counts) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        int d = 0;
        for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) d += idx[k] < i ? 1 : 0;
        for (i64 k = tptr[i]; k < tptr[i + 1]; ++k) d += trow[k] < i ? 1 : 0;
        indeg[i] = d;
        if (d == 0) {
            level[i] = 0;
            order[atomicAdd(&counts[0], 1)] = (i32)i;
        }
    }
}

// offs[l] = first position of level l in order[]; counts[l] = its rows (both final when the launch for level l starts)
__global__ void k_gsp_frontier(int l, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, const i64 *__restrict__ tptr,
                               const i32 *__restrict__ trow, int *__restrict__ indeg, i32 *__restrict__ level, i32 *__restrict__ order,
                               int *__restrict__ counts, i64 *__restrict__ offs) {
    const i64 first = offs[l], cnt = counts[l], next = first + cnt;
    if (blockIdx.x == 0 && threadIdx.x == 0) offs[l + 1] = next;
    for (i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x; f < cnt; f += (i64)gridDim.x * blockDim.x) {
        const i32 r = order[first + f];
        for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) {
            const i32 j = idx[k];
            if (j > r && atomicSub(&indeg[j], 1) == 1) {
                level[j] = l + 1;
                order[next + atomicAdd(&counts[l + 1], 1)] = j;
            }
        }
        for (i64 k = tptr[r]; k < tptr[r + 1]; ++k) {
            const i32 j = trow[k];
            if (j > r && atomicSub(&indeg[j], 1) == 1) {
                level[j] = l + 1;
                order[next + atomicAdd(&counts[l + 1], 1)] = j;
            }
        }
    }
}

// The same for a RUN of narrow levels in ONE workgroup: while a level has at most kGspNarrow rows the workgroup takes level after
// level with a barrier in between (a launch per level costs ~6 us, a barrier ~1: the Potts 256^2 matrix has 511 levels of
// ~900 rows).  Stops at the first wider (or empty) level, or after `max_levels`; *reached = the level it stopped in front of.
constexpr int kGspNarrow = 4096;
__global__ __launch_bounds__(1024) void k_gsp_frontier_run(int l0, int max_levels, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                                                           const i64 *__restrict__ tptr, const i32 *__restrict__ trow, int *indeg, i32 *level,
                                                           i32 *order, int *counts, i64 *offs, int *__restrict__ reached) {
    int l = l0;
    for (int it = 0; it < max_levels; ++it, ++l) {
        // (agent-scope loads: what the previous level's atomics and stores left must come from the L2, not from a line this
        // compute unit cached while its neighbours in the line were read a level earlier)
        const i64 first = __hip_atomic_load(&offs[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const i64 cnt = __hip_atomic_load(&counts[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cnt == 0 || cnt > kGspNarrow) break;
        const i64 next = first + cnt;
        for (i64 f = threadIdx.x; f < cnt; f += blockDim.x) {
            const i32 r = __hip_atomic_load(&order[first + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) {
                const i32 j = idx[k];
                if (j > r && atomicSub(&indeg[j], 1) == 1) {
                    level[j] = l + 1;
                    __hip_atomic_store(&order[next + atomicAdd(&counts[l + 1], 1)], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            for (i64 k = tptr[r]; k < tptr[r + 1]; ++k) {
                const i32 j = trow[k];
                if (j > r && atomicSub(&indeg[j], 1) == 1) {
                    level[j] = l + 1;
                    __hip_atomic_store(&order[next + atomicAdd(&counts[l + 1], 1)], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (threadIdx.x == 0) __hip_atomic_store(&offs[l + 1], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
    }
    if (threadIdx.x == 0) *reached = l;
}

// coupled[min(i, j)] = 1 for every stored entry (i, j), j != i: the row has a coupled row of higher index
__global__ void k_gsp_coupled(i64 n, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, unsigned char *__restrict__ coupled) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x)
        for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) {
            const i64 j = idx[k];
            if (j != i) coupled[j < i ? j : i] = 1;
        }
}

// counts the rows nothing waits for (not coupled upwards, level > 0)
__global__ void k_gsp_count_sinks(i64 n, const unsigned char *__restrict__ coupled, const i32 *__restrict__ level,
                                  unsigned long long *__restrict__ sinks) {
    unsigned long long c = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) c += (!coupled[i] && level[i] > 0) ? 1 : 0;
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(sinks, c);
}

__global__ void k_gsp_move_sinks(i64 n, const unsigned char *__restrict__ coupled, i32 sink_level, i32 *__restrict__ level,
                                 i32 *__restrict__ present) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        if (!coupled[i] && level[i] > 0) level[i] = sink_level;
        present[level[i]] = 1;
    }
}

__global__ void k_gsp_remap(i64 n, const i32 *__restrict__ remap, i32 *__restrict__ level) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) level[i] = remap[level[i]];
}

__global__ void k_gsp_iota(i64 n, i32 *__restrict__ v) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) v[i] = (i32)i;
}

// pos[rows[t]] = t ; rlen[t] = entries of the row at position t
__global__ void k_gsp_positions(i64 n, const i32 *__restrict__ rows, const i64 *__restrict__ ptr, i32 *__restrict__ pos, i64 *__restrict__ rlen) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i32 i = rows[t];
        pos[i] = (i32)t;
        rlen[t] = ptr[i + 1] - ptr[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) rlen[n] = 0;
}

// the matrix in position order: entries of position t at p2[t] .. p2[t + 1]
__global__ void k_gsp_permute(i64 n, const i32 *__restrict__ rows, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                              const double *__restrict__ val, const i64 *__restrict__ p2, i32 *__restrict__ j2, double *__restrict__ v2) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 s = ptr[rows[t]], len = ptr[rows[t] + 1] - s, o = p2[t];
        for (i64 k = 0; k < len; ++k) {
            j2[o + k] = idx[s + k];
            v2[o + k] = val[s + k];
        }
    }
}

// ---- lane slots -----------------------------------------------------------------------------------------------------------
// A unit = the rows one workgroup sweeps between two barriers: a narrow level, or one band's share of a level.  The host lays
// the unit table out in the order the host plan walks (levels ascending; a run with bands: band after band, level after level).
struct GspUnit {
    i64 beg, end;        // row positions
    i64 slot0;           // first lane slot (filled in after the layout scan)
    int level;           // global level (plain unit) / the band's own level number (band unit)
    int seg_first;       // plain: first level of the run of narrow levels the unit belongs to
    int band;            // -1: plain unit; else the band
    int run;             // band unit: index of its run (band tables below)
    int fits_ring;       // the unit's results go to the LDS ring
    int pad;
};

// One thread per unit packs the unit's rows into lane slots, in position order (gs_plan's rules): a row's lanes stay inside one
// group of 16 lane slots; (plain units) at most 1024 lane slots between two barrier-free steps, every step padded to whole
// waves; the unit ends on a wave boundary.  slot_rel[t] = the row's first lane slot relative to the unit's; total[u] = its slots.
__global__ void k_gsp_layout(int nunits, const GspUnit *__restrict__ units, const i64 *__restrict__ rlen, i32 *__restrict__ slot_rel,
                             i64 *__restrict__ total) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nunits) return;
    const GspUnit un = units[u];
    i64 cur = 0, first = 0;
    for (i64 t = un.beg; t < un.end; ++t) {
        const i64 len = rlen[t];
        const i64 nl = len > kGsEntries ? (len + kGsEntries - 1) / kGsEntries : 1;
        i64 used = cur - first;
        if ((used & 15) + nl > 16) { cur += (16 - (used & 15)) & 15; used = cur - first; }
        if (un.band < 0 && used + nl > 1024) {   // next step of the same level: no barrier in between
            cur = first + ((used + 63) & ~(i64)63);
            first = cur;
        }
        slot_rel[t] = (i32)cur;
        cur += nl;
    }
    cur = first + ((cur - first + 63) & ~(i64)63);
    total[u] = cur;
}

__global__ void k_gsp_fill_idle(i64 S, GsEnt *__restrict__ ents, GsLane *__restrict__ lanes, i32 *__restrict__ lane_row) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < S; k += (i64)gridDim.x * blockDim.x) {
        GsEnt en;
        GsLane cd;
#pragma unroll
        for (int e = 0; e < kGsEntries; ++e) { en.idx[e] = 0; en.val[e] = 0.0; cd.code[e] = (unsigned short)kGsOne; }
        cd.info = 0;
        cd.ldsw = 0xffff;
        ents[k] = en;
        lanes[k] = cd;
        lane_row[k] = 0;
    }
}

// per run with bands (device tables): see gs_plan's BandRun
struct GspRun {
    i64 L0, L1;                 // global levels of the run
    int P;                      // bands
    int pad;
    const i64 *blptr;           // [sum over bands of (nlev + 1)] row positions of every band's levels; band p starts at blptr_off[p]
    const int *blptr_off;       // [P + 1]
    const i32 *ext;             // sorted distinct external rows, cell after cell
    const i64 *ext_ptr;         // [sum over bands of nlev + 1] cell (band, level) -> ext[ext_ptr[c] .. ext_ptr[c + 1]); cell = blptr_off[p] - p + ll
};

// The per-lane-slot records of every row of every unit: entries in storage order, the class of every entry (static: multiply with
// the ring cell that holds 1.0 -- the term is folded before the kernel; near: the ring cell of a row updated in one of the last
// three levels; far: gathered; external, bands: the cell the fetch wave fills), the lane's place in its row, where the row's result
// goes in the ring.  One thread per row position.
__global__ void k_gsp_records(i64 n, int nunits, const GspUnit *__restrict__ units, const GspRun *__restrict__ runs,
                              const i32 *__restrict__ rows, const i64 *__restrict__ ptr, const i32 *__restrict__ idx,
                              const double *__restrict__ val, const i32 *__restrict__ level, const i32 *__restrict__ pos,
                              const i64 *__restrict__ lptr, const i32 *__restrict__ band_of, const i32 *__restrict__ blev,
                              const i32 *__restrict__ slot_rel, GsEnt *__restrict__ ents, GsLane *__restrict__ lanes,
                              i32 *__restrict__ lane_row, int *__restrict__ has_far) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        int lo = 0, hi = nunits - 1, u = -1;   // the unit whose position range holds t (units are in position order)
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            if (units[mid].end <= t) lo = mid + 1;
            else if (units[mid].beg > t) hi = mid - 1;
            else { u = mid; break; }
        }
        if (u < 0) continue;   // a row of a level that is swept chip-wide
        const GspUnit un = units[u];
        const i32 i = rows[t];
        const i64 s = ptr[i], len = ptr[i + 1] - s;
        const int nl = (int)(len > kGsEntries ? (len + kGsEntries - 1) / kGsEntries : 1);
        const int ldsw = un.fits_ring ? (int)((un.level % kGsWinLevels) * kGsWide + (t - un.beg)) : -1;
        for (int j = 0; j < nl; ++j) {
            const i64 left = len - (i64)j * kGsEntries;
            const int cnt = (int)(left < 0 ? 0 : (left < kGsEntries ? left : kGsEntries));
            GsEnt en;
            GsLane cd;
#pragma unroll
            for (int e = 0; e < kGsEntries; ++e) { en.idx[e] = 0; en.val[e] = 0.0; cd.code[e] = (unsigned short)kGsOne; }
            unsigned info = (unsigned)j | (j == nl - 1 ? 16u : 0u) | ((unsigned)cnt << 5);
            for (int e = 0; e < cnt; ++e) {
                const i32 js = idx[s + (i64)j * kGsEntries + e];
                en.idx[e] = js;
                en.val[e] = val[s + (i64)j * kGsEntries + e];
                const i64 lj = level[js];
                if (un.band < 0) {
                    const i64 l = un.level;
                    if (lj < un.seg_first || lj >= l) {
                        // static: final before this run starts, or not touched before this row's step
                    } else if (lj > l - kGsWinLevels && lptr[lj + 1] - lptr[lj] <= kGsWide) {
                        cd.code[e] = (unsigned short)((lj % kGsWinLevels) * kGsWide + (pos[js] - lptr[lj]));   // near
                    } else {
                        info |= 1u << (8 + e);   // far: gathered from the results; the position is read from the entry record
                        en.idx[e] = pos[js];
                        *has_far = 1;
                    }
                } else {
                    const GspRun rn = runs[un.run];
                    if (js >= i || lj < rn.L0 || lj >= rn.L1) continue;   // static
                    const int p = un.band, ll = un.level;
                    const i64 *bl = rn.blptr + rn.blptr_off[p];
                    if (band_of[js] == p) {
                        const i32 bj = blev[js];
                        if (bj > ll - kGsWinLevels && bl[bj + 1] - bl[bj] <= kGsWide) {
                            cd.code[e] = (unsigned short)((bj % kGsWinLevels) * kGsWide + (pos[js] - bl[bj]));
                        } else {
                            info |= 1u << (8 + e);
                            en.idx[e] = pos[js];
                            *has_far = 1;
                        }
                    } else {   // a lower band's row: the fetch wave has put it into this level's generation
                        const i64 c = (i64)(rn.blptr_off[p] - p) + ll;
                        const i32 *ex = rn.ext + rn.ext_ptr[c];
                        int a = 0, b = (int)(rn.ext_ptr[c + 1] - rn.ext_ptr[c]);
                        while (a < b) {
                            const int mid = (a + b) >> 1;
                            if (ex[mid] < js) a = mid + 1;
                            else b = mid;
                        }
                        cd.code[e] = (unsigned short)(kGsExt + (ll & 1) * (kGsFetchK * 64) + a);
                    }
                }
            }
            cd.info = (unsigned short)info;
            cd.ldsw = (unsigned short)(ldsw >= 0 ? ldsw : 0xffff);
            const i64 slot = un.slot0 + slot_rel[t] + j;
            ents[slot] = en;
            lanes[slot] = cd;
            lane_row[slot] = (i32)t;
        }
    }
}

// per wave slot (64 lane slots): chain rounds | far mask << 8 -- all the header lists need of the lane records
__global__ void k_gsp_wave_summary(i64 nw, const GsLane *__restrict__ lanes, unsigned short *__restrict__ out) {
    const i64 w = ((i64)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x & 63;
    if (w >= nw) return;
    const unsigned info = lanes[w * 64 + lane].info;
    unsigned r = (info & 15u) + 1u, f = info >> 8;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned ro = __shfl_xor(r, d), fo = __shfl_xor(f, d);
        r = ro > r ? ro : r;
        f |= fo;
    }
    if (lane == 0) out[w] = (unsigned short)(r | ((f & 0xffu) << 8));
}

}  // namespace slp

// ---- bands (gs_plan's BandRun analysis) -----------------------------------------------------------------------------------------
namespace slp {

// per row position (level order): lane slots of the row; long rows mark their level
__global__ void k_gsp_long_levels(i64 n, const i32 *__restrict__ rows, const i64 *__restrict__ ptr, const i32 *__restrict__ level,
                                  unsigned char *__restrict__ long_level, i64 *__restrict__ lanes) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i32 i = rows[t];
        const i64 len = ptr[i + 1] - ptr[i];
        if (len > (i64)kGsMaxSeg * kGsEntries) long_level[level[i]] = 1;
        lanes[t] = len > kGsEntries ? (len + kGsEntries - 1) / kGsEntries : 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) lanes[n] = 0;
}

// out[l] = scan[lptr[l]] : the running lane count at every level boundary
__global__ void k_gsp_at_levels(i64 nlev1, const i64 *__restrict__ lptr, const i64 *__restrict__ scan, i64 *__restrict__ out) {
    for (i64 l = (i64)blockIdx.x * blockDim.x + threadIdx.x; l < nlev1; l += (i64)gridDim.x * blockDim.x) out[l] = scan[lptr[l]];
}

__global__ void k_gsp_row_lanes(i64 cnt, const i32 *__restrict__ rr, const i64 *__restrict__ ptr, i64 *__restrict__ rl) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) {
        const i64 len = ptr[rr[k] + 1] - ptr[rr[k]];
        rl[k] = len > kGsEntries ? (len + kGsEntries - 1) / kGsEntries : 1;
    }
}

// index ranges with equal numbers of lane slots; marks which (band, level) cells have rows
__global__ void k_gsp_band_of(i64 cnt, const i32 *__restrict__ rr, const i64 *__restrict__ acc, i64 lanes_total, int P, i64 l0, i64 L,
                              const i32 *__restrict__ level, i32 *__restrict__ band_of, i32 *__restrict__ ordm) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) {
        const i32 i = rr[k];
        i64 b = acc[k] * P / (lanes_total > 1 ? lanes_total : 1);
        b = b < P - 1 ? b : P - 1;
        band_of[i] = (i32)b;
        ordm[b * L + (level[i] - l0)] = 1;
    }
}

// per band: its own level numbers = the global levels it has rows in, counted up (one thread per band)
__global__ void k_gsp_band_levels(int P, i64 L, i32 *__restrict__ ordm, i32 *__restrict__ nlev) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    i32 cnt = 0;
    for (i64 l = 0; l < L; ++l) {
        const i32 has = ordm[(i64)p * L + l];
        ordm[(i64)p * L + l] = cnt;
        cnt += has;
    }
    nlev[p] = cnt;
}

// blev; lane slots and rows per (band, level) cell; how many references to lower bands' rows the row makes
__global__ void k_gsp_blev(i64 cnt, const i32 *__restrict__ rr, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, i64 l0, i64 l1, i64 L,
                           const i32 *__restrict__ level, const i32 *__restrict__ band_of, const i32 *__restrict__ ordm,
                           const i32 *__restrict__ cellbase, i32 *__restrict__ blev, unsigned long long *__restrict__ lanes_pl,
                           unsigned long long *__restrict__ rows_pl, i64 *__restrict__ nref) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) {
        const i32 i = rr[k], p = band_of[i];
        const i32 lv = ordm[(i64)p * L + (level[i] - l0)];
        blev[i] = lv;
        const i64 len = ptr[i + 1] - ptr[i];
        atomicAdd(&lanes_pl[cellbase[p] + lv], (unsigned long long)(len > kGsEntries ? (len + kGsEntries - 1) / kGsEntries : 1));
        atomicAdd(&rows_pl[cellbase[p] + lv], 1ull);
        i64 c = 0;
        for (i64 q = ptr[i]; q < ptr[i + 1]; ++q) {
            const i32 j = idx[q];
            if (j < i && level[j] >= l0 && level[j] < l1 && band_of[j] < p) ++c;
        }
        nref[k] = c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) nref[cnt] = 0;
}

__global__ void k_gsp_refs(i64 cnt, const i32 *__restrict__ rr, const i64 *__restrict__ ptr, const i32 *__restrict__ idx, i64 l0, i64 l1,
                           const i32 *__restrict__ level, const i32 *__restrict__ band_of, const i32 *__restrict__ blev,
                           const i64 *__restrict__ off, unsigned long long *__restrict__ refs) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) {
        const i32 i = rr[k], p = band_of[i], lv = blev[i];
        i64 o = off[k];
        for (i64 q = ptr[i]; q < ptr[i + 1]; ++q) {
            const i32 j = idx[q];
            if (j < i && level[j] >= l0 && level[j] < l1 && band_of[j] < p)
                refs[o++] = (unsigned long long)p << 60 | (unsigned long long)lv << 32 | (unsigned int)j;
        }
    }
}

// heads of runs of equal keys in the sorted references
__global__ void k_gsp_heads(i64 cnt, const unsigned long long *__restrict__ key, i64 *__restrict__ head) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) head[k] = (k == 0 || key[k] != key[k - 1]) ? 1 : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) head[cnt] = 0;
}

// distinct references: external row, its cell, and what the lower band must have stored for it
__global__ void k_gsp_unique(i64 cnt, const unsigned long long *__restrict__ key, const i64 *__restrict__ head, const i64 *__restrict__ at,
                             const i32 *__restrict__ cellbase, const i32 *__restrict__ band_of, const i32 *__restrict__ blev,
                             i32 *__restrict__ ext, unsigned int *__restrict__ cell, i32 *__restrict__ need) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) {
        if (!head[k]) continue;
        const unsigned long long r = key[k];
        const int p = (int)(r >> 60);
        const i32 lv = (i32)((r >> 32) & 0x0fffffffu), j = (i32)(r & 0xffffffffu);
        const i64 c = (i64)cellbase[p] + lv;
        ext[at[k]] = j;
        cell[at[k]] = (unsigned int)c;
        atomicMax(&need[c * 16 + band_of[j]], blev[j] + 1);
    }
}

// cumulative requirement rows (per band and lower band: one thread each), and the largest cell
__global__ void k_gsp_need_scan(int P, const i32 *__restrict__ cellbase, const i32 *__restrict__ nlev, i32 *__restrict__ need,
                                const i64 *__restrict__ ext_ptr, int *__restrict__ maxcell) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P * 16) return;
    const int p = t / 16, q = t % 16;
    i32 run = 0, mc = 0;
    for (i32 lv = 0; lv < nlev[p]; ++lv) {
        const i64 c = (i64)cellbase[p] + lv;
        const i32 v = need[c * 16 + q];
        run = v > run ? v : run;
        need[c * 16 + q] = run;
        if (q == 0) { const i32 sz = (i32)(ext_ptr[c + 1] - ext_ptr[c]); mc = sz > mc ? sz : mc; }
    }
    if (q == 0) atomicMax(maxcell, mc);
}

__global__ void k_gsp_copy_rows(i64 cnt, const i32 *__restrict__ rr, const i32 *__restrict__ a, const i32 *__restrict__ b,
                                i32 *__restrict__ a2, i32 *__restrict__ b2) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) { a2[rr[k]] = a[rr[k]]; b2[rr[k]] = b[rr[k]]; }
}

__global__ void k_gsp_cell_keys(i64 cnt, const i32 *__restrict__ rr, const i32 *__restrict__ cellbase, const i32 *__restrict__ band_of,
                                const i32 *__restrict__ blev, unsigned int *__restrict__ key) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (i64)gridDim.x * blockDim.x) key[k] = (unsigned int)(cellbase[band_of[rr[k]]] + blev[rr[k]]);
}

// the fetch wave's read positions of one band: per level kGsFetchK * 64, padded by 3 kGsFetchD levels (idle reads: the band's first position)
__global__ void k_gsp_fsrc(i32 nlev, i64 cell0, const i64 *__restrict__ ext_ptr, const i32 *__restrict__ ext, const i32 *__restrict__ pos,
                           i32 idle, i32 *__restrict__ out) {
    const i64 total = (i64)(nlev + 3 * kGsFetchD) * (kGsFetchK * 64);
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (i64)gridDim.x * blockDim.x) {
        const i64 ll = k / (kGsFetchK * 64), c = k % (kGsFetchK * 64);
        i32 v = idle;
        if (ll < nlev) {
            const i64 a = ext_ptr[cell0 + ll], b = ext_ptr[cell0 + ll + 1];
            if (c < b - a) v = pos[ext[a + c]];
        }
        out[k] = v;
    }
}

struct GspBandRun {
    i64 level_first = 0, level_count = 0;
    int P = 0;
    std::vector<i32> nlev;                     // per band
    std::vector<i32> cellbase;                 // per band: first cell (cells = (band, level) pairs, band after band)
    std::vector<std::vector<i64>> lptr;        // per band: row positions of its levels (nlev + 1, absolute)
    std::vector<i32> need;                     // cells x 16, cumulative
    DevBuf<i32> ext;                           // sorted distinct external rows, cell after cell
    DevBuf<i64> ext_ptr;                       // cells + 1
};

template <class T>
static std::vector<T> gsp_download(const T *p, size_t count) {
    std::vector<T> h(count);
    if (count) SLP_HIP(hipMemcpyAsync(h.data(), p, count * sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
    SLP_HIP(hipStreamSynchronize(ctx().stream));
    return h;
}

static int gsp_bits(i64 v) {   // bits needed for keys 0 .. v
    int b = 1;
    while (((i64)1 << b) <= v) ++b;
    return b;
}

// gs_plan's band analysis with the per-row and per-entry work on the device; the timing model runs on the host over the per-cell
// tables (lane slots per band and level, requirement rows).  Reorders `rows` inside the runs that get bands.
static void gsp_bands(GsPlan &g, i64 n, const i64 *dptr, const i32 *didx, const DevBuf<i32> &level, DevBuf<i32> &rows,
                      const std::vector<char> &is_launch, const std::vector<unsigned long long> &lanes_per_level,
                      std::vector<GspBandRun> &band_runs, DevBuf<i32> &band_best, DevBuf<i32> &blev_best) {
    hipStream_t st = ctx().stream;
    const char *eb = getenv("SLP_GS_BANDS");
    const int want = eb ? atoi(eb) : -1;
    if (want == 0) return;
    DevBuf<i32> band_of, blev;
    for (i64 l0 = 0; l0 < g.nlevels;) {
        if (is_launch[(size_t)l0]) { ++l0; continue; }
        i64 l1 = l0;
        while (l1 < g.nlevels && !is_launch[(size_t)l1]) ++l1;
        const i64 t0 = g.lptr[(size_t)l0], t1 = g.lptr[(size_t)l1], cnt = t1 - t0, L = l1 - l0;
        i64 lanes_total = 0;
        // (the constants: see gs_plan)
        const double tau1 = 0.47, tauP = 0.32, per_slot = 0.075, per_slot_band = 0.09, hop = 3.0, flag_latency = 6.0;  // us
        double t_single = 0;
        for (i64 l = l0; l < l1; ++l) {
            lanes_total += (i64)lanes_per_level[(size_t)l];
            t_single += tau1 + per_slot * (double)(((i64)lanes_per_level[(size_t)l] + 63) / 64);
        }
        std::vector<int> cand;
        if (L >= ((i64)1 << 20)) cand.clear();
        else if (want > 0) cand.push_back(std::min(want, kGsMaxBands));
        else if (lanes_total >= 32768 && L >= 64) cand = {4, 8, 16};
        if (cand.empty()) { l0 = l1; continue; }
        if (!band_of.p) {
            band_of.alloc((size_t)n); blev.alloc((size_t)n); band_best.alloc((size_t)n); blev_best.alloc((size_t)n);
            SLP_HIP(hipMemsetAsync(band_best.p, 0xff, (size_t)n * sizeof(i32), st));
            blev_best.zero();
        }
        // the run's rows in increasing index, their lane counts and the running sum
        DevBuf<i32> rr((size_t)cnt);
        DevBuf<i64> rl((size_t)cnt + 1), acc((size_t)cnt + 1);
        {
            size_t bytes = 0;
            const unsigned bits = (unsigned)gsp_bits(n);
            SLP_HIP(rocprim::radix_sort_keys(nullptr, bytes, reinterpret_cast<const unsigned int *>(rows.p + t0), reinterpret_cast<unsigned int *>(rr.p), (size_t)cnt, 0u, bits, st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::radix_sort_keys(tmp.p, bytes, reinterpret_cast<const unsigned int *>(rows.p + t0), reinterpret_cast<unsigned int *>(rr.p), (size_t)cnt, 0u, bits, st));
            hipLaunchKernelGGL(k_gsp_row_lanes, dim3(grid_for(cnt, kGspBlock)), dim3(kGspBlock), 0, st, cnt, rr.p, dptr, rl.p);
            SLP_HIP(hipMemsetAsync(rl.p + cnt, 0, sizeof(i64), st));
            bytes = 0;
            SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, rl.p, acc.p, (i64)0, (size_t)cnt + 1, rocprim::plus<i64>(), st));
            DevBuf<char> tmp2(bytes);
            SLP_HIP(rocprim::exclusive_scan(tmp2.p, bytes, rl.p, acc.p, (i64)0, (size_t)cnt + 1, rocprim::plus<i64>(), st));
            SLP_HIP(hipStreamSynchronize(st));
        }
        GspBandRun best;
        double t_best = want > 0 ? 1e300 : t_single / 1.1 - 5.0;
        for (int P : cand) {
            if (cnt < P) continue;
            GspBandRun br;
            br.level_first = l0; br.level_count = L; br.P = P;
            DevBuf<i32> ordm((size_t)P * (size_t)L), dnlev((size_t)P);
            ordm.zero();
            hipLaunchKernelGGL(k_gsp_band_of, dim3(grid_for(cnt, kGspBlock)), dim3(kGspBlock), 0, st, cnt, rr.p, acc.p, lanes_total, P, l0, L, level.p,
                               band_of.p, ordm.p);
            hipLaunchKernelGGL(k_gsp_band_levels, dim3(1), dim3(64), 0, st, P, L, ordm.p, dnlev.p);
            SLP_HIP(hipGetLastError());
            br.nlev = gsp_download(dnlev.p, (size_t)P);
            bool ok = true;
            for (int p = 0; p < P; ++p) ok = ok && br.nlev[(size_t)p] > 0 && br.nlev[(size_t)p] < (1 << 24);
            if (!ok) continue;
            br.cellbase.assign((size_t)P + 1, 0);
            for (int p = 0; p < P; ++p) br.cellbase[(size_t)p + 1] = br.cellbase[(size_t)p] + br.nlev[(size_t)p];
            const i64 cells = br.cellbase[(size_t)P];
            DevBuf<i32> dcellbase;
            dcellbase.upload(br.cellbase.data(), br.cellbase.size());
            DevBuf<unsigned long long> lanes_pl((size_t)cells), rows_pl((size_t)cells);
            DevBuf<i64> nref((size_t)cnt + 1), roff((size_t)cnt + 1);
            lanes_pl.zero();
            rows_pl.zero();
            hipLaunchKernelGGL(k_gsp_blev, dim3(grid_for(cnt, kGspBlock)), dim3(kGspBlock), 0, st, cnt, rr.p, dptr, didx, l0, l1, L, level.p, band_of.p,
                               ordm.p, dcellbase.p, blev.p, lanes_pl.p, rows_pl.p, nref.p);
            SLP_HIP(hipGetLastError());
            i64 nrefs = 0;
            {
                size_t bytes = 0;
                SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, nref.p, roff.p, (i64)0, (size_t)cnt + 1, rocprim::plus<i64>(), st));
                DevBuf<char> tmp(bytes);
                SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, nref.p, roff.p, (i64)0, (size_t)cnt + 1, rocprim::plus<i64>(), st));
                SLP_HIP(hipMemcpyAsync(&nrefs, roff.p + cnt, sizeof(i64), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipStreamSynchronize(st));
            }
            DevBuf<i32> need((size_t)cells * 16);
            need.zero();
            br.ext_ptr.alloc((size_t)cells + 1);
            i64 nuniq = 0;
            if (nrefs > 0) {
                DevBuf<unsigned long long> refs((size_t)nrefs), sref((size_t)nrefs);
                hipLaunchKernelGGL(k_gsp_refs, dim3(grid_for(cnt, kGspBlock)), dim3(kGspBlock), 0, st, cnt, rr.p, dptr, didx, l0, l1, level.p, band_of.p,
                                   blev.p, roff.p, refs.p);
                size_t bytes = 0;
                SLP_HIP(rocprim::radix_sort_keys(nullptr, bytes, refs.p, sref.p, (size_t)nrefs, 0u, 64u, st));
                DevBuf<char> tmp(bytes);
                SLP_HIP(rocprim::radix_sort_keys(tmp.p, bytes, refs.p, sref.p, (size_t)nrefs, 0u, 64u, st));
                DevBuf<i64> head((size_t)nrefs + 1), at((size_t)nrefs + 1);
                hipLaunchKernelGGL(k_gsp_heads, dim3(grid_for(nrefs, kGspBlock)), dim3(kGspBlock), 0, st, nrefs, sref.p, head.p);
                bytes = 0;
                SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, head.p, at.p, (i64)0, (size_t)nrefs + 1, rocprim::plus<i64>(), st));
                DevBuf<char> tmp2(bytes);
                SLP_HIP(rocprim::exclusive_scan(tmp2.p, bytes, head.p, at.p, (i64)0, (size_t)nrefs + 1, rocprim::plus<i64>(), st));
                SLP_HIP(hipMemcpyAsync(&nuniq, at.p + nrefs, sizeof(i64), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipStreamSynchronize(st));
                br.ext.alloc((size_t)nuniq);
                DevBuf<unsigned int> cell((size_t)nuniq);
                hipLaunchKernelGGL(k_gsp_unique, dim3(grid_for(nrefs, kGspBlock)), dim3(kGspBlock), 0, st, nrefs, sref.p, head.p, at.p, dcellbase.p,
                                   band_of.p, blev.p, br.ext.p, cell.p, need.p);
                hipLaunchKernelGGL(k_gsp_ptr_from_sorted, dim3(grid_for(std::max<i64>(nuniq, cells + 1), kGspBlock)), dim3(kGspBlock), 0, st, nuniq, cells,
                                   cell.p, br.ext_ptr.p);
                SLP_HIP(hipGetLastError());
                SLP_HIP(hipStreamSynchronize(st));
            } else {
                br.ext.alloc(1);
                br.ext_ptr.zero();
            }
            DevBuf<int> maxcell(1);
            maxcell.zero();
            DevBuf<i32> dnl;
            dnl.upload(br.nlev.data(), br.nlev.size());
            hipLaunchKernelGGL(k_gsp_need_scan, dim3((unsigned)((P * 16 + 63) / 64)), dim3(64), 0, st, P, dcellbase.p, dnl.p, need.p, br.ext_ptr.p, maxcell.p);
            SLP_HIP(hipGetLastError());
            int hmax = 0;
            maxcell.download(&hmax, 1);
            if (hmax > kGsFetchK * 64) continue;   // a level of a band reads more external rows than the fetch wave carries
            br.need = gsp_download(need.p, (size_t)cells * 16);
            const std::vector<unsigned long long> hl = gsp_download(lanes_pl.p, (size_t)cells), hr = gsp_download(rows_pl.p, (size_t)cells);
            // the pipeline's finish time (gs_plan): a band's level starts when its own previous level is done and the lower bands
            // have stored what the reads issued in it (for kGsFetchD levels ahead) want, seen one flag latency later
            std::vector<std::vector<double>> fin((size_t)P);
            double t_all = 0;
            for (int p = 0; p < P; ++p) {
                const i32 nl = br.nlev[(size_t)p];
                fin[(size_t)p].assign((size_t)nl, 0.0);
                double t = 0;
                for (i32 lv = 0; lv < nl; ++lv) {
                    const i32 ahead = std::min<i32>(lv + kGsFetchD, nl - 1);
                    for (int q = 0; q < p; ++q) {
                        const i32 nd = br.need[((size_t)br.cellbase[(size_t)p] + (size_t)ahead) * 16 + (size_t)q];
                        if (nd > 0) t = std::max(t, fin[(size_t)q][(size_t)nd - 1] + flag_latency);
                    }
                    t += tauP + per_slot_band * (double)(((i64)hl[(size_t)br.cellbase[(size_t)p] + (size_t)lv] + 63) / 64);
                    fin[(size_t)p][(size_t)lv] = t;
                }
                t_all = std::max(t_all, t + hop);
            }
            if (getenv("SLP_GS_VERBOSE"))
                fprintf(stderr, "gauss-seidel bands (device plan): levels %lld..%lld, %lld lane slots: one workgroup %.0f us, %d bands %.0f us (levels of band 0: %d)\n",
                        (long long)l0, (long long)l1, (long long)lanes_total, t_single, P, t_all, (int)br.nlev[0]);
            if (t_all < t_best) {
                t_best = t_all;
                br.lptr.resize((size_t)P);
                i64 at = t0;
                for (int p = 0; p < P; ++p) {
                    const i32 nl = br.nlev[(size_t)p];
                    br.lptr[(size_t)p].assign((size_t)nl + 1, 0);
                    for (i32 lv = 0; lv < nl; ++lv) {
                        br.lptr[(size_t)p][(size_t)lv] = at;
                        at += (i64)hr[(size_t)br.cellbase[(size_t)p] + (size_t)lv];
                    }
                    br.lptr[(size_t)p][(size_t)nl] = at;
                }
                hipLaunchKernelGGL(k_gsp_copy_rows, dim3(grid_for(cnt, kGspBlock)), dim3(kGspBlock), 0, st, cnt, rr.p, band_of.p, blev.p, band_best.p, blev_best.p);
                SLP_HIP(hipGetLastError());
                SLP_HIP(hipStreamSynchronize(st));
                best = std::move(br);
            }
        }
        if (best.P > 0) {
            // positions: band after band, level after level, rows ascending inside a level = a stable sort of the run's rows by cell
            DevBuf<unsigned int> key((size_t)cnt), skey((size_t)cnt);
            DevBuf<i32> dcb;
            dcb.upload(best.cellbase.data(), best.cellbase.size());
            hipLaunchKernelGGL(k_gsp_cell_keys, dim3(grid_for(cnt, kGspBlock)), dim3(kGspBlock), 0, st, cnt, rr.p, dcb.p, band_best.p, blev_best.p, key.p);
            size_t bytes = 0;
            const unsigned bits = (unsigned)gsp_bits(best.cellbase[(size_t)best.P]);
            SLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, skey.p, rr.p, rows.p + t0, (size_t)cnt, 0u, bits, st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, skey.p, rr.p, rows.p + t0, (size_t)cnt, 0u, bits, st));
            SLP_HIP(hipStreamSynchronize(st));
            band_runs.push_back(std::move(best));
        }
        l0 = l1;
    }
}

// the device tables k_gsp_records reads for one run with bands
static GspRun gsp_run_tables(const GspBandRun &br, std::vector<DevBuf<i64>> &keep_i64, std::vector<DevBuf<int>> &keep_int) {
    std::vector<i64> bl;
    std::vector<int> off;
    for (int p = 0; p < br.P; ++p) {
        off.push_back((int)bl.size());
        bl.insert(bl.end(), br.lptr[(size_t)p].begin(), br.lptr[(size_t)p].end());
    }
    off.push_back((int)bl.size());
    keep_i64.emplace_back();
    keep_i64.back().upload(bl.data(), bl.size());
    keep_int.emplace_back();
    keep_int.back().upload(off.data(), off.size());
    GspRun r;
    r.L0 = br.level_first; r.L1 = br.level_first + br.level_count; r.P = br.P; r.pad = 0;
    r.blptr = keep_i64.back().p;
    r.blptr_off = keep_int.back().p;
    r.ext = br.ext.p;
    r.ext_ptr = br.ext_ptr.p;
    return r;
}

}  // namespace slp

// ---- host side of the device plan -----------------------------------------------------------------------------------------------
namespace slp {

// The plan of gs_plan() for the matrix at dptr / didx / dval (DEVICE arrays), built on the device.  Returns false where only the
// host plan serves (the round-1 sweep kernel without the LDS window, sizes beyond the 32-bit slot / position indices): the caller
// then downloads the matrix and plans on the host.
static bool gs_plan_device(GsPlan &g, i64 n, i64 nnz, const i64 *dptr, const i32 *didx, const double *dval) {
    hipStream_t st = ctx().stream;
    {
        const char *ew = getenv("SLP_GS_WINDOW");
        if (ew && ew[0] == '0') return false;
    }
    if (n <= 0 || n >= ((i64)1 << 30) || nnz >= ((i64)1 << 31)) return false;
    g.n = n;
    g.nnz = nnz;
    { const char *es = getenv("SLP_GS_BANDS_SAFE"); g.bands_safe = es && es[0] == '1'; }
    const int gn = grid_for(n, kGspBlock);

    // ---- levels over the symmetrised pattern: the pattern of M^T by one stable sort of (column, row)
    DevBuf<i32> level((size_t)n);
    {
        DevBuf<int> bad(1);
        bad.zero();
        hipLaunchKernelGGL(k_gsp_validate, dim3(gn), dim3(kGspBlock), 0, st, n, nnz, dptr, didx, bad.p);
        SLP_HIP(hipGetLastError());
        int hbad = 0;
        bad.download(&hbad, 1);
        SLP_REQUIRE(!(hbad & 2), "gauss-seidel: bad row pointer");
        SLP_REQUIRE(!(hbad & 1), "gauss-seidel: column index out of range");
    }
    {
        Phase ph("  gsp: levels (pattern of M^T, level-by-level sweep, sinks)");
        DevBuf<i32> rowid((size_t)std::max<i64>(nnz, 1)), trow((size_t)std::max<i64>(nnz, 1));
        DevBuf<unsigned int> scol((size_t)std::max<i64>(nnz, 1));
        DevBuf<i64> tptr((size_t)n + 1);
        hipLaunchKernelGGL(k_gsp_rowid, dim3(gn), dim3(kGspBlock), 0, st, n, dptr, rowid.p);
        if (nnz) {
            size_t bytes = 0;
            const unsigned bits = (unsigned)gsp_bits(n);
            SLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, reinterpret_cast<const unsigned int *>(didx), scol.p, rowid.p, trow.p, (size_t)nnz, 0u, bits, st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, reinterpret_cast<const unsigned int *>(didx), scol.p, rowid.p, trow.p, (size_t)nnz, 0u, bits, st));
            SLP_HIP(hipStreamSynchronize(st));
        }
        hipLaunchKernelGGL(k_gsp_ptr_from_sorted, dim3(grid_for(std::max<i64>(nnz, n + 1), kGspBlock)), dim3(kGspBlock), 0, st, nnz, n, scol.p, tptr.p);
        i32 maxlev = 0;
        {
            DevBuf<int> indeg((size_t)n), counts((size_t)n + 2);
            DevBuf<i32> order((size_t)n);
            DevBuf<i64> offs((size_t)n + 2);
            counts.zero();
            SLP_HIP(hipMemsetAsync(offs.p, 0, sizeof(i64), st));
            hipLaunchKernelGGL(k_gsp_indeg, dim3(gn), dim3(kGspBlock), 0, st, n, dptr, didx, tptr.p, trow.p, indeg.p, level.p, order.p, counts.p);
            // Wide levels: a chip-wide launch per level.  Runs of narrow levels: one workgroup, a barrier per level.  The host looks
            // at where the sweep stands after every step (a run of up to 4096 narrow levels, or one wide level).
            DevBuf<int> reached(1);
            i64 l = 0;
            int c = 0;
            SLP_HIP(hipMemcpyAsync(&c, counts.p, sizeof(int), hipMemcpyDeviceToHost, st));
            SLP_HIP(hipStreamSynchronize(st));
            while (c > 0 && l < n) {
                if (c > kGspNarrow) {
                    hipLaunchKernelGGL(k_gsp_frontier, dim3(grid_for(c, kGspBlock)), dim3(kGspBlock), 0, st, (int)l, dptr, didx, tptr.p, trow.p, indeg.p,
                                       level.p, order.p, counts.p, offs.p);
                    l += 1;
                } else {
                    hipLaunchKernelGGL(k_gsp_frontier_run, dim3(1), dim3(1024), 0, st, (int)l, (int)std::min<i64>(4096, n - l), dptr, didx, tptr.p, trow.p,
                                       indeg.p, level.p, order.p, counts.p, offs.p, reached.p);
                    int hr = 0;
                    SLP_HIP(hipMemcpyAsync(&hr, reached.p, sizeof(int), hipMemcpyDeviceToHost, st));
                    SLP_HIP(hipStreamSynchronize(st));
                    l = hr;
                }
                SLP_HIP(hipGetLastError());
                i64 done_rows = 0;
                c = 0;
                if (l <= n) SLP_HIP(hipMemcpyAsync(&c, counts.p + l, sizeof(int), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipMemcpyAsync(&done_rows, offs.p + l, sizeof(i64), hipMemcpyDeviceToHost, st));
                SLP_HIP(hipStreamSynchronize(st));
                // long chains of very narrow levels (a banded matrix: ~ n levels of a few rows): the host's one pass over the rows
                // is the better tool
                if (l >= 65536 && done_rows < 32 * l) return false;
            }
            // the last level with rows: counts[] is positive up to it
            const std::vector<int> hc = gsp_download(counts.p, (size_t)std::min<i64>(l + 1, n + 1));
            i64 total = 0;
            for (size_t k = 0; k < hc.size() && hc[k] > 0; ++k) { maxlev = (i32)k; total += hc[k]; }
            SLP_REQUIRE(total == n, "gauss-seidel: the dependency levels do not cover every row");
        }
        DevBuf<unsigned char> coupled((size_t)n);
        coupled.zero();
        hipLaunchKernelGGL(k_gsp_coupled, dim3(gn), dim3(kGspBlock), 0, st, n, dptr, didx, coupled.p);
        DevBuf<unsigned long long> sinks(1);
        sinks.zero();
        hipLaunchKernelGGL(k_gsp_count_sinks, dim3(gn), dim3(kGspBlock), 0, st, n, coupled.p, level.p, sinks.p);
        SLP_HIP(hipGetLastError());
        unsigned long long hsinks = 0;
        sinks.download(&hsinks, 1);
        const char *es = getenv("SLP_GS_SINKS");
        if (hsinks > 4096 && !(es && es[0] == '0')) {
            // rows that nothing waits for: one level of their own behind all others (see gs_plan); levels that held nothing else go
            DevBuf<i32> present((size_t)maxlev + 2), remap((size_t)maxlev + 2);
            present.zero();
            hipLaunchKernelGGL(k_gsp_move_sinks, dim3(gn), dim3(kGspBlock), 0, st, n, coupled.p, maxlev + 1, level.p, present.p);
            size_t bytes = 0;
            SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, present.p, remap.p, (i32)0, (size_t)maxlev + 2, rocprim::plus<i32>(), st));
            DevBuf<char> tmp(bytes);
            SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, present.p, remap.p, (i32)0, (size_t)maxlev + 2, rocprim::plus<i32>(), st));
            hipLaunchKernelGGL(k_gsp_remap, dim3(gn), dim3(kGspBlock), 0, st, n, remap.p, level.p);
            SLP_HIP(hipGetLastError());
            i32 last[2];
            SLP_HIP(hipMemcpyAsync(&last[0], remap.p + maxlev + 1, sizeof(i32), hipMemcpyDeviceToHost, st));
            SLP_HIP(hipMemcpyAsync(&last[1], present.p + maxlev + 1, sizeof(i32), hipMemcpyDeviceToHost, st));
            SLP_HIP(hipStreamSynchronize(st));
            maxlev = last[0] + last[1] - 1;
        }
        g.nlevels = (i64)maxlev + 1;
    }

    // ---- rows by level (stable: increasing row inside a level), level pointer, positions
    DevBuf<i32> rows((size_t)n), pos((size_t)n);
    DevBuf<i64> lptr_dev((size_t)g.nlevels + 1), rlen((size_t)n + 1);
    DevBuf<unsigned char> long_level((size_t)g.nlevels);
    std::vector<unsigned long long> hlanes((size_t)g.nlevels, 0);
    {
        Phase ph("  gsp: rows by level");
        DevBuf<i32> iota((size_t)n);
        DevBuf<unsigned int> skey((size_t)n);
        hipLaunchKernelGGL(k_gsp_iota, dim3(gn), dim3(kGspBlock), 0, st, n, iota.p);
        size_t bytes = 0;
        const unsigned bits = (unsigned)gsp_bits(g.nlevels);
        SLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, reinterpret_cast<const unsigned int *>(level.p), skey.p, iota.p, rows.p, (size_t)n, 0u, bits, st));
        DevBuf<char> tmp(bytes);
        SLP_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, reinterpret_cast<const unsigned int *>(level.p), skey.p, iota.p, rows.p, (size_t)n, 0u, bits, st));
        hipLaunchKernelGGL(k_gsp_ptr_from_sorted, dim3(gn), dim3(kGspBlock), 0, st, n, g.nlevels, skey.p, lptr_dev.p);
        long_level.zero();
        DevBuf<i64> lanes((size_t)n + 1), lscan((size_t)n + 1), lat((size_t)g.nlevels + 1);
        hipLaunchKernelGGL(k_gsp_long_levels, dim3(gn), dim3(kGspBlock), 0, st, n, rows.p, dptr, level.p, long_level.p, lanes.p);
        bytes = 0;
        SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, lanes.p, lscan.p, (i64)0, (size_t)n + 1, rocprim::plus<i64>(), st));
        DevBuf<char> tmp2(bytes);
        SLP_HIP(rocprim::exclusive_scan(tmp2.p, bytes, lanes.p, lscan.p, (i64)0, (size_t)n + 1, rocprim::plus<i64>(), st));
        hipLaunchKernelGGL(k_gsp_at_levels, dim3(grid_for(g.nlevels + 1, kGspBlock)), dim3(kGspBlock), 0, st, g.nlevels + 1, lptr_dev.p, lscan.p, lat.p);
        SLP_HIP(hipGetLastError());
        const std::vector<i64> hat = gsp_download(lat.p, (size_t)g.nlevels + 1);
        for (i64 l = 0; l < g.nlevels; ++l) hlanes[(size_t)l] = (unsigned long long)(hat[(size_t)l + 1] - hat[(size_t)l]);
    }
    g.lptr = gsp_download(lptr_dev.p, (size_t)g.nlevels + 1);
    const std::vector<unsigned char> hlong = gsp_download(long_level.p, (size_t)g.nlevels);
    g.max_width = 0;
    for (i64 l = 0; l < g.nlevels; ++l) g.max_width = std::max(g.max_width, g.lptr[(size_t)l + 1] - g.lptr[(size_t)l]);

    // ---- which levels go where: gs_plan's rules on the level table
    g.one_block = (g.max_width <= 2048) && (g.nnz <= 200000);
    const char *ep = getenv("SLP_GS_PIPELINED");
    const bool forced = ep && ep[0] == '1';
    const i64 wide = forced ? ((i64)1 << 40) : 4096;
    std::vector<char> is_launch((size_t)g.nlevels, 0);
    i64 narrow_levels = 0;
    for (i64 l = 0; l < g.nlevels; ++l) {
        const bool launch = g.lptr[(size_t)l + 1] - g.lptr[(size_t)l] > wide || hlong[(size_t)l];
        is_launch[(size_t)l] = launch ? 1 : 0;
        narrow_levels += launch ? 0 : 1;
    }
    const bool pipeline = !(ep && ep[0] == '0') && (forced || !g.one_block) && (forced || narrow_levels >= 16) && narrow_levels > 0;

    g.pipelined = false;
    g.windowed = false;
    g.has_far = false;
    g.nbands = 0;
    g.segments.clear();

    // ---- bands: which runs of narrow levels are cut into row ranges (reorders the rows inside such runs)
    std::vector<GspBandRun> band_runs;
    DevBuf<i32> band_of, blev;
    if (pipeline && n < ((i64)1 << 25)) {
        Phase ph("  gsp: bands");
        gsp_bands(g, n, dptr, didx, level, rows, is_launch, hlanes, band_runs, band_of, blev);
    }

    // ---- the matrix in level order, its inverted diagonal
    {
        Phase ph("  gsp: matrix in level order");
        hipLaunchKernelGGL(k_gsp_positions, dim3(gn), dim3(kGspBlock), 0, st, n, rows.p, dptr, pos.p, rlen.p);
        g.ptr.alloc((size_t)n + 1);
        size_t bytes = 0;
        SLP_HIP(rocprim::exclusive_scan(nullptr, bytes, rlen.p, g.ptr.p, (i64)0, (size_t)n + 1, rocprim::plus<i64>(), st));
        DevBuf<char> tmp(bytes);
        SLP_HIP(rocprim::exclusive_scan(tmp.p, bytes, rlen.p, g.ptr.p, (i64)0, (size_t)n + 1, rocprim::plus<i64>(), st));
        g.idx.alloc((size_t)nnz + kGsEntries);
        g.val.alloc((size_t)nnz + kGsEntries);
        g.idx.zero();
        g.val.zero();
        hipLaunchKernelGGL(k_gsp_permute, dim3(gn), dim3(kGspBlock), 0, st, n, rows.p, dptr, didx, dval, g.ptr.p, g.idx.p, g.val.p);
        g.invd.alloc((size_t)n);
        g.diag.alloc((size_t)n);
        hipLaunchKernelGGL(k_invert_diag, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, n, rows.p, g.ptr.p, g.idx.p, g.val.p, g.invd.p, g.diag.p);
        SLP_HIP(hipGetLastError());
        SLP_HIP(hipStreamSynchronize(st));
    }
    if (pipeline) {

        // ---- units in the order gs_plan walks: levels ascending, a run with bands band after band
        std::vector<GspUnit> units;
        std::vector<GsPlan::Segment> segs;
        struct Group { bool band; size_t seg; int run, p; std::vector<size_t> units; };
        std::vector<Group> groups;   // header groups: bands first (in run order), then the plain segments
        std::vector<Group> plain_groups;
        size_t next_run = 0;
        i64 seg_first_level = 0;
        i64 scratch_cells = (i64)64 * kGsWaves;
        for (i64 l = 0; l < g.nlevels; ++l) {
            if (is_launch[(size_t)l]) {
                GsPlan::Segment sg;
                sg.launch = true; sg.first = l; sg.count = 1; sg.slot_first = 0; sg.slot_count = 0;
                sg.level_first = l; sg.level_count = 1; sg.hoff = 0; sg.waves = 0;
                segs.push_back(sg);
                continue;
            }
            if (next_run < band_runs.size() && band_runs[next_run].level_first == l) {
                const GspBandRun &br = band_runs[next_run];
                GsPlan::Segment sg;
                sg.launch = false; sg.first = 0; sg.count = 0; sg.slot_first = 0; sg.slot_count = 0;
                sg.level_first = br.level_first; sg.level_count = br.level_count; sg.hoff = 0; sg.waves = 0;
                sg.bands = br.P; sg.band_first = 0;
                segs.push_back(sg);
                for (int p = 0; p < br.P; ++p) {
                    Group gr;
                    gr.band = true; gr.seg = segs.size() - 1; gr.run = (int)next_run; gr.p = p;
                    const std::vector<i64> &bl = br.lptr[(size_t)p];
                    for (size_t ll = 0; ll + 1 < bl.size(); ++ll) {
                        GspUnit un;
                        un.beg = bl[ll]; un.end = bl[ll + 1]; un.slot0 = 0; un.level = (int)ll; un.seg_first = 0; un.band = p; un.run = (int)next_run;
                        un.fits_ring = un.end - un.beg <= kGsWide; un.pad = 0;
                        gr.units.push_back(units.size());
                        units.push_back(un);
                    }
                    groups.push_back(std::move(gr));
                }
                scratch_cells = std::max<i64>(scratch_cells, (i64)br.P * 1024);
                ++next_run;
                l = br.level_first + br.level_count - 1;
                continue;
            }
            if (segs.empty() || segs.back().launch || segs.back().bands) {
                seg_first_level = l;
                GsPlan::Segment sg;
                sg.launch = false; sg.first = 0; sg.count = 0; sg.slot_first = 0; sg.slot_count = 0;
                sg.level_first = l; sg.level_count = 0; sg.hoff = 0; sg.waves = 0;
                segs.push_back(sg);
                Group gr;
                gr.band = false; gr.seg = segs.size() - 1; gr.run = -1; gr.p = -1;
                plain_groups.push_back(std::move(gr));
            }
            GspUnit un;
            un.beg = g.lptr[(size_t)l]; un.end = g.lptr[(size_t)l + 1]; un.slot0 = 0; un.level = (int)l; un.seg_first = (int)seg_first_level;
            un.band = -1; un.run = -1; un.fits_ring = un.end - un.beg <= kGsWide; un.pad = 0;
            plain_groups.back().units.push_back(units.size());
            units.push_back(un);
            segs.back().level_count += 1;
        }
        if (!units.empty()) {
            Phase ph("  gsp: lane slots, records, headers");
            const int nunits = (int)units.size();
            DevBuf<GspUnit> dunits;
            dunits.upload(units.data(), units.size());
            DevBuf<i32> slot_rel((size_t)n);
            DevBuf<i64> totals((size_t)nunits);
            hipLaunchKernelGGL(k_gsp_layout, dim3((unsigned)((nunits + 63) / 64)), dim3(64), 0, st, nunits, dunits.p, rlen.p, slot_rel.p, totals.p);
            SLP_HIP(hipGetLastError());
            const std::vector<i64> ht = gsp_download(totals.p, (size_t)nunits);
            i64 S = 0;
            for (int u = 0; u < nunits; ++u) { units[(size_t)u].slot0 = S; S += ht[(size_t)u]; }
            // lane slots beyond the 32-bit slot indices (or units without a single slot): declined -- the caller resets the plan and
            // takes the host's, whose decisions for such a matrix are the ones the sweep kernels were written against
            if (S <= 0 || S >= ((i64)1 << 31)) return false;
            {
                dunits.upload(units.data(), units.size());
                g.ents.alloc((size_t)S);
                g.lanes.alloc((size_t)S);
                g.lane_row.alloc((size_t)S);
                hipLaunchKernelGGL(k_gsp_fill_idle, dim3(grid_for(S, kGspBlock)), dim3(kGspBlock), 0, st, S, g.ents.p, g.lanes.p, g.lane_row.p);
                // device tables of the runs with bands
                std::vector<GspRun> hruns;
                std::vector<DevBuf<i64>> keep_i64;
                std::vector<DevBuf<int>> keep_int;
                keep_i64.reserve(band_runs.size());   // (the tables' addresses go into hruns)
                keep_int.reserve(band_runs.size());
                for (const GspBandRun &br : band_runs) hruns.push_back(gsp_run_tables(br, keep_i64, keep_int));
                DevBuf<GspRun> druns;
                if (!hruns.empty()) druns.upload(hruns.data(), hruns.size());
                DevBuf<int> far_flag(1);
                far_flag.zero();
                hipLaunchKernelGGL(k_gsp_records, dim3(gn), dim3(kGspBlock), 0, st, n, nunits, dunits.p, druns.p, rows.p, dptr, didx, dval, level.p,
                                   pos.p, lptr_dev.p, band_of.p, blev.p, slot_rel.p, g.ents.p, g.lanes.p, g.lane_row.p, far_flag.p);
                SLP_HIP(hipGetLastError());
                const i64 nw = S / 64;
                DevBuf<unsigned short> wsum((size_t)nw);
                hipLaunchKernelGGL(k_gsp_wave_summary, dim3((unsigned)((nw * 64 + kGspBlock - 1) / kGspBlock)), dim3(kGspBlock), 0, st, nw, g.lanes.p, wsum.p);
                SLP_HIP(hipGetLastError());
                const std::vector<unsigned short> hw = gsp_download(wsum.p, (size_t)nw);
                int hfar = 0;
                far_flag.download(&hfar, 1);
                g.has_far = hfar != 0;
                // ---- header lists (per wave slot data only): the bands' groups first, then the plain segments, as gs_plan emits them
                std::vector<GsStepW> hd;
                std::vector<int> hoffs;
                auto info = [&](size_t q0, unsigned *rounds, unsigned *far) {
                    const unsigned v = hw[q0 / 64];
                    *rounds = v & 0xffu;
                    *far = v >> 8;
                    if (*far) g.has_far = true;
                };
                auto ranges_of = [&](const Group &gr) {
                    std::vector<std::pair<i64, i64>> lv;
                    for (size_t u : gr.units) lv.push_back({units[u].slot0, units[u].slot0 + ht[u]});
                    return lv;
                };
                std::vector<GsBand> bands;
                std::vector<int> freq;
                i64 fsrc_total = 0;
                for (const Group &gr : groups) fsrc_total += (i64)((i64)gr.units.size() + 3 * kGsFetchD) * (kGsFetchK * 64);
                if (fsrc_total) g.fsrc.alloc((size_t)fsrc_total);
                i64 fsrc_at = 0;
                for (const Group &gr : groups) {
                    const GspBandRun &br = band_runs[(size_t)gr.run];
                    GsPlan::Segment &sg = segs[gr.seg];
                    if (gr.p == 0) { sg.band_first = (i64)bands.size(); sg.slot_first = units[gr.units.front()].slot0; }
                    int waves = 1;
                    GsBand bd;
                    bd.nlev = (int)gr.units.size();
                    // the fetch wave's read positions (on the device, from the run's external-row lists) ...
                    bd.fsrc = (int)fsrc_at;
                    const i64 fcount = (i64)(bd.nlev + 3 * kGsFetchD) * (kGsFetchK * 64);
                    hipLaunchKernelGGL(k_gsp_fsrc, dim3(grid_for(fcount, kGspBlock)), dim3(kGspBlock), 0, st, bd.nlev, (i64)br.cellbase[(size_t)gr.p],
                                       br.ext_ptr.p, br.ext.p, pos.p, (i32)br.lptr[(size_t)gr.p][0], g.fsrc.p + fsrc_at);
                    SLP_HIP(hipGetLastError());
                    fsrc_at += fcount;
                    // ... and what the lower bands must have stored before (row 0: levels 0 .. D - 1; row l + 1: levels up to l + D)
                    bd.freq = (int)freq.size();
                    for (i32 r = 0; r < bd.nlev + 1 + 2 * kGsFetchD; ++r) {
                        const i32 upto = std::min<i32>(bd.nlev - 1, r == 0 ? kGsFetchD - 1 : r - 1 + kGsFetchD);
                        for (int q = 0; q < 16; ++q)
                            freq.push_back(q < gr.p ? br.need[((size_t)br.cellbase[(size_t)gr.p] + (size_t)upto) * 16 + (size_t)q] : 0);
                    }
                    bd.hoff = gs_emit_headers(hd, hoffs, ranges_of(gr), kGsBandWaves, true, g.bands_safe, &waves, info);
                    bd.waves = waves;
                    bd.scratch = gr.p * 1024;
                    bd.pad0 = bd.pad1 = 0;
                    bands.push_back(bd);
                    sg.waves = std::max(sg.waves, waves);
                    sg.slot_count = units[gr.units.back()].slot0 + ht[gr.units.back()] - sg.slot_first;
                }
                for (const Group &gr : plain_groups) {
                    GsPlan::Segment &sg = segs[gr.seg];
                    sg.slot_first = units[gr.units.front()].slot0;
                    sg.slot_count = units[gr.units.back()].slot0 + ht[gr.units.back()] - sg.slot_first;
                    int waves = 1;
                    sg.hoff = gs_emit_headers(hd, hoffs, ranges_of(gr), kGsWaves, false, g.bands_safe, &waves, info);
                    sg.waves = waves;
                }
                g.pipelined = true;
                g.one_block = false;
                g.windowed = true;
                g.segments = segs;
                g.stepsw.upload(hd.data(), hd.size());
                g.hoffs.upload(hoffs.data(), hoffs.size());
                g.packed.alloc((size_t)n);
                g.dyn.alloc((size_t)S);
                g.rowsw.alloc((size_t)n);
                g.scratch.alloc((size_t)scratch_cells);
                g.xpos.alloc((size_t)n);
                if (!bands.empty()) {
                    g.nbands = (int)bands.size();
                    g.bands.upload(bands.data(), bands.size());
                    g.freq.upload(freq.data(), freq.size());
                    g.prog.alloc((size_t)kGsMaxBands * kGsProgStride);
                }
                SLP_HIP(hipStreamSynchronize(st));   // (the runs' tables go back to the cache)
            }
        }
    }
    g.rows = std::move(rows);
    g.lptr_dev = std::move(lptr_dev);
    SLP_HIP(hipStreamSynchronize(st));
    return true;
}

}  // namespace slp

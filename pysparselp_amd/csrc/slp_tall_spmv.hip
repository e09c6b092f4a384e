// slp_tall_spmv.hip -- the product kernel over tall cells (format, builder and the reasons for both: slp_tall.hip).
// Reference products: `a * x` / `y * a` of ChambollePockPPD.py:206,216,235,240 and ADMM.py:148,220,262.
// A translation unit of its own because it is compiled with another instruction-scheduling strategy than the rest of the
// library (Makefile: -mllvm -amdgpu-sched-strategy=max-ilp; measured on the 2.5e6 x 1e7 slice: A x 3.89 -> 3.73 ms, A^T y
// 3.91 -> 3.77 ms; the same flag makes the sort and the strip kernels slower).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "slp_common.h"
#include "slp_kernels.h"
#include "slp_tall.h"

namespace slp {


// ---- the product -------------------------------------------------------------------------------------------------------
template <bool DICT>
struct TallRegs {
    unsigned int lo[kTallSlots];
    unsigned int hi[DICT ? 2 : 1];
    double val[DICT ? 1 : kTallSlots];
    double x[4];
};

// DICT: depth 4 (20 KB of payload per packet); fp64 entries: depth 2 (48 KB per packet, 16 more registers per packet)
// POW (fp64 entries only): every stored value v enters as |v|^pw * 1.0 -- the sums behind the Chambolle-Pock preconditioners
// (slp_cp.hip, strip_spmv_abs_pow) over a copy without a value table; the same chain of additions as the CSR walk.
// A workgroup runs `nseg` packet streams one after the other over ONE set of running sums (descriptor of segment s of workgroup v:
// wgs[s * gridDim.x + v]): nseg = 1 for a row block of an ordinary copy -- and for the row blocks of ALL chunks of a chunked
// matrix in one grid (A x in one launch); nseg = the chunks for the copy of A^T of a chunked matrix, where the workgroup of a
// column block walks chunk 0, 1, ... in order and the column sums simply stay in LDS between them (A^T y in one launch: the chain
// of additions of the unchunked product, as the chunk-by-chunk launches formed it through `out`).
// C: columns per strip = doubles of an x-tile (4096, 2048 or 1024: the build halves the strips of denser matrices until a cell
// holds ~0.3-0.6 entries per row -- the regime the dealing of a cell's rows to the lanes is made for; items keep their 12-bit
// column field).
template <bool DICT, bool ACC, bool POW = false, int C = kTallC>
__global__ __launch_bounds__(kTallT) void k_tall_spmv(int nseg, const TallWg *__restrict__ wgs, const double *__restrict__ dict_arg,
                                                      const double *__restrict__ x, double *__restrict__ out, double pw) {
    constexpr int kDepth = DICT ? kTallDepth : 2;
    // acc[0] is a scratch cell: the row field of an item is its local row + 1, and whatever is not a lane's item -- a slot
    // beyond its list (the load past the buffer descriptor returns 0), a skip item -- decodes to row 0, value id 0, column 0
    // and adds into the scratch cell: no predicate on the stores, none on the sums
    __shared__ double acc[kTallRmax + 1];
    __shared__ double dv[kTallDictMax];
    static_assert(C == 4096 || C == 2048 || C == 1024, "tall cells: strip width");
    __shared__ double xt[2][C];
    constexpr int kPieces = C / 2048 ? C / 2048 : 1;     // 16-byte pieces of a tile per lane
    const int p = threadIdx.x;
    const bool tile_lane = C >= 2048 || p < C / 2;       // (1024 columns: the first 512 lanes carry the tile)
    // (wave-uniform by construction; told to the compiler, so that tests on it are scalar branches, not EXEC masks)
    const unsigned int wbase = (unsigned int)__builtin_amdgcn_readfirstlane(p & ~(kWave - 1));
    const int v = blockIdx.x;
    int cur = 0;
    TallWg wg = wgs[v];
    // ACC: the sums continue from what `out` holds -- a row chunk of a chunked matrix carrying on the column sums of the chunks
    // before it, launch by launch (the strip-range split takes it in k_tall_combine instead: its descriptors point into the
    // partial-sum array).  A template parameter: loads under a run-time condition in front of the pipeline made the compiler
    // wait for vmcnt(0) in the loop.
    for (int r = p; r <= wg.nrows; r += kTallT) acc[r] = (ACC && r > 0) ? out[wg.row0 + r - 1] : 0.0;
  for (int seg = 0; seg < nseg; ++seg) {
    if (seg > 0) wg = wgs[(i64)seg * gridDim.x + v];
    if (DICT) {
        const double *__restrict__ dsrc = dict_arg ? dict_arg : wg.dict;
        for (int q = p; q < wg.D; q += kTallT) dv[q] = dsrc[q];
    }
    // this lane's dword of every header -- through a GLOBAL-address-space pointer: a pointer read from a structure in memory is
    // a generic one to the compiler, its loads become flat_load (complete out of order: every wait a vmcnt(0), the pipeline gone)
    typedef const unsigned int __attribute__((address_space(1))) *gptr_t;
    const gptr_t hd = (gptr_t)(unsigned long long)wg.dir + (p & 7);
    const int npk = (int)wg.npk - 2 * kTallDepth;                             // the last 2 x depth packets are prefetch targets only
    // buffer descriptors: lanes without work address past num_records (the load returns 0 without a memory request)
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x + wg.x0), 0, (int)(wg.ncol * 8), 0x00020000);

    // A slot as a buffer of its own: `width` words from word `so` of the payload -- the lanes past the slot's width lie behind the
    // descriptor's end (the load returns 0, no memory request) with no per-lane predicate: the slot's base and size are wave-uniform,
    // so the descriptor is a few SCALAR instructions where a compare and a select per load and lane used to be (round 6: 224 of the
    // ~1200 vector instructions of eight packets)
    auto slot_pay = [&](unsigned int so, unsigned int width) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(wg.pay) + so, 0, (int)(width * 4u), 0x00020000);
    };
    auto slot_val = [&](unsigned int so, unsigned int width) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(DICT ? x : wg.val) + so, 0, (int)(width * 8u), 0x00020000);
    };

    TallRegs<DICT> regs[kDepth];
    unsigned int hw[2 * kDepth];

    auto widths = [&](unsigned int h, unsigned int *c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int v = (unsigned int)__builtin_amdgcn_readlane((int)h, 2 + i);
            c[2 * i] = v & 0xffffu;
            c[2 * i + 1] = v >> 16;
        }
    };

    // this lane's two 16-byte pieces of an x-tile (doubles 2p, 2p + 1 of each tile half).  A piece that straddles the end of x
    // (odd width) may fetch the 8 bytes behind it: every device block carries 16 bytes of slack, and no item names that column
    auto load_tile = [&](TallRegs<DICT> &g, const unsigned int xo) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xo + (unsigned int)i * (unsigned int)(C * 4), 0, 0);
            g.x[2 * i] = __hiloint2double((int)v[1], (int)v[0]);
            g.x[2 * i + 1] = __hiloint2double((int)v[3], (int)v[2]);
        }
    };

    auto issue = [&](TallRegs<DICT> &g, unsigned int h) {
        const unsigned int off = (unsigned int)__builtin_amdgcn_readlane((int)h, 0);
        const unsigned int xs = (unsigned int)__builtin_amdgcn_readlane((int)h, 1) & 0x7fffffffu;
        unsigned int c[kTallSlots];
        widths(h, c);
        unsigned int so = off;   // (words)
        const unsigned int mine = (unsigned int)p * 4u;
#pragma unroll
        for (int k = 0; k < kTallSlots; ++k) {
            g.lo[k] = __builtin_amdgcn_raw_buffer_load_b32(slot_pay(so, c[k]), mine, 0, 2);  // streamed once
            if (!DICT) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(slot_val(so, c[k]), 2u * mine, 0, 2);
                g.val[k] = __hiloint2double((int)v[1], (int)v[0]);
            }
            so += c[k];
        }
        if (DICT) {
            g.hi[0] = __builtin_amdgcn_raw_buffer_load_b32(slot_pay(so, c[0]), mine, 0, 2);
            so += c[0];
            g.hi[DICT ? 1 : 0] = __builtin_amdgcn_raw_buffer_load_b32(slot_pay(so, c[4]), mine, 0, 2);
        }
        // columns past ncol read 0: the displacement of each double is part of the voffset, which the descriptor's range check
        // covers (an soffset is not checked on gfx9 raw buffers)
        // lane p carries doubles 2p, 2p + 1 of the tile's first half and of its second half: consecutive lanes then write
        // consecutive 16-byte pieces of the LDS tile (no bank conflicts; with 4p .. 4p + 3 per lane the two 16-byte writes of a
        // lane pair collided -- the tile write was the largest single item of the kernel's ablation, 0.9 of 4.1 ms)
        const unsigned int xo = (xs == kNoTile || !tile_lane) ? kOob : (xs + 2u * (unsigned int)p) * 8u;
        load_tile(g, xo);
    };

    // issue() in two halves: the registers of slots 0-3 are free once the first four items are done, so their loads for the packet
    // `depth` ahead go out BETWEEN the two groups of items (in the shadow of the LDS latency) instead of behind all of them
    auto issue_part = [&](TallRegs<DICT> &g, unsigned int h, const int part) {
        const unsigned int off = (unsigned int)__builtin_amdgcn_readlane((int)h, 0);
        unsigned int c[kTallSlots];
        widths(h, c);
        const unsigned int mine = (unsigned int)p * 4u;
        unsigned int so = off;   // (words)
        if (part == 1) so += c[0] + c[1] + c[2] + c[3];
        // A packet without a fifth item (lists of <= 4 items: the metric's density, where a cell is one such packet) issues no
        // loads for slots 4-7 and their fifth bytes -- a uniform branch; consume() takes the second group under the same test.  The
        // compiler then counts the loads in flight by the path without them (vmcnt(24-26) instead of 36-41): exact for those
        // packets, a shorter prefetch distance behind a packet that has the second group.  3.30 / 3.30 -> 3.25 / 3.22 ms on the
        // slice, unchanged where lists are five long (profiles/r05_tall_half_packets.log).
        if (part == 0 || wbase < c[4]) {   // (per wave: the waves past the fifth items' lanes skip them too)
#pragma unroll
            for (int k = 4 * part; k < 4 * part + 4; ++k) {
                g.lo[k] = __builtin_amdgcn_raw_buffer_load_b32(slot_pay(so, c[k]), mine, 0, 2);
                if (!DICT) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(slot_val(so, c[k]), 2u * mine, 0, 2);
                    g.val[DICT ? 0 : k] = __hiloint2double((int)v[1], (int)v[0]);
                }
                so += c[k];
            }
            if (DICT) {
                if (part == 0) so += c[4] + c[5] + c[6] + c[7];   // the fifth bytes follow the eight slots: slots 0-3, then slots 4-7
                else so += c[0];
                g.hi[DICT ? part : 0] = __builtin_amdgcn_raw_buffer_load_b32(slot_pay(so, c[4 * part]), mine, 0, 2);
            }
        }
        if (part == 1) {   // the tile of the packet `depth` ahead: behind all of this packet's items
            const unsigned int xs = (unsigned int)__builtin_amdgcn_readlane((int)h, 1) & 0x7fffffffu;
            const unsigned int xo = (xs == kNoTile || !tile_lane) ? kOob : (xs + 2u * (unsigned int)p) * 8u;
            load_tile(g, xo);
        }
    };

    auto consume = [&](TallRegs<DICT> &g, unsigned int h, unsigned int hnext, auto &&between) {
        const unsigned int xw = (unsigned int)__builtin_amdgcn_readlane((int)h, 1);
        const unsigned int c4 = (unsigned int)__builtin_amdgcn_readlane((int)h, 4) & 0xffffu;   // lanes with a fifth item
        if (xw & kPktNewCell) {
            // sums of the previous cell (other lanes owned these rows there) and the x-tile: LDS only, loads stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
        }
        auto stage_tile = [&]() {
            if ((xw & 0x7fffffffu) != kNoTile) {
                // (Bringing the tile in by LDS-DMA -- __builtin_amdgcn_global_load_lds, no registers, no ds_write -- was built and
                // measured: 4.71 ms against 3.99.  With two tile buffers the DMA can only start at the cell's barrier and must
                // have landed by the next one, 1.6 us later; the register path issues the loads four cells ahead.  A third buffer
                // does not fit beside 78 KB of running sums.)
                double *dst = &xt[cur ^ 1][2 * p];
                if (tile_lane) *reinterpret_cast<double2 *>(dst) = make_double2(g.x[0], g.x[1]);
                if (kPieces == 2) *reinterpret_cast<double2 *>(dst + C / 2) = make_double2(g.x[2], g.x[3]);
            }
        };
        // WHERE the wave stores its share of the NEXT cell's x-tile (it only has to be there by the next cell's barrier; lab
        // patch: tools/lab/patches).  Round 5, with the predicate-free item code: right behind the barrier (0) 3.28-3.29 ms on the
        // 2.5e6 x 1e7 slice; behind the first four items (1) 3.74-3.77; between the two item groups' loads (2) 3.44-3.46; behind all
        // of the packet's items (3) 3.44 (profiles/r05_tall_stage_position.log).  Round 4's kernel -- a select and a compare
        // per store -- had it the other way round (behind the items 3.92 against 3.99 behind the barrier: there every wave's
        // gathers queued behind 32 KB of stores): with the shorter item code the stores are out of the way before the first
        // gathers are ready to issue, and they no longer sit between a packet's items and the next packet's.
        stage_tile();
        const double *__restrict__ tile = xt[cur];
        // Four slots at a time: all twelve LDS reads (running sums, values, x) are issued together, the products do not
        // depend on the sums, and a sum that the lane has just updated is carried in a register (a lane's items of one
        // row are consecutive) -- the LDS latency is paid once per group, not once per item.  The additions of a row
        // stay one chain in list order.
        auto group = [&](const int k0) {
            unsigned int w[4], hb[4], row[4];
            double ar[4], pr[4], t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[k] = g.lo[k0 + k];
                if (DICT) {
                    hb[k] = (g.hi[DICT ? (k0 >> 2) : 0] >> (8 * k)) & 0xffu;
                    row[k] = (w[k] >> 23) | (hb[k] << 9);          // local row + 1; 0 = the scratch cell
                } else {
                    hb[k] = 0;
                    row[k] = w[k] >> kTallColBits;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ar[k] = acc[row[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pr[k] = DICT ? dv[w[k] & ((1u << kTallIdBits) - 1)] * tile[(w[k] >> kTallIdBits) & (kTallC - 1)]
                             : (POW ? abs_pow(g.val[DICT ? 0 : k0 + k], pw) * 1.0 : g.val[DICT ? 0 : k0 + k]) * tile[w[k] & (kTallC - 1)];
            }
            t[0] = ar[0] + pr[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) t[k] = ((row[k] == row[k - 1]) ? t[k - 1] : ar[k]) + pr[k];
            // Unconditional stores, no predicate at all: what is not this lane's item lands in the scratch cell acc[0] by its own
            // decoding -- a packet's items stay ONE basic block, which the instruction scheduler may interleave with the loads of
            // the packets ahead (rounds 3-4: a branch, then a select, per store: 3 vector instructions per item more)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[row[k]] = t[k];
        };
        group(0);   // (a wave without items -- nearly never -- reads cell 0 of the arrays and stores into the scratch cell)
        between();
        if (wbase < c4) group(4);
    };

    // prologue: headers of the first 2 x depth packets, payload of the first depth
#pragma unroll
    for (int u = 0; u < 2 * kDepth; ++u) hw[u] = hd[(i64)u * 8];
#pragma unroll
    for (int u = 0; u < kDepth; ++u) issue(regs[u], hw[u]);
    __syncthreads();
    for (int jj = 0; jj < npk; jj += 2 * kDepth) {
#pragma unroll
        for (int u = 0; u < 2 * kDepth; ++u) {
            consume(regs[u % kDepth], hw[u], hw[(u + kDepth) % (2 * kDepth)], [&]() { issue_part(regs[u % kDepth], hw[(u + kDepth) % (2 * kDepth)], 0); });   // packet jj + u
            issue_part(regs[u % kDepth], hw[(u + kDepth) % (2 * kDepth)], 1);                                               // (payload of packet jj + u + depth)
            hw[u] = hd[(i64)(jj + u + 2 * kDepth) * 8];                     // header of packet jj + u + 2 depth
        }
    }
    __syncthreads();   // (every wave is done with this segment's tiles and value table)
  }
    // (S > 1: row0 points into the partial-sum array -- the sums of this strip range, added up in range order by k_tall_combine)
    for (int r = p; r < wg.nrows; r += kTallT) out[wg.row0 + r] = acc[r + 1];
}


__global__ void k_tall_combine(i64 nrow, int S, const double *__restrict__ part, double *__restrict__ out, int accum) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (i64)gridDim.x * blockDim.x) {
        double a = accum ? out[r] + part[r] : part[r];
        for (int s = 1; s < S; ++s) a += part[(i64)s * nrow + r];  // strip ranges in order: deterministic
        out[r] = a;
    }
}

// the instantiation for the copy's strip width
template <bool DICT, bool ACC, bool POW>
static void tall_launch(int C, unsigned grid, int nseg, const TallWg *wgs, const double *dict, const double *x, double *out, double pw) {
    switch (C) {
    case 4096: hipLaunchKernelGGL((k_tall_spmv<DICT, ACC, POW, 4096>), dim3(grid), dim3(kTallT), 0, ctx().stream, nseg, wgs, dict, x, out, pw); break;
    case 2048: hipLaunchKernelGGL((k_tall_spmv<DICT, ACC, POW, 2048>), dim3(grid), dim3(kTallT), 0, ctx().stream, nseg, wgs, dict, x, out, pw); break;
    case 1024: hipLaunchKernelGGL((k_tall_spmv<DICT, ACC, POW, 1024>), dim3(grid), dim3(kTallT), 0, ctx().stream, nseg, wgs, dict, x, out, pw); break;
    default: SLP_REQUIRE(false, "tall cells: strip width");
    }
}

void tall_spmv(const StripJds &f, const double *x, double *out, int accum) {
    double *dst = f.S > 1 ? f.part.p : out;
    const unsigned grid = (unsigned)(f.B * f.S);
    const bool acc = accum && f.S == 1;
    const double *none = nullptr;
    if (f.D > 0) { if (acc) tall_launch<true, true, false>((int)f.C, grid, 1, f.tall_wg.p, f.dict, x, dst, 0.0); else tall_launch<true, false, false>((int)f.C, grid, 1, f.tall_wg.p, f.dict, x, dst, 0.0); }
    else { if (acc) tall_launch<false, true, false>((int)f.C, grid, 1, f.tall_wg.p, none, x, dst, 0.0); else tall_launch<false, false, false>((int)f.C, grid, 1, f.tall_wg.p, none, x, dst, 0.0); }
    if (f.S > 1)
        hipLaunchKernelGGL(k_tall_combine, dim3(grid_for(f.nrow, kBlock)), dim3(kBlock), 0, ctx().stream, f.nrow, f.S, f.part.p, out, accum);
    SLP_HIP(hipGetLastError());
}

// A composite of tall-cell copies (the chunks of a chunked matrix) in ONE launch: the descriptor table tall_fuse() laid out.
// Rows orientation: a grid over the row blocks of all chunks (2048 workgroups drained by 256 compute units: no launch ends at
// its slowest workgroup eight times per product).  Columns orientation: one workgroup per column block walks the chunks in
// order, the column sums stay in LDS (no hand-over of the sums through `out` between launches).
void tall_spmv_fused(const StripJds &f, const double *x, double *out) {
    const int nseg = f.parts_cols ? (int)f.parts.size() : 1;
    const unsigned grid = (unsigned)(f.tall_wg.n / (size_t)nseg);
    if (f.D > 0) tall_launch<true, false, false>((int)f.C, grid, nseg, f.tall_wg.p, nullptr, x, out, 0.0);
    else tall_launch<false, false, false>((int)f.C, grid, nseg, f.tall_wg.p, nullptr, x, out, 0.0);
    SLP_HIP(hipGetLastError());
}

// The fused descriptor table of a composite whose parts are all tall-cell copies of one kind (all with a dictionary or all
// without, no strip-range split; columns orientation: the same row blocks in every chunk).  False: the composite keeps its
// chunk-by-chunk launches.
bool tall_fuse(StripJds &f) {
    f.tall_wg.release();
    const char *e = getenv("SLP_TALL_FUSE");
    if (e && e[0] == '0') return false;
    if (f.parts.empty()) return false;
    const StripJds &f0 = *f.parts[0];
    for (const StripJds *g : f.parts)
        if (!g->ok || !g->tall || g->S != 1 || g->C != f0.C || (g->D > 0) != (f0.D > 0) || (f.parts_cols && (g->B != f0.B || g->tall_R != f0.tall_R || g->nrow != f0.nrow)))
            return false;
    std::vector<TallWg> all;
    for (size_t k = 0; k < f.parts.size(); ++k) {
        const StripJds &g = *f.parts[k];
        std::vector<TallWg> wg((size_t)g.B);
        SLP_HIP(hipMemcpy(wg.data(), g.tall_wg.p, wg.size() * sizeof(TallWg), hipMemcpyDeviceToHost));
        for (TallWg &w : wg) {
            w.dict = g.dict;                                   // every chunk has a value table of its own
            if (f.parts_cols) w.x0 = f.part_off[k];            // the chunk multiplies its slice of x ...
            else w.row0 += f.part_off[k];                      // ... or writes its own rows
            all.push_back(w);
        }
    }
    f.tall_wg.upload(all.data(), all.size());
    f.C = f0.C;
    return true;
}

// out = |A|^pw x over a tall-cell copy with fp64 entries (strip_spmv_abs_pow)
void tall_spmv_pow(const StripJds &f, double pw, const double *x, double *out, int accum) {
    SLP_REQUIRE(f.ok && f.tall && f.D == 0, "tall_spmv_pow: not a tall-cell copy with fp64 entries");
    double *dst = f.S > 1 ? f.part.p : out;
    const unsigned grid = (unsigned)(f.B * f.S);
    if (accum && f.S == 1) tall_launch<false, true, true>((int)f.C, grid, 1, f.tall_wg.p, nullptr, x, dst, pw);
    else tall_launch<false, false, true>((int)f.C, grid, 1, f.tall_wg.p, nullptr, x, dst, pw);
    if (f.S > 1)
        hipLaunchKernelGGL(k_tall_combine, dim3(grid_for(f.nrow, kBlock)), dim3(kBlock), 0, ctx().stream, f.nrow, f.S, f.part.p, out, accum);
    SLP_HIP(hipGetLastError());
}
}  // namespace slp

// slp_tall.h -- the tall-cell format shared by its builder (slp_tall.hip) and its product kernel (slp_tall_spmv.hip);
// the format itself is described at the top of slp_tall.hip.
#pragma once
#include "slp_common.h"

namespace slp {

constexpr int kTallC = 4096;       // columns per strip (12-bit column inside the strip); 32 KB of x
constexpr int kTallT = 1024;       // threads per workgroup = lanes a cell's rows are dealt to
constexpr int kTallSlots = 8;      // list positions per packet
constexpr int kTallDepth = 4;      // packets of payload in flight per lane; headers run 2 x this ahead
constexpr int kTallDictMax = 2048;
constexpr int kTallBuckets = 6;    // rows are ordered by min(entries in the cell, 6), descending
constexpr int kTallIdBits = 11, kTallColBits = 12, kTallRowBits = 14;
constexpr int kTallCellShift = kTallIdBits + kTallColBits + kTallRowBits;  // sort key: cell | row | column | id
static_assert(kTallBuckets * 32 == 3 * kWave, "k_tall_build scans the (count, bank class) buckets three per lane of one wave");
static_assert(kTallRmax < (1 << kTallRowBits) && kTallC == (1 << kTallColBits) && kTallDictMax == (1 << kTallIdBits), "tall geometry");
// LDS of the product kernel: sums (+ the scratch cell) + value table + two x-tiles
static_assert((kTallRmax + 1) * 8 + kTallDictMax * 8 + 2 * kTallC * 8 <= 160 * 1024, "tall cells: LDS budget");

constexpr unsigned int kPktNewCell = 0x80000000u;  // first packet of a cell: barrier, then the other x-tile
constexpr unsigned int kNoTile = 0x7fffffffu;
constexpr unsigned int kOob = 0xffffff00u;         // byte offset past every buffer descriptor: the load returns 0, no request

// 32-byte packet header (8 dwords; lane l & 7 of a wave loads dword l & 7)
struct TallPkt {
    unsigned int off;     // payload offset of the packet inside its row block (words)
    unsigned int xsrc;    // kPktNewCell | first column of the strip whose x-tile this packet carries for the NEXT cell (or kNoTile)
    unsigned int c[4];    // c[i] = w[2i] | w[2i+1] << 16,  w[k] = lanes whose list is longer than slot k of this packet
    unsigned int strip;   // strip index (diagnostics)
    unsigned int pad;
};
static_assert(sizeof(TallPkt) == 32, "packet header");

__host__ __device__ inline unsigned long long tall_value_key(unsigned long long bits) {  // as value_key() of slp_strip.hip
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}

}  // namespace slp

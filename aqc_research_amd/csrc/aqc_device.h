// Device-side descriptors shared by the host runtime and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace aqc {

constexpr int kCoefStride = 12;     // doubles per coefficient record
constexpr int kSlotsPerGroup = 5;   // inner-product slots reserved per gate group
constexpr int kMaxTileBits = 14;    // dlo: 6 bits, dhi: 8 bits
constexpr int kMaxBits = 31;

// One gate group as the kernel sees it: bit positions are LOCAL to the stage's tile.
struct DevOp {
    int32_t type;    // 0 FRONT, 1 BLOCK
    int32_t p0;      // FRONT: qubit bit; BLOCK: control bit
    int32_t p1;      // BLOCK: target bit
    int32_t flags;   // bit0 pre Rz(-pi/2) on control, bit1 post Rz(+pi/2) on target
    int32_t coef;    // coefficient record index
    int32_t slot;    // first inner-product slot (= group index * kSlotsPerGroup)
    int32_t jblock;  // block index mod L (block_range test), -1 for FRONT
    int32_t pad;
};

// One stage: which address bits are local and which groups run while the tile sits in LDS.
struct DevStage {
    int32_t k;            // local bits
    int32_t nops;         // groups in this stage
    int32_t op_begin;     // first DevOp
    int32_t ntiles;       // 2^(nbits-k)
    int32_t nub;          // number of non-local bits
    int32_t pad[3];
    int32_t ubits[32];    // non-local address bits, ascending
    uint32_t dlo[64];     // element offset of local index low 6 bits
    uint32_t dhi[256];    // element offset of local index bits 6..13
};

}  // namespace aqc

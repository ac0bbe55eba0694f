// Device-side descriptors shared by the host runtime and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace aqc {

// Coefficient record (doubles): [0..9] five (cos, sin) pairs -- half angles of t0..t3, full angle of the
// CP phase; [12..21] the same rotations in lifting form (-tan(phi/2), sin(phi)) after normalising to
// cos(phi) >= 0 (a rotation by phi equals minus the rotation by phi -+ pi); [22] product of the signs
// dropped by that normalisation.  Record n+L is constant: Rz(+-pi/2) of the Trotter decoration, and
// [2] of it holds the lane's overall sign.
constexpr int kCoefStride = 24;     // doubles per coefficient record
constexpr int kLiftOffset = 12;
constexpr int kSignOffset = 22;
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
    int32_t sub_begin;    // first DevSub (register-blocked kernels)
    int32_t nsubs;
    uint32_t fresh_nonlocal;   // sweep plans: element-offset mask of the non-local address bits that no EARLIER stage had local ("fresh":
                               // a sweep from basis states leaves w zero wherever such a bit differs from the basis index)
    int32_t ubits[32];    // non-local address bits, ascending
    uint32_t dlo[64];     // element offset of local index low 6 bits
    uint32_t dhi[256];    // element offset of local index bits 6..13
};

// ---- register-blocked ("v2") kernels ------------------------------------------------------------
// A sub-stage keeps 2^r amplitudes per thread in registers (r register bits out of the tile's local
// bits) and runs a list of micro-ops on them before going back to LDS.
enum MopKind { MOP_RY = 0, MOP_RZ = 1, MOP_RX = 2, MOP_CX = 3, MOP_CZ = 4, MOP_CP = 5, MOP_REDUCE = 6 };
enum MopFlags { MOPF_NEG_S = 1 };  // use (c, -s): inverse rotation / Rz(-pi/2) decoration

struct DevMop {
    int32_t kind;    // MopKind
    int32_t p;       // register bit (rotations) or control register bit (entanglers)
    int32_t p2;      // target register bit (entanglers)
    int32_t flags;   // MopFlags
    int32_t coef;    // offset (in doubles) of the (c, s) pair inside the lane's coefficient array
    int32_t slot;    // inner-product slot fed by this micro-op, -1 if none
    int32_t jblock;  // block index mod L for the block_range test, -1 = front layer
    int32_t pad;     // index of the theta behind this micro-op (matrix-core path: sincos straight from the thetas);
                     // -1 = the constant Trotter Rz(+-pi/2), -2 = no angle (CX, CZ, MOP_REDUCE)
    // MOP_REDUCE: reduces the (up to 4) most recent inner products of the thread; `slot` is the slot of
    // the newest one, p / p2 / coef those of the 2nd / 3rd / 4th newest (-1 = none); `flags` packs the
    // MopKind of each producer (4 bits each, newest first) for the 0.5 / 0.5j / -1j factor.
};

struct DevSub {
    int32_t bits[5];    // register bits as local bit positions, ascending
    int32_t nbits;
    int32_t nmops;
    int32_t mop_begin;
};

// Matrix-core kernels (aqc_kernels3.hip): LDS slot tables of one sub-stage.  The local index of (amplitude a, chunk c)
// is deposit(a -> register bits) | deposit(c -> the other local bits); the tables hold swz3() of the three parts
// (the swizzle is GF(2)-linear, so slot = dep_a[a] ^ dep_clo[c & 15] ^ dep_chi[c >> 4]).
// One gate group of a sub-stage as the small kernels (U builder, gradient walker) see it: the group is
// (C (x) T) ENT on register bits (pc, pt) -- or its inverse -- with 2 x 2 matrices C, T built from the thetas.
struct DevGrp {
    int32_t type;     // 0 front (Rz Ry Rz on pc; pt is any other register bit), 1 unit-block
    int32_t pc, pt;   // register bits (0..3) of control / target
    int32_t flags;    // bit0 Trotter Rz(-pi/2) on control before the block, bit1 Rz(+pi/2) on target after it
    int32_t theta0;   // index of the group's first theta
    int32_t slot0;    // first inner-product slot of the group (sweep), -1 otherwise
    int32_t jblock;   // block index mod L for the block_range test, -1 = front layer
    int32_t pad;
};

struct DevSub3 {
    int32_t mop_begin, nmops;
    int32_t bits[4];        // register bits (local bit positions, ascending)
    int32_t grp_begin, ngrp;  // gate groups of the sub-stage in execution order (DevGrp)
    uint16_t dep_a[16];
    uint16_t dep_clo[16];
    uint16_t dep_chi[64];   // up to 2^14-amplitude tiles
    // the same tables in the form the stage kernels consume (so that a sub-stage's address set-up is one vector load and
    // a few scalar loads instead of 16 loads, 13 readfirstlanes and ~70 scalar operations):
    uint32_t lane12[64];    // per lane: low half = L1 slot (dep_clo[l % 16] ^ dep_a[l / 16]), high half = L2 slot
    uint32_t kk[16][8];     // per group g: [s] = (dep_a[4 s] ^ dep_chi[g]) << 4 (L1, K-step s), [4 + r] = (dep_clo[4 r] ^ dep_chi[g]) << 4
    // Sweep from basis states (sparse lhs): which index bits of this sub-stage are still FRESH -- local bits that neither an earlier
    // stage nor an earlier sub-stage of this stage has mixed -- so that w is zero wherever they differ from the basis index:
    //   bits 0..3   fresh bits of the GROUP index (group g = high chunk bits, dep_chi)      -> whole groups of 16 chunks with w = 0
    //   bits 4..7   fresh bits of the AMPLITUDE index (register bits); bits 6, 7 select the K-step of the W product
    //   bits 8..23  local bit position (4 bits each) behind group-index bit 0..3;  bits 24..31: behind amplitude bits 2, 3
    uint32_t skipinfo;
    uint32_t pad3[3];
};

constexpr int kMaxMopsPerSub = 64;   // 8 gate groups x (7 micro-ops + 1 reduction)
constexpr int kMaxReducePerSub = 8;
constexpr int kMaxOpsPerSub = 8;     // (CP entangler: 2 reductions per block => 4 groups per sub-stage)

}  // namespace aqc

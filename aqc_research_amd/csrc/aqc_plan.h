// Gate program + stage planner (host only, no HIP dependency).
//
// The ansatz is flattened into "gate groups" in forward program order
// (core_operations.py:671-708): n front groups (Rz Ry Rz on one qubit) followed
// by L_eff unit-blocks (optional Trotter Rz, entangler, 4 rotations, optional
// Trotter Rz).  The planner cuts that program into *stages*: a stage owns a set
// of k "local" address bits; a workgroup holds the 2^k amplitudes that differ
// only in those bits in LDS and applies every group of the stage before the
// tile goes back to HBM.  One HBM round trip per stage instead of one per gate
// group is what the design is about.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace aqc {

enum GroupType { GROUP_FRONT = 0, GROUP_BLOCK = 1 };
enum GroupFlags { FLAG_PRE_RZ = 1, FLAG_POST_RZ = 2 };

struct GateGroup {
    int type;    // GroupType
    int q0, q1;  // FRONT: qubit q0 (q1 = -1); BLOCK: control q0, target q1
    int coef;    // coefficient record: FRONT q -> q, BLOCK j -> n + j (j = i mod L)
    int theta0;  // first theta index of the group
    int jblock;  // BLOCK: i mod L, else -1 (block_range test uses this, core_operations.py:969)
    int flags;   // GroupFlags (Trotter decorations, core_operations.py:962-964,1015-1017)
};

struct Program {
    int n = 0, entangler = 0, num_blocks = 0, tail_blocks = 0, tpb = 4;
    bool trotter = false, second_order = false;
    std::vector<int32_t> blocks;    // [2][L]
    std::vector<GateGroup> groups;  // forward order, size n + L_eff
    int num_thetas() const { return 3 * n + tpb * num_blocks; }
};

// Builds the forward gate program; returns "" or an error message
// (parametric_circuit.py:234-254,391-423 validity rules).
std::string build_program(int n, int entangler, const int32_t* blocks, int L, bool trotter,
                          bool second_order, Program& out);

struct SubStage {
    std::vector<int> bits;  // register bits as LOCAL bit positions of the stage (ascending, <= 4)
    std::vector<int> ops;   // indices into Program::groups, in execution order
};

struct Stage {
    std::vector<int> bits;  // local address bits, ascending; local index bit j <-> bits[j]
    std::vector<int> ops;   // indices into Program::groups, in execution order
    std::vector<SubStage> subs;  // register-blocked partition of `ops` (same overall order per qubit)
};

struct Plan {
    int nbits = 0;      // address bits = col_bits + n
    int col_bits = 0;   // low address bits that index matrix columns (0 for state vectors)
    int tile_bits = 0;  // k (every stage has min(k, nbits) local bits)
    bool inverse = false;
    std::vector<Stage> stages;
    int num_tiles() const { return 1 << (nbits - (int)stages.front().bits.size()); }
};

// Cuts the program (forward order, or reversed when inverse) into stages of at most
// tile_bits local bits; the lowest `low_bits` address bits are always local so that
// every HBM access is a run of 2^low_bits contiguous complex128.
// `subset` (optional): plan only these gate groups (indices in forward program order) -- a sub-circuit.
// `nbits` (optional): size of the register when it is not col_bits + prog.n (a sub-circuit moved to a register of its own).
Plan make_plan(const Program& prog, int col_bits, int tile_bits, int low_bits, bool inverse, const std::vector<int>* subset = nullptr, int nbits = 0);

// Partitions every stage into sub-stages: each sub-stage touches at most `reg_bits` of the stage's
// local bits (held in registers by one thread: 2^reg_bits amplitudes) and at most `max_ops` groups.
void split_substages(const Program& prog, Plan& plan, int reg_bits, int max_ops, int beam_width = 64);   // beam_width: candidates kept per level of the sub-stage search

// Throws nothing; returns "" if the plan executes every group exactly once, in an order
// compatible with per-qubit program order, using only local bits.
std::string check_plan(const Program& prog, const Plan& plan, const std::vector<int>* subset = nullptr);   // subset: exactly these groups are scheduled

}  // namespace aqc

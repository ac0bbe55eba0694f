// Lane exchanges on the vector ALU (gfx950): sums over a wave or a half-wave without the LDS crossbar (ds_bpermute) and
// without lgkmcnt waits on the critical path.  Shared by the coordinate-descent kernel and the small Jacobi SVD.
#pragma once
#include <hip/hip_runtime.h>

namespace aqc {

// (no LDS crossbar, no lgkmcnt waits on the critical path of a step)
// data-parallel-primitive moves inside a row of 16 lanes, gfx950's row / half swaps across rows
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    // (every pattern used here reads a valid lane: with bound_ctrl set the "old" operand is dead and costs no v_mov to initialise)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// v_permlane16_swap (v, v): first result = rows (0, 0, 2, 2) of v, second = rows (1, 1, 3, 3): their sum is v[l] + v[l ^ 16];
// v_permlane32_swap (v, v): first = halves (0, 0), second = halves (1, 1): sum = v[l] + v[l ^ 32]
__device__ __forceinline__ double add_xor16(double v) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double add_xor32(double v) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// Sums of four per-lane values over the wave in 7 exchange steps instead of 24: the first two steps hand half of the values to
// the partner lane while adding the other half (lane bit 0 ends up with the {g, p} choice, bit 1 with {re, im}), the other four
// add within the lanes of equal low bits.  Afterwards lane l holds the wave's total of value (l & 3): 0 gr, 1 pr, 2 gi, 3 pi.
__device__ __forceinline__ double wave_sum4(double gr, double gi, double pr, double pi, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    // step xor 1: even lanes keep (gr, gi), odd lanes keep (pr, pi)
    const double s0 = b0 ? gr : pr, s1 = b0 ? gi : pi;            // what goes to the partner
    double k0 = b0 ? pr : gr, k1 = b0 ? pi : gi;                 // what stays
    k0 += dpp<0xB1>(s0);                       // quad_perm [1, 0, 3, 2]: lane ^ 1
    k1 += dpp<0xB1>(s1);
    // step xor 2: bit 1 clear keeps the real part, set keeps the imaginary part
    double v = (b1 ? k1 : k0) + dpp<0x4E>(b1 ? k0 : k1);   // quad_perm [2, 3, 0, 1]: lane ^ 2
    v += dpp<0x124>(v);                        // row_ror:4 and row_ror:8: the four lanes of equal (lane & 3) in a row of 16
    v += dpp<0x128>(v);
    v = add_xor16(v);
    v = add_xor32(v);
    return v;
}


// The same over a HALF-wave of 32 lanes, with BITWISE IDENTICAL totals in all lanes of equal (lane & 3): every step pairs a lane
// with a partner through an involution and adds commutatively, so both partners -- and by induction all lanes of a slot -- form the
// same binary tree of additions (rotations, as in wave_sum4, give every lane its own summation order: fine where one lane's value is
// used, not where 32 lanes must apply the SAME rotation angle).  gfx9 DPP has no row_xmask; lane ^ 4 is row_half_mirror followed by
// the quad reversal (7 - i, then ^ 3), lane ^ 12 is row_mirror followed by the quad reversal (15 - i, then ^ 3).
// Afterwards lane l holds the half-wave's total of value (l & 3) -- 0 v0, 1 v2, 2 v1, 3 v3 in the argument order (v0, v1, v2, v3) --
// and quad_bcast<k> hands value slot k of a quad to all of its lanes.
__device__ __forceinline__ double lane_xor4(double v) { return dpp<0x1B>(dpp<0x141>(v)); }    // row_half_mirror, quad_perm [3, 2, 1, 0]
__device__ __forceinline__ double lane_xor12(double v) { return dpp<0x1B>(dpp<0x140>(v)); }   // row_mirror, quad_perm [3, 2, 1, 0]
__device__ __forceinline__ double halfwave_sum4(double v0, double v1, double v2, double v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const double s0 = b0 ? v0 : v2, s1 = b0 ? v1 : v3;
    double k0 = b0 ? v2 : v0, k1 = b0 ? v3 : v1;
    k0 += dpp<0xB1>(s0);
    k1 += dpp<0xB1>(s1);
    double v = (b1 ? k1 : k0) + dpp<0x4E>(b1 ? k0 : k1);
    v += lane_xor4(v);
    v += lane_xor12(v);
    return add_xor16(v);
}
// ... and over a ROW of 16 lanes (the same steps without the last, cross-row one): for work that fits 16 lanes, four groups per wave
__device__ __forceinline__ double row_sum4(double v0, double v1, double v2, double v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const double s0 = b0 ? v0 : v2, s1 = b0 ? v1 : v3;
    double k0 = b0 ? v2 : v0, k1 = b0 ? v3 : v1;
    k0 += dpp<0xB1>(s0);
    k1 += dpp<0xB1>(s1);
    double v = (b1 ? k1 : k0) + dpp<0x4E>(b1 ? k0 : k1);
    v += lane_xor4(v);
    v += lane_xor12(v);
    return v;
}
template <int K> __device__ __forceinline__ double quad_bcast(double v) { return dpp<K * 0x55>(v); }   // quad_perm [K, K, K, K]

}  // namespace aqc

// Device-side complex helpers and 2x2 rotations shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

namespace aqc {

typedef double2 cplx;  // x = re, y = im

static constexpr double kR = 0.70710678118654752440;  // cos(pi/4) = sin(pi/4)

__device__ __forceinline__ cplx cmul(cplx a, double c, double s) {  // a * (c + i s)
    return make_double2(a.x * c - a.y * s, a.y * c + a.x * s);
}
// acc += conj(a) * b
__device__ __forceinline__ void cmacc(cplx& acc, cplx a, cplx b) {
    acc.x += a.x * b.x + a.y * b.y;
    acc.y += a.x * b.y - a.y * b.x;
}
// acc -= conj(a) * b
__device__ __forceinline__ void cmsub(cplx& acc, cplx a, cplx b) {
    acc.x -= a.x * b.x + a.y * b.y;
    acc.y -= a.x * b.y - a.y * b.x;
}
// Ry = [[c,-s],[s,c]]  (elementary_operations.py:204-210)
__device__ __forceinline__ void ry2(cplx& a0, cplx& a1, double c, double s) {
    const cplx t0 = make_double2(c * a0.x - s * a1.x, c * a0.y - s * a1.y);
    const cplx t1 = make_double2(s * a0.x + c * a1.x, s * a0.y + c * a1.y);
    a0 = t0;
    a1 = t1;
}
// Rz = diag(c - i s, c + i s)  (elementary_operations.py:246-251)
__device__ __forceinline__ void rz2(cplx& a0, cplx& a1, double c, double s) {
    a0 = make_double2(c * a0.x + s * a0.y, c * a0.y - s * a0.x);
    a1 = make_double2(c * a1.x - s * a1.y, c * a1.y + s * a1.x);
}
// Rx = [[c,-is],[-is,c]]  (elementary_operations.py:159-165)
__device__ __forceinline__ void rx2(cplx& a0, cplx& a1, double c, double s) {
    const cplx t0 = make_double2(c * a0.x + s * a1.y, c * a0.y - s * a1.x);
    const cplx t1 = make_double2(s * a0.y + c * a1.x, c * a1.y - s * a0.x);
    a0 = t0;
    a1 = t1;
}

__device__ __forceinline__ unsigned insert_zero(unsigned g, int pos) {
    const unsigned lo = g & ((1u << pos) - 1u);
    return ((g >> pos) << (pos + 1)) | lo;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// DPP move of a double (two dword moves); CTRL is a DPP control word.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// lane ^ 4 inside a row of 16: banks 0/2 read 4 lanes up (row_shl:4), banks 1/3 read 4 lanes down (row_shr:4)
__device__ __forceinline__ double dpp_xor4(double v) {
    const long long bits = __double_as_longlong(v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), 0x104, 0xf, 0x5, false);
    lo = __builtin_amdgcn_update_dpp(lo, (int)(bits & 0xffffffffll), 0x114, 0xf, 0xa, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), 0x104, 0xf, 0x5, false);
    hi = __builtin_amdgcn_update_dpp(hi, (int)(bits >> 32), 0x114, 0xf, 0xa, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// Sum over the 64 lanes with DPP only (no LDS traffic); the total lands in lane 63.
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_mov<0x140, 0xf>(v);  // row_mirror        -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3 -> lane 63 = total
    return v;
}

// Sums eight per-thread doubles over each ROW of 16 lanes with a transposing butterfly made of DPP moves
// only (no LDS traffic, no ds_bpermute latency): after the three exchange steps lane l owns value (l & 7),
// one more step folds the two halves of the row.  Lanes 0..7 of every row then hold the row totals.
__device__ __forceinline__ double reduce8_row(const double (&v)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    double r[4], q[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        r[i] = keep + dpp_mov<0xB1, 0xf>(send);  // lane ^ 1
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b1 ? r[2 * i + 1] : r[2 * i], send = b1 ? r[2 * i] : r[2 * i + 1];
        q[i] = keep + dpp_mov<0x4E, 0xf>(send);  // lane ^ 2
    }
    const double keep = b2 ? q[1] : q[0], send = b2 ? q[0] : q[1];
    double u = keep + dpp_xor4(send);             // lane ^ 4
    u += dpp_mov<0x128, 0xf>(u);                  // row_ror:8 = lane ^ 8
    return u;
}

}  // namespace aqc

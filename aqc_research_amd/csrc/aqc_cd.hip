// Coordinate-descent sweep (core_op_matrix.py:765-917): per parameter one reduction pass
// (grad = 0.5j<P w|z>, prod = <w|z>) and one update pass that derives the Newton / gradient
// step on the device (every workgroup recomputes the same scalar from the same partials in the
// same order) and applies the rotation with the OLD angle to z and the NEW angle to w.
// Parameters are strictly sequential (Gauss-Seidel), so the sweep is a chain of small launches
// with no host round trip in between.
#include <hip/hip_runtime.h>

#include "aqc_lanes.h"
#include "aqc_launch.h"

namespace aqc {

typedef double2 cplx;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ size_t pair_index(size_t g, int hbit) {
    const size_t lo = g & (((size_t)1 << hbit) - 1);
    return ((g >> hbit) << (hbit + 1)) | lo;
}

// kind: 0 = Y (ry), 1 = Z (rz), 2 = X (rx).  part[blk] = {sum for grad, sum for prod}
__global__ __launch_bounds__(256) void cd_dot_kernel(const cplx* __restrict__ w, const cplx* __restrict__ z, size_t npairs,
                                                     int hbit, int kind, cplx* part) {
    __shared__ double sm[4][4];
    const size_t h = (size_t)1 << hbit;
    double gr = 0, gi = 0, pr = 0, pi = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = pair_index(g, hbit);
        const cplx w0 = w[i0], w1 = w[i0 + h], z0 = z[i0], z1 = z[i0 + h];
        // conj(a)*b = (ax bx + ay by) + i (ax by - ay bx)
        const double c00r = w0.x * z0.x + w0.y * z0.y, c00i = w0.x * z0.y - w0.y * z0.x;
        const double c11r = w1.x * z1.x + w1.y * z1.y, c11i = w1.x * z1.y - w1.y * z1.x;
        pr += c00r + c11r;
        pi += c00i + c11i;
        if (kind == 1) {
            gr += c00r - c11r;
            gi += c00i - c11i;
        } else {
            const double c01r = w0.x * z1.x + w0.y * z1.y, c01i = w0.x * z1.y - w0.y * z1.x;
            const double c10r = w1.x * z0.x + w1.y * z0.y, c10i = w1.x * z0.y - w1.y * z0.x;
            if (kind == 0) { gr += c01r - c10r; gi += c01i - c10i; } else { gr += c01r + c10r; gi += c01i + c10i; }
        }
    }
    gr = wsum(gr); gi = wsum(gi); pr = wsum(pr); pi = wsum(pi);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wave][0] = gr; sm[wave][1] = gi; sm[wave][2] = pr; sm[wave][3] = pi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0, d = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += sm[i][0]; b += sm[i][1]; c += sm[i][2]; d += sm[i][3]; }
        part[2 * blockIdx.x] = make_double2(a, b);
        part[2 * blockIdx.x + 1] = make_double2(c, d);
    }
}

__device__ __forceinline__ void rot_pair(int kind, cplx& a0, cplx& a1, double c, double s) {
    if (kind == 0) {  // Ry
        const cplx t0 = make_double2(c * a0.x - s * a1.x, c * a0.y - s * a1.y);
        const cplx t1 = make_double2(s * a0.x + c * a1.x, s * a0.y + c * a1.y);
        a0 = t0; a1 = t1;
    } else if (kind == 1) {  // Rz
        a0 = make_double2(c * a0.x + s * a0.y, c * a0.y - s * a0.x);
        a1 = make_double2(c * a1.x - s * a1.y, c * a1.y + s * a1.x);
    } else {  // Rx
        const cplx t0 = make_double2(c * a0.x + s * a1.y, c * a0.y - s * a1.x);
        const cplx t1 = make_double2(s * a0.y + c * a1.x, c * a1.y - s * a0.x);
        a0 = t0; a1 = t1;
    }
}

__global__ __launch_bounds__(256) void cd_update_kernel(cplx* __restrict__ w, cplx* __restrict__ z, size_t npairs, int hbit,
                                                        int kind, const cplx* part, int nparts, const double* theta_in,
                                                        double* theta_out, int tindex, double dim) {
    __shared__ double sh[4];
    if (threadIdx.x == 0) {  // identical in every workgroup: same partials, same order
        double gr = 0, gi = 0, pr = 0, pi = 0;
        for (int i = 0; i < nparts; ++i) { gr += part[2 * i].x; gi += part[2 * i].y; pr += part[2 * i + 1].x; pi += part[2 * i + 1].y; }
        // grad = f * S with f = 0.5 (Y) or 0.5j (Z, X)   (core_op_matrix.py:284-389)
        double g_re, g_im;
        if (kind == 0) { g_re = 0.5 * gr; g_im = 0.5 * gi; } else { g_re = -0.5 * gi; g_im = 0.5 * gr; }
        // _delta_theta (core_op_matrix.py:833-850)
        const double d2n = dim * dim;
        double d1 = (-2.0 * (pr * g_re + pi * g_im)) / d2n;
        const double d2 = (-2.0 * (g_re * g_re + g_im * g_im) + 0.5 * (pr * pr + pi * pi)) / d2n;
        const double tol = 1.4901161193847656e-08, lr = 0.19634954084936207, maxdt = 0.78539816339744831;
        double dt;
        if (d2 < tol) { d1 /= fmax(fabs(d1), 1.0); dt = -lr * d1; } else { dt = -d1 / d2; }
        const double r = fabs(dt / maxdt);
        if (!(r <= 1.0)) dt = dt / r;
        const double t_old = theta_in[tindex], t_new = t_old + dt;
        double s, c;
        sincos(0.5 * t_old, &s, &c); sh[0] = c; sh[1] = s;
        sincos(0.5 * t_new, &s, &c); sh[2] = c; sh[3] = s;
        if (blockIdx.x == 0) theta_out[tindex] = t_new;
    }
    __syncthreads();
    const double co = sh[0], so = sh[1], cn = sh[2], sn = sh[3];
    const size_t h = (size_t)1 << hbit;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = pair_index(g, hbit);
        cplx z0 = z[i0], z1 = z[i0 + h], w0 = w[i0], w1 = w[i0 + h];
        rot_pair(kind, z0, z1, co, so);  // z <- R(theta_old) z
        rot_pair(kind, w0, w1, cn, sn);  // w <- R(theta_new) w
        z[i0] = z0; z[i0 + h] = z1; w[i0] = w0; w[i0 + h] = w1;
    }
}

// CX (ent 0) or CZ (ent 1) on both operands
__global__ __launch_bounds__(256) void cd_entangle_kernel(cplx* __restrict__ w, cplx* __restrict__ z, size_t ngroups, int cbit,
                                                          int tbit, int ent) {
    const int lo = min(cbit, tbit), hi = max(cbit, tbit);
    const size_t ic = (size_t)1 << cbit, it = (size_t)1 << tbit;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = pair_index(pair_index(g, lo), hi);
        if (ent == 0) {
            cplx t = w[i0 + ic]; w[i0 + ic] = w[i0 + ic + it]; w[i0 + ic + it] = t;
            t = z[i0 + ic]; z[i0 + ic] = z[i0 + ic + it]; z[i0 + ic + it] = t;
        } else {
            cplx t = w[i0 + ic + it]; w[i0 + ic + it] = make_double2(-t.x, -t.y);
            t = z[i0 + ic + it]; z[i0 + ic + it] = make_double2(-t.x, -t.y);
        }
    }
}

// ---- the whole sweep -- or many sweeps -- as ONE persistent launch ------------------------------------------------------
// Up to 6 qubits the two d x d operands of the walk fit one workgroup's LDS (2 x 16 KiB at d = 32, 2 x 64 KiB at d = 64), so a
// workgroup per lane holds w and z for the whole walk: no launch per parameter, no HBM round trip between parameters.  Lanes
// are independent problems (random restarts of one ansatz, different targets).  Per sweep and lane:
//   theta -> (cos, sin) of every half angle (all threads in parallel);
//   z <- target (HBM -> LDS), z <- V(theta)^H z in LDS (v_dagger_mul_mat, core_op_matrix.py:562-642), w <- I;
//   the Gauss-Seidel walk of core_op_matrix.py:852-912;
//   fobj = 1 - |<w|z>|^2 / d^2 (:917).
// The walk is cut into SEGMENTS: a front-layer qubit (3 parameters) or a unit block (entangler + 4 parameters).  All gates of a
// segment act on two address bits (a, b), so a thread takes the 4 elements of w and of z that differ in exactly those bits into
// registers ONCE per segment: the entangler is a register permutation, each parameter is
//   partial inner products on the thread's two pairs -> wave reduction (4 sums in 7 exchange steps: the butterfly transposes
//   while it adds) -> the waves' partials through LDS, one barrier -> the Newton / gradient step worked out by EVERY thread from
//   the same numbers in the same order (identical arithmetic, nothing to broadcast) -> z rotated by the old angle, w by the new
//   one, in registers;
// and the group goes back to LDS when the segment ends (first version: every parameter read both operands from LDS twice and
// wrote them once -- 96 KiB of LDS traffic per parameter and lane at d = 32; now 64 KiB per SEGMENT).
struct CdSeg {
    int32_t ha, hb;       // address bits of the segment's two qubits: a = control / the front-layer qubit, b = target / any other qubit
    int32_t ent;          // 0 none (front layer), 1 CX, 2 CZ -- applied to both operands before the rotations
    int32_t nrot;         // parameters of the segment (3 or 4)
    int32_t kind[4];      // 0 Ry, 1 Rz, 2 Rx
    int32_t on_b[4];      // rotated qubit: 0 = a, 1 = b
    int32_t tindex[4];    // index of the parameter
};

__device__ __forceinline__ void cd_delta(int kind, double gr, double gi, double pr, double pi, double inv_d2n, double& dt_out) {
    // grad = f * S with f = 0.5 (Y) or 0.5j (Z, X)   (core_op_matrix.py:284-389)
    double g_re, g_im;
    if (kind == 0) { g_re = 0.5 * gr; g_im = 0.5 * gi; } else { g_re = -0.5 * gi; g_im = 0.5 * gr; }
    // _delta_theta (core_op_matrix.py:833-850); d^2 is a power of two: multiplying by its reciprocal IS the division
    double d1 = (-2.0 * (pr * g_re + pi * g_im)) * inv_d2n;
    const double d2 = (-2.0 * (g_re * g_re + g_im * g_im) + 0.5 * (pr * pr + pi * pi)) * inv_d2n;
    const double tol = 1.4901161193847656e-08, lr = 0.19634954084936207, maxdt = 0.78539816339744831;
    double dt;
    if (d2 < tol) { d1 /= fmax(fabs(d1), 1.0); dt = -lr * d1; } else { dt = -d1 / d2; }
    // |dt| <= max_delta_theta: dt / |dt / maxdt| is maxdt with the sign of dt (:849-850); NaN steps are left alone like there
    if (fabs(dt) > maxdt) dt = copysign(maxdt, dt);
    dt_out = dt;
}

// cos / sin of x for |x| <= pi / 8 (half of a step that is clamped to pi / 4): Taylor polynomials in x^2, remainders < 1e-20
__device__ __forceinline__ void sincos_small(double x, double& s, double& c) {
    const double x2 = x * x;
    double ps = -7.6471637318198164759e-13;           // -1/15!
    double pc = -1.1470745597729724714e-11;           // -1/14!
    ps = fma(ps, x2, 1.6059043836821614599e-10);      //  1/13!
    pc = fma(pc, x2, 2.0876756987868098979e-09);      //  1/12!
    ps = fma(ps, x2, -2.5052108385441718775e-08);     // -1/11!
    pc = fma(pc, x2, -2.7557319223985890653e-07);     // -1/10!
    ps = fma(ps, x2, 2.7557319223985890653e-06);      //  1/9!
    pc = fma(pc, x2, 2.4801587301587301587e-05);      //  1/8!
    ps = fma(ps, x2, -1.9841269841269841270e-04);     // -1/7!
    pc = fma(pc, x2, -1.3888888888888888889e-03);     // -1/6!
    ps = fma(ps, x2, 8.3333333333333333333e-03);      //  1/5!
    pc = fma(pc, x2, 4.1666666666666666667e-02);      //  1/4!
    ps = fma(ps, x2, -1.6666666666666666667e-01);     // -1/3!
    pc = fma(pc, x2, -0.5);                           // -1/2!
    s = fma(ps * x2, x, x);
    c = fma(pc, x2, 1.0);
}

// rotation of the thread's two pairs: (0,1),(2,3) for a gate on bit a, (0,2),(1,3) on bit b
template <int KIND, bool ON_B>
__device__ __forceinline__ void rot_group(cplx (&e)[4], double c, double s) {
    if (ON_B) { rot_pair(KIND, e[0], e[2], c, s); rot_pair(KIND, e[1], e[3], c, s); }
    else      { rot_pair(KIND, e[0], e[1], c, s); rot_pair(KIND, e[2], e[3], c, s); }
}

struct CdShared {          // what a parameter step needs besides the thread's registers
    double* th;            // LDS: thetas of the lane
    const double2* cs;     // LDS: (cos, sin) of every half angle at the start of the sweep
    double* red;           // LDS: [2 parities][4 waves][4] partial sums
    double inv_d2n;
    int tid, wave, wl;
    unsigned parity;
};

// One parameter of the walk (core_op_matrix.py:855-912) with the gate kind and the rotated bit known at compile time.
template <int KIND, bool ON_B, int G>
__device__ __forceinline__ void cd_param(cplx (&ww)[G][4], cplx (&zz)[G][4], const bool (&live)[G], int tix, CdShared& sh) {
    double gr = 0, gi = 0, pr = 0, pi = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        if (!live[j]) continue;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int i0 = ON_B ? p : 2 * p, i1 = ON_B ? p + 2 : 2 * p + 1;
            const cplx w0 = ww[j][i0], w1 = ww[j][i1], z0 = zz[j][i0], z1 = zz[j][i1];
            const double c00r = w0.x * z0.x + w0.y * z0.y, c00i = w0.x * z0.y - w0.y * z0.x;
            const double c11r = w1.x * z1.x + w1.y * z1.y, c11i = w1.x * z1.y - w1.y * z1.x;
            pr += c00r + c11r;
            pi += c00i + c11i;
            if (KIND == 1) {
                gr += c00r - c11r;
                gi += c00i - c11i;
            } else {
                const double c01r = w0.x * z1.x + w0.y * z1.y, c01i = w0.x * z1.y - w0.y * z1.x;
                const double c10r = w1.x * z0.x + w1.y * z0.y, c10i = w1.x * z0.y - w1.y * z0.x;
                if (KIND == 0) { gr += c01r - c10r; gi += c01i - c10i; } else { gr += c01r + c10r; gi += c01i + c10i; }
            }
        }
    }
    const double v = wave_sum4(gr, gi, pr, pi, sh.wl);
    double* rd = sh.red + 16 * (sh.parity & 1);   // double-buffered: the next parameter's partials never overtake a reader
    ++sh.parity;
    if (sh.wl < 4) rd[4 * sh.wave + sh.wl] = v;   // [wave][0 gr, 1 pr, 2 gi, 3 pi]
    const double2 co = sh.cs[tix];
#pragma unroll
    for (int j = 0; j < G; ++j) rot_group<KIND, ON_B>(zz[j], co.x, co.y);   // z <- R(theta_old) z: does not wait for the sums
    __syncthreads();
    gr = (rd[0] + rd[4]) + (rd[8] + rd[12]);
    pr = (rd[1] + rd[5]) + (rd[9] + rd[13]);
    gi = (rd[2] + rd[6]) + (rd[10] + rd[14]);
    pi = (rd[3] + rd[7]) + (rd[11] + rd[15]);
    double dt;
    cd_delta(KIND, gr, gi, pr, pi, sh.inv_d2n, dt);
    double sd, cd;
    sincos_small(0.5 * dt, sd, cd);
    const double cn = co.x * cd - co.y * sd, sn = co.y * cd + co.x * sd;   // half angle of theta_old + dt
#pragma unroll
    for (int j = 0; j < G; ++j) rot_group<KIND, ON_B>(ww[j], cn, sn);       // w <- R(theta_new) w
    if (sh.tid == 0) sh.th[tix] += dt;   // (read again at the start of the next sweep only)
}

template <int G>
__device__ __forceinline__ void cd_persistent_body(const CdSeg* __restrict__ prog, int nsegs, int nbits, int col_bits,
                                                   const cplx* __restrict__ target, size_t lane_stride, double* thetas, int T,
                                                   double* fobj, int nsweeps, int max_steps) {
    extern __shared__ double cd_lds[];
    const int N = 1 << nbits, ngroups = N >> 2;
    cplx* w = reinterpret_cast<cplx*>(cd_lds);
    cplx* z = w + N;
    double* th = reinterpret_cast<double*>(z + N);
    double2* cs = reinterpret_cast<double2*>(th + ((T + 1) & ~1));
    const int tid = threadIdx.x, lane = blockIdx.x;
    const double dim = (double)(1 << (nbits - col_bits));
    CdShared sh{th, cs, reinterpret_cast<double*>(cs + T), 1.0 / (dim * dim), tid, tid >> 6, tid & 63, 0u};
    const int cmask = (1 << col_bits) - 1;
    double* my_thetas = thetas + (size_t)lane * T;
    const cplx* y = target + (size_t)lane * lane_stride;
    for (int t = tid; t < T; t += 256) th[t] = my_thetas[t];
    __syncthreads();
    for (int sweep = 0; sweep < nsweeps; ++sweep) {
        for (int t = tid; t < T; t += 256) { double s, c; sincos(0.5 * th[t], &s, &c); cs[t] = make_double2(c, s); }
        for (int e = tid; e < N; e += 256) {
            z[e] = y[e];
            w[e] = make_double2(((e >> col_bits) == (e & cmask)) ? 1.0 : 0.0, 0.0);
        }
        __syncthreads();
        // ---- z <- V^H z: the segments in reverse order, inside a segment the rotations in reverse order with inverted angles,
        // then the (self-inverse) entangler; one LDS round trip and one barrier per segment
        for (int sg = nsegs - 1; sg >= 0; --sg) {
            const CdSeg& seg = prog[sg];
            const int ha = __builtin_amdgcn_readfirstlane(seg.ha), hb = __builtin_amdgcn_readfirstlane(seg.hb);
            const int ent = __builtin_amdgcn_readfirstlane(seg.ent);
            const int t0 = __builtin_amdgcn_readfirstlane(seg.tindex[0]), t1 = __builtin_amdgcn_readfirstlane(seg.tindex[1]);
            const int t2 = __builtin_amdgcn_readfirstlane(seg.tindex[2]), t3 = __builtin_amdgcn_readfirstlane(seg.tindex[3]);
            const int lo = min(ha, hb), hi = max(ha, hb), ia = 1 << ha, ib = 1 << hb;
            cplx zz[G][4];
            int base[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int g = tid + 256 * j;
                base[j] = (int)pair_index(pair_index((size_t)(g < ngroups ? g : 0), lo), hi);
                zz[j][0] = z[base[j]]; zz[j][1] = z[base[j] + ia]; zz[j][2] = z[base[j] + ib]; zz[j][3] = z[base[j] + ia + ib];
            }
            const double2 c0 = cs[t0], c1 = cs[t1], c2 = cs[t2], c3 = cs[ent ? t3 : t0];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (ent == 0) {          // front layer: Rz(t0)^-1 ... after Ry(t1)^-1 after Rz(t2)^-1 in reverse order of the walk
                    rot_group<1, false>(zz[j], c2.x, -c2.y); rot_group<0, false>(zz[j], c1.x, -c1.y); rot_group<1, false>(zz[j], c0.x, -c0.y);
                } else {
                    if (ent == 1) rot_group<2, true>(zz[j], c3.x, -c3.y); else rot_group<1, true>(zz[j], c3.x, -c3.y);
                    rot_group<0, true>(zz[j], c2.x, -c2.y); rot_group<1, false>(zz[j], c1.x, -c1.y); rot_group<0, false>(zz[j], c0.x, -c0.y);
                    if (ent == 1) { const cplx t = zz[j][1]; zz[j][1] = zz[j][3]; zz[j][3] = t; }
                    else zz[j][3] = make_double2(-zz[j][3].x, -zz[j][3].y);
                }
                if (tid + 256 * j < ngroups) {
                    z[base[j]] = zz[j][0]; z[base[j] + ia] = zz[j][1]; z[base[j] + ib] = zz[j][2]; z[base[j] + ia + ib] = zz[j][3];
                }
            }
            __syncthreads();
        }
        // ---- the walk
        int left = max_steps >= 0 ? max_steps : 0x7fffffff;   // (tests: stop after a given number of parameter steps)
        for (int sg = 0; sg < nsegs && left > 0; ++sg) {
            const CdSeg& seg = prog[sg];
            const int ha = __builtin_amdgcn_readfirstlane(seg.ha), hb = __builtin_amdgcn_readfirstlane(seg.hb);
            const int ent = __builtin_amdgcn_readfirstlane(seg.ent);
            const int t0 = __builtin_amdgcn_readfirstlane(seg.tindex[0]), t1 = __builtin_amdgcn_readfirstlane(seg.tindex[1]);
            const int t2 = __builtin_amdgcn_readfirstlane(seg.tindex[2]), t3 = __builtin_amdgcn_readfirstlane(seg.tindex[3]);
            const int lo = min(ha, hb), hi = max(ha, hb), ia = 1 << ha, ib = 1 << hb;
            cplx ww[G][4], zz[G][4];
            int base[G];
            bool live[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int g = tid + 256 * j;
                live[j] = g < ngroups;
                base[j] = (int)pair_index(pair_index((size_t)(live[j] ? g : 0), lo), hi);
                ww[j][0] = w[base[j]]; ww[j][1] = w[base[j] + ia]; ww[j][2] = w[base[j] + ib]; ww[j][3] = w[base[j] + ia + ib];
                zz[j][0] = z[base[j]]; zz[j][1] = z[base[j] + ia]; zz[j][2] = z[base[j] + ib]; zz[j][3] = z[base[j] + ia + ib];
            }
            if (ent == 0) {                       // front layer of one qubit: Rz(t2), Ry(t1), Rz(t0)  (tindex = t2, t1, t0 in walk order)
                cd_param<1, false, G>(ww, zz, live, t0, sh);
                if (--left > 0) { cd_param<0, false, G>(ww, zz, live, t1, sh);
                if (--left > 0) { cd_param<1, false, G>(ww, zz, live, t2, sh); --left; } }
            } else {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (ent == 1) {
                        cplx t = zz[j][1]; zz[j][1] = zz[j][3]; zz[j][3] = t;
                        t = ww[j][1]; ww[j][1] = ww[j][3]; ww[j][3] = t;
                    } else {
                        zz[j][3] = make_double2(-zz[j][3].x, -zz[j][3].y);
                        ww[j][3] = make_double2(-ww[j][3].x, -ww[j][3].y);
                    }
                }
                cd_param<0, false, G>(ww, zz, live, t0, sh);
                if (--left > 0) { cd_param<1, false, G>(ww, zz, live, t1, sh);
                if (--left > 0) { cd_param<0, true, G>(ww, zz, live, t2, sh);
                if (--left > 0) { if (ent == 1) cd_param<2, true, G>(ww, zz, live, t3, sh); else cd_param<1, true, G>(ww, zz, live, t3, sh); --left; } } }
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (live[j]) {
                    w[base[j]] = ww[j][0]; w[base[j] + ia] = ww[j][1]; w[base[j] + ib] = ww[j][2]; w[base[j] + ia + ib] = ww[j][3];
                    z[base[j]] = zz[j][0]; z[base[j] + ia] = zz[j][1]; z[base[j] + ib] = zz[j][2]; z[base[j] + ia + ib] = zz[j][3];
                }
            }
            __syncthreads();
        }
        // ---- fobj = 1 - |<w|z> / d|^2
        double pr = 0, pi = 0;
        for (int e = tid; e < N; e += 256) {
            const cplx a = w[e], b = z[e];
            pr += a.x * b.x + a.y * b.y;
            pi += a.x * b.y - a.y * b.x;
        }
        pr = wsum(pr); pi = wsum(pi);
        double* rd = sh.red + 16 * (sh.parity & 1);
        ++sh.parity;
        if (sh.wl == 0) { rd[4 * sh.wave] = pr; rd[4 * sh.wave + 1] = pi; }
        __syncthreads();
        if (tid == 0) {
            const double a = (rd[0] + rd[4]) + (rd[8] + rd[12]), b = (rd[1] + rd[5]) + (rd[9] + rd[13]);
            fobj[(size_t)lane * nsweeps + sweep] = 1.0 - (a * a + b * b) * sh.inv_d2n;
        }
        __syncthreads();
    }
    for (int t = tid; t < T; t += 256) my_thetas[t] = th[t];
}

// up to 5 qubits: one 4-element group per thread; three workgroups per CU (50 KiB of LDS each at 5 qubits and 735 parameters),
// i.e. three waves per SIMD: the register budget is set accordingly
__global__ __launch_bounds__(256, 3) void cd_persistent_kernel_g1(const CdSeg* __restrict__ prog, int nsegs, int nbits, int col_bits,
                                                                  const cplx* __restrict__ target, size_t lane_stride, double* thetas, int T,
                                                                  double* fobj, int nsweeps, int max_steps) {
    cd_persistent_body<1>(prog, nsegs, nbits, col_bits, target, lane_stride, thetas, T, fobj, nsweeps, max_steps);
}
// 6 qubits: four groups per thread, one workgroup per CU (128 KiB of LDS)
__global__ __launch_bounds__(256) void cd_persistent_kernel_g4(const CdSeg* __restrict__ prog, int nsegs, int nbits, int col_bits,
                                                               const cplx* __restrict__ target, size_t lane_stride, double* thetas, int T,
                                                               double* fobj, int nsweeps, int max_steps) {
    cd_persistent_body<4>(prog, nsegs, nbits, col_bits, target, lane_stride, thetas, T, fobj, nsweeps, max_steps);
}

size_t cd_persistent_lds_bytes(int nbits, int T) {
    return ((size_t)2 << nbits) * sizeof(cplx) + (size_t)((T + 1) & ~1) * sizeof(double) + (size_t)T * sizeof(double2) + 32 * sizeof(double);
}

hipError_t launch_cd_persistent(const void* prog, int nsegs, int nbits, int col_bits, const void* target, size_t lane_stride, double* thetas,
                                int T, double* fobj, int nsweeps, int max_steps, int batch, hipStream_t s) {
    const size_t lds = cd_persistent_lds_bytes(nbits, T);
    const bool big = ((size_t)1 << nbits) / 4 > 256;    // more than one 4-element group per thread (6 qubits: 4)
    static size_t granted_all[64][2] = {};   // hipFuncSetAttribute applies to the current device: one record per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    size_t (&granted)[2] = granted_all[dev];
    if (lds > granted[big] || dev == 0) {   // (device 0 doubles as the catch-all slot: always set there -- the call is cheap)
        hipError_t e = big ? hipFuncSetAttribute(reinterpret_cast<const void*>(cd_persistent_kernel_g4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(cd_persistent_kernel_g1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted[big] = lds;
    }
    if (big)
        cd_persistent_kernel_g4<<<batch, 256, lds, s>>>(static_cast<const CdSeg*>(prog), nsegs, nbits, col_bits, static_cast<const cplx*>(target),
                                                        lane_stride, thetas, T, fobj, nsweeps, max_steps);
    else
        cd_persistent_kernel_g1<<<batch, 256, lds, s>>>(static_cast<const CdSeg*>(prog), nsegs, nbits, col_bits, static_cast<const cplx*>(target),
                                                        lane_stride, thetas, T, fobj, nsweeps, max_steps);
    return hipGetLastError();
}

static unsigned cd_blocks(size_t items) { return (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (items + 255) / 256)); }

int cd_num_parts(size_t npairs) { return (int)cd_blocks(npairs); }

hipError_t launch_cd_dot(const void* w, const void* z, size_t npairs, int hbit, int kind, void* part, hipStream_t s) {
    cd_dot_kernel<<<cd_blocks(npairs), 256, 0, s>>>(static_cast<const cplx*>(w), static_cast<const cplx*>(z), npairs, hbit, kind,
                                                     static_cast<cplx*>(part));
    return hipGetLastError();
}
hipError_t launch_cd_update(void* w, void* z, size_t npairs, int hbit, int kind, const void* part, int nparts,
                            const double* theta_in, double* theta_out, int tindex, double dim, hipStream_t s) {
    cd_update_kernel<<<cd_blocks(npairs), 256, 0, s>>>(static_cast<cplx*>(w), static_cast<cplx*>(z), npairs, hbit, kind,
                                                        static_cast<const cplx*>(part), nparts, theta_in, theta_out, tindex, dim);
    return hipGetLastError();
}
hipError_t launch_cd_entangle(void* w, void* z, size_t ngroups, int cbit, int tbit, int ent, hipStream_t s) {
    cd_entangle_kernel<<<cd_blocks(ngroups), 256, 0, s>>>(static_cast<cplx*>(w), static_cast<cplx*>(z), ngroups, cbit, tbit, ent);
    return hipGetLastError();
}

}  // namespace aqc

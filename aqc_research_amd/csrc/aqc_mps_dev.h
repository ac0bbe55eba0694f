// Device bodies of the MPS engine's small-bond kernels, shared by the single-lane kernels (aqc_svd.hip: one matrix / one site pair per
// launch) and the lockstep lanes (aqc_mps_lanes.hip: a workgroup per lane): the Jacobi sweeps on a work matrix in LDS, the two-site tensor
// with its gate, the split into new site tensors, the environment steps.  Private to csrc/.
#pragma once
#include <hip/hip_runtime.h>

#include "aqc_lanes.h"
#include "aqc_math.h"

namespace aqc {

// A column whose norm is below 1e-15 of the matrix's Frobenius norm is a numerically zero singular direction (the engine
// drops singular values below 1e-14 of the largest as rank deficiency anyway) and is left alone: rotating such a column
// again and again shrinks it geometrically until its norm SQUARED underflows (1e-153 seen on a two-site tensor of a 32-qubit
// Trotter state with singular values from 0.86 down to 1e-22), where the rotation formulas lose all accuracy, the pair never
// becomes orthogonal to the relative tolerance and the sweeps never end ("no convergence within 60 sweeps").
constexpr double kNegligible2 = 1e-30;

// Whole SVD in ONE launch when the work matrix and V fit into the LDS of one workgroup (rows, cols <= 64: bond
// dimensions up to 32): 32 half-waves, one column pair each per round, __syncthreads between rounds, the sweep
// loop and the convergence test inside the kernel.  Replaces ~500 launches of jacobi_round_kernel.
constexpr int kSmallMax = 64;
// 1 / sqrt(x) and 1 / y for normal positive arguments: hardware estimate + STEPS Newton steps (one: ~1e-14, two: the last bit or two)
template <int STEPS>
__device__ __forceinline__ double rsqrt_refined(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) r = fma(r, fma(-hx * r, r, 0.5), r);
    return r;
}
template <int STEPS>
__device__ __forceinline__ double rcp_refined(double y) {
    double r = __builtin_amdgcn_rcp(y);
#pragma unroll
    for (int i = 0; i < STEPS; ++i) r = fma(r, fma(-y, r, 1.0), r);
    return r;
}
// The sweeps on a work matrix sw [cols][rows] and its rotation accumulator sv [cols][cols] (identity on entry) that live in LDS; a half-wave
// per column pair.  Returns the number of sweeps used (max_sweeps: no convergence).  Any block size (whole waves) works: with fewer than
// cols / 2 half-waves a round takes several passes.  fro2: Frobenius norm squared of the matrix (scale of kNegligible2).
// Pairing: the round-robin tournament over n2 = even(cols) seats (seat 0 stays, the others move one seat per round; a seat >= cols is a
// bye), worked out from (round, pair) -- a table in global memory put a load of several hundred cycles at the head of every round.
// G lanes per column pair: 32 (a half-wave), or 16 (a row: four pairs per wave) for work matrices of up to 16 rows -- most of a walk's
// matrices at bonds <= 8 --, which halves the waves that compete for the vector pipes once several workgroups share a CU.
template <int G>
__device__ __forceinline__ int jacobi_lds_sweeps(cplx* __restrict__ sw, cplx* __restrict__ sv, int rows, int cols, double tol, int max_sweeps, double fro2) {
    __shared__ int rotated;
    const int tid = threadIdx.x, grp = tid / G, lane = tid % G, ngrp = blockDim.x / G;
    const int n2 = cols + (cols & 1), rounds = n2 - 1, per_round = n2 >> 1;
    const double negligible = kNegligible2 * fro2;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int r = 0; r < rounds; ++r) {
            for (int pg = grp; pg < per_round; pg += ngrp) {
                int2 pq;
                {
                    int ja = pg - 1 - r, jb = n2 - 2 - pg - r;              // seats pg and n2 - 1 - pg, r rounds ago
                    ja += ja < 0 ? rounds : 0; jb += jb < 0 ? rounds : 0;   // (both lie in (-rounds, rounds))
                    const int pa = pg == 0 ? 0 : ja + 1, pb = jb + 1;
                    pq = make_int2(min(pa, pb), max(pa, pb));
                }
                if (pq.y < cols) {
                    cplx* wp = sw + pq.x * rows;   // (32-bit index arithmetic: these are LDS addresses)
                    cplx* wq = sw + pq.y * rows;
                    double a = 0.0, b = 0.0, gr = 0.0, gi = 0.0;
                    for (int i = lane; i < rows; i += G) {
                        const cplx x = wp[i], y = wq[i];
                        a += x.x * x.x + x.y * x.y;
                        b += y.x * y.x + y.y * y.y;
                        gr += x.x * y.x + x.y * y.y;
                        gi += x.x * y.y - x.y * y.x;
                    }
                    {   // the four sums over the half-wave in 6 exchange steps on the vector ALU (aqc_lanes.h; it was a 5-step
                        // butterfly of four values through the LDS crossbar: 40 ds_bpermute per round), then every lane takes all four
                        const double v = G == 32 ? halfwave_sum4(a, b, gr, gi, lane) : row_sum4(a, b, gr, gi, lane);   // slots of a quad: 0 a, 1 gr, 2 b, 3 gi
                        a = quad_bcast<0>(v); gr = quad_bcast<1>(v); b = quad_bcast<2>(v); gi = quad_bcast<3>(v);
                    }
                    const double g2 = gr * gr + gi * gi;
                    if (g2 > tol * tol * a * b && g2 != 0.0 && fmin(a, b) > negligible) {
                        if (lane == 0) rotated = 1;   // (a flag: every writer stores the same value)
                        // Rotation (x, y) <- (c x - s e y, conj(s e) x + c y), e = conj(gamma) / |gamma|, tan = t = sign(d) 2|gamma| / (|d| + h),
                        // h = sqrt(d^2 + 4|gamma|^2), d = b - a: diagonalises the pair's Gram matrix.  The parameters come from the hardware
                        // reciprocal / reciprocal-square-root estimates refined by Newton steps (a few fused multiply-adds each) in place of
                        // six dependent IEEE divisions and square roots (~20 instructions each: more than half of a round).  With
                        // u = 1 / (|d| + h): t^2 = 4 |gamma|^2 u^2 and s e = +-2 c u conj(gamma), so |gamma| itself is never needed, and
                        // c^2 (1 + t^2) = 1 to the last bits whatever the error of u -- a slightly imperfect angle is taken up by the next
                        // sweep, a non-unitary rotation would not be.
                        const double d = b - a, hx = d * d + 4.0 * g2;
                        const double h = hx * rsqrt_refined<1>(hx);                 // (the angle: a Newton step each is plenty)
                        const double u2 = 2.0 * rcp_refined<1>(fabs(d) + h);
                        const double c = rsqrt_refined<2>(1.0 + g2 * u2 * u2);
                        const double f = d >= 0.0 ? c * u2 : -c * u2;
                        const double sr = f * gr, si = -f * gi;                     // s e
                        cplx* vp = sv + pq.x * cols;
                        cplx* vq = sv + pq.y * cols;
                        for (int i = lane; i < rows + cols; i += G) {
                            cplx* xp = i < rows ? wp + i : vp + (i - rows);
                            cplx* yp = i < rows ? wq + i : vq + (i - rows);
                            const cplx x = *xp, y = *yp;
                            *xp = make_double2(c * x.x - (sr * y.x - si * y.y), c * x.y - (sr * y.y + si * y.x));
                            *yp = make_double2(c * y.x + (sr * x.x + si * x.y), c * y.y + (sr * x.y - si * x.x));
                        }
                    }
                }
            }
            __syncthreads();
        }
        const int any = rotated;
        __syncthreads();
        if (any == 0) { ++sweep; break; }
    }
    return sweep;
}
__device__ __forceinline__ int jacobi_lds_core(cplx* __restrict__ sw, cplx* __restrict__ sv, int rows, int cols, double tol, int max_sweeps, double fro2) {
    return rows <= 16 ? jacobi_lds_sweeps<16>(sw, sv, rows, cols, tol, max_sweeps, fro2) : jacobi_lds_sweeps<32>(sw, sv, rows, cols, tol, max_sweeps, fro2);
}
// |column c| of sw, fixed order (the singular value once the sweeps have converged)
__device__ __forceinline__ double lds_column_norm(const cplx* __restrict__ sw, int rows, int c) {
    double a = 0.0;
    for (int i = 0; i < rows; ++i) { const cplx x = sw[(size_t)c * rows + i]; a += x.x * x.x + x.y * x.y; }
    return sqrt(a);
}
// sum over the workgroup (whole waves, at most 16), every thread gets it; fixed order
__device__ __forceinline__ double lds_block_total(double v) {
    __shared__ double part[16], total;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
        total = t;
    }
    __syncthreads();
    return total;
}

// The same with the product T_q . [T_{q+1}[0] | T_{q+1}[1]] formed on the fly (small bonds: the two zgemm launches that produced theta0
// cost more than the arithmetic): one thread per (l, r) takes its four length-chi_m dot products, scales, applies the gate and writes the
// Jacobi work matrix.  tq: [2][chi_l][chi_m], tq1: [2][chi_m][chi_r].
template <typename G>
__device__ __forceinline__ void mps_theta_fused_body(const cplx* __restrict__ tq, const cplx* __restrict__ tq1, const double* __restrict__ lam_left, int chil,
                                                     int chim, int chir, const G& g, int mode, cplx* __restrict__ work, int idx) {
    if (idx >= chil * chir) return;
    const int l = idx / chir, r = idx - l * chir;
    const int m = 2 * chil, n = 2 * chir;
    const double sc = lam_left ? lam_left[l] : 1.0;
    cplx in[4];
    {   // the four dot products over the middle bond in one pass: four loads and four independent accumulations per k (every sum in the
        // order of k, as four separate loops would form it)
        const cplx* x0 = tq + l * chim;
        const cplx* x1 = tq + (chil + l) * chim;
        const cplx* y0 = tq1 + r;
        const cplx* y1 = tq1 + chim * chir + r;
        double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < chim; ++k) {
            const cplx u0 = x0[k], u1 = x1[k], v0 = y0[k * chir], v1 = y1[k * chir];
            re[0] += u0.x * v0.x - u0.y * v0.y; im[0] += u0.x * v0.y + u0.y * v0.x;
            re[1] += u0.x * v1.x - u0.y * v1.y; im[1] += u0.x * v1.y + u0.y * v1.x;
            re[2] += u1.x * v0.x - u1.y * v0.y; im[2] += u1.x * v0.y + u1.y * v0.x;
            re[3] += u1.x * v1.x - u1.y * v1.y; im[3] += u1.x * v1.y + u1.y * v1.x;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) in[i] = make_double2(sc * re[i], sc * im[i]);
    }
    cplx out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            re += g.m[4 * i + j].x * in[j].x - g.m[4 * i + j].y * in[j].y;
            im += g.m[4 * i + j].x * in[j].y + g.m[4 * i + j].y * in[j].x;
        }
        out[i] = make_double2(re, im);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const size_t row = a * chil + l, col = b * chir + r;
            const cplx v = out[2 * a + b];
            if (mode == 0) work[col * m + row] = v;
            else work[row * n + col] = make_double2(v.x, -v.y);
        }
}

// New site tensors from the converged Jacobi pair (W, V), keeping columns ord[0..k):
// mode 0: W = theta' Vj  =>  U S = W, V^H = Vj^H;   mode 1: W = theta'^H Vj  =>  U = Vj, S V^H = W^H.
// T_q'[a][l][j] = (U S)[(a,l), j] / lam_left[l];  T_{q+1}'[b][j][r] = V^H[j, (b,r)]  (lambda_{q+1} is already inside).
__device__ __forceinline__ void mps_split_body(const cplx* __restrict__ W, const cplx* __restrict__ V, const int* __restrict__ ord,
                                               const double* __restrict__ sigma, const double* __restrict__ lam_left, int chil, int chir, int k,
                                               int mode, double rescale, cplx* __restrict__ tq, cplx* __restrict__ tq1,
                                               const double* __restrict__ lam_new, double* __restrict__ lam_dst, size_t idx) {
    const int m = 2 * chil, n = 2 * chir;
    if (lam_new && idx < (size_t)k) lam_dst[idx] = lam_new[idx];   // the bond's new Schmidt values ride along (they were an upload of their own)
    const size_t n_left = (size_t)m * k, n_right = (size_t)k * n;
    if (idx < n_left) {
        const int row = (int)(idx / k), j = (int)(idx - (size_t)row * k);
        const int l = row % chil, c = ord[j];
        const double inv = 1.0 / (lam_left ? lam_left[l] : 1.0);
        cplx v;
        if (mode == 0) { v = W[(size_t)c * m + row]; v.x *= rescale * inv; v.y *= rescale * inv; }
        else { v = V[(size_t)c * m + row]; const double f = sigma[c] * rescale * inv; v.x *= f; v.y *= f; }
        tq[idx] = v;
    } else if (idx < n_left + n_right) {
        const size_t e = idx - n_left;                       // e = (b * k + j) * chir + r
        const int r = (int)(e % chir), bj = (int)(e / chir), j = bj % k, b = bj / k;
        const int c = ord[j], col = b * chir + r;
        cplx v;
        if (mode == 0) { v = V[(size_t)c * n + col]; v.y = -v.y; }
        else { v = W[(size_t)c * n + col]; const double f = 1.0 / sigma[c]; v.x *= f; v.y *= -f; }
        tq1[e] = v;
    }
}

// ---- environment steps of <(ops) w|z> for small bonds (every dimension <= 64): ONE launch per site instead of four zgemms (+ one
// gate1q when an operator sits on the site).  One workgroup; the intermediate of a bit lives in LDS.
struct Gate4c { cplx m[4]; };
// out[u][v] = sum_bit sum_x conj(A[bit][x][u]) (sum_y in[x][y] B'[bit][y][v]),  B'[bit] = B[bit], or gh[bit][0] B[0] + gh[bit][1] B[1]
template <typename G>
__device__ __forceinline__ void mps_env_left_body(const cplx* __restrict__ in, const cplx* __restrict__ A, const cplx* __restrict__ B,
                                                  int xa, int ua, int yb, int vb, int has_op, const G& gh, cplx* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char env_smem[];
    cplx* t = reinterpret_cast<cplx*>(env_smem);   // [xa][vb]
    const int tid = threadIdx.x;
    for (int bit = 0; bit < 2; ++bit) {
        for (int e = tid; e < xa * vb; e += 256) {
            const int x = e / vb, v = e - x * vb;
            double re = 0.0, im = 0.0;
            for (int y = 0; y < yb; ++y) {
                const cplx a = in[(size_t)x * yb + y];
                cplx b = B[((size_t)bit * yb + y) * vb + v];
                if (has_op) {
                    const cplx b0 = B[(size_t)y * vb + v], b1 = B[((size_t)yb + y) * vb + v], g0 = gh.m[2 * bit], g1 = gh.m[2 * bit + 1];
                    b = make_double2(g0.x * b0.x - g0.y * b0.y + g1.x * b1.x - g1.y * b1.y, g0.x * b0.y + g0.y * b0.x + g1.x * b1.y + g1.y * b1.x);
                }
                re += a.x * b.x - a.y * b.y;
                im += a.x * b.y + a.y * b.x;
            }
            t[e] = make_double2(re, im);
        }
        __syncthreads();
        for (int e = tid; e < ua * vb; e += 256) {
            const int u = e / vb, v = e - u * vb;
            double re = 0.0, im = 0.0;
            if (bit) { const cplx o = out[e]; re = o.x; im = o.y; }   // (written by this very thread in the first pass)
            for (int x = 0; x < xa; ++x) {
                const cplx a = A[((size_t)bit * xa + x) * ua + u], b = t[x * vb + v];
                re += a.x * b.x + a.y * b.y;      // conj(a) b
                im += a.x * b.y - a.y * b.x;
            }
            out[e] = make_double2(re, im);
        }
        __syncthreads();
    }
}
// out[x][y] = sum_bit sum_u A[bit][x][u] (sum_v Rc[u][v] conj(B[bit][y][v]))
__device__ __forceinline__ void mps_env_right_body(const cplx* __restrict__ rc, const cplx* __restrict__ A, const cplx* __restrict__ B,
                                                   int xa, int ua, int yb, int vb, cplx* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char env_smem[];
    cplx* t = reinterpret_cast<cplx*>(env_smem);   // [ua][yb]
    const int tid = threadIdx.x;
    for (int bit = 0; bit < 2; ++bit) {
        for (int e = tid; e < ua * yb; e += 256) {
            const int u = e / yb, y = e - u * yb;
            double re = 0.0, im = 0.0;
            for (int v = 0; v < vb; ++v) {
                const cplx a = rc[(size_t)u * vb + v], b = B[((size_t)bit * yb + y) * vb + v];
                re += a.x * b.x + a.y * b.y;      // a conj(b)
                im += a.y * b.x - a.x * b.y;
            }
            t[e] = make_double2(re, im);
        }
        __syncthreads();
        for (int e = tid; e < xa * yb; e += 256) {
            const int x = e / yb, y = e - x * yb;
            double re = 0.0, im = 0.0;
            if (bit) { const cplx o = out[e]; re = o.x; im = o.y; }
            for (int u = 0; u < ua; ++u) {
                const cplx a = A[((size_t)bit * xa + x) * ua + u], b = t[u * yb + y];
                re += a.x * b.x - a.y * b.y;
                im += a.x * b.y + a.y * b.x;
            }
            out[e] = make_double2(re, im);
        }
        __syncthreads();
    }
}

}  // namespace aqc

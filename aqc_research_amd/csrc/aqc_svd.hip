// Complex SVD by one-sided (Hestenes) Jacobi on the device, for the truncated 2-qubit gates of the MPS engine
// (the step qiskit-aer performs inside AerSimulator(method="matrix_product_state"), reached from
// mps_dot_objective.py:380-468 / mps_operations.py:252-257).  The work matrix W is column-major (rows x cols,
// every column contiguous); plane rotations make its columns mutually orthogonal and are accumulated in the
// column-major unitary V (cols x cols): on exit W = A V, sigma_j = |W_j|, so A = sum_j W_j V_j^H.
// One launch = one round of the round-robin tournament: cols/2 disjoint column pairs, one workgroup per pair.
// Everything is deterministic: fixed pair order, fixed-order block reductions, no float atomics.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "aqc_lanes.h"
#include "aqc_launch.h"
#include "aqc_math.h"

namespace aqc {

constexpr int kSvdThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = kSvdThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// A column whose norm is below 1e-15 of the matrix's Frobenius norm is a numerically zero singular direction (the engine
// drops singular values below 1e-14 of the largest as rank deficiency anyway) and is left alone: rotating such a column
// again and again shrinks it geometrically until its norm SQUARED underflows (1e-153 seen on a two-site tensor of a 32-qubit
// Trotter state with singular values from 0.86 down to 1e-22), where the rotation formulas lose all accuracy, the pair never
// becomes orthogonal to the relative tolerance and the sweeps never end ("no convergence within 60 sweeps").
constexpr double kNegligible2 = 1e-30;

// Frobenius norm squared of W (fixed-order sum): the scale of kNegligible2 for the multi-launch / multi-workgroup kernels
__global__ __launch_bounds__(1024) void svd_fro2_kernel(const cplx* __restrict__ W, size_t n, double* __restrict__ out) {
    __shared__ double red[16];
    double a = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) a += W[i].x * W[i].x + W[i].y * W[i].y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = t;
    }
}

// One wave per column pair (4 pairs per workgroup): no barriers, the four sums go through one shuffle butterfly.
__global__ __launch_bounds__(kSvdThreads) void jacobi_round_kernel(cplx* __restrict__ W, int rows, cplx* __restrict__ V, int cols,
                                                                   const int2* __restrict__ pairs, int npairs, double tol,
                                                                   const double* __restrict__ fro2, int* __restrict__ rotations) {
    const int lane = threadIdx.x & 63, pair = blockIdx.x * (kSvdThreads / 64) + (threadIdx.x >> 6);
    if (pair >= npairs) return;
    const int2 pq = pairs[pair];
    if (pq.x < 0 || pq.y < 0 || pq.x >= cols || pq.y >= cols) return;   // bye of an odd tournament
    cplx* wp = W + (size_t)pq.x * rows;
    cplx* wq = W + (size_t)pq.y * rows;
    double a = 0.0, b = 0.0, gr = 0.0, gi = 0.0;
    for (int i = lane; i < rows; i += 64) {
        const cplx x = wp[i], y = wq[i];
        a += x.x * x.x + x.y * x.y;
        b += y.x * y.x + y.y * y.y;
        gr += x.x * y.x + x.y * y.y;      // conj(x) * y
        gi += x.x * y.y - x.y * y.x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {   // xor butterfly: every lane ends with the same totals, fixed order
        a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64);
        gr += __shfl_xor(gr, off, 64); gi += __shfl_xor(gi, off, 64);
    }
    const double g2 = gr * gr + gi * gi;
    if (g2 <= tol * tol * a * b || g2 == 0.0) return;                    // already orthogonal (or a zero column)
    if (fmin(a, b) <= kNegligible2 * fro2[0]) return;                    // a numerically zero column: see kNegligible2
    if (lane == 0) atomicAdd(rotations, 1);
    const double g = sqrt(g2);
    const double zeta = (b - a) / (2.0 * g);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    const double er = gr / g, ei = -gi / g;                              // e^{-i phi}, phi = arg(gamma)
    // (x, y) <- (c x - s e^{-i phi} y,  s x + c e^{-i phi} y): makes the pair's Gram matrix diagonal
    cplx* vp = V + (size_t)pq.x * cols;
    cplx* vq = V + (size_t)pq.y * cols;
    for (int i = lane; i < rows + cols; i += 64) {
        cplx* xp = i < rows ? wp + i : vp + (i - rows);
        cplx* yp = i < rows ? wq + i : vq + (i - rows);
        const cplx x = *xp, y0 = *yp;
        const cplx y = make_double2(y0.x * er - y0.y * ei, y0.x * ei + y0.y * er);
        *xp = make_double2(c * x.x - s * y.x, c * x.y - s * y.y);
        *yp = make_double2(s * x.x + c * y.x, s * x.y + c * y.y);
    }
}

__global__ void svd_identity_kernel(cplx* V, int cols) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)cols * cols) V[i] = make_double2((i / cols) == (i % cols) ? 1.0 : 0.0, 0.0);
}

__global__ __launch_bounds__(kSvdThreads) void svd_norms_kernel(const cplx* __restrict__ W, int rows, double* __restrict__ sigma) {
    __shared__ double red[kSvdThreads];
    const cplx* w = W + (size_t)blockIdx.x * rows;
    double a = 0.0;
    for (int i = threadIdx.x; i < rows; i += kSvdThreads) a += w[i].x * w[i].x + w[i].y * w[i].y;
    a = block_sum(a, red);
    if (threadIdx.x == 0) sigma[blockIdx.x] = sqrt(a);
}

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
__device__ __forceinline__ int jacobi_lds_core(cplx* __restrict__ sw, cplx* __restrict__ sv, int rows, int cols, double tol, int max_sweeps, double fro2) {
    __shared__ int rotated;
    const int tid = threadIdx.x, grp = tid >> 5, lane = tid & 31, ngrp = blockDim.x >> 5;
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
                    for (int i = lane; i < rows; i += 32) {
                        const cplx x = wp[i], y = wq[i];
                        a += x.x * x.x + x.y * x.y;
                        b += y.x * y.x + y.y * y.y;
                        gr += x.x * y.x + x.y * y.y;
                        gi += x.x * y.y - x.y * y.x;
                    }
                    {   // the four sums over the half-wave in 6 exchange steps on the vector ALU (aqc_lanes.h; it was a 5-step
                        // butterfly of four values through the LDS crossbar: 40 ds_bpermute per round), then every lane takes all four
                        const double v = halfwave_sum4(a, b, gr, gi, lane);   // slots of a quad: 0 a, 1 gr, 2 b, 3 gi
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
                        for (int i = lane; i < rows + cols; i += 32) {
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
__device__ __forceinline__ void jacobi_small_body(cplx* __restrict__ W, int rows, cplx* __restrict__ V, int cols, double tol,
                                                  int max_sweeps, int* __restrict__ sweeps_out, double* __restrict__ sigma_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* sw = reinterpret_cast<cplx*>(smem);            // [cols][rows]
    cplx* sv = sw + (size_t)cols * rows;                  // [cols][cols]
    const int tid = threadIdx.x;
    double fr = 0.0;
    for (int i = tid; i < rows * cols; i += blockDim.x) { const cplx v = W[i]; sw[i] = v; fr += v.x * v.x + v.y * v.y; }
    for (int i = tid; i < cols * cols; i += blockDim.x) sv[i] = make_double2((i / cols) == (i % cols) ? 1.0 : 0.0, 0.0);
    const double fro2 = lds_block_total(fr);
    const int sweep = jacobi_lds_core(sw, sv, rows, cols, tol, max_sweeps, fro2);
    for (int i = tid; i < rows * cols; i += blockDim.x) W[i] = sw[i];
    for (int i = tid; i < cols * cols; i += blockDim.x) V[i] = sv[i];
    if (sigma_out)   // the column norms = singular values, while the columns are still in LDS (it was a launch of its own)
        for (int c = tid; c < cols; c += blockDim.x) sigma_out[c] = lds_column_norm(sw, rows, c);
    if (tid == 0) { if (sweeps_out) *sweeps_out = sweep; if (sigma_out) sigma_out[cols + 1] = (double)sweep; }   // (behind sigma and fro2: one read-back for the host)
}
__global__ __launch_bounds__(1024) void jacobi_small_kernel(cplx* __restrict__ W, int rows, cplx* __restrict__ V, int cols, double tol, int max_sweeps,
                                                            int* __restrict__ sweeps_out, double* __restrict__ sigma_out) {
    jacobi_small_body(W, rows, V, cols, tol, max_sweeps, sweeps_out, sigma_out);
}

// Two-level (block) Jacobi for work matrices beyond one workgroup's LDS (64 < columns, rows <= 512: bond dimensions
// up to 256, the cap of config 3).  The columns are cut into blocks of 8; one outer round pairs the blocks by a
// round-robin tournament and gives every block pair to one workgroup, which pulls its 16 columns of W into LDS,
// runs a complete inner tournament over them (15 rounds x 8 column pairs, one wave per pair, workgroup barriers
// only), and accumulates the 16x16 unitary Q of its rotations.  W goes back once; the 16 columns of V are updated
// as V <- V Q on the fp64 matrix cores.  The whole SVD -- all sweeps, all outer rounds, the convergence test -- is
// ONE cooperative launch: the workgroups are persistent and meet at a grid barrier after every outer round
// (511 dependent launches per sweep at 512 columns before, 63 barriers now).  Every wait is bounded: a barrier that
// does not fill raises `fail` and every workgroup leaves.
constexpr int kBlk = 8, kBlk2 = 2 * kBlk, kBlockThreads = 64 * kBlk, kBlockMaxRows = 512, kBlockMaxSweeps = 64;
typedef double double4_t __attribute__((ext_vector_type(4)));

struct BlockJacobi {
    cplx* W; cplx* V;
    const int2* bpairs;   // [rounds][per_round] block pairs; .y = -1: the block plays alone (odd number of blocks)
    int rows, cols, rounds, per_round, max_sweeps;
    double tol;
    const double* fro2;   // Frobenius norm squared of the matrix (svd_fro2_kernel): scale of kNegligible2
    int* rot;             // [kBlockMaxSweeps] rotations per sweep, zeroed by the host
    unsigned* bar;        // grid barrier counter, zeroed by the host
    int* status;          // [0] sweeps used, [1] != 0: a barrier timed out
    int debug;            // tuning builds: 1 no inner rounds, 2 no V update, 4 no grid barrier, 8 no W traffic, 16 run 10 sweeps regardless
};
#ifdef AQC_TUNING
#define SVD_DBG(a, bit) ((a).debug & (bit))
#else
#define SVD_DBG(a, bit) 0
#endif

__device__ __forceinline__ int block_col(int2 bp, int lc, int cols) {
    const int blk = lc < kBlk ? bp.x : bp.y;
    const int c = blk * kBlk + (lc & (kBlk - 1));
    return (blk >= 0 && c < cols) ? c : -1;
}

// One outer round of one block pair.  `full`: the inner tournament covers all 120 pairs of the 16 columns (first outer
// round of a sweep: this is where the pairs inside a block are met); otherwise only the 64 cross pairs (8 rounds), so
// that a sweep visits every column pair exactly once.  The rotation count goes to *s_rot.
__device__ void block_round(const BlockJacobi& a, int2 bp, bool full, cplx* sw, cplx* sq, int* s_rot) {
    constexpr int kIt = kBlockMaxRows / 64, kVt = kBlockMaxRows / (16 * (kBlockThreads / 64));
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, rows = a.rows, cols = a.cols;
    // LDS columns are padded to a multiple of 64 rows (zeros): the loops below run whole waves, without lane guards.
    // Global loads use clamped (always valid) addresses and select afterwards, so that all of them are in flight at once.
    const int nit = (rows + 63) >> 6, ldw = nit << 6;
    {   // the two columns of this wave
        cplx ld[2][kIt];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = SVD_DBG(a, 8) ? -1 : block_col(bp, 2 * w + h, cols);
            const cplx* src = a.W + (size_t)(c >= 0 ? c : 0) * rows;
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int i = lane + 64 * it;
                const cplx val = src[i < rows ? i : rows - 1];
                ld[h][it] = (c >= 0 && i < rows) ? val : make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int it = 0; it < kIt; ++it)
                if (it < nit) sw[(2 * w + h) * ldw + lane + 64 * it] = ld[h][it];
    }
    if (tid < kBlk2 * kBlk2) sq[tid] = make_double2((tid / kBlk2) == (tid % kBlk2) ? 1.0 : 0.0, 0.0);
    // the V operands of the update at the end do not depend on the inner sweep: fetch them now, under its shadow.
    // B[k][row] = V[col(k)][row]; lane l feeds B[l/16 (+4 kk)][l%16]; wave w owns the row tiles w, w + 8, ...
    int ck[4];
    cplx v[kVt][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ck[kk] = block_col(bp, 4 * kk + (lane >> 4), cols);
#pragma unroll
    for (int vt = 0; vt < kVt; ++vt) {
        const int row = 16 * (w + (kBlockThreads / 64) * vt) + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const cplx val = a.V[(size_t)(ck[kk] >= 0 ? ck[kk] : 0) * cols + (row < cols ? row : cols - 1)];
            v[vt][kk] = (ck[kk] >= 0 && row < cols && !SVD_DBG(a, 2)) ? val : make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
    const int nrounds = SVD_DBG(a, 1) ? 0 : (full ? kBlk2 - 1 : kBlk);
    // Cross rounds: wave w keeps column w of the first block (x) in registers for all 8 rounds -- only the partner
    // column goes through LDS -- and tracks |x|^2 through the rotations' own update.
    cplx x[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) x[it] = (!full && it < nit) ? sw[w * ldw + lane + 64 * it] : make_double2(0.0, 0.0);
    for (int r = 0; r < nrounds; ++r) {
        int p, q;
        if (full) {   // circle method: player 15 stays, the others rotate
            p = w == 0 ? kBlk2 - 1 : (r + w) % (kBlk2 - 1);
            q = w == 0 ? r : (r + (kBlk2 - 1) - w) % (kBlk2 - 1);
        } else {      // column w of the first block against column (w + r) mod 8 of the second
            p = w;
            q = kBlk + ((w + r) & (kBlk - 1));
        }
        if (block_col(bp, p, cols) >= 0 && block_col(bp, q, cols) >= 0) {
            cplx* wp = sw + p * ldw + lane;
            cplx* wq = sw + q * ldw + lane;
            cplx y[kIt];
            double sa = 0.0, sb = 0.0, gr = 0.0, gi = 0.0;
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                if (full && it < nit) x[it] = wp[64 * it];
                y[it] = it < nit ? wq[64 * it] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                sa += x[it].x * x[it].x + x[it].y * x[it].y;
                sb += y[it].x * y[it].x + y[it].y * y[it].y;
                gr += x[it].x * y[it].x + x[it].y * y[it].y;      // conj(x) * y
                gi += x[it].x * y[it].y - x[it].y * y[it].x;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sa += __shfl_xor(sa, off, 64); sb += __shfl_xor(sb, off, 64);
                gr += __shfl_xor(gr, off, 64); gi += __shfl_xor(gi, off, 64);
            }
            const double g2 = gr * gr + gi * gi;
            if (g2 > a.tol * a.tol * sa * sb && g2 != 0.0 && fmin(sa, sb) > kNegligible2 * a.fro2[0]) {
                if (lane == 0) atomicAdd(s_rot, 1);
                const double inv_g = rsqrt(g2);
                const double zeta = 0.5 * (sb - sa) * inv_g;
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = rsqrt(1.0 + t * t), s = c * t;
                const double er = gr * inv_g, ei = -gi * inv_g;             // e^{-i phi}, phi = arg(gamma)
#pragma unroll
                for (int it = 0; it < kIt; ++it) {
                    if (it < nit) {
                        const cplx yy = make_double2(y[it].x * er - y[it].y * ei, y[it].x * ei + y[it].y * er);
                        const cplx xn = make_double2(c * x[it].x - s * yy.x, c * x[it].y - s * yy.y);
                        if (full) wp[64 * it] = xn;
                        wq[64 * it] = make_double2(s * x[it].x + c * yy.x, s * x[it].y + c * yy.y);
                        x[it] = xn;
                    }
                }
                if (lane < kBlk2) {   // the same rotation on the columns of Q
                    const cplx xx = sq[p * kBlk2 + lane], y0 = sq[q * kBlk2 + lane];
                    const cplx yy = make_double2(y0.x * er - y0.y * ei, y0.x * ei + y0.y * er);
                    sq[p * kBlk2 + lane] = make_double2(c * xx.x - s * yy.x, c * xx.y - s * yy.y);
                    sq[q * kBlk2 + lane] = make_double2(s * xx.x + c * yy.x, s * xx.y + c * yy.y);
                }
            }
        }
        __syncthreads();
    }
    if (!full && nrounds > 0) {   // column w goes back to LDS for the store below
#pragma unroll
        for (int it = 0; it < kIt; ++it)
            if (it < nit) sw[w * ldw + lane + 64 * it] = x[it];
        __syncthreads();
    }
    if (!SVD_DBG(a, 8))
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = block_col(bp, 2 * w + h, cols);
            if (c >= 0)
                for (int i = lane; i < rows; i += 64) a.W[(size_t)c * rows + i] = sw[(2 * w + h) * ldw + i];
        }
    if ((*s_rot == 0 && !SVD_DBG(a, 16)) || SVD_DBG(a, 2)) return;   // Q = 1 (uniform: read after the barrier above)
    // V[:, col(n)] <- sum_k V[:, col(k)] Q[k][n] on the matrix cores: D[n][row] = sum_k A[n][k] B[k][row] with
    // A[n][k] = Q[k][n] = sq[n * 16 + k]; lane l feeds A[l%16][l/16] and owns D[4 j + l/16][l%16].  Complex by four
    // real products.
    double qr[4], qi[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const cplx qv = sq[(lane & 15) * kBlk2 + 4 * kk + (lane >> 4)];
        qr[kk] = qv.x; qi[kk] = qv.y;
    }
#pragma unroll
    for (int vt = 0; vt < kVt; ++vt) {
        const int row0 = 16 * (w + (kBlockThreads / 64) * vt);
        if (row0 >= cols) break;
        const int row = row0 + (lane & 15);
        double4_t dre = {0.0, 0.0, 0.0, 0.0}, dim = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            dre = __builtin_amdgcn_mfma_f64_16x16x4f64(qr[kk], v[vt][kk].x, dre, 0, 0, 0);
            dre = __builtin_amdgcn_mfma_f64_16x16x4f64(-qi[kk], v[vt][kk].y, dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f64_16x16x4f64(qi[kk], v[vt][kk].x, dim, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f64_16x16x4f64(qr[kk], v[vt][kk].y, dim, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)   // D rows 4 j + l/16 are the columns ck[j]
            if (ck[j] >= 0 && row < cols) a.V[(size_t)ck[j] * cols + row] = make_double2(dre[j], dim[j]);
    }
}

__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, int* s_fail) {
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: this workgroup's W and V columns are visible device-wide (L2 written back) before the arrival counts
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 4000000L) { *s_fail = 1; break; }   // bounded: seconds, a round takes microseconds
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave drops its stale lines of the other workgroups' columns
    return *s_fail == 0;
}

__global__ __launch_bounds__(kBlockThreads) void jacobi_block_kernel(const BlockJacobi a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* sw = reinterpret_cast<cplx*>(smem);       // [16][rows]
    cplx* sq = sw + (size_t)kBlk2 * (((a.rows + 63) >> 6) << 6);   // [16][16], behind the padded columns
    __shared__ int s_rot, s_fail;
    if (threadIdx.x == 0) { s_rot = 0; s_fail = 0; }
    __syncthreads();
    unsigned epoch = 0;
    int sweep = 0;
    for (; sweep < a.max_sweeps; ++sweep) {
        for (int r = 0; r < a.rounds; ++r) {
            const int2 bp = a.bpairs[(size_t)r * a.per_round + blockIdx.x];
            if (bp.x >= 0) block_round(a, bp, r == 0, sw, sq, &s_rot);
            __syncthreads();
            if (threadIdx.x == 0 && s_rot) { atomicAdd(a.rot + sweep, s_rot); s_rot = 0; }
            ++epoch;
            if (!SVD_DBG(a, 4) && !grid_barrier(a.bar, epoch * gridDim.x, &s_fail)) {
                if (threadIdx.x == 0) a.status[1] = 1;
                return;
            }
        }
        const int any = __hip_atomic_load(a.rot + sweep, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);   // the same on every workgroup
        if (SVD_DBG(a, 16)) { if (sweep == 9) { ++sweep; break; } continue; }
        if (any == 0) { ++sweep; break; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.status[0] = sweep;
}

bool svd_fits_block(int rows, int cols) { return rows <= kBlockMaxRows && cols <= rows && cols >= 2; }
int svd_block_size() { return kBlk; }
hipError_t launch_svd_fro2(const void* W, size_t n, double* out, hipStream_t s) {
    svd_fro2_kernel<<<1, 1024, 0, s>>>(static_cast<const cplx*>(W), n, out);
    return hipGetLastError();
}

hipError_t launch_jacobi_block(void* W, int rows, void* V, int cols, const void* bpairs, int rounds, int per_round, double tol, int max_sweeps,
                               const double* fro2, int* rot, unsigned* bar, int* status, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(cplx) * (kBlk2 * kBlockMaxRows + kBlk2 * kBlk2)));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (max_sweeps > kBlockMaxSweeps) max_sweeps = kBlockMaxSweeps;
    BlockJacobi a{static_cast<cplx*>(W), static_cast<cplx*>(V), static_cast<const int2*>(bpairs), rows, cols, rounds, per_round, max_sweeps, tol, fro2, rot, bar, status, 0};
#ifdef AQC_TUNING
    if (const char* e = getenv("AQC_SVD_DEBUG")) a.debug = atoi(e);
#endif
    void* args[] = {&a};
    const size_t lds = sizeof(cplx) * ((size_t)kBlk2 * (((rows + 63) >> 6) << 6) + kBlk2 * kBlk2);
    // cooperative: the runtime refuses the launch unless all workgroups are resident together (the barrier needs that)
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(jacobi_block_kernel), dim3(per_round), dim3(kBlockThreads), args, (unsigned)lds, s);
}

// Generic entry (aqc_svd): A row-major (m x n) -> Jacobi work matrix, and (W, V, order, sigma) -> U (m x k), Vh (k x n).
// mode 0 (n <= m): work = column-major A;  mode 1: work = column-major A^H (the conjugate of row-major A).
__global__ void svd_load_kernel(const cplx* __restrict__ a, int m, int n, int mode, cplx* __restrict__ work) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)m * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    const cplx v = a[idx];
    if (mode == 0) work[(size_t)j * m + i] = v;
    else work[idx] = make_double2(v.x, -v.y);
}
__global__ void svd_assemble_kernel(const cplx* __restrict__ W, const cplx* __restrict__ V, const int* __restrict__ ord,
                                    const double* __restrict__ sigma, int m, int n, int k, int mode, cplx* __restrict__ u,
                                    cplx* __restrict__ vh, double* __restrict__ s_sorted) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nu = (size_t)m * k, nv = (size_t)k * n;
    if (idx < nu) {
        const int i = (int)(idx / k), j = (int)(idx - (size_t)i * k), c = ord[j];
        const double inv = sigma[c] > 0.0 ? 1.0 / sigma[c] : 0.0;
        cplx v = mode == 0 ? W[(size_t)c * m + i] : V[(size_t)c * m + i];
        if (mode == 0) { v.x *= inv; v.y *= inv; }
        u[idx] = v;
    } else if (idx < nu + nv) {
        const size_t e = idx - nu;
        const int j = (int)(e / n), i = (int)(e - (size_t)j * n), c = ord[j];
        const double inv = sigma[c] > 0.0 ? 1.0 / sigma[c] : 0.0;
        cplx v = mode == 0 ? V[(size_t)c * n + i] : W[(size_t)c * n + i];
        if (mode == 1) { v.x *= inv; v.y *= inv; }
        vh[e] = make_double2(v.x, -v.y);
    } else if (idx < nu + nv + (size_t)k) {
        const int j = (int)(idx - nu - nv);
        s_sorted[j] = sigma[ord[j]];
    }
}
hipError_t launch_svd_load(const void* a, int m, int n, int mode, void* work, hipStream_t s) {
    const size_t total = (size_t)m * n;
    svd_load_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(a), m, n, mode, static_cast<cplx*>(work));
    return hipGetLastError();
}
hipError_t launch_svd_assemble(const void* W, const void* V, const int* ord, const double* sigma, int m, int n, int k, int mode, void* u, void* vh,
                               double* s_sorted, hipStream_t s) {
    const size_t total = (size_t)m * k + (size_t)k * n + k;
    svd_assemble_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(W), static_cast<const cplx*>(V), ord, sigma, m, n, k, mode,
                                                                        static_cast<cplx*>(u), static_cast<cplx*>(vh), s_sorted);
    return hipGetLastError();
}

bool svd_fits_small(int rows, int cols) { return rows <= kSmallMax && cols <= kSmallMax && cols >= 2; }
hipError_t launch_jacobi_small(void* W, int rows, void* V, int cols, double tol, int max_sweeps, int* sweeps_out, double* sigma_out, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * kSmallMax * kSmallMax * (int)sizeof(cplx));   // + the static flag < 160 KiB
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t lds = sizeof(cplx) * ((size_t)cols * rows + (size_t)cols * cols);
    const int per_round = (cols + 1) / 2;
    const int threads = std::min(1024, std::max(64, (32 * per_round + 63) & ~63));   // a half-wave per column pair of a round: no idle waves at the barrier
    jacobi_small_kernel<<<1, threads, lds, s>>>(static_cast<cplx*>(W), rows, static_cast<cplx*>(V), cols, tol, max_sweeps, sweeps_out, sigma_out);
    return hipGetLastError();
}

hipError_t launch_svd_identity(void* V, int cols, hipStream_t s) {
    const size_t total = (size_t)cols * cols;
    svd_identity_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(V), cols);
    return hipGetLastError();
}
hipError_t launch_jacobi_round(void* W, int rows, void* V, int cols, const void* pairs, int npairs, double tol, const double* fro2, int* rotations,
                               hipStream_t s) {
    jacobi_round_kernel<<<(npairs + kSvdThreads / 64 - 1) / (kSvdThreads / 64), kSvdThreads, 0, s>>>(
        static_cast<cplx*>(W), rows, static_cast<cplx*>(V), cols, static_cast<const int2*>(pairs), npairs, tol, fro2, rotations);
    return hipGetLastError();
}
hipError_t launch_svd_norms(const void* W, int rows, int cols, double* sigma, hipStream_t s) {
    svd_norms_kernel<<<cols, kSvdThreads, 0, s>>>(static_cast<const cplx*>(W), rows, sigma);
    return hipGetLastError();
}

}  // namespace aqc

// ---- pieces of a 2-qubit gate on two adjacent MPS sites ------------------------------------------------
namespace aqc {

struct Gate16 { cplx m[16]; };

// theta0: row-major (2 chi_l) x (2 chi_r), rows (a, l), columns (b, r) = T_q . [T_{q+1}[0] | T_{q+1}[1]].
// theta'[(a',l),(b',r)] = lam_left[l] * sum_{ab} G[2a'+b'][2a+b] theta0[(a,l),(b,r)], written as the Jacobi work
// matrix: mode 0 (columns <= rows) column-major theta'; mode 1 column-major theta'^H (= conj of row-major theta').
__global__ void mps_theta_kernel(const cplx* __restrict__ theta0, const double* __restrict__ lam_left, int chil, int chir, Gate16 g,
                                 int mode, cplx* __restrict__ work) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= chil * chir) return;
    const int l = idx / chir, r = idx - l * chir;
    const int m = 2 * chil, n = 2 * chir;
    const double sc = lam_left ? lam_left[l] : 1.0;
    cplx in[4], out[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const cplx v = theta0[(size_t)(a * chil + l) * n + b * chir + r];
            in[2 * a + b] = make_double2(sc * v.x, sc * v.y);
        }
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
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const cplx* x = tq + ((size_t)a * chil + l) * chim;
            const cplx* y = tq1 + (size_t)b * chim * chir + r;
            double re = 0.0, im = 0.0;
            for (int k = 0; k < chim; ++k) {
                const cplx u = x[k], v = y[(size_t)k * chir];
                re += u.x * v.x - u.y * v.y;
                im += u.x * v.y + u.y * v.x;
            }
            in[2 * a + b] = make_double2(sc * re, sc * im);
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
__global__ void mps_theta_fused_kernel(const cplx* __restrict__ tq, const cplx* __restrict__ tq1, const double* __restrict__ lam_left, int chil,
                                       int chim, int chir, Gate16 g, int mode, cplx* __restrict__ work) {
    mps_theta_fused_body(tq, tq1, lam_left, chil, chim, chir, g, mode, work, (int)(blockIdx.x * blockDim.x + threadIdx.x));
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
__global__ void mps_split_kernel(const cplx* __restrict__ W, const cplx* __restrict__ V, const int* __restrict__ ord,
                                 const double* __restrict__ sigma, const double* __restrict__ lam_left, int chil, int chir, int k,
                                 int mode, double rescale, cplx* __restrict__ tq, cplx* __restrict__ tq1,
                                 const double* __restrict__ lam_new, double* __restrict__ lam_dst) {
    mps_split_body(W, V, ord, sigma, lam_left, chil, chir, k, mode, rescale, tq, tq1, lam_new, lam_dst, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// t[row][col] *= (mul ? lam[col] : 1 / lam[col])   -- import (Gamma -> Gamma lambda) and export of MPS tensors
__global__ void mps_colscale_kernel(cplx* t, const double* lam, size_t rows, int cols, int mul) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const double s = mul ? lam[i % cols] : 1.0 / lam[i % cols];
    t[i].x *= s;
    t[i].y *= s;
}

hipError_t launch_mps_theta(const void* theta0, const double* lam_left, int chil, int chir, const double* g16, int mode, void* work, hipStream_t s) {
    Gate16 g;
    for (int i = 0; i < 16; ++i) g.m[i] = make_double2(g16[2 * i], g16[2 * i + 1]);
    const int total = chil * chir;
    mps_theta_kernel<<<(total + 255) / 256, 256, 0, s>>>(static_cast<const cplx*>(theta0), lam_left, chil, chir, g, mode, static_cast<cplx*>(work));
    return hipGetLastError();
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
__global__ __launch_bounds__(256) void mps_env_left_kernel(const cplx* __restrict__ in, const cplx* __restrict__ A, const cplx* __restrict__ B,
                                                           int xa, int ua, int yb, int vb, int has_op, Gate4c gh, cplx* __restrict__ out) {
    mps_env_left_body(in, A, B, xa, ua, yb, vb, has_op, gh, out);
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
__global__ __launch_bounds__(256) void mps_env_right_kernel(const cplx* __restrict__ rc, const cplx* __restrict__ A, const cplx* __restrict__ B,
                                                            int xa, int ua, int yb, int vb, cplx* __restrict__ out) {
    mps_env_right_body(rc, A, B, xa, ua, yb, vb, out);
}
bool mps_env_fits_small(int xa, int ua, int yb, int vb) { return xa <= 64 && ua <= 64 && yb <= 64 && vb <= 64; }
hipError_t launch_mps_env_left(const void* in, const void* A, const void* B, int xa, int ua, int yb, int vb, const double* gh8, void* out, hipStream_t s) {
    Gate4c g;
    for (int i = 0; i < 4; ++i) g.m[i] = gh8 ? make_double2(gh8[2 * i], gh8[2 * i + 1]) : make_double2(0.0, 0.0);
    mps_env_left_kernel<<<1, 256, sizeof(cplx) * (size_t)xa * vb, s>>>(static_cast<const cplx*>(in), static_cast<const cplx*>(A), static_cast<const cplx*>(B),
                                                                       xa, ua, yb, vb, gh8 ? 1 : 0, g, static_cast<cplx*>(out));
    return hipGetLastError();
}
hipError_t launch_mps_env_right(const void* rc, const void* A, const void* B, int xa, int ua, int yb, int vb, void* out, hipStream_t s) {
    mps_env_right_kernel<<<1, 256, sizeof(cplx) * (size_t)ua * yb, s>>>(static_cast<const cplx*>(rc), static_cast<const cplx*>(A), static_cast<const cplx*>(B),
                                                                        xa, ua, yb, vb, static_cast<cplx*>(out));
    return hipGetLastError();
}

hipError_t launch_mps_theta_fused(const void* tq, const void* tq1, const double* lam_left, int chil, int chim, int chir, const double* g16, int mode,
                                  void* work, hipStream_t s) {
    Gate16 g;
    for (int i = 0; i < 16; ++i) g.m[i] = make_double2(g16[2 * i], g16[2 * i + 1]);
    const int total = chil * chir;
    mps_theta_fused_kernel<<<(total + 127) / 128, 128, 0, s>>>(static_cast<const cplx*>(tq), static_cast<const cplx*>(tq1), lam_left, chil, chim, chir, g,
                                                               mode, static_cast<cplx*>(work));
    return hipGetLastError();
}
hipError_t launch_mps_split(const void* W, const void* V, const int* ord, const double* sigma, const double* lam_left, int chil, int chir,
                            int k, int mode, double rescale, void* tq, void* tq1, const double* lam_new, double* lam_dst, hipStream_t s) {
    const size_t total = (size_t)2 * chil * k + (size_t)k * 2 * chir;
    mps_split_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<const cplx*>(W), static_cast<const cplx*>(V), ord, sigma, lam_left,
                                                                     chil, chir, k, mode, rescale, static_cast<cplx*>(tq), static_cast<cplx*>(tq1),
                                                                     lam_new, lam_dst);
    return hipGetLastError();
}
hipError_t launch_mps_colscale(void* t, const double* lam, size_t rows, int cols, int mul, hipStream_t s) {
    const size_t total = rows * cols;
    mps_colscale_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(t), lam, rows, cols, mul);
    return hipGetLastError();
}

// ---- device-resident lanes (aqc_mps_batch.cpp).  blockIdx = lane; bond dimensions, thetas and tensors of the lane are read on the device.
struct GateRef { const cplx* m; };
__device__ __forceinline__ void lane_rot(const LaneRot& r, const double* __restrict__ th, cplx* o) {   // o = 2 x 2 of one rotation
    const double t = r.idx >= 0 ? r.scale * th[r.idx] : r.scale;
    double s, c;
    sincos(0.5 * t, &s, &c);
    if (r.kind == 1) { o[0] = make_double2(c, -s); o[1] = make_double2(0.0, 0.0); o[2] = make_double2(0.0, 0.0); o[3] = make_double2(c, s); }
    else if (r.kind == 2) { o[0] = make_double2(c, 0.0); o[1] = make_double2(-s, 0.0); o[2] = make_double2(s, 0.0); o[3] = make_double2(c, 0.0); }
    else { o[0] = make_double2(c, 0.0); o[1] = make_double2(0.0, -s); o[2] = make_double2(0.0, -s); o[3] = make_double2(c, 0.0); }
}
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
// u = r[0] r[1] r[2] of the lane
__device__ __forceinline__ void lane_gate1_matrix(const LaneGate1& g, const double* __restrict__ th, cplx* u) {
    lane_rot(g.r[0], th, u);
#pragma unroll
    for (int k = 1; k < 3; ++k)
        if (g.r[k].kind) {
            cplx v[4];
            lane_rot(g.r[k], th, v);
            const cplx p0 = cadd(cmul(u[0], v[0]), cmul(u[1], v[2])), p1 = cadd(cmul(u[0], v[1]), cmul(u[1], v[3]));
            const cplx p2 = cadd(cmul(u[2], v[0]), cmul(u[3], v[2])), p3 = cadd(cmul(u[2], v[1]), cmul(u[3], v[3]));
            u[0] = p0; u[1] = p1; u[2] = p2; u[3] = p3;
        }
}
__device__ __forceinline__ void lane_apply_gate1(cplx* __restrict__ t, int ne, const cplx* u, int first, int step) {   // T[2][ne] <- u T
    for (int i = first; i < ne; i += step) {
        const cplx a0 = t[i], a1 = t[ne + i];
        t[i] = cadd(cmul(u[0], a0), cmul(u[1], a1));
        t[ne + i] = cadd(cmul(u[2], a0), cmul(u[3], a1));
    }
}
__global__ __launch_bounds__(128) void lanes_gate1_kernel(LaneMps a, LaneMps b, const LaneOp1* __restrict__ ops, LaneOp1 one, const double* __restrict__ thetas,
                                                          int T, int lanes) {   // blockIdx.y = state and lane, blockIdx.z = gate of the table
    const int l = blockIdx.y % lanes;
    const LaneMps& m = blockIdx.y < (unsigned)lanes ? a : b;
    const LaneOp1 op = ops ? ops[blockIdx.z] : one;
    const int* dims = m.dims + (size_t)l * (m.n + 1);
    cplx u[4];
    lane_gate1_matrix(op.g, thetas + (size_t)l * T, u);
    lane_apply_gate1(static_cast<cplx*>(m.T) + ((size_t)l * m.n + op.q) * kLaneSite, dims[op.q] * dims[op.q + 1], u, blockIdx.x * blockDim.x + threadIdx.x,
                     gridDim.x * blockDim.x);
}

#ifdef AQC_TUNING   // in-kernel stamps (diagnostic builds only): where the time of lanes_gate2_kernel goes (workgroup 0)
__device__ unsigned long long g_gate2_stamps[16];
#define G2_STAMP(slot) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if (slot) atomicAdd(&g_gate2_stamps[slot], now_ - last_); else atomicAdd(&g_gate2_stamps[0], 1ull); last_ = now_; } } while (0)
#define G2_COUNT(slot, v) atomicAdd(&g_gate2_stamps[slot], (unsigned long long)(v))
#else
#define G2_STAMP(slot) do { } while (0)
#define G2_COUNT(slot, v) do { } while (0)
#endif
// One truncated 2-qubit gate on the sites (q, q + 1) of every lane, the whole of it in ONE workgroup per lane: two-site tensor with the gate
// -> Jacobi work matrix in LDS, the sweeps, singular values, order / rank / truncation (the rule of gate_adjacent, aqc_mps_engine.cpp, on the
// lane's own values), new site tensors, Schmidt values and bond dimension.
__global__ __launch_bounds__(1024) void lanes_gate2_kernel(LaneMps m0, LaneMps m1, int lanes, const LaneOp2* __restrict__ ops, LaneOp2 one,
                                                           const double* __restrict__ thetas, int T, double trunc_thr, int max_bond, double tol, int max_sweeps,
                                                           int* __restrict__ status, int* __restrict__ peak, unsigned lds_elems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ cplx gm[16];
    __shared__ double sig[2 * kLaneCap];
    __shared__ int ord[2 * kLaneCap];
    __shared__ double sorted[2 * kLaneCap];
    __shared__ int k_sh;
    __shared__ double rescale_sh;
    const int l = blockIdx.x % lanes, tid = threadIdx.x;
    const LaneMps& m = blockIdx.x < (unsigned)lanes ? m0 : m1;   // (the two operands of the gradient walk take every gate in one launch)
    const LaneOp2 op = ops ? ops[blockIdx.y] : one;              // (blockIdx.y: the gates of one layer, on disjoint sites)
    const int n = m.n, q = op.q;
    const LaneGate2& g = op.g;
#ifdef AQC_TUNING
    unsigned long long last_ = 0;
#endif
    G2_STAMP(0);   // (tuning builds stamp workgroup 0 of the first gate)
    int* dims = m.dims + (size_t)l * (n + 1);
    const int chil = dims[q], chim = dims[q + 1], chir = dims[q + 2];
    const int rows = 2 * chil, cols = 2 * chir, mode = cols <= rows ? 0 : 1;
    const int wr = mode == 0 ? rows : cols, wc = mode == 0 ? cols : rows;
    // LDS: the work matrix, then V -- and, before the sweeps start, the two site tensors in V's place
    const int nq = 2 * chil * chim, nq1 = 2 * chim * chir;
    if ((unsigned)(wr * wc + max(wc * wc, nq + nq1)) > lds_elems) {   // the launch was sized for smaller bonds: the host repeats the evaluation at full size
        if (tid == 0) atomicOr(&status[l], kLaneLdsShort);
        return;
    }
    if (tid < 16) {   // the 4 x 4 gate, index 2 bit_q + bit_{q+1}
        const int i = tid >> 2, j = tid & 3;
        const int si = g.flip ? ((i & 1) << 1 | (i >> 1)) : i, sj = g.flip ? ((j & 1) << 1 | (j >> 1)) : j;
        cplx v = make_double2(0.0, 0.0);
        if (g.kind == 0) v.x = (si == sj && (si == 0 || si == 3)) || (si == 1 && sj == 2) || (si == 2 && sj == 1) ? 1.0 : 0.0;
        else if (g.kind == 1) v.x = (si == sj && si < 2) || (si == 2 && sj == 3) || (si == 3 && sj == 2) ? 1.0 : 0.0;
        else if (g.kind == 2) v.x = si == sj ? (si == 3 ? -1.0 : 1.0) : 0.0;
        else if (si == sj) {
            if (si < 3) v.x = 1.0;
            else { const double t = g.scale * thetas[(size_t)l * T + g.idx]; v = make_double2(cos(t), sin(t)); }
        }
        gm[tid] = v;
    }
    cplx* sw = reinterpret_cast<cplx*>(smem);   // [wc][wr]
    cplx* sv = sw + (size_t)wc * wr;             // [wc][wc]
    cplx* tq = static_cast<cplx*>(m.T) + ((size_t)l * n + q) * kLaneSite;
    cplx* tq1 = tq + kLaneSite;
    const int nb = n > 1 ? n - 1 : 1;
    const double* lam_left = q > 0 ? m.lam + ((size_t)l * nb + (q - 1)) * kLaneCap : nullptr;
    // the two site tensors come in once, coalesced (the dot products of the two-site tensor read every element 2 chi times: from global
    // memory, one dependent load after the other, that was a third of the kernel)
    for (int i = tid; i < nq; i += blockDim.x) sv[i] = tq[i];
    for (int i = tid; i < nq1; i += blockDim.x) sv[nq + i] = tq1[i];
    __syncthreads();
    G2_STAMP(1);
    const GateRef gref{gm};
    for (int idx = tid; idx < chil * chir; idx += blockDim.x) mps_theta_fused_body(sv, sv + nq, lam_left, chil, chim, chir, gref, mode, sw, idx);
    __syncthreads();
    G2_STAMP(2);
    for (int i = tid; i < wc * wc; i += blockDim.x) sv[i] = make_double2((i / wc) == (i % wc) ? 1.0 : 0.0, 0.0);
    double fr = 0.0;
    for (int i = tid; i < wr * wc; i += blockDim.x) { const cplx v = sw[i]; fr += v.x * v.x + v.y * v.y; }
    const double fro2 = lds_block_total(fr);
    G2_STAMP(3);
    const int sweeps = jacobi_lds_core(sw, sv, wr, wc, tol, max_sweeps, fro2);
    G2_STAMP(4);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { G2_COUNT(8, sweeps); G2_COUNT(9, wc); G2_COUNT(10, wr); G2_COUNT(12, sweeps * (wc + (wc & 1) - 1)); }
    for (int c = tid; c < wc; c += blockDim.x) sig[c] = lds_column_norm(sw, wr, c);
    __syncthreads();
    G2_STAMP(5);
    for (int c = tid; c < wc; c += blockDim.x) {   // stable descending order by counting
        const double sc = sig[c];
        int r = 0;
        for (int j = 0; j < wc; ++j) { const double sj = sig[j]; r += (sj > sc || (sj == sc && j < c)) ? 1 : 0; }
        ord[r] = c;
        sorted[r] = sc;
    }
    __syncthreads();
    G2_STAMP(6);
    if (tid < 64) {   // rank and truncation by the rule of gate_adjacent (aqc_mps_engine.cpp), on the first wave: lane j holds the j-th largest
                      // singular value (at most 64); sums by butterflies and a suffix scan instead of one thread's loops over LDS (which were
                      // a quarter of the kernel) -- the sums may differ from the host's sequential ones in the last bit
        const int j = tid;
        const double v = j < wc ? sorted[j] : 0.0, v2 = v * v;
        const double smax = __shfl(v, 0, 64);
        int flags = sweeps >= max_sweeps ? kLaneNoConv : 0;
        int k = 1;
        double rescale = 1.0;
        if (!(smax > 0.0) || !isfinite(smax)) flags |= kLaneZero;
        else {
            double total = v2;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
            k = __popcll(__ballot(v > 1e-14 * smax));                   // values above the rank-deficiency floor (they are sorted)
            if (max_bond > 0) k = min(k, max_bond);
            if (trunc_thr > 0.0) {                                       // drop the tail while its weight stays below the threshold
                double tail = j < k ? v2 : 0.0;                          // -> sum of the values j .. k - 1
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_down(tail, off, 64); if (j + off < 64) tail += t; }
                k -= __popcll(__ballot(j >= 1 && j < k && tail < trunc_thr));
            }
            if (k > kLaneCap) { flags |= kLaneOverflow; k = kLaneCap; }
            double kept = j < k ? v2 : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
            rescale = kept > 0.0 ? sqrt(total / kept) : 1.0;
            if (j == 0) atomicAdd(&m.discarded[l], total - kept);   // (the gates of a layer run side by side: several may add to the lane's weight)
        }
        if (j == 0) {
            k_sh = k; rescale_sh = rescale;
            if (flags) atomicOr(&status[l], flags);
            atomicMax(&peak[l], k);
        }
    }
    __syncthreads();
    G2_STAMP(7);
    const int k = k_sh;
    const double rescale = rescale_sh;
    const size_t total_el = (size_t)rows * k + (size_t)k * cols;
    for (size_t idx = tid; idx < total_el; idx += blockDim.x)
        mps_split_body(sw, sv, ord, sig, lam_left, chil, chir, k, mode, rescale, tq, tq1, nullptr, nullptr, idx);
    double* lam_dst = m.lam + ((size_t)l * nb + q) * kLaneCap;
    for (int j = tid; j < k; j += blockDim.x) lam_dst[j] = sorted[j] * rescale;
    if (tid == 0) dims[q + 1] = k;
    G2_STAMP(11);
}
#ifdef AQC_TUNING
extern "C" void aqc_dbg_gate2_stamps() {
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gate2_stamps), sizeof h) != hipSuccess || h[0] == 0) return;
    const double n = (double)h[0];
    fprintf(stderr, "aqc_hip stamps: lanes_gate2 workgroup 0, %.0f launches: mean work matrix %.1f x %.1f, %.2f sweeps; cycles: stage %.0f theta %.0f fro2 %.0f "
            "sweeps %.0f (%.0f per round) sigma %.0f sort %.0f decision %.0f split %.0f\n", n, h[10] / n, h[9] / n, h[8] / n, h[1] / n, h[2] / n, h[3] / n,
            h[4] / n, (double)h[4] / (double)std::max<unsigned long long>(h[12], 1), h[5] / n, h[6] / n, h[7] / n, h[11] / n);
}
#endif

__global__ __launch_bounds__(256) void lanes_env_left_kernel(LaneMps w, LaneMps z, int p, const cplx* __restrict__ in, size_t in_stride, cplx* __restrict__ out,
                                                             size_t out_stride, int has_op, Gate4c gh) {
    const int l = blockIdx.x, n = w.n;
    const int* dw = w.dims + (size_t)l * (n + 1);
    const int* dz = z.dims + (size_t)l * (n + 1);
    mps_env_left_body(in + (size_t)l * in_stride, static_cast<const cplx*>(w.T) + ((size_t)l * n + p) * kLaneSite,
                      static_cast<const cplx*>(z.T) + ((size_t)l * n + p) * kLaneSite, dw[p], dw[p + 1], dz[p], dz[p + 1], has_op, gh, out + (size_t)l * out_stride);
}
__global__ __launch_bounds__(256) void lanes_env_right_kernel(LaneMps w, LaneMps z, int p, const cplx* __restrict__ in, size_t in_stride, cplx* __restrict__ out,
                                                              size_t out_stride) {
    const int l = blockIdx.x, n = w.n;
    const int* dw = w.dims + (size_t)l * (n + 1);
    const int* dz = z.dims + (size_t)l * (n + 1);
    mps_env_right_body(in + (size_t)l * in_stride, static_cast<const cplx*>(w.T) + ((size_t)l * n + p) * kLaneSite,
                       static_cast<const cplx*>(z.T) + ((size_t)l * n + p) * kLaneSite, dw[p], dw[p + 1], dz[p], dz[p + 1], out + (size_t)l * out_stride);
}
// vals[lane][slot] = sum_i e[i] conj(rc[i]) over the dims[hi + 1] of both operands (fixed order)
__global__ __launch_bounds__(256) void lanes_env_dot_kernel(LaneMps w, LaneMps z, int hi, const cplx* __restrict__ e, size_t e_stride, const cplx* __restrict__ rc,
                                                            size_t rc_stride, cplx* __restrict__ vals, int nvals, int slot) {
    const int l = blockIdx.x, n = w.n;
    const int count = w.dims[(size_t)l * (n + 1) + hi + 1] * z.dims[(size_t)l * (n + 1) + hi + 1];
    __shared__ double sr[256], si[256];
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) {
        const cplx a = e[(size_t)l * e_stride + i], b = rc[(size_t)l * rc_stride + i];
        re += a.x * b.x + a.y * b.y;
        im += a.y * b.x - a.x * b.y;
    }
    sr[threadIdx.x] = re; si[threadIdx.x] = im;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sr[threadIdx.x] += sr[threadIdx.x + s]; si[threadIdx.x] += si[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) vals[(size_t)l * nvals + slot] = make_double2(sr[0], si[0]);
}
// One parameter of the gradient walk in ONE launch: the rotation on site q of both operands, then <P w|z> with the Pauli P on that site from
// the environments on either side: E = step(L[q], site q seen through P^H), vals[lane][slot] = sum E conj(R[q]).  (They were three launches.)
__global__ __launch_bounds__(256) void lanes_grad_step_kernel(LaneMps w, LaneMps z, int q, LaneGate1 g, const double* __restrict__ thetas, int T,
                                                              const cplx* __restrict__ env_l, size_t l_stride, const cplx* __restrict__ env_r, size_t r_stride,
                                                              Gate4c gh, cplx* __restrict__ scratch, cplx* __restrict__ vals, int nvals, int slot) {
    const int l = blockIdx.x, n = w.n, tid = threadIdx.x;
    const int* dw = w.dims + (size_t)l * (n + 1);
    const int* dz = z.dims + (size_t)l * (n + 1);
    const int xa = dw[q], ua = dw[q + 1], yb = dz[q], vb = dz[q + 1];
    cplx* A = static_cast<cplx*>(w.T) + ((size_t)l * n + q) * kLaneSite;
    cplx* B = static_cast<cplx*>(z.T) + ((size_t)l * n + q) * kLaneSite;
    {
        cplx u[4];
        lane_gate1_matrix(g, thetas + (size_t)l * T, u);
        lane_apply_gate1(A, xa * ua, u, tid, 256);
        lane_apply_gate1(B, yb * vb, u, tid, 256);
    }
    __syncthreads();   // (workgroup-scope release / acquire: the environment step below reads what other threads have just written)
    cplx* e = scratch + (size_t)l * kLaneEnv;
    mps_env_left_body(env_l + (size_t)l * l_stride, A, B, xa, ua, yb, vb, 1, gh, e);
    __shared__ double sr[256], si[256];
    const cplx* rc = env_r + (size_t)l * r_stride;
    double re = 0.0, im = 0.0;
    for (int i = tid; i < ua * vb; i += 256) {
        const cplx a = e[i], b = rc[i];
        re += a.x * b.x + a.y * b.y;
        im += a.y * b.x - a.x * b.y;
    }
    sr[tid] = re; si[tid] = im;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sr[tid] += sr[tid + s]; si[tid] += si[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) vals[(size_t)l * nvals + slot] = make_double2(sr[0], si[0]);
}
// every lane <- the computational-basis state bits[lane][site]: site tensors [2][1][1], Schmidt values 1, bond dimensions 1
__global__ void lanes_basis_kernel(LaneMps m, const unsigned char* __restrict__ bits, int lanes) {
    const int l = blockIdx.x, n = m.n, nb = n > 1 ? n - 1 : 1;
    for (int q = threadIdx.x; q <= n; q += blockDim.x) {
        m.dims[(size_t)l * (n + 1) + q] = 1;
        if (q == n) break;
        cplx* t = static_cast<cplx*>(m.T) + ((size_t)l * n + q) * kLaneSite;
        const int bit = bits[(size_t)l * n + q] ? 1 : 0;
        t[0] = make_double2(bit ? 0.0 : 1.0, 0.0);
        t[1] = make_double2(bit ? 1.0 : 0.0, 0.0);
        if (q < n - 1) m.lam[((size_t)l * nb + q) * kLaneCap] = 1.0;
    }
    if (threadIdx.x == 0) m.discarded[l] = 0.0;
}
__global__ void lanes_env_init_kernel(cplx* env_l, size_t l_stride, cplx* env_r_last, size_t r_stride, int lanes) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes) return;
    env_l[(size_t)l * l_stride] = make_double2(1.0, 0.0);
    env_r_last[(size_t)l * r_stride] = make_double2(1.0, 0.0);
}

hipError_t launch_lanes_gate1(const LaneMps& a, const LaneMps* b, const LaneOp1* ops, int nops, const LaneOp1& one, const double* thetas, int T, int lanes,
                              int bond_hint, hipStream_t s) {
    const int blocks = std::max(1, (bond_hint * bond_hint + 127) / 128);
    lanes_gate1_kernel<<<dim3(blocks, lanes * (b ? 2 : 1), ops ? nops : 1), 128, 0, s>>>(a, b ? *b : a, ops, one, thetas, T, lanes);
    return hipGetLastError();
}
hipError_t launch_lanes_gate2(const LaneMps& m, const LaneMps* m2, const LaneOp2* ops, int nops, const LaneOp2& one, const double* thetas, int T,
                              double trunc_thr, int max_bond, int* status, int* peak, int lanes, int bond_hint, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lanes_gate2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * kSmallMax * kSmallMax * (int)sizeof(cplx));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int h = std::min(kLaneCap, std::max(2, bond_hint));
    const unsigned lds_elems = 2u * (2 * h) * (2 * h);
    // A half-wave per column pair of the largest work matrix the launch is sized for while the lanes do not fill the chip (shortest
    // rounds); half of that once there are several workgroups per CU (typical matrices are well below the largest, idle waves only cost
    // barrier time and occupancy: 6.4 k -> 7.4 k evals/s at 1024 lanes of the 32-qubit workload, 7.6 k -> 9.6 k at 4096)
    const int groups = lanes * (m2 ? 2 : 1), gates = ops ? nops : 1;
    const int per_bond = groups * gates > 512 ? 16 : 32;
    const int threads = std::min(1024, std::max(64, (per_bond * h + 63) & ~63));
    lanes_gate2_kernel<<<dim3(groups, gates), threads, lds_elems * sizeof(cplx), s>>>(m, m2 ? *m2 : m, lanes, ops, one, thetas, T, trunc_thr, max_bond, 1e-15, 60,
                                                                                      status, peak, lds_elems);
    return hipGetLastError();
}
hipError_t launch_lanes_env_left(const LaneMps& w, const LaneMps& z, int p, const void* in, size_t in_stride, void* out, size_t out_stride,
                                 const double* gh8, int lanes, hipStream_t s) {
    Gate4c g;
    for (int i = 0; i < 4; ++i) g.m[i] = gh8 ? make_double2(gh8[2 * i], gh8[2 * i + 1]) : make_double2(0.0, 0.0);
    lanes_env_left_kernel<<<lanes, 256, sizeof(cplx) * kLaneEnv, s>>>(w, z, p, static_cast<const cplx*>(in), in_stride, static_cast<cplx*>(out), out_stride,
                                                                       gh8 ? 1 : 0, g);
    return hipGetLastError();
}
hipError_t launch_lanes_env_right(const LaneMps& w, const LaneMps& z, int p, const void* in, size_t in_stride, void* out, size_t out_stride, int lanes,
                                  hipStream_t s) {
    lanes_env_right_kernel<<<lanes, 256, sizeof(cplx) * kLaneEnv, s>>>(w, z, p, static_cast<const cplx*>(in), in_stride, static_cast<cplx*>(out), out_stride);
    return hipGetLastError();
}
hipError_t launch_lanes_env_dot(const LaneMps& w, const LaneMps& z, int hi, const void* e, size_t e_stride, const void* rc, size_t rc_stride, void* vals,
                                int nvals, int slot, int lanes, hipStream_t s) {
    lanes_env_dot_kernel<<<lanes, 256, 0, s>>>(w, z, hi, static_cast<const cplx*>(e), e_stride, static_cast<const cplx*>(rc), rc_stride,
                                               static_cast<cplx*>(vals), nvals, slot);
    return hipGetLastError();
}
hipError_t launch_lanes_grad_step(const LaneMps& w, const LaneMps& z, int q, const LaneGate1& g, const double* thetas, int T, const void* env_l, size_t l_stride,
                                  const void* env_r, size_t r_stride, const double* gh8, void* scratch, void* vals, int nvals, int slot, int lanes, hipStream_t s) {
    Gate4c gh;
    for (int i = 0; i < 4; ++i) gh.m[i] = make_double2(gh8[2 * i], gh8[2 * i + 1]);
    lanes_grad_step_kernel<<<lanes, 256, sizeof(cplx) * kLaneEnv, s>>>(w, z, q, g, thetas, T, static_cast<const cplx*>(env_l), l_stride,
                                                                        static_cast<const cplx*>(env_r), r_stride, gh, static_cast<cplx*>(scratch),
                                                                        static_cast<cplx*>(vals), nvals, slot);
    return hipGetLastError();
}
hipError_t launch_lanes_basis(const LaneMps& m, const unsigned char* bits, int lanes, hipStream_t s) {
    lanes_basis_kernel<<<lanes, 64, 0, s>>>(m, bits, lanes);
    return hipGetLastError();
}
hipError_t launch_lanes_env_init(void* env_l, size_t l_stride, void* env_r_last, size_t r_stride, int lanes, hipStream_t s) {
    lanes_env_init_kernel<<<(lanes + 127) / 128, 128, 0, s>>>(static_cast<cplx*>(env_l), l_stride, static_cast<cplx*>(env_r_last), r_stride, lanes);
    return hipGetLastError();
}

}  // namespace aqc

// Complex SVD by one-sided (Hestenes) Jacobi on the device, for the truncated 2-qubit gates of the MPS engine
// (the step qiskit-aer performs inside AerSimulator(method="matrix_product_state"), reached from
// mps_dot_objective.py:380-468 / mps_operations.py:252-257).  The work matrix W is column-major (rows x cols,
// every column contiguous); plane rotations make its columns mutually orthogonal and are accumulated in the
// column-major unitary V (cols x cols): on exit W = A V, sigma_j = |W_j|, so A = sum_j W_j V_j^H.
// One launch = one round of the round-robin tournament: cols/2 disjoint column pairs, one workgroup per pair.
// Everything is deterministic: fixed pair order, fixed-order block reductions, no float atomics.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "aqc_lanes.h"
#include "aqc_launch.h"
#include "aqc_math.h"
#include "aqc_mps_dev.h"

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

__global__ void mps_theta_fused_kernel(const cplx* __restrict__ tq, const cplx* __restrict__ tq1, const double* __restrict__ lam_left, int chil,
                                       int chim, int chir, Gate16 g, int mode, cplx* __restrict__ work) {
    mps_theta_fused_body(tq, tq1, lam_left, chil, chim, chir, g, mode, work, (int)(blockIdx.x * blockDim.x + threadIdx.x));
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
__global__ __launch_bounds__(256) void mps_env_left_kernel(const cplx* __restrict__ in, const cplx* __restrict__ A, const cplx* __restrict__ B,
                                                           int xa, int ua, int yb, int vb, int has_op, Gate4c gh, cplx* __restrict__ out) {
    mps_env_left_body(in, A, B, xa, ua, yb, vb, has_op, gh, out);
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

}  // namespace aqc

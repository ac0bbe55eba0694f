// MPS helpers on the GPU: Gamma*lambda folding, <mps1|mps2> by transfer matrices and
// MPS -> dense state.  Reference: mps_operations.py:126-213.  Both contractions are chains of
// small complex GEMMs; one LDS-tiled fp64 MFMA zgemm kernel serves them.
#include <hip/hip_runtime.h>

#include "aqc_launch.h"

namespace aqc {

typedef double2 cplx;

// gamma[b][l][r] *= lambda[r]   (mps_operations.py:146-149)
__global__ void mps_scale_kernel(cplx* g, const double* lam, int rows, int cols) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * cols) {
        const double s = lam[i % cols];
        g[i].x *= s;
        g[i].y *= s;
    }
}

// all sites of an MPS at once: gamma_q[b][l][r] *= lambda_q[r] for q < n - 1 (one launch instead of n - 1)
__global__ void mps_scale_all_kernel(cplx* t, const double* lam, MpsSites sites) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sites.total) return;
    int q = 0;
    while (q + 1 < sites.n && i >= sites.offset[q + 1]) ++q;
    if (q >= sites.n - 1) return;   // the last site has no Schmidt vector on its right
    const size_t local = i - sites.offset[q];
    const double sc = lam[sites.lam_offset[q] + (int)(local % (size_t)sites.cols[q])];
    t[i].x *= sc;
    t[i].y *= sc;
}
// C[M x N] (+)= op(A) * B, row-major.  CONJ_T: op(A)[m][k] = conj(A[k][m]) with A stored (K x M).
//
// fp64 matrix cores: one v_mfma_f64_16x16x4_f64 multiplies a 16x4 by a 4x16 real tile.  Operand layout
// (probed on gfx950, tools/ubench/mfma_f64_probe.hip): lane l supplies A[l%16][l/16] and B[l/16][l%16] and
// receives D[4r + l/16][l%16] in accumulator register r.  A complex product is four real ones; the planes
// are split (re / im) while the tiles are staged into LDS so that every operand read is one b64.
// Workgroup = 4 waves, 64 x 64 outputs; wave w owns rows [16w, 16w+16) and all four 16-column tiles.
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int TM = 64, TN = 64, TK = 16;
// Batching: product z = blockIdx.z uses operand pointers advanced by z * stride; with pointer tables (ZTables, device
// arrays) product z = outer * inner + i takes its bases from entry `outer` of each table and advances them by i * stride --
// the lanes of a batched MPS contraction live in unrelated allocations.
struct ZTables { const cplx* const* a; const cplx* const* b; cplx* const* c; int inner; };
template <bool CONJ_T, bool ACCUM>
__global__ __launch_bounds__(256) void zgemm_kernel(int M, int N, int K, const cplx* __restrict__ A, int lda,
                                                    const cplx* __restrict__ B, int ldb, cplx* __restrict__ C, int ldc,
                                                    size_t stride_a, size_t stride_b, size_t stride_c, int b_herm, ZTables tab) {
    if (tab.inner > 0) {
        const int outer = blockIdx.z / tab.inner, i = blockIdx.z % tab.inner;
        A = tab.a[outer] + (size_t)i * stride_a; B = tab.b[outer] + (size_t)i * stride_b; C = tab.c[outer] + (size_t)i * stride_c;
    } else {
        A += (size_t)blockIdx.z * stride_a; B += (size_t)blockIdx.z * stride_b; C += (size_t)blockIdx.z * stride_c;   // batched
    }
    __shared__ double sar[TK][TM + 4], sai[TK][TM + 4];
    __shared__ double sbr[TK][TN + 4], sbi[TK][TN + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    double4_t cre[4], cim[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { cre[t] = double4_t{0, 0, 0, 0}; cim[t] = double4_t{0, 0, 0, 0}; }
    for (int k0 = 0; k0 < K; k0 += TK) {
        for (int e = threadIdx.x; e < TK * TM; e += 256) {
            int kk, mm;
            if (CONJ_T) { kk = e / TM; mm = e % TM; } else { mm = e / TK; kk = e % TK; }
            const int gm = m0 + mm, gk = k0 + kk;
            cplx v = make_double2(0.0, 0.0);
            if (gm < M && gk < K) {
                if (CONJ_T) { v = A[(size_t)gk * lda + gm]; v.y = -v.y; } else { v = A[(size_t)gm * lda + gk]; }
            }
            sar[kk][mm] = v.x;
            sai[kk][mm] = v.y;
        }
        for (int e = threadIdx.x; e < TK * TN; e += 256) {
            const int kk = e / TN, nn = e % TN;
            const int gk = k0 + kk, gn = n0 + nn;
            cplx v = make_double2(0.0, 0.0);
            if (gk < K && gn < N) {
                // b_herm 1: B holds (N x K) and is used as B^H; 2: the same storage used as B^T (no conjugation)
                if (b_herm) { v = B[(size_t)gn * ldb + gk]; if (b_herm == 1) v.y = -v.y; } else { v = B[(size_t)gk * ldb + gn]; }
            }
            sbr[kk][nn] = v.x;
            sbi[kk][nn] = v.y;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < TK; ks += 4) {
            const double ar = sar[ks + lk][16 * wave + li];
            const double ai = sai[ks + lk][16 * wave + li];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double br = sbr[ks + lk][16 * t + li];
                const double bi = sbi[ks + lk][16 * t + li];
                cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cre[t], 0, 0, 0);
                cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, cim[t], 0, 0, 0);
                cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cre[t], 0, 0, 0);
                cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, cim[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = m0 + 16 * wave + 4 * r + lk, gn = n0 + 16 * t + li;
            if (gm < M && gn < N) {
                cplx* c = C + (size_t)gm * ldc + gn;
                if (ACCUM) { c->x += cre[t][r]; c->y += cim[t][r]; } else { *c = make_double2(cre[t][r], cim[t][r]); }
            }
        }
}

hipError_t launch_mps_scale(void* g, const double* lam, int rows, int cols, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    mps_scale_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(g), lam, rows, cols);
    return hipGetLastError();
}

hipError_t launch_mps_scale_all(void* t, const double* lam, const MpsSites& sites, hipStream_t s) {
    if (sites.total == 0) return hipSuccess;
    mps_scale_all_kernel<<<(unsigned)((sites.total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(t), lam, sites);
    return hipGetLastError();
}
namespace {
hipError_t zgemm_launch(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, size_t sa,
                        size_t sb, size_t sc, int nbatch, int b_herm, hipStream_t s, ZTables tab = ZTables{nullptr, nullptr, nullptr, 0}) {
    if (M <= 0 || N <= 0 || nbatch <= 0) return hipSuccess;
    const dim3 grid((N + TN - 1) / TN, (M + TM - 1) / TM, nbatch);
    const cplx* a = static_cast<const cplx*>(A);
    const cplx* b = static_cast<const cplx*>(B);
    cplx* c = static_cast<cplx*>(C);
    if (conj_t) {
        if (accum) zgemm_kernel<true, true><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc, sa, sb, sc, b_herm, tab);
        else zgemm_kernel<true, false><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc, sa, sb, sc, b_herm, tab);
    } else {
        if (accum) zgemm_kernel<false, true><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc, sa, sb, sc, b_herm, tab);
        else zgemm_kernel<false, false><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc, sa, sb, sc, b_herm, tab);
    }
    return hipGetLastError();
}
}  // namespace

hipError_t launch_zgemm(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                        void* C, int ldc, hipStream_t s) {
    return zgemm_launch(conj_t, accum, M, N, K, A, lda, B, ldb, C, ldc, 0, 0, 0, 1, 0, s);
}
// C (M x N) = op(A) . B^H with B stored (N x K) row-major
hipError_t launch_zgemm_bh(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                           void* C, int ldc, hipStream_t s) {
    return zgemm_launch(conj_t, accum, M, N, K, A, lda, B, ldb, C, ldc, 0, 0, 0, 1, 1, s);
}

// `nbatch` products in one launch: operand / result pointers advance by the strides (in elements) per product
hipError_t launch_zgemm_batched(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                void* C, int ldc, size_t sa, size_t sb, size_t sc, int nbatch, hipStream_t s) {
    return zgemm_launch(conj_t, accum, M, N, K, A, lda, B, ldb, C, ldc, sa, sb, sc, nbatch, 0, s);
}

// `outer` x `inner` products in one launch: bases from the device pointer tables, advanced by the strides per inner index
hipError_t launch_zgemm_tables(int M, int N, int K, const void* const* a_tab, int lda, const void* const* b_tab, int ldb, void* const* c_tab, int ldc,
                               size_t sa, size_t sb, size_t sc, int outer, int inner, hipStream_t s, int b_transposed) {
    const ZTables tab{reinterpret_cast<const cplx* const*>(a_tab), reinterpret_cast<const cplx* const*>(b_tab), reinterpret_cast<cplx* const*>(c_tab), inner};
    return zgemm_launch(false, false, M, N, K, nullptr, lda, nullptr, ldb, nullptr, ldc, sa, sb, sc, outer * inner, b_transposed ? 2 : 0, s, tab);
}

// product states (every bond dimension 1): out[i] = prod_q t_q[bit q of i], t_q = the two numbers of site q (2 q, 2 q + 1 of the
// lane's packed tensors); one launch per batch instead of the whole contraction chain
__global__ void mps_product_kernel(const cplx* const* t_tab, cplx* const* out_tab, int n) {
    const cplx* t = t_tab[blockIdx.y];
    cplx* out = out_tab[blockIdx.y];
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ((size_t)1 << n)) return;
    cplx v = t[e & 1];
    for (int q = 1; q < n; ++q) {
        const cplx f = t[2 * q + ((e >> q) & 1)];
        v = make_double2(v.x * f.x - v.y * f.y, v.x * f.y + v.y * f.x);
    }
    out[e] = v;
}
hipError_t launch_mps_product(const void* const* t_tab, void* const* out_tab, int n, int count, hipStream_t s) {
    const size_t total = (size_t)1 << n;
    mps_product_kernel<<<dim3((unsigned)((total + 255) / 256), count), 256, 0, s>>>(reinterpret_cast<const cplx* const*>(t_tab),
                                                                                    reinterpret_cast<cplx* const*>(out_tab), n);
    return hipGetLastError();
}

// out = sum_i e[i] conj(rc[i]): closes an inner product between a left environment and a (conjugated) right one;
// one workgroup, fixed-order reduction
__global__ __launch_bounds__(256) void mps_env_dot_kernel(const cplx* __restrict__ e, const cplx* __restrict__ rc, size_t count, cplx* __restrict__ out) {
    __shared__ double sr[256], si[256];
    double re = 0.0, im = 0.0;
    for (size_t i = threadIdx.x; i < count; i += 256) {
        const cplx a = e[i], b = rc[i];
        re += a.x * b.x + a.y * b.y;
        im += a.y * b.x - a.x * b.y;
    }
    sr[threadIdx.x] = re; si[threadIdx.x] = im;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sr[threadIdx.x] += sr[threadIdx.x + s]; si[threadIdx.x] += si[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = make_double2(sr[0], si[0]);
}
hipError_t launch_mps_env_dot(const void* e, const void* rc, size_t count, void* out, hipStream_t s) {
    mps_env_dot_kernel<<<1, 256, 0, s>>>(static_cast<const cplx*>(e), static_cast<const cplx*>(rc), count, static_cast<cplx*>(out));
    return hipGetLastError();
}

}  // namespace aqc

// MPS helpers on the GPU: Gamma*lambda folding, <mps1|mps2> by transfer matrices and
// MPS -> dense state.  Reference: mps_operations.py:126-213.  Both contractions are chains of
// small complex GEMMs; one LDS-tiled fp64 zgemm kernel serves them.
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

// C[M x N] (+)= op(A) * B, row-major.  CONJ_T: op(A)[m][k] = conj(A[k][m]) with A stored (K x M).
constexpr int TM = 64, TN = 64, TK = 8;
template <bool CONJ_T, bool ACCUM>
__global__ __launch_bounds__(256) void zgemm_kernel(int M, int N, int K, const cplx* __restrict__ A, int lda,
                                                    const cplx* __restrict__ B, int ldb, cplx* __restrict__ C, int ldc) {
    __shared__ cplx sa[TK][TM + 1];
    __shared__ cplx sb[TK][TN + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    cplx acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = make_double2(0.0, 0.0);
    for (int k0 = 0; k0 < K; k0 += TK) {
        for (int e = threadIdx.x; e < TK * TM; e += 256) {
            int kk, mm;
            if (CONJ_T) { kk = e / TM; mm = e % TM; } else { mm = e / TK; kk = e % TK; }
            const int gm = m0 + mm, gk = k0 + kk;
            cplx v = make_double2(0.0, 0.0);
            if (gm < M && gk < K) {
                if (CONJ_T) { v = A[(size_t)gk * lda + gm]; v.y = -v.y; } else { v = A[(size_t)gm * lda + gk]; }
            }
            sa[kk][mm] = v;
        }
        for (int e = threadIdx.x; e < TK * TN; e += 256) {
            const int kk = e / TN, nn = e % TN;
            const int gk = k0 + kk, gn = n0 + nn;
            sb[kk][nn] = (gk < K && gn < N) ? B[(size_t)gk * ldb + gn] : make_double2(0.0, 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            cplx a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sa[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sb[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j].x += a[i].x * b[j].x - a[i].y * b[j].y;
                    acc[i][j].y += a[i].x * b[j].y + a[i].y * b[j].x;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gm = m0 + ty + 16 * i, gn = n0 + tx + 16 * j;
            if (gm < M && gn < N) {
                cplx* c = C + (size_t)gm * ldc + gn;
                if (ACCUM) { c->x += acc[i][j].x; c->y += acc[i][j].y; } else { *c = acc[i][j]; }
            }
        }
}

hipError_t launch_mps_scale(void* g, const double* lam, int rows, int cols, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    mps_scale_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(static_cast<cplx*>(g), lam, rows, cols);
    return hipGetLastError();
}

hipError_t launch_zgemm(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                        void* C, int ldc, hipStream_t s) {
    const dim3 grid((N + TN - 1) / TN, (M + TM - 1) / TM);
    const cplx* a = static_cast<const cplx*>(A);
    const cplx* b = static_cast<const cplx*>(B);
    cplx* c = static_cast<cplx*>(C);
    if (conj_t) {
        if (accum) zgemm_kernel<true, true><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc);
        else zgemm_kernel<true, false><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc);
    } else {
        if (accum) zgemm_kernel<false, true><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc);
        else zgemm_kernel<false, false><<<grid, 256, 0, s>>>(M, N, K, a, lda, b, ldb, c, ldc);
    }
    return hipGetLastError();
}

}  // namespace aqc

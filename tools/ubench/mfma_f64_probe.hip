// Probes the operand layout of v_mfma_f64_16x16x4_f64 on gfx950: D = A(16x4) B(4x16) with one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {  // A[16][4], B[4][16] row-major, D[16][16]
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // hypothesis: lane holds A[i = l%16][k = l/16]
    const double b = B[(l / 16) * 16 + l % 16];  //             lane holds B[k = l/16][j = l%16]
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];  // raw dump: D[lane][r]
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + 0.37 * i + (i % 5) * 0.11; hB[i] = 2 - 0.21 * i + (i % 7) * 0.13; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // find, for every (lane, r), the (i, j) of the reference product it equals
    for (int l = 0; l < 64; l += 1) for (int r = 0; r < 4; ++r) {
        int fi = -1, fj = -1, cnt = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i * 16 + j] - hD[l * 4 + r]) < 1e-9) { fi = i; fj = j; ++cnt; }
        if (l < 20 || l % 16 == 0) printf("lane %2d r %d -> D[%d][%d] (%d matches)\n", l, r, fi, fj, cnt);
    }
    return 0;
}

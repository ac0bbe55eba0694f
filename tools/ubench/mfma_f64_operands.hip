// v_mfma_f64_16x16x4_f64 issue rate as a function of how the A / B operand registers change between
// consecutive instructions (one wave per SIMD unless argv[1] = 2).  6 independent accumulators in every variant.
//   V0 same (a, b) every time          V1 alternate (a, b) / (b, a)
//   V2 six distinct (a_i, b_i) pairs   V3 A distinct, B shared    V4 A shared, B distinct
//   V5 pairs of instructions share both operands (a0,b0),(a0,b0),(a1,b1),(a1,b1),...
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define M(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512) void rate(double* out, const double* in, int iters) {
    const int l = threadIdx.x;
    double a[6], b[6];
    for (int i = 0; i < 6; ++i) { a[i] = in[l + 64 * i]; b[i] = in[l + 64 * (i + 6)]; }
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { M(a[0], b[0], c0); M(a[0], b[0], c1); M(a[0], b[0], c2); M(a[0], b[0], c3); M(a[0], b[0], c4); M(a[0], b[0], c5); }
        if (V == 1) { M(a[0], b[0], c0); M(b[0], a[0], c1); M(a[0], b[0], c2); M(b[0], a[0], c3); M(a[0], b[0], c4); M(b[0], a[0], c5); }
        if (V == 2) { M(a[0], b[0], c0); M(a[1], b[1], c1); M(a[2], b[2], c2); M(a[3], b[3], c3); M(a[4], b[4], c4); M(a[5], b[5], c5); }
        if (V == 3) { M(a[0], b[0], c0); M(a[1], b[0], c1); M(a[2], b[0], c2); M(a[3], b[0], c3); M(a[4], b[0], c4); M(a[5], b[0], c5); }
        if (V == 4) { M(a[0], b[0], c0); M(a[0], b[1], c1); M(a[0], b[2], c2); M(a[0], b[3], c3); M(a[0], b[4], c4); M(a[0], b[5], c5); }
        if (V == 5) { M(a[0], b[0], c0); M(a[0], b[0], c1); M(a[1], b[1], c2); M(a[1], b[1], c3); M(a[2], b[2], c4); M(a[2], b[2], c5); }
    }
    double s = 0;
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r];
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int threads = 256 * wps, blocks = 256 * 4, iters = 4000;
    double *d, *din;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipMalloc(&din, sizeof(double) * 1024);
    double h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0.5 + 0.001 * ((i * 7919) % 1000);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 6; ++v) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            switch (v) {
                case 0: rate<0><<<blocks, threads>>>(d, din, iters); break;
                case 1: rate<1><<<blocks, threads>>>(d, din, iters); break;
                case 2: rate<2><<<blocks, threads>>>(d, din, iters); break;
                case 3: rate<3><<<blocks, threads>>>(d, din, iters); break;
                case 4: rate<4><<<blocks, threads>>>(d, din, iters); break;
                default: rate<5><<<blocks, threads>>>(d, din, iters); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        const double mfmas = (double)blocks * (threads / 64) * iters * 6;
        printf("waves/SIMD %d V%d: %.3f ms  %.2f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", wps, v, best, mfmas * 2048.0 / best / 1e9,
               best * 1e-3 * 2.4e9 / (mfmas / 1024.0));
    }
    return 0;
}

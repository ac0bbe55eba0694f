// Issue-rate probes for v_mfma_f64_16x16x4_f64 on gfx950 (how the sweep's MFMA formulation can be fed):
//   mode 0: NACC independent accumulators, back to back (peak issue rate)
//   mode 1: one dependent accumulation chain
//   mode 2: 3 chains + one ds_read_b128 and one v_add_f64 per MFMA (the kernel's steady state mix)
// Usage: mfma_f64_rate [waves_per_simd=1|2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void rate(double* out, int iters) {
    __shared__ double2 lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += blockDim.x) lds[i] = make_double2(1e-3 * i, 1.0);
    __syncthreads();
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double2 v0 = lds[(l + 64 * (it & 31) + 7 * j) & 4095];
                const double2 v1 = lds[(l + 64 * (it & 31) + 7 * j + 2048) & 4095];
                const double2 v2 = lds[(l * 3 + 64 * (it & 31) + 7 * j + 1024) & 4095];
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0.x, b, c0, 0, 0, 0);
                acc += v0.y;
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1.x, b, c1, 0, 0, 0);
                acc += v1.y;
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(v2.x, b, c2, 0, 0, 0);
                acc += v2.y;
            }
        }
    }
    double s = acc;
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r];
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int threads = 256 * wps, blocks = 256 * 4, iters = 4000;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<blocks, threads>>>(d, iters);
            else if (mode == 1) rate<1><<<blocks, threads>>>(d, iters);
            else rate<2><<<blocks, threads>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double mfmas = (double)blocks * (threads / 64) * iters * 6;
        const double flops = mfmas * 2048.0;
        // cycles per MFMA per SIMD at 2.4 GHz nominal: each SIMD runs blocks*waves/1024 waves in sequence
        const double per_simd = mfmas / 1024.0;
        printf("waves/SIMD %d mode %d: %.3f ms  %.2f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", wps, mode, best, flops / best / 1e9,
               best * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}

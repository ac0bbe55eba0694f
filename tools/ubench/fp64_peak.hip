// Microbenchmark: sustained v_fma_f64 rate and LDS b128 read/write rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void fma_kernel(double* out, int iters, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
            x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void lds_kernel(double* out, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int n = 4096;
    for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = make_double2(i, -i);
    __syncthreads();
    double2 acc = make_double2(0, 0);
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (mode == 0) { double2 v = sm[i]; acc.x += v.x; acc.y += v.y; }
            else { sm[i] = make_double2(acc.x + it, acc.y + i); }
        }
        __syncthreads();
    }
    if (mode == 1) acc = sm[threadIdx.x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}

int main() {
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 1024}) {
        for (int wgs_per_cu : {1, 2}) {
            const int blocks = 256 * wgs_per_cu, iters = 20000;
            fma_kernel<<<blocks, threads>>>(d, 100, 1.0000001, 1e-9);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            fma_kernel<<<blocks, threads>>>(d, iters, 1.0000001, 1e-9);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 64.0 * iters * (double)blocks * threads;
            printf("fma_f64: %4d thr x %d WG/CU: %.3f ms  %.1f TFLOP/s  (%.2f clk@2.4GHz per wave-instr per SIMD)\n", threads, wgs_per_cu, ms,
                   flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (64.0 * iters * (threads / 64) * wgs_per_cu / 4.0));
        }
    }
    for (int mode : {0, 1}) {
        for (int threads : {256, 512}) {
            const int blocks = 256, iters = 2000;
            lds_kernel<<<blocks, threads, 65536>>>(d, 10, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            lds_kernel<<<blocks, threads, 65536>>>(d, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 16.0 * 4096 * iters;  // per CU
            printf("lds %s b128: %4d thr: %.3f ms  %.1f B/clk/CU @2.4GHz\n", mode ? "write" : "read ", threads, ms, bytes / (ms * 1e-3 * 2.4e9));
        }
    }
    return 0;
}

// Does the sustained v_mfma_f64_16x16x4_f64 rate depend on the operand data (DVFS under load)?
//   mode 0: constant operands, mode 1: pseudo-random operands refreshed every 6 MFMAs (cheap integer VALU).
// Reports TFLOP/s and the in-kernel clock (s_memtime ticks per s_memrealtime 100 MHz tick), median over workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(double* out, unsigned long long* clk, int iters) {
    const int l = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (l + 1 + 1024ull * blockIdx.x);
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {   // mantissa bits from a xorshift; exponent fixed => values in [1, 2)
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            a = __longlong_as_double((long long)((x >> 12) | 0x3FF0000000000000ull));
            b = __longlong_as_double((long long)(((x * 0x2545F4914F6CDD1Dull) >> 12) | 0x3FF0000000000000ull)) - 1.5;
            a -= 1.5;
        }
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c5, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r];
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
    if (l == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int threads = 256, blocks = 256 * (argc > 1 ? atoi(argv[1]) : 8), iters = argc > 2 ? atoi(argv[2]) : 20000;
    double* d; unsigned long long* dc;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipMalloc(&dc, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<blocks, threads>>>(d, dc, iters); else rate<1><<<blocks, threads>>>(d, dc, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), dc, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        std::vector<double> ghz;
        for (int i = 0; i < blocks; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        const double flops = (double)blocks * (threads / 64) * iters * 6 * 2048.0;
        printf("blocks/CU %d iters %d mode %d (%s operands): %.3f ms  %.2f TFLOP/s  in-kernel clock %.3f GHz (median)\n", blocks / 256, iters, mode, mode ? "random" : "constant", best,
               flops / best / 1e9, ghz[blocks / 2]);
    }
    return 0;
}

// How many independent vector instructions fit in the shadow of one v_mfma_f64_16x16x4_f64 issued by the SAME wave?
// Loop body: 6 MFMAs (independent accumulators), each followed by N v_add_f64 (FP64) or N v_xor_b32 (INT) or N ds_read_b128 (LDS).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int N, int KIND>
__global__ __launch_bounds__(256) void shadow(double* out, int iters) {
    __shared__ double2 lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 256) lds[i] = make_double2(i, 1.0);
    __syncthreads();
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double4_t c[6];
    for (int i = 0; i < 6; ++i) c[i] = double4_t{0, 0, 0, 0};
    double f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    unsigned x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    d2_t ld[4];
    const unsigned addr = (unsigned)(size_t)(&lds[l & 255]);
    const unsigned addr2 = (unsigned)(size_t)(&lds[((l & 15) * 16 + ((l >> 4) & 15)) & 255]);   // transposed: lanes 0..15 are 256 B apart
    ld[0] = ld[1] = ld[2] = ld[3] = d2_t{1.0, 2.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[k & 7]) : "v"(a));
                else if (KIND == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(l));
                else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[k & 3]) : "v"(addr));
                else if (KIND == 3) asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(ld[k & 3]) : "memory");
                else if (KIND == 4) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(f[k & 7]) : "memory");
                else asm volatile("ds_write_b128 %0, %1" : : "v"(addr2), "v"(ld[k & 3]) : "memory");   // 16-lane stride pattern
            }
        }
        if (KIND >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = 0;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
    for (int k = 0; k < 8; ++k) s += f[k] + x[k];
    if (KIND == 2) for (int k = 0; k < 4; ++k) s += ld[k].x;
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
}

static int g_blocks_per_cu = 4;
template <int N, int KIND>
void run(double* d, const char* name) {
    const int blocks = 256 * g_blocks_per_cu, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        shadow<N, KIND><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double per_simd = (double)blocks * 4 * iters * 6 / 1024.0;
    printf("waves/SIMD %d: %s x %2d per MFMA: %.1f cycles per MFMA per SIMD (2.4 GHz)\n", g_blocks_per_cu, name, N, best * 1e-3 * 2.4e9 / per_simd);
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks_per_cu = atoi(argv[1]);
    double* d;
    hipMalloc(&d, sizeof(double) * 2048 * 256);
    run<0, 0>(d, "v_add_f64   "); run<2, 0>(d, "v_add_f64   "); run<4, 0>(d, "v_add_f64   "); run<8, 0>(d, "v_add_f64   "); run<12, 0>(d, "v_add_f64   ");
    run<16, 0>(d, "v_add_f64   ");
    run<4, 1>(d, "v_xor_b32   "); run<8, 1>(d, "v_xor_b32   "); run<16, 1>(d, "v_xor_b32   "); run<32, 1>(d, "v_xor_b32   ");
    run<1, 2>(d, "ds_read_b128"); run<2, 2>(d, "ds_read_b128"); run<4, 2>(d, "ds_read_b128");
    run<1, 3>(d, "ds_write_b128"); run<2, 3>(d, "ds_write_b128"); run<4, 3>(d, "ds_write_b128");
    run<2, 4>(d, "ds_write_b64 "); run<4, 4>(d, "ds_write_b64 "); run<8, 4>(d, "ds_write_b64 ");
    run<1, 5>(d, "ds_write_b128 transposed"); run<2, 5>(d, "ds_write_b128 transposed");
    return 0;
}

// What do the fp64 adds of the 3-multiplication complex product cost next to the matrix work?  One iteration = one "group" of the V / V^H
// stage kernel: 12 v_mfma_f64_16x16x4_f64 (three accumulator chains k1, k2, k3 over four K-steps) and 12 v_add_f64 (4 operand sums, 8
// that combine the accumulators: re = k1 - k3, im = k1 + k2).  Patterns:
//   0  adds independent of the accumulators (control: what tools/ubench/mfma_f64_shadow.hip measured)
//   1  the kernel's order: [12 MFMA] [4 sums, 8 combining adds reading the accumulators just written] -- one accumulator set
//   2  two accumulator sets: the combining adds of run j - 1 sit in the MIDDLE of run j (after its 4th MFMA), as one bunch
//   3  two accumulator sets: the combining adds of run j - 1 directly BEFORE run j (but run j - 1 was issued a whole run earlier)
//   4  pattern 1 with 12 independent v_xor_b32 between the run and the dependent adds (does time alone cure it?)
//   7  as 2, after the 3rd MFMA
//   5  no adds at all (12 MFMAs only)
//   6  16 MFMAs, no adds (the 4-multiplication form)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef double double4_t __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA0(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
#define ADD(d, x, y) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define SUB(d, x, y) asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(x), "v"(y))

template <int P>
__global__ __launch_bounds__(256) void combine(double* out, int iters) {
    const int l = threadIdx.x;
    double u0[4], u1[4], u2[4], vx[4], vy[4], sv[4], f[8];
    for (int s = 0; s < 4; ++s) { u0[s] = 1e-3 * (l + s); u1[s] = 2e-3 * (l - s); u2[s] = 1e-3 * s; vx[s] = 1.0 + 1e-9 * l; vy[s] = 1.0 - 1e-9 * (l + s); sv[s] = 2.0; }
    for (int k = 0; k < 8; ++k) f[k] = k;
    unsigned x[12];
    for (int k = 0; k < 12; ++k) x[k] = k;
    double4_t k1[2], k2[2], k3[2], k4[2];
    for (int b = 0; b < 2; ++b) { k1[b] = double4_t{0, 0, 0, 0}; k2[b] = k1[b]; k3[b] = k1[b]; k4[b] = k1[b]; }
    double ox[4] = {0, 0, 0, 0}, oy[4] = {0, 0, 0, 0};
    auto run = [&](auto B_, auto PB_, int it) __attribute__((always_inline)) {   // accumulator sets by compile-time index
        constexpr int b = decltype(B_)::value, pb = decltype(PB_)::value;
        if (P == 3 && it > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { SUB(ox[r], k1[pb][r], k3[pb][r]); ADD(oy[r], k1[pb][r], k2[pb][r]); }
        }
        if (P != 5 && P != 6) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ADD(sv[s], vx[s], vy[s]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s == 0) { MFMA0(k1[b], sv[s], u0[s]); MFMA0(k2[b], vx[s], u1[s]); MFMA0(k3[b], vy[s], u2[s]); if (P == 6) MFMA0(k4[b], vy[s], u0[s]); }
            else { MFMA(k1[b], sv[s], u0[s]); MFMA(k2[b], vx[s], u1[s]); MFMA(k3[b], vy[s], u2[s]); if (P == 6) MFMA(k4[b], vy[s], u0[s]); }
            if ((P == 2 && s == 1 && it > 0) || (P == 7 && s == 0 && it > 0)) {   // after the 6th (P 2) / 3rd (P 7) MFMA of this run: combine the PREVIOUS run
#pragma unroll
                for (int r = 0; r < 4; ++r) { SUB(ox[r], k1[pb][r], k3[pb][r]); ADD(oy[r], k1[pb][r], k2[pb][r]); }
            }
        }
        if (P == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ADD(f[k], f[k], vx[k & 3]);
        }
        if (P == 4) {
#pragma unroll
            for (int k = 0; k < 12; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[k]) : "v"(l));
        }
        if (P == 1 || P == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { SUB(ox[r], k1[b][r], k3[b][r]); ADD(oy[r], k1[b][r], k2[b][r]); }
        }
        // (the results feed the next run's operands, as the tile does through LDS in the kernel -- without extra instructions)
        if (P == 1 || P == 2 || P == 3 || P == 4 || P == 7) { vx[0] = ox[0]; vy[1] = oy[0]; vx[2] = ox[2]; vy[3] = oy[3]; }
    };
    for (int it = 0; it < iters; it += 2) {
        if (P == 2 || P == 3 || P == 7) {
            run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, it);
            run(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, it + 1);
        } else {
            run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, it);
            run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, it + 1);
        }
    }
    double s = 0;
    for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) s += k1[b][r] + k2[b][r] + k3[b][r] + k4[b][r];
    for (int r = 0; r < 4; ++r) s += ox[r] + oy[r] + sv[r];
    for (int k = 0; k < 8; ++k) s += f[k];
    for (int k = 0; k < 12; ++k) s += x[k];
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
}

static int g_blocks_per_cu = 4;
template <int P>
void run(double* d, const char* name) {
    const int blocks = 256 * g_blocks_per_cu, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        combine<P><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double per_simd = (double)blocks * 4 * iters / 1024.0;   // groups per SIMD
    printf("waves/SIMD %d: pattern %d %-72s %7.1f cycles per group per SIMD (2.4 GHz)\n", g_blocks_per_cu, P, name, best * 1e-3 * 2.4e9 / per_simd);
}


// Patterns 8 / 9 / 10: the loop of the V / V^H stage kernel instruction for instruction -- LDS reads of the next group issued, 4 operand sums,
// 8 combining adds on the accumulators of the run that has just been issued, then the next run of 12 MFMAs with the 4 LDS writes of the
// combined results after its first four MFMAs.  9: without the LDS writes; 10: without the LDS reads; 11: combining adds replaced by
// nothing (LDS writes read the accumulators); 12: as 8 with the combining adds AFTER the next run's first 4 MFMAs (second accumulator set)
typedef double d2_t __attribute__((ext_vector_type(2)));
template <int P>
__global__ __launch_bounds__(256) void kernel_like(double* out, int iters) {
    __shared__ double2 lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 256) lds[i] = make_double2(1.0 + 1e-6 * i, 1.0);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(&lds[l]);
    double u0[4], u1[4], u2[4];
    for (int s = 0; s < 4; ++s) { u0[s] = 1e-3 * (l + s); u1[s] = 2e-3 * (l - s); u2[s] = 1e-3 * s; }
    d2_t v[2][4];
    for (int b = 0; b < 2; ++b) for (int s = 0; s < 4; ++s) v[b][s] = d2_t{1.0, 2.0};
    double4_t k1[2], k2[2], k3[2];
    for (int b = 0; b < 2; ++b) { k1[b] = double4_t{0, 0, 0, 0}; k2[b] = k1[b]; k3[b] = k1[b]; }
    double sv[4] = {0, 0, 0, 0};
    d2_t o[4] = {d2_t{0, 0}, d2_t{0, 0}, d2_t{0, 0}, d2_t{0, 0}};
    auto body = [&](auto B_, auto PB_, auto NB_) __attribute__((always_inline)) {
        constexpr int b = decltype(B_)::value, pb = decltype(PB_)::value, nb = decltype(NB_)::value;   // accumulator set of this run / the previous run, operand buffer
        if (P != 10) {
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[nb ^ 1][s]) : "v"(base), "n"(4096 * 0) : "memory");
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) ADD(sv[s], v[nb][s].x, v[nb][s].y);
        if (P != 11 && P != 12) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { SUB(o[r].x, k1[pb][r], k3[pb][r]); ADD(o[r].y, k1[pb][r], k2[pb][r]); }
        }
        if (P == 11) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r].x = k1[pb][r]; o[r].y = k2[pb][r]; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s == 0) { MFMA0(k1[b], sv[s], u0[s]); } else { MFMA(k1[b], sv[s], u0[s]); }
            if (P != 9 && P != 12) asm volatile("ds_write_b128 %0, %1" : : "v"(base), "v"(o[s]) : "memory");
            if (s == 0) { MFMA0(k2[b], v[nb][s].x, u1[s]); MFMA0(k3[b], v[nb][s].y, u2[s]); } else { MFMA(k2[b], v[nb][s].x, u1[s]); MFMA(k3[b], v[nb][s].y, u2[s]); }
            if (P == 12 && s == 0) {   // the previous run's accumulators (other set) are combined here, 3 MFMAs into this run
#pragma unroll
                for (int r = 0; r < 4; ++r) { SUB(o[r].x, k1[pb][r], k3[pb][r]); ADD(o[r].y, k1[pb][r], k2[pb][r]); }
            }
            if (P == 12 && s >= 1 && s <= 3) asm volatile("ds_write_b128 %0, %1" : : "v"(base), "v"(o[s]) : "memory");
        }
        if (P == 12) asm volatile("ds_write_b128 %0, %1" : : "v"(base), "v"(o[0]) : "memory");
        if (P != 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int it = 0; it < iters; it += 2) {
        if (P == 12) {
            body(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            body(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        } else {
            body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        }
    }
    double s = 0;
    for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) s += k1[b][r] + k2[b][r] + k3[b][r];
    for (int r = 0; r < 4; ++r) s += o[r].x + o[r].y + sv[r];
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
}
template <int P>
void run_like(double* d, const char* name) {
    const int blocks = 256 * g_blocks_per_cu, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kernel_like<P><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("waves/SIMD %d: pattern %2d %-72s %7.1f cycles per group per SIMD (2.4 GHz)\n", g_blocks_per_cu, P, name, best * 1e-3 * 2.4e9 / per_simd);
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks_per_cu = atoi(argv[1]);
    double* d;
    if (hipMalloc(&d, sizeof(double) * 2048 * 256) != hipSuccess) return 1;
    run<5>(d, "12 MFMAs, no adds");
    run<0>(d, "12 MFMAs + 12 adds independent of the accumulators");
    run<1>(d, "12 MFMAs + 4 sums + 8 adds on the accumulators just written (kernel)");
    run<4>(d, "... with 12 independent v_xor_b32 before the dependent adds");
    run<3>(d, "two accumulator sets, combining adds of run j-1 right before run j");
    run<2>(d, "two accumulator sets, combining adds of run j-1 after the 6th MFMA of run j");
    run<7>(d, "two accumulator sets, combining adds of run j-1 after the 3rd MFMA of run j");
    run<6>(d, "16 MFMAs, no adds (4-multiplication form)");
    run_like<8>(d, "kernel loop: LDS reads, 4 sums, 8 combining adds, 12 MFMAs + 4 LDS writes");
    run_like<9>(d, "... without the LDS writes");
    run_like<10>(d, "... without the LDS reads");
    run_like<11>(d, "... without the combining adds (LDS writes read the accumulators)");
    run_like<12>(d, "... combining adds moved 3 MFMAs into the next run (second accumulator set)");
    return 0;
}

// Host-side helpers shared by the single-lane MPS engine (aqc_mps_engine.cpp) and the lockstep lanes (aqc_mps_batch.cpp): error
// reporting, 2 x 2 gate algebra, entangler matrices, the block list of an ansatz.  Private to csrc/ (everything has internal linkage).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/aqc_hip.h"
#include "aqc_launch.h"

namespace aqc {
// device pointers of site q of a single-lane MPS (T_q, and the Schmidt vector of bond q when q < n - 1) after its stream has drained:
// how the lockstep lanes take a copy of a state built by the single-lane engine (defined in aqc_mps_engine.cpp)
int mps_peek(const aqc_mps* m, int q, const void** site, const double** lam);
// the reverse: a new single-lane MPS from tensors that live on the device (site q: [2][dims[q]][dims[q+1]] complex, T_q = Gamma_q lambda_q;
// lams[q]: the dims[q+1] Schmidt values of bond q), copied
int mps_adopt(int device, int n, const int* dims, const void* const* sites, const double* const* lams, double discarded, aqc_mps** out);
}  // namespace aqc

namespace {

[[maybe_unused]] int failf(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return aqc::set_error(buf);
}

#define HIP_OK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return failf("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)


[[maybe_unused]] void permute_gate(const double* g, bool flip, double* out) {   // flip: swap the roles of the two qubits (index 2a+b -> 2b+a)
    static const int p[4] = {0, 2, 1, 3};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const int si = flip ? p[i] : i, sj = flip ? p[j] : j;
            out[2 * (4 * i + j)] = g[2 * (4 * si + sj)];
            out[2 * (4 * i + j) + 1] = g[2 * (4 * si + sj) + 1];
        }
}



typedef std::complex<double> cd;
struct M2 { cd m[4]; };   // row-major 2x2
[[maybe_unused]] M2 operator*(const M2& x, const M2& y) {
    return {{x.m[0] * y.m[0] + x.m[1] * y.m[2], x.m[0] * y.m[1] + x.m[1] * y.m[3], x.m[2] * y.m[0] + x.m[3] * y.m[2], x.m[2] * y.m[1] + x.m[3] * y.m[3]}};
}
[[maybe_unused]] M2 rz_m(double t) { return {{std::polar(1.0, -0.5 * t), 0.0, 0.0, std::polar(1.0, 0.5 * t)}}; }
[[maybe_unused]] M2 ry_m(double t) { const double c = std::cos(0.5 * t), s = std::sin(0.5 * t); return {{c, -s, s, c}}; }
[[maybe_unused]] M2 rx_m(double t) { const double c = std::cos(0.5 * t), s = std::sin(0.5 * t); return {{c, cd(0, -s), cd(0, -s), c}}; }
[[maybe_unused]] const M2 kPauliX = {{0.0, 1.0, 1.0, 0.0}}, kPauliY = {{0.0, cd(0, -1), cd(0, 1), 0.0}}, kPauliZ = {{1.0, 0.0, 0.0, -1.0}}, kProj1 = {{0.0, 0.0, 0.0, 1.0}};
[[maybe_unused]] void pack(const M2& g, double* out8) { for (int i = 0; i < 4; ++i) { out8[2 * i] = g.m[i].real(); out8[2 * i + 1] = g.m[i].imag(); } }

[[maybe_unused]] void entangler_matrix(int ent, double angle, double* out32) {   // |0><0| x I + |1><1| x {X, Z, diag(1, e^{i angle})}, index 2 c + t
    std::fill(out32, out32 + 32, 0.0);
    out32[0] = 1.0; out32[2 * 5] = 1.0;
    if (ent == AQC_CX) { out32[2 * 11] = 1.0; out32[2 * 14] = 1.0; }
    else if (ent == AQC_CZ) { out32[2 * 10] = 1.0; out32[2 * 15] = -1.0; }
    else { out32[2 * 10] = 1.0; out32[2 * 15] = std::cos(angle); out32[2 * 15 + 1] = std::sin(angle); }
}

struct BlockRef { int i, j, c, t; };   // running index, parameter block, control, target
[[maybe_unused]] int check_circuit(const aqc_circuit* c, int n) {
    if (!c || !c->blocks) return failf("null circuit description");
    if (c->num_qubits != n) return failf("circuit and MPS differ in the number of qubits");
    if (c->entangler != AQC_CX && c->entangler != AQC_CZ && c->entangler != AQC_CP) return failf("unknown entangler");
    if (c->num_blocks < 0) return failf("negative number of blocks");
    for (int b = 0; b < c->num_blocks; ++b) {
        const int ct = c->blocks[b], tg = c->blocks[c->num_blocks + b];
        if (ct < 0 || ct >= n || tg < 0 || tg >= n || ct == tg) return failf("block %d couples invalid qubits", b);
    }
    return 0;
}
// incl. the virtual trailing half-layer of a 2nd-order Trotter ansatz (parametric_circuit.py:328-333)
[[maybe_unused]] std::vector<BlockRef> blocks_of(const aqc_circuit* c) {
    const int L = c->num_blocks, tail = (c->trotter && c->second_order) ? 3 * (c->num_qubits / 2) : 0;
    std::vector<BlockRef> out;
    for (int i = 0; i < L + tail && L > 0; ++i) out.push_back({i, i % L, c->blocks[i % L], c->blocks[L + i % L]});
    return out;
}


}  // namespace

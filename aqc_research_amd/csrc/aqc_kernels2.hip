// Register-blocked stage kernels (gfx950).
//
// Same tiling as aqc_kernels.hip (one workgroup = one LDS tile of 2^k amplitudes of one lane), but
// inside a stage the gate groups are executed in *sub-stages*: every thread pulls the 2^r amplitudes
// that differ only in r "register bits" out of LDS (r = 4 for the w/z sweep: 16 + 16 complex128 =
// 128 VGPRs; r = 5 for V / V^H), runs every micro-op (single rotation or entangler) of the sub-stage
// on registers, and writes them back.  LDS traffic drops from one round trip per gate group to one
// per sub-stage (4-7 gate groups for brickwork circuits), which moves the kernel from LDS-latency
// bound to fp64-VALU bound.  The tile is stored XOR-swizzled (slot = l ^ ((l >> 4) & 15)) so that the
// strided register-bit accesses of 16 consecutive lanes fall into 16 different 16-byte bank groups.
//
// Inner products: each rotation micro-op reduces its 0.5j<P w|z> with a DPP-only wave butterfly
// (no LDS traffic), lane 63 drops the value into a per-wave LDS slot, and the slots are summed in a
// fixed order after the sub-stage's barrier (bit-reproducible, no float atomics).
#include <hip/hip_runtime.h>

#include "aqc_device.h"
#include "aqc_launch.h"
#include "aqc_math.h"

#include <type_traits>
#include <utility>

#ifndef AQC_OPT_PREFETCH
#define AQC_OPT_PREFETCH 1
#endif
#ifndef AQC_OPT_UNROLL
#define AQC_OPT_UNROLL 1
#endif
#ifndef AQC_OPT_DPPRED
#define AQC_OPT_DPPRED 1
#endif

namespace aqc {

__device__ __forceinline__ unsigned swz(unsigned l) { return l ^ ((l >> 4) & 15u); }

// ---- in-place arithmetic --------------------------------------------------------------------------
// Every update below is a v_fma_f64 whose destination is tied to its addend ("+v"), so the 64 doubles
// of a chunk stay in the registers they were loaded into: without the tie the compiler materialises
// each micro-op's results in fresh registers and copies them back at the dispatch merge point
// (v_mov_b64 count ~ fp64 op count in the first version of this kernel).
__device__ __forceinline__ void fma_ip(double& acc, double a, double b) {   // acc += a * b
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void fnma_ip(double& acc, double a, double b) {  // acc -= a * b
    asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// Plane rotation by phi, |phi| <= pi/2, as three shears: (x, y) <- (c x - s y, s x + c y) with
// nt = -tan(phi/2), s = sin(phi).  3 FMAs instead of 2 mul + 2 fma, and no temporaries.
__device__ __forceinline__ void lift_pos(double& x, double& y, double nt, double s) {
    fma_ip(x, nt, y); fma_ip(y, s, x); fma_ip(x, nt, y);
}
__device__ __forceinline__ void lift_neg(double& x, double& y, double nt, double s) {  // rotation by -phi
    fnma_ip(x, nt, y); fnma_ip(y, s, x); fnma_ip(x, nt, y);
}
template <int KIND>
__device__ __forceinline__ void rot_pair_ip(cplx& a0, cplx& a1, double nt, double s) {
    if (KIND == MOP_RY) {          // [[c,-s],[s,c]] on real and imaginary parts
        lift_pos(a0.x, a1.x, nt, s); lift_pos(a0.y, a1.y, nt, s);
    } else if (KIND == MOP_RZ) {   // a0 *= e^{-i phi}, a1 *= e^{+i phi}
        lift_neg(a0.x, a0.y, nt, s); lift_pos(a1.x, a1.y, nt, s);
    } else {                       // Rx: (a0.x, a1.y) rotate by -phi, (a0.y, a1.x) by +phi
        lift_neg(a0.x, a1.y, nt, s); lift_pos(a0.y, a1.x, nt, s);
    }
}
__device__ __forceinline__ void cmacc_ip(cplx& d, const cplx& a, const cplx& b) {  // d += conj(a) b
    fma_ip(d.x, a.x, b.x); fma_ip(d.x, a.y, b.y); fma_ip(d.y, a.x, b.y); fnma_ip(d.y, a.y, b.x);
}
__device__ __forceinline__ void cmsub_ip(cplx& d, const cplx& a, const cplx& b) {  // d -= conj(a) b
    fnma_ip(d.x, a.x, b.x); fnma_ip(d.x, a.y, b.y); fnma_ip(d.y, a.x, b.y); fma_ip(d.y, a.y, b.x);
}
__device__ __forceinline__ void swap_ip(cplx& a, cplx& b) { const cplx t = a; a = b; b = t; }

// Compile-time loop: the index arrives as an integral_constant, so every array subscript is a constant
// in the very first IR and the arrays are promoted to SSA values before any CFG transformation runs
// (with `#pragma unroll` loops the promotion has to wait for the unroller, and by then SimplifyCFG may
// have merged look-alike arms through pointer phis, which pins the arrays in scratch memory).
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---- micro-ops on register-resident amplitudes ---------------------------------------------------
// one vector
template <int R, int P, int KIND>
__device__ __forceinline__ void rot_regs1(cplx (&v)[1 << R], double nt, double s) {
    static_for<(1 << (R - 1))>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int i0 = ((g >> P) << (P + 1)) | (g & ((1 << P) - 1)), i1 = i0 | (1 << P);
        rot_pair_ip<KIND>(v[i0], v[i1], nt, s);
    });
}
template <int R, int PC, int PT, int KIND>
__device__ __forceinline__ void ent_regs1(cplx (&v)[1 << R], double c, double s) {
    constexpr int LO = PC < PT ? PC : PT, HI = PC < PT ? PT : PC;
    static_for<(1 << (R - 2))>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int b0 = ((g >> LO) << (LO + 1)) | (g & ((1 << LO) - 1));
        constexpr int b = ((b0 >> HI) << (HI + 1)) | (b0 & ((1 << HI) - 1));
        constexpr int i2 = b | (1 << PC), i3 = i2 | (1 << PT);
        if (KIND == MOP_CX) swap_ip(v[i2], v[i3]);
        else if (KIND == MOP_CZ) { v[i3].x = -v[i3].x; v[i3].y = -v[i3].y; }
        else v[i3] = cmul(v[i3], c, s);
    });
}
// two vectors + inner product
template <int R, int P, int KIND, bool DOT>
__device__ __forceinline__ void rot_regs2(cplx (&w)[1 << R], cplx (&z)[1 << R], double nt, double s, cplx& d) {
    static_for<(1 << (R - 1))>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int i0 = ((g >> P) << (P + 1)) | (g & ((1 << P) - 1)), i1 = i0 | (1 << P);
        rot_pair_ip<KIND>(w[i0], w[i1], nt, s);
        rot_pair_ip<KIND>(z[i0], z[i1], nt, s);
        if (DOT) {
            if (KIND == MOP_RY) { cmacc_ip(d, w[i0], z[i1]); cmsub_ip(d, w[i1], z[i0]); }        // <Y w|z> / i
            else if (KIND == MOP_RZ) { cmacc_ip(d, w[i0], z[i0]); cmsub_ip(d, w[i1], z[i1]); }   // <Z w|z>
            else { cmacc_ip(d, w[i1], z[i0]); cmacc_ip(d, w[i0], z[i1]); }                       // <X w|z>
        }
    });
}
template <int R, int PC, int PT, int KIND, bool DOT>
__device__ __forceinline__ void ent_regs2(cplx (&w)[1 << R], cplx (&z)[1 << R], double c, double s, cplx& d) {
    constexpr int LO = PC < PT ? PC : PT, HI = PC < PT ? PT : PC;
    static_for<(1 << (R - 2))>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int b0 = ((g >> LO) << (LO + 1)) | (g & ((1 << LO) - 1));
        constexpr int b = ((b0 >> HI) << (HI + 1)) | (b0 & ((1 << HI) - 1));
        constexpr int i2 = b | (1 << PC), i3 = i2 | (1 << PT);
        if (KIND == MOP_CX) {
            swap_ip(w[i2], w[i3]); swap_ip(z[i2], z[i3]);
        } else if (KIND == MOP_CZ) {
            w[i3].x = -w[i3].x; w[i3].y = -w[i3].y; z[i3].x = -z[i3].x; z[i3].y = -z[i3].y;
        } else {
            if (DOT) cmacc_ip(d, w[i3], z[i3]);  // -i<P11 w|z>, pre-gate (core_op_matrix.py:430-477)
            w[i3] = cmul(w[i3], c, s); z[i3] = cmul(z[i3], c, s);
        }
    });
}

// ---- dispatch -----------------------------------------------------------------------------------------
// The dispatch is a sequence of ONE-SIDED, wave-uniform ifs.  A switch (or if/else chain) is lowered
// to a multi-exit region that StructurizeCFG linearises with "Flow" blocks; the phis of those blocks
// keep the pre-dispatch copy of every live value alive across each arm, so every arm starts by copying
// the whole register tile (measured: ~1 v_mov_b64 per fp64 op).  Guarded in-place updates have no such
// problem: each arm is a simple diamond and the values never leave their registers.
// Conditions are single-bit tests of ONE-HOT masks (kind, position, target, dot): the compiler cannot
// prove two bit tests exclusive, so it cannot fold the sequence back into a switch / if-else chain.
#define AQC_ROT1_ARM(K, P) if ((P) < R && (pm & (1 << (P)))) rot_regs1<R, (P) < R ? (P) : 0, K>(v, c, s);
#define AQC_ENT1_ARM(K, PC, PT) if (tm & (1 << (PT))) ent_regs1<R, PC, PT, K>(v, c, s);

template <int R, int ENT>
__device__ __forceinline__ void run_mop1(int km, int pm, int tm, cplx (&v)[1 << R], double c, double s) {
    if (km & (1 << MOP_RY)) { AQC_ROT1_ARM(MOP_RY, 0) AQC_ROT1_ARM(MOP_RY, 1) AQC_ROT1_ARM(MOP_RY, 2) AQC_ROT1_ARM(MOP_RY, 3) }
    if (km & (1 << MOP_RZ)) { AQC_ROT1_ARM(MOP_RZ, 0) AQC_ROT1_ARM(MOP_RZ, 1) AQC_ROT1_ARM(MOP_RZ, 2) AQC_ROT1_ARM(MOP_RZ, 3) }
    if (ENT == 0 && (km & (1 << MOP_RX))) { AQC_ROT1_ARM(MOP_RX, 0) AQC_ROT1_ARM(MOP_RX, 1) AQC_ROT1_ARM(MOP_RX, 2) AQC_ROT1_ARM(MOP_RX, 3) }
    constexpr int K = ENT == 0 ? MOP_CX : (ENT == 1 ? MOP_CZ : MOP_CP);
    if (km & (1 << K)) {
        if (pm & 1) { AQC_ENT1_ARM(K, 0, 1) AQC_ENT1_ARM(K, 0, 2) AQC_ENT1_ARM(K, 0, 3) }
        if (pm & 2) { AQC_ENT1_ARM(K, 1, 0) AQC_ENT1_ARM(K, 1, 2) AQC_ENT1_ARM(K, 1, 3) }
        if (pm & 4) { AQC_ENT1_ARM(K, 2, 0) AQC_ENT1_ARM(K, 2, 1) AQC_ENT1_ARM(K, 2, 3) }
        if (pm & 8) { AQC_ENT1_ARM(K, 3, 0) AQC_ENT1_ARM(K, 3, 1) AQC_ENT1_ARM(K, 3, 2) }
    }
}

// dm: one-hot {1: no inner product, 2: inner product}
#define AQC_ROT2_ARM(K, P) \
    if ((P) < R && (pm & (1 << (P)))) { \
        if (dm & 2) rot_regs2<R, (P) < R ? (P) : 0, K, true>(w, z, c, s, d); \
        if (dm & 1) rot_regs2<R, (P) < R ? (P) : 0, K, false>(w, z, c, s, d); }
#define AQC_ENT2_ARM(K, PC, PT) \
    if ((PC) < R && (PT) < R && (tm & (1 << (PT)))) { \
        if (dm & 2) ent_regs2<R, (PC) < R ? (PC) : 0, (PT) < R ? (PT) : 1, K, true>(w, z, c, s, d); \
        if (dm & 1) ent_regs2<R, (PC) < R ? (PC) : 0, (PT) < R ? (PT) : 1, K, false>(w, z, c, s, d); }

template <int ENT, int R>
__device__ __forceinline__ void run_mop2(int km, int pm, int tm, int dm, cplx (&w)[1 << R], cplx (&z)[1 << R], double c, double s, cplx& d) {
    if (km & (1 << MOP_RY)) { AQC_ROT2_ARM(MOP_RY, 0) AQC_ROT2_ARM(MOP_RY, 1) AQC_ROT2_ARM(MOP_RY, 2) AQC_ROT2_ARM(MOP_RY, 3) }
    if (km & (1 << MOP_RZ)) { AQC_ROT2_ARM(MOP_RZ, 0) AQC_ROT2_ARM(MOP_RZ, 1) AQC_ROT2_ARM(MOP_RZ, 2) AQC_ROT2_ARM(MOP_RZ, 3) }
    if (ENT == 0 && (km & (1 << MOP_RX))) { AQC_ROT2_ARM(MOP_RX, 0) AQC_ROT2_ARM(MOP_RX, 1) AQC_ROT2_ARM(MOP_RX, 2) AQC_ROT2_ARM(MOP_RX, 3) }
    constexpr int K = ENT == 0 ? MOP_CX : (ENT == 1 ? MOP_CZ : MOP_CP);
    if (km & (1 << K)) {
        if (pm & 1) { AQC_ENT2_ARM(K, 0, 1) AQC_ENT2_ARM(K, 0, 2) AQC_ENT2_ARM(K, 0, 3) }
        if (pm & 2) { AQC_ENT2_ARM(K, 1, 0) AQC_ENT2_ARM(K, 1, 2) AQC_ENT2_ARM(K, 1, 3) }
        if (pm & 4) { AQC_ENT2_ARM(K, 2, 0) AQC_ENT2_ARM(K, 2, 1) AQC_ENT2_ARM(K, 2, 3) }
        if (pm & 8) { AQC_ENT2_ARM(K, 3, 0) AQC_ENT2_ARM(K, 3, 1) AQC_ENT2_ARM(K, 3, 2) }
    }
}

// Micro-op as the inner loop sees it: decoded once per sub-stage into LDS (one thread per micro-op)
// so that the per-micro-op dispatch costs one LDS broadcast read instead of two dependent global loads.
struct __attribute__((aligned(16))) SMop {
    int km;        // one-hot: 1 << MopKind
    int pm;        // one-hot: 1 << register bit (control bit); MOP_REDUCE: producer kinds, 4 bits each, newest first
    int tm;        // one-hot: 1 << target bit (entanglers); MOP_REDUCE: index of this reduction inside the sub-stage
    int dm;        // one-hot: 2 if the inner product(s) of this micro-op are wanted, else 1
    union {
        struct { double c, s; };  // rotation: (-tan(phi/2), sin(phi)); CP: (cos, sin); sign applied
        int slots[4];             // MOP_REDUCE: enabled slots of the 4 newest inner products (-1 = none)
    };
};

struct Tile2 {
    size_t base;
    unsigned* dlo;
    unsigned* dhi;
};
__device__ __forceinline__ Tile2 tile_setup2(const DevStage* st, unsigned* tables) {
    Tile2 t;
    t.dlo = tables;
    t.dhi = tables + 64;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) t.dlo[i] = st->dlo[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) t.dhi[i] = st->dhi[i];
    size_t base = 0;
    const unsigned tile = blockIdx.x;
    for (int i = 0; i < st->nub; ++i) base |= (size_t)((tile >> i) & 1u) << st->ubits[i];
    t.base = base;
    return t;
}

// chunk -> local index of its amplitude 0 (zeros inserted at the register bits, ascending)
template <int R>
__device__ __forceinline__ unsigned chunk_base(unsigned chunk, const DevSub& sub) {
    unsigned b = chunk;
#pragma unroll
    for (int i = 0; i < R; ++i) b = insert_zero(b, sub.bits[i]);
    return b;
}
template <int R>
__device__ __forceinline__ unsigned amp_offset(int j, const DevSub& sub) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (j >> i & 1) o |= 1u << sub.bits[i];
    return o;
}

// ---- V / V^H, r = 5 ---------------------------------------------------------------------------------
template <int ENT>
__global__ __launch_bounds__(512) void apply_stage_kernel2(StageArgs a) {
    constexpr int R = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage* st = a.stage;
    const unsigned tsize = 1u << st->k;
    cplx* tile = reinterpret_cast<cplx*>(smem);
    unsigned* tables = reinterpret_cast<unsigned*>(smem + (size_t)tsize * sizeof(cplx));
    SMop* smops = reinterpret_cast<SMop*>(tables + 320);  // [2][kMaxMopsPerSub]
    const Tile2 tc = tile_setup2(st, tables);
    const double* coef = a.coef + (size_t)blockIdx.y * a.ncoef * kCoefStride;
    // Decoding sub-stage si+1 (one thread per micro-op: two dependent global loads) is split into `fetch`, issued
    // at the start of sub-stage si, and `commit` (the LDS write) after si's micro-op loop: the ~1 us of load
    // latency hides behind the arithmetic instead of stalling every sub-stage.
    SMop pend;
    bool pend_valid = false;
    auto fetch_mop = [&](int si) {
        pend_valid = false;
        if (si >= st->nsubs) return;
        const DevSub sub = a.subs[st->sub_begin + si];
        if ((int)threadIdx.x < sub.nmops) {
            const DevMop m = a.mops[sub.mop_begin + threadIdx.x];
            pend.km = 1 << m.kind; pend.pm = 1 << m.p; pend.tm = 1 << m.p2; pend.dm = 1;
            const double sg = (m.flags & MOPF_NEG_S) ? -1.0 : 1.0;
            pend.c = m.kind <= MOP_RX ? sg * coef[m.coef] : coef[m.coef];   // rotations: (-t, s) flip together
            pend.s = sg * coef[m.coef + 1];
            pend_valid = true;
        }
    };
    auto commit_mop = [&](int si) {
        if (pend_valid) smops[(si & 1) * kMaxMopsPerSub + threadIdx.x] = pend;
    };
    fetch_mop(0);
    commit_mop(0);
    __syncthreads();
    const size_t lane_off = (size_t)blockIdx.y * a.lane_stride + tc.base;
    const cplx* src = a.in0 + lane_off;
#if AQC_OPT_UNROLL
#pragma unroll 8
#endif
    for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) tile[swz(l)] = src[tc.dlo[l & 63u] + tc.dhi[l >> 6]];

    const unsigned nchunks = tsize >> R;
    const bool active = threadIdx.x < nchunks;
    for (int si = 0; si < st->nsubs; ++si) {
        const DevSub sub = a.subs[st->sub_begin + si];
        __syncthreads();
        fetch_mop(si + 1);
        cplx v[1 << R];
        if (active) {
            const unsigned b = chunk_base<R>(threadIdx.x, sub);
#pragma unroll
            for (int j = 0; j < (1 << R); ++j) v[j] = tile[swz(b | amp_offset<R>(j, sub))];
            const SMop* sm = smops + (si & 1) * kMaxMopsPerSub;
            SMop mnext = sm[0];
            for (int i = 0; i < sub.nmops; ++i) {
                const SMop m = mnext;
                mnext = sm[i + 1 < sub.nmops ? i + 1 : i];
                run_mop1<R, ENT>(__builtin_amdgcn_readfirstlane(m.km), __builtin_amdgcn_readfirstlane(m.pm),
                                 __builtin_amdgcn_readfirstlane(m.tm), v, m.c, m.s);
            }
#pragma unroll
            for (int j = 0; j < (1 << R); ++j) tile[swz(b | amp_offset<R>(j, sub))] = v[j];
        }
        commit_mop(si + 1);
    }
    __syncthreads();
    cplx* dst = a.out0 + lane_off;
    // rotations with cos(phi) < 0 were applied as minus the rotation by phi -+ pi: undo the lane's sign
    const double sign = a.final_stage ? coef[(size_t)(a.ncoef - 1) * kCoefStride + 2] : 1.0;
    for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) {
        cplx v = tile[swz(l)];
        v.x *= sign; v.y *= sign;
        dst[tc.dlo[l & 63u] + tc.dhi[l >> 6]] = v;
    }
}

// ---- forward w/z sweep with in-flight inner products, r = 4 -----------------------------------------
// w and z live in registers for the whole stage (2 x 2^R amplitudes per thread); LDS only carries them from one
// sub-stage's register layout to the next, and the two vectors take turns in ONE tile buffer.  That halves the
// LDS footprint (64 KB at k = 12), so two workgroups share a CU and each SIMD has a second wave to issue from
// while the first waits on LDS, a barrier or a DPP reduction.
template <int ENT, int R>
__global__ __launch_bounds__(R == 4 ? 256 : 512) void sweep_stage_kernel2(StageArgs a) {
    constexpr int NA = 1 << R;  // amplitudes per thread and vector
    constexpr int kMaxReducePerSub = R == 4 ? aqc::kMaxReducePerSub : aqc::kMaxReducePerSub / 2;  // shadows: 8 waves at R = 3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage* st = a.stage;
    const unsigned tsize = 1u << st->k;
    cplx* tt = reinterpret_cast<cplx*>(smem);
    unsigned* tables = reinterpret_cast<unsigned*>(tt + tsize);
    const int nwaves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* scratch = reinterpret_cast<double*>(tables + 320);  // [kMaxReducePerSub][nwaves * 4 rows][8]
    SMop* smops = reinterpret_cast<SMop*>(scratch + (size_t)kMaxReducePerSub * nwaves * 32);  // [kMaxMopsPerSub]
    const Tile2 tc = tile_setup2(st, tables);
    const double* coef = a.coef + (size_t)blockIdx.y * a.ncoef * kCoefStride;
    auto enabled = [&](int jblock) { return jblock < 0 ? (a.front != 0) : (jblock >= a.from && jblock < a.to); };
    SMop pend;                 // decode of the next sub-stage: fetched early, committed to LDS late (see apply kernel)
    bool pend_valid = false;
    auto fetch_mop = [&](int si) {
        pend_valid = false;
        if (si >= st->nsubs) return;
        const DevSub sub = a.subs[st->sub_begin + si];
        const int t = threadIdx.x;
        if (t < sub.nmops) {
            const DevMop m = a.mops[sub.mop_begin + t];
            const bool on = enabled(m.jblock);
            pend.km = 1 << m.kind;
#ifdef AQC_TUNING
            if ((a.debug & 1) && m.kind != MOP_REDUCE) pend.km = 1 << 20;
            if ((a.debug & 2) && m.kind == MOP_REDUCE) pend.km = 1 << 20;
#endif
            if (m.kind == MOP_REDUCE) {
                pend.pm = m.flags;
                pend.slots[0] = on ? m.slot : -1; pend.slots[1] = on ? m.p : -1;
                pend.slots[2] = on ? m.p2 : -1;   pend.slots[3] = on ? m.coef : -1;
                pend.dm = on ? 2 : 1;
                int r = 0;  // index among the reductions of this sub-stage
                for (int u = 0; u < t; ++u) r += a.mops[sub.mop_begin + u].kind == MOP_REDUCE;
                pend.tm = r;
            } else {
                pend.pm = 1 << m.p; pend.tm = 1 << m.p2;
                pend.dm = (on && m.slot >= 0) ? 2 : 1;
                const double sg = (m.flags & MOPF_NEG_S) ? -1.0 : 1.0;
                pend.c = m.kind <= MOP_RX ? sg * coef[m.coef] : coef[m.coef];
                pend.s = sg * coef[m.coef + 1];
            }
            pend_valid = true;
        }
    };
    auto commit_mop = [&]() {
        if (pend_valid) smops[threadIdx.x] = pend;
    };
    const size_t lane_off = (size_t)blockIdx.y * a.lane_stride + tc.base;
    const cplx* sw = a.in0 + lane_off;
    const cplx* sz = a.in1 + lane_off;
    cplx* dw = a.out0 + lane_off;
    cplx* dz = a.out1 + lane_off;
#ifdef AQC_TUNING
    const int nsubs_run = (a.debug & 4) ? 0 : st->nsubs;   // timing experiment: memory phases only
#else
    const int nsubs_run = st->nsubs;
#endif
    if (nsubs_run == 0) {   // nothing to do in this stage: plain copy
        __syncthreads();
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) {
            const unsigned off = tc.dlo[l & 63u] + tc.dhi[l >> 6];
            dw[off] = sw[off];
            dz[off] = sz[off];
        }
        return;
    }
    cplx* partial = a.partial + (size_t)blockIdx.y * a.nslots * a.ntiles_max;
    const unsigned nchunks = tsize >> R;
    const bool active = threadIdx.x < nchunks;
    auto load_tile = [&](const cplx* src) {   // HBM -> LDS, coalesced runs of the stage's low bits
#if AQC_OPT_UNROLL
#pragma unroll 8
#endif
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) tt[swz(l)] = src[tc.dlo[l & 63u] + tc.dhi[l >> 6]];
    };
    auto store_tile = [&](cplx* dst) {
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) dst[tc.dlo[l & 63u] + tc.dhi[l >> 6]] = tt[swz(l)];
    };
    auto regs_from_tile = [&](cplx (&v)[NA], const DevSub& sub) {
        if (active) {
            const unsigned b = chunk_base<R>(threadIdx.x, sub);
#pragma unroll
            for (int j = 0; j < NA; ++j) v[j] = tt[swz(b | amp_offset<R>(j, sub))];
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) v[j] = make_double2(0.0, 0.0);
        }
    };
    auto regs_to_tile = [&](const cplx (&v)[NA], const DevSub& sub) {
        if (active) {
            const unsigned b = chunk_base<R>(threadIdx.x, sub);
#pragma unroll
            for (int j = 0; j < NA; ++j) tt[swz(b | amp_offset<R>(j, sub))] = v[j];
        }
    };
    auto flush = [&](int count) {  // fixed-order cross-wave sums of the finished sub-stage's reductions
        for (int t = threadIdx.x; t < count; t += blockDim.x) {
            const SMop m = smops[t];
            if (!(m.km & (1 << MOP_REDUCE)) || !(m.dm & 2)) continue;
            const double* sc = scratch + (size_t)m.tm * nwaves * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (m.slots[q] < 0) continue;
                double re = 0.0, im = 0.0;
                for (int rw = 0; rw < nwaves * 4; ++rw) { re += sc[rw * 8 + 2 * q]; im += sc[rw * 8 + 2 * q + 1]; }
                const int kind = (m.pm >> (4 * q)) & 15;
                cplx r;  // factors: Ry -> 0.5, Rz / Rx -> 0.5j, CP -> -1j  (core_operations.py:267-351,972-975)
                if (kind == MOP_RY) r = make_double2(0.5 * re, 0.5 * im);
                else if (kind == MOP_CP) r = make_double2(im, -re);
                else r = make_double2(-0.5 * im, 0.5 * re);
                partial[(size_t)m.slots[q] * a.ntiles_max + blockIdx.x] = r;
            }
        }
    };

    cplx w[NA], z[NA];
    DevSub sub = a.subs[st->sub_begin];
    fetch_mop(0);
    commit_mop();
    __syncthreads();          // offset tables (tile_setup2) and the first descriptors are in LDS
    load_tile(sw);
    __syncthreads();
    regs_from_tile(w, sub);
    __syncthreads();
    load_tile(sz);
    __syncthreads();
    regs_from_tile(z, sub);

    for (int si = 0; si < nsubs_run; ++si) {
        const bool last = si + 1 == nsubs_run;
        fetch_mop(si + 1);        // registers only; written to smops once this sub-stage's flush has read them
        cplx d0 = make_double2(0.0, 0.0), d1 = d0, d2 = d0, d3 = d0;  // newest ... oldest pending inner products
        SMop mnext = smops[0];
        for (int i = 0; i < sub.nmops; ++i) {
            const SMop m = mnext;
#if AQC_OPT_PREFETCH
            mnext = smops[i + 1 < sub.nmops ? i + 1 : i];   // fetch the next descriptor under this micro-op's arithmetic
#else
            if (i + 1 < sub.nmops) mnext = smops[i + 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            const int km = __builtin_amdgcn_readfirstlane(m.km);  // wave-uniform by construction
            const int mp = __builtin_amdgcn_readfirstlane(m.pm), mp2 = __builtin_amdgcn_readfirstlane(m.tm);
            const int dm = __builtin_amdgcn_readfirstlane(m.dm);
            if (km & (1 << MOP_REDUCE)) {
                if (dm & 2) {
                    const double v[8] = {d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
                    const double tot = reduce8_row(v, lane);
                    if ((lane & 15) < 8)
                        scratch[(((size_t)mp2) * nwaves * 4 + wave * 4 + (lane >> 4)) * 8 + (lane & 7)] = tot;
                }
                d0 = d1 = d2 = d3 = make_double2(0.0, 0.0);
            }
            if (km & ((1 << MOP_REDUCE) - 1)) {
                cplx d = make_double2(0.0, 0.0);
                run_mop2<ENT, R>(km, mp, mp2, dm, w, z, m.c, m.s, d);   // idle lanes carry zeros: harmless
                if (dm & 2) { d3 = d2; d2 = d1; d1 = d0; d0 = d; }
            }
        }
        // hand over to the next sub-stage's register layout (or to HBM): w first, then z, through the one tile
        const DevSub next = a.subs[st->sub_begin + (last ? si : si + 1)];
        __syncthreads();          // every wave finished the loop (scratch complete) and its earlier reads of the tile
        regs_to_tile(w, sub);
        flush(sub.nmops);
        __syncthreads();
        commit_mop();             // flush has read this sub-stage's descriptors; visible after the next barrier
        if (last) store_tile(dw); else regs_from_tile(w, next);
        __syncthreads();
        regs_to_tile(z, sub);
        __syncthreads();
        if (last) store_tile(dz); else regs_from_tile(z, next);
        sub = next;
    }
}

// ---- launchers -----------------------------------------------------------------------------------------
size_t apply2_lds_bytes(int k) { return ((size_t)16 << k) + 320 * sizeof(unsigned) + (size_t)2 * kMaxMopsPerSub * sizeof(SMop); }
size_t sweep2_lds_bytes(int k, int threads, int reg_bits) {
    const int maxr = reg_bits == 4 ? kMaxReducePerSub : kMaxReducePerSub / 2;
    return ((size_t)16 << k) + 320 * sizeof(unsigned) + (size_t)maxr * (threads / 64) * 32 * sizeof(double) +
           (size_t)kMaxMopsPerSub * sizeof(SMop);
}
int apply2_threads(int k) { return std::max(64, 1 << (k - 4)); }
int sweep2_threads(int k, int r) { return std::max(64, 1 << (k - r)); }

template <typename K>
static hipError_t allow_big_lds2(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t init_kernels2() {
    hipError_t e;
#define AQC_TRY(x) if ((e = (x)) != hipSuccess) return e
    AQC_TRY(allow_big_lds2(apply_stage_kernel2<0>)); AQC_TRY(allow_big_lds2(apply_stage_kernel2<1>)); AQC_TRY(allow_big_lds2(apply_stage_kernel2<2>));
    AQC_TRY(allow_big_lds2(sweep_stage_kernel2<0, 4>)); AQC_TRY(allow_big_lds2(sweep_stage_kernel2<1, 4>)); AQC_TRY(allow_big_lds2(sweep_stage_kernel2<2, 4>));
    AQC_TRY(allow_big_lds2(sweep_stage_kernel2<0, 3>)); AQC_TRY(allow_big_lds2(sweep_stage_kernel2<1, 3>)); AQC_TRY(allow_big_lds2(sweep_stage_kernel2<2, 3>));
#undef AQC_TRY
    return hipSuccess;
}

hipError_t launch_apply2(int ent, int ntiles, int batch, int k, hipStream_t s, const StageArgs& a) {
    const dim3 grid(ntiles, batch), block(apply2_threads(k));
    const size_t lds = apply2_lds_bytes(k);
    switch (ent) {
        case 0: apply_stage_kernel2<0><<<grid, block, lds, s>>>(a); break;
        case 1: apply_stage_kernel2<1><<<grid, block, lds, s>>>(a); break;
        default: apply_stage_kernel2<2><<<grid, block, lds, s>>>(a); break;
    }
    return hipGetLastError();
}
hipError_t launch_sweep2(int ent, int ntiles, int batch, int k, int reg_bits, hipStream_t s, const StageArgs& a) {
    const int threads = sweep2_threads(k, reg_bits);
    const dim3 grid(ntiles, batch), block(threads);
    const size_t lds = sweep2_lds_bytes(k, threads, reg_bits);
    switch (ent * 2 + (reg_bits == 3 ? 1 : 0)) {
        case 0: sweep_stage_kernel2<0, 4><<<grid, block, lds, s>>>(a); break;
        case 1: sweep_stage_kernel2<0, 3><<<grid, block, lds, s>>>(a); break;
        case 2: sweep_stage_kernel2<1, 4><<<grid, block, lds, s>>>(a); break;
        case 3: sweep_stage_kernel2<1, 3><<<grid, block, lds, s>>>(a); break;
        case 4: sweep_stage_kernel2<2, 4><<<grid, block, lds, s>>>(a); break;
        default: sweep_stage_kernel2<2, 3><<<grid, block, lds, s>>>(a); break;
    }
    return hipGetLastError();
}

}  // namespace aqc

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

namespace aqc {

__device__ __forceinline__ unsigned swz(unsigned l) { return l ^ ((l >> 4) & 15u); }

// ---- micro-ops on register-resident amplitudes ---------------------------------------------------
// one vector
template <int R, int P, int KIND>
__device__ __forceinline__ void rot_regs1(cplx (&v)[1 << R], double c, double s) {
#pragma unroll
    for (int g = 0; g < (1 << (R - 1)); ++g) {
        const int i0 = ((g >> P) << (P + 1)) | (g & ((1 << P) - 1)), i1 = i0 | (1 << P);
        if (KIND == MOP_RY) ry2(v[i0], v[i1], c, s);
        else if (KIND == MOP_RZ) rz2(v[i0], v[i1], c, s);
        else rx2(v[i0], v[i1], c, s);
    }
}
template <int R, int PC, int PT, int KIND>
__device__ __forceinline__ void ent_regs1(cplx (&v)[1 << R], double c, double s) {
    constexpr int LO = PC < PT ? PC : PT, HI = PC < PT ? PT : PC;
#pragma unroll
    for (int g = 0; g < (1 << (R - 2)); ++g) {
        int b = ((g >> LO) << (LO + 1)) | (g & ((1 << LO) - 1));
        b = ((b >> HI) << (HI + 1)) | (b & ((1 << HI) - 1));
        const int i2 = b | (1 << PC), i3 = i2 | (1 << PT);
        if (KIND == MOP_CX) { const cplx t = v[i2]; v[i2] = v[i3]; v[i3] = t; }
        else if (KIND == MOP_CZ) { v[i3].x = -v[i3].x; v[i3].y = -v[i3].y; }
        else v[i3] = cmul(v[i3], c, s);
    }
}
// two vectors + inner product
template <int R, int P, int KIND, bool DOT>
__device__ __forceinline__ void rot_regs2(cplx (&w)[1 << R], cplx (&z)[1 << R], double c, double s, cplx& d) {
#pragma unroll
    for (int g = 0; g < (1 << (R - 1)); ++g) {
        const int i0 = ((g >> P) << (P + 1)) | (g & ((1 << P) - 1)), i1 = i0 | (1 << P);
        if (KIND == MOP_RY) {
            ry2(w[i0], w[i1], c, s); ry2(z[i0], z[i1], c, s);
            if (DOT) { cmacc(d, w[i0], z[i1]); cmsub(d, w[i1], z[i0]); }   // <Y w|z> / i
        } else if (KIND == MOP_RZ) {
            rz2(w[i0], w[i1], c, s); rz2(z[i0], z[i1], c, s);
            if (DOT) { cmacc(d, w[i0], z[i0]); cmsub(d, w[i1], z[i1]); }   // <Z w|z>
        } else {
            rx2(w[i0], w[i1], c, s); rx2(z[i0], z[i1], c, s);
            if (DOT) { cmacc(d, w[i1], z[i0]); cmacc(d, w[i0], z[i1]); }   // <X w|z>
        }
    }
}
template <int R, int PC, int PT, int KIND, bool DOT>
__device__ __forceinline__ void ent_regs2(cplx (&w)[1 << R], cplx (&z)[1 << R], double c, double s, cplx& d) {
    constexpr int LO = PC < PT ? PC : PT, HI = PC < PT ? PT : PC;
#pragma unroll
    for (int g = 0; g < (1 << (R - 2)); ++g) {
        int b = ((g >> LO) << (LO + 1)) | (g & ((1 << LO) - 1));
        b = ((b >> HI) << (HI + 1)) | (b & ((1 << HI) - 1));
        const int i2 = b | (1 << PC), i3 = i2 | (1 << PT);
        if (KIND == MOP_CX) {
            cplx t = w[i2]; w[i2] = w[i3]; w[i3] = t;
            t = z[i2]; z[i2] = z[i3]; z[i3] = t;
        } else if (KIND == MOP_CZ) {
            w[i3].x = -w[i3].x; w[i3].y = -w[i3].y; z[i3].x = -z[i3].x; z[i3].y = -z[i3].y;
        } else {
            if (DOT) cmacc(d, w[i3], z[i3]);  // -i<P11 w|z>, pre-gate (core_op_matrix.py:430-477)
            w[i3] = cmul(w[i3], c, s); z[i3] = cmul(z[i3], c, s);
        }
    }
}

#define AQC_ROT1_CASE(K, P) case (K) * 8 + (P): if ((P) < R) rot_regs1<R, (P) < R ? (P) : 0, K>(v, c, s); break;
#define AQC_ENT1_CASE(K, PC, PT) case (PC) * 8 + (PT): if ((PC) < R && (PT) < R) ent_regs1<R, (PC) < R ? (PC) : 0, (PT) < R ? (PT) : 1, K>(v, c, s); break;

template <int R, int ENT>
__device__ __forceinline__ void run_mop1(int code, cplx (&v)[1 << R], double c, double s) {
    if (code < 64) {
        switch (code) {
            AQC_ROT1_CASE(MOP_RY, 0) AQC_ROT1_CASE(MOP_RY, 1) AQC_ROT1_CASE(MOP_RY, 2) AQC_ROT1_CASE(MOP_RY, 3) AQC_ROT1_CASE(MOP_RY, 4)
            AQC_ROT1_CASE(MOP_RZ, 0) AQC_ROT1_CASE(MOP_RZ, 1) AQC_ROT1_CASE(MOP_RZ, 2) AQC_ROT1_CASE(MOP_RZ, 3) AQC_ROT1_CASE(MOP_RZ, 4)
            AQC_ROT1_CASE(MOP_RX, 0) AQC_ROT1_CASE(MOP_RX, 1) AQC_ROT1_CASE(MOP_RX, 2) AQC_ROT1_CASE(MOP_RX, 3) AQC_ROT1_CASE(MOP_RX, 4)
            default: break;
        }
    } else {
        constexpr int K = ENT == 0 ? MOP_CX : (ENT == 1 ? MOP_CZ : MOP_CP);
        switch (code - 64) {
            AQC_ENT1_CASE(K, 0, 1) AQC_ENT1_CASE(K, 0, 2) AQC_ENT1_CASE(K, 0, 3) AQC_ENT1_CASE(K, 0, 4)
            AQC_ENT1_CASE(K, 1, 0) AQC_ENT1_CASE(K, 1, 2) AQC_ENT1_CASE(K, 1, 3) AQC_ENT1_CASE(K, 1, 4)
            AQC_ENT1_CASE(K, 2, 0) AQC_ENT1_CASE(K, 2, 1) AQC_ENT1_CASE(K, 2, 3) AQC_ENT1_CASE(K, 2, 4)
            AQC_ENT1_CASE(K, 3, 0) AQC_ENT1_CASE(K, 3, 1) AQC_ENT1_CASE(K, 3, 2) AQC_ENT1_CASE(K, 3, 4)
            AQC_ENT1_CASE(K, 4, 0) AQC_ENT1_CASE(K, 4, 1) AQC_ENT1_CASE(K, 4, 2) AQC_ENT1_CASE(K, 4, 3)
            default: break;
        }
    }
}

#define AQC_ROT2_CASE(K, P) \
    case (K) * 16 + (P) * 2: rot_regs2<4, P, K, false>(w, z, c, s, d); break; \
    case (K) * 16 + (P) * 2 + 1: rot_regs2<4, P, K, true>(w, z, c, s, d); break;
#define AQC_ENT2_CASE(K, PC, PT) \
    case ((PC) * 4 + (PT)) * 2: ent_regs2<4, PC, PT, K, false>(w, z, c, s, d); break; \
    case ((PC) * 4 + (PT)) * 2 + 1: ent_regs2<4, PC, PT, K, true>(w, z, c, s, d); break;

template <int ENT>
__device__ __forceinline__ void run_mop2(int code, cplx (&w)[16], cplx (&z)[16], double c, double s, cplx& d) {
    if (code < 64) {
        switch (code) {
            AQC_ROT2_CASE(MOP_RY, 0) AQC_ROT2_CASE(MOP_RY, 1) AQC_ROT2_CASE(MOP_RY, 2) AQC_ROT2_CASE(MOP_RY, 3)
            AQC_ROT2_CASE(MOP_RZ, 0) AQC_ROT2_CASE(MOP_RZ, 1) AQC_ROT2_CASE(MOP_RZ, 2) AQC_ROT2_CASE(MOP_RZ, 3)
            AQC_ROT2_CASE(MOP_RX, 0) AQC_ROT2_CASE(MOP_RX, 1) AQC_ROT2_CASE(MOP_RX, 2) AQC_ROT2_CASE(MOP_RX, 3)
            default: break;
        }
    } else {
        constexpr int K = ENT == 0 ? MOP_CX : (ENT == 1 ? MOP_CZ : MOP_CP);
        switch (code - 64) {
            AQC_ENT2_CASE(K, 0, 1) AQC_ENT2_CASE(K, 0, 2) AQC_ENT2_CASE(K, 0, 3)
            AQC_ENT2_CASE(K, 1, 0) AQC_ENT2_CASE(K, 1, 2) AQC_ENT2_CASE(K, 1, 3)
            AQC_ENT2_CASE(K, 2, 0) AQC_ENT2_CASE(K, 2, 1) AQC_ENT2_CASE(K, 2, 3)
            AQC_ENT2_CASE(K, 3, 0) AQC_ENT2_CASE(K, 3, 1) AQC_ENT2_CASE(K, 3, 2)
            default: break;
        }
    }
}

// Micro-op as the inner loop sees it: decoded once per sub-stage into LDS (one thread per micro-op)
// so that the per-micro-op dispatch costs one LDS broadcast read instead of two dependent global loads.
struct __attribute__((aligned(16))) SMop {
    int code;      // dispatch code (kind / register bits / dot flag folded in)
    int kind;      // MopKind (for the inner-product factor)
    double c, s;   // rotation coefficients, sign already applied
    int slot;      // enabled inner-product slot or -1
    int pad;
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
    auto stage_mops = [&](int si) {  // decode sub-stage si into smops[si & 1]
        if (si >= st->nsubs) return;
        const DevSub sub = a.subs[st->sub_begin + si];
        for (int t = threadIdx.x; t < sub.nmops; t += blockDim.x) {
            const DevMop m = a.mops[sub.mop_begin + t];
            SMop sm;
            sm.kind = m.kind;
            sm.code = m.kind <= MOP_RX ? m.kind * 8 + m.p : 64 + m.p * 8 + m.p2;
            sm.c = coef[m.coef];
            sm.s = (m.flags & MOPF_NEG_S) ? -coef[m.coef + 1] : coef[m.coef + 1];
            sm.slot = -1;
            sm.pad = 0;
            smops[(si & 1) * kMaxMopsPerSub + t] = sm;
        }
    };
    stage_mops(0);
    __syncthreads();
    const size_t lane_off = (size_t)blockIdx.y * a.lane_stride + tc.base;
    const cplx* src = a.in0 + lane_off;
    for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) tile[swz(l)] = src[tc.dlo[l & 63u] + tc.dhi[l >> 6]];

    const unsigned nchunks = tsize >> R;
    const bool active = threadIdx.x < nchunks;
    for (int si = 0; si < st->nsubs; ++si) {
        const DevSub sub = a.subs[st->sub_begin + si];
        __syncthreads();
        stage_mops(si + 1);
        cplx v[1 << R];
        if (active) {
            const unsigned b = chunk_base<R>(threadIdx.x, sub);
#pragma unroll
            for (int j = 0; j < (1 << R); ++j) v[j] = tile[swz(b | amp_offset<R>(j, sub))];
            const SMop* sm = smops + (si & 1) * kMaxMopsPerSub;
            for (int i = 0; i < sub.nmops; ++i) run_mop1<R, ENT>(sm[i].code, v, sm[i].c, sm[i].s);
#pragma unroll
            for (int j = 0; j < (1 << R); ++j) tile[swz(b | amp_offset<R>(j, sub))] = v[j];
        }
    }
    __syncthreads();
    cplx* dst = a.out0 + lane_off;
    for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) dst[tc.dlo[l & 63u] + tc.dhi[l >> 6]] = tile[swz(l)];
}

// ---- forward w/z sweep with in-flight inner products, r = 4 -----------------------------------------
template <int ENT>
__global__ __launch_bounds__(256) void sweep_stage_kernel2(StageArgs a) {
    constexpr int R = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage* st = a.stage;
    const unsigned tsize = 1u << st->k;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    cplx* tz = tw + tsize;
    unsigned* tables = reinterpret_cast<unsigned*>(tz + tsize);
    const int nwaves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    cplx* scratch = reinterpret_cast<cplx*>(tables + 320);  // [2][kMaxMopsPerSub][nwaves]
    SMop* smops = reinterpret_cast<SMop*>(scratch + (size_t)2 * kMaxMopsPerSub * nwaves);  // [2][kMaxMopsPerSub]
    const Tile2 tc = tile_setup2(st, tables);
    const double* coef = a.coef + (size_t)blockIdx.y * a.ncoef * kCoefStride;
    auto enabled = [&](const DevMop& m) {
        return m.slot >= 0 && (m.jblock < 0 ? (a.front != 0) : (m.jblock >= a.from && m.jblock < a.to));
    };
    auto stage_mops = [&](int si) {  // decode sub-stage si into smops[si & 1]
        if (si >= st->nsubs) return;
        const DevSub sub = a.subs[st->sub_begin + si];
        for (int t = threadIdx.x; t < sub.nmops; t += blockDim.x) {
            const DevMop m = a.mops[sub.mop_begin + t];
            const bool dot = enabled(m);
            SMop sm;
            sm.kind = m.kind;
            sm.code = m.kind <= MOP_RX ? m.kind * 16 + m.p * 2 + (dot ? 1 : 0) : 64 + (m.p * 4 + m.p2) * 2 + (dot ? 1 : 0);
            sm.c = coef[m.coef];
            sm.s = (m.flags & MOPF_NEG_S) ? -coef[m.coef + 1] : coef[m.coef + 1];
            sm.slot = dot ? m.slot : -1;
            sm.pad = 0;
            smops[(si & 1) * kMaxMopsPerSub + t] = sm;
        }
    };
    stage_mops(0);
    __syncthreads();
    const size_t lane_off = (size_t)blockIdx.y * a.lane_stride + tc.base;
    {
        const cplx* sw = a.in0 + lane_off;
        const cplx* sz = a.in1 + lane_off;
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) {
            const unsigned off = tc.dlo[l & 63u] + tc.dhi[l >> 6], p = swz(l);
            tw[p] = sw[off];
            tz[p] = sz[off];
        }
    }
    cplx* partial = a.partial + (size_t)blockIdx.y * a.nslots * a.ntiles_max;
    const unsigned nchunks = tsize >> R;
    const bool active = threadIdx.x < nchunks;

    int prev_n = 0;
    auto flush = [&](int par) {  // fixed-order cross-wave sum of the previous sub-stage's inner products
        for (int t = threadIdx.x; t < prev_n; t += blockDim.x) {
            const int slot = smops[par * kMaxMopsPerSub + t].slot;
            if (slot < 0) continue;
            const cplx* sc = scratch + ((size_t)par * kMaxMopsPerSub + t) * nwaves;
            cplx acc = sc[0];
            for (int wv = 1; wv < nwaves; ++wv) { acc.x += sc[wv].x; acc.y += sc[wv].y; }
            partial[(size_t)slot * a.ntiles_max + blockIdx.x] = acc;
        }
    };

    for (int si = 0; si < st->nsubs; ++si) {
        const DevSub sub = a.subs[st->sub_begin + si];
        const int par = si & 1;
        __syncthreads();
        flush(par ^ 1);
        __syncthreads();          // the slots of smops[par ^ 1] are consumed before they are overwritten
        stage_mops(si + 1);
        cplx w[16], z[16];
        const unsigned b = chunk_base<R>(threadIdx.x, sub);
        if (active) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { const unsigned p = swz(b | amp_offset<R>(j, sub)); w[j] = tw[p]; z[j] = tz[p]; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) { w[j] = make_double2(0.0, 0.0); z[j] = make_double2(0.0, 0.0); }
        }
        const SMop* smp = smops + par * kMaxMopsPerSub;
        for (int i = 0; i < sub.nmops; ++i) {
            const SMop m = smp[i];
            const bool dot = m.slot >= 0;
            cplx d = make_double2(0.0, 0.0);
            run_mop2<ENT>(m.code, w, z, m.c, m.s, d);   // idle lanes carry zeros: harmless and keeps the wave uniform
            if (dot) {
                const double re = wave_sum_dpp(d.x), im = wave_sum_dpp(d.y);
                if (lane == 63) {
                    cplx r;  // factors: Ry -> 0.5, Rz / Rx -> 0.5j, CP -> -1j
                    if (m.kind == MOP_RY) r = make_double2(0.5 * re, 0.5 * im);
                    else if (m.kind == MOP_CP) r = make_double2(im, -re);
                    else r = make_double2(-0.5 * im, 0.5 * re);
                    scratch[((size_t)par * kMaxMopsPerSub + i) * nwaves + wave] = r;
                }
            }
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { const unsigned p = swz(b | amp_offset<R>(j, sub)); tw[p] = w[j]; tz[p] = z[j]; }
        }
        prev_n = sub.nmops;
    }
    __syncthreads();
    flush((st->nsubs - 1) & 1);
    {
        cplx* dw = a.out0 + lane_off;
        cplx* dz = a.out1 + lane_off;
        for (unsigned l = threadIdx.x; l < tsize; l += blockDim.x) {
            const unsigned off = tc.dlo[l & 63u] + tc.dhi[l >> 6], p = swz(l);
            dw[off] = tw[p];
            dz[off] = tz[p];
        }
    }
}

// ---- launchers -----------------------------------------------------------------------------------------
size_t apply2_lds_bytes(int k) { return ((size_t)16 << k) + 320 * sizeof(unsigned) + (size_t)2 * kMaxMopsPerSub * sizeof(SMop); }
size_t sweep2_lds_bytes(int k, int threads) {
    return ((size_t)32 << k) + 320 * sizeof(unsigned) + (size_t)2 * kMaxMopsPerSub * (threads / 64) * sizeof(cplx) +
           (size_t)2 * kMaxMopsPerSub * sizeof(SMop);
}
int apply2_threads(int k) { return std::max(64, 1 << (k - 4)); }
int sweep2_threads(int k) { return std::max(64, 1 << (k - 4)); }

template <typename K>
static hipError_t allow_big_lds2(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t init_kernels2() {
    hipError_t e;
#define AQC_TRY(x) if ((e = (x)) != hipSuccess) return e
    AQC_TRY(allow_big_lds2(apply_stage_kernel2<0>)); AQC_TRY(allow_big_lds2(apply_stage_kernel2<1>)); AQC_TRY(allow_big_lds2(apply_stage_kernel2<2>));
    AQC_TRY(allow_big_lds2(sweep_stage_kernel2<0>)); AQC_TRY(allow_big_lds2(sweep_stage_kernel2<1>)); AQC_TRY(allow_big_lds2(sweep_stage_kernel2<2>));
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
hipError_t launch_sweep2(int ent, int ntiles, int batch, int k, hipStream_t s, const StageArgs& a) {
    const int threads = sweep2_threads(k);
    const dim3 grid(ntiles, batch), block(threads);
    const size_t lds = sweep2_lds_bytes(k, threads);
    switch (ent) {
        case 0: sweep_stage_kernel2<0><<<grid, block, lds, s>>>(a); break;
        case 1: sweep_stage_kernel2<1><<<grid, block, lds, s>>>(a); break;
        default: sweep_stage_kernel2<2><<<grid, block, lds, s>>>(a); break;
    }
    return hipGetLastError();
}

}  // namespace aqc

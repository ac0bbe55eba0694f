// Matrix-core stage kernels (gfx950): the sweep on v_mfma_f64_16x16x4_f64.
//
// Same tiling as the other two families (one workgroup = one LDS tile of 2^k amplitudes of one lane,
// every gate group of the stage applied before the tile goes back to HBM) and the same sub-stages
// (aqc_plan.cpp: split_substages, 4 "register bits" per sub-stage).  What changes is how a sub-stage is
// executed.  All its gates act on the 4 register bits, so the whole sub-stage is ONE 16 x 16 complex
// unitary U per lane (ubuild_kernel multiplies the micro-ops into it, a few microseconds for all
// sub-stages of all lanes).  The tile is a 16 x C matrix W (row = value of the register bits, column =
// "chunk" = the other k - 4 local bits) and the sub-stage is W <- U W: a complex GEMM on the matrix cores.
// Every inner product of the sub-stage is a function of R = Z W^H (16 x 16, another GEMM with the chunks
// as the summed dimension): <P_g w_g | z_g> = Tr(P_g R_g) with R_g = U_g R U_g^H, so the per-gate
// work on the 2^n-vectors disappears; rgrad_kernel walks the micro-ops over the 16 x 16 matrix instead.
//
// MFMA operand layout (tools/ubench/mfma_f64_probe.hip): lane l feeds A[l%16][l/16] and B[l/16][l%16]
// and owns D[4r + l/16][l%16], r = 0..3.  Per group of 16 chunks:
//   product 1  W'^T = W^T U^T : A = W^T (M = chunk, K = input amplitude), B = U^T, D = W'^T
//              input  ("L1") lane l, K-step s : amplitude 4s + l/16 of chunk l%16
//              output ("L2") lane l, register r: amplitude l%16 of chunk 4r + l/16
//   product 2  R += Z' W'^H   : A = Z' (M = amplitude of z, K = chunk), B = W'^H -- exactly the L2 registers,
//              so the outputs of product 1 feed product 2 without any data movement.
// Complex products use the 3-multiplication form (T1 = ar br, T2 = ai bi, T3 = (ar + ai)(br + bi)):
// 9 real 16x16x16 products per group of 16 chunks for the sweep, 3 for V / V^H; the handful of fp64 adds
// rides on the otherwise idle vector ALU.  Sums are accumulated in a fixed order (MFMA K-chains, then a
// fixed-order sum over waves and tiles): run-to-run identical, no float atomics.
#include <hip/hip_runtime.h>

#include <cstring>

#include <type_traits>

#include "aqc_device.h"
#include "aqc_launch.h"
#include "aqc_math.h"

namespace aqc {

typedef double double4_t __attribute__((ext_vector_type(4)));

// LDS slot of local index l: XOR-folds index bits 4..11 into the low nibble so that both access patterns
// of a sub-stage (16 lanes along the chunk bits, 16 lanes along the register bits) spread over the 16-byte
// bank groups.  GF(2)-linear: swz3(x ^ y) == swz3(x) ^ swz3(y), which the host tables rely on.
__host__ __device__ __forceinline__ unsigned swz3(unsigned l) { return l ^ ((l >> 4) & 15u) ^ ((l >> 8) & 15u); }

__device__ __forceinline__ double4_t mfma(double a, double b, double4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// o = U v for one group of 16 chunks, in two halves so that the callers can software-pipeline: the 12 MFMAs
// (u_mfma: v[s] = L1 operand of K-step s) and the fp64 adds that turn the three real products into the complex
// result (u_combine: o[r] = L2 result register r).  Three-multiplication form with the U-side sums precomputed by
// ubuild_kernel (planes u0 = Re U, u1 = Im U - Re U, u2 = Re U + Im U):
//   K1 = (vx + vy) u0, K2 = vx u1, K3 = vy u2;  Re = K1 - K3, Im = K1 + K2      -- 3 fp64 adds per amplitude.
// fp64 MFMA and the vector ALU do NOT overlap on gfx950 (tools/ubench/mfma_f64_shadow.hip: every v_add_f64 issued
// between MFMAs costs its full 4 cycles), so every vector instruction saved here is matrix time gained.
struct Acc3 { double4_t k1, k2, k3; };
__device__ __forceinline__ void u_mfma(const cplx (&v)[4], const double (&u0)[4], const double (&u1)[4], const double (&u2)[4], Acc3& t) {
    t.k1 = double4_t{0.0, 0.0, 0.0, 0.0}; t.k2 = t.k1; t.k3 = t.k1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        t.k1 = mfma(v[s].x + v[s].y, u0[s], t.k1);
        t.k2 = mfma(v[s].x, u1[s], t.k2);
        t.k3 = mfma(v[s].y, u2[s], t.k3);
    }
}
__device__ __forceinline__ void u_combine(const Acc3& t, cplx (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r].x = t.k1[r] - t.k3[r];
        o[r].y = t.k1[r] + t.k2[r];
    }
}

#ifdef AQC_TUNING   // in-kernel stamps (diagnostic builds only): where a workgroup's time goes
#define AQC_STAMP(slot) do { if (a.stamps && threadIdx.x == 0 && (slot) < kStampSlots) \
    a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kStampSlots + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef AQC_TUNING_NODBG   // stamps only: the work-skipping switches cost registers (the 2^12 sweep spills with them)
#define AQC_DBG_AND(x)
#define AQC_DBG_TEST(x) false
#else
#define AQC_DBG_AND(x) && (x)
#define AQC_DBG_TEST(x) (x)
#endif
#else
#define AQC_STAMP(slot) do { } while (0)
#define AQC_DBG_AND(x)
#define AQC_DBG_TEST(x) false
#endif

// sweep kernels: R scratch double-buffered (one barrier per sub-stage) where the second buffer costs no occupancy: the
// persistent 2^12 kernel (one workgroup per CU anyway) and tiles up to 2^9 (scratch of a few KiB)
__host__ __device__ constexpr bool sweep_scratch_double(int k) { return k >= 12 || k <= 9; }
#ifndef AQC_SWEEP_SPREAD   // (variant builds: make variant NAME=s2 EXTRA=-DAQC_SWEEP_SPREAD=2; sweep pair 1.012 / 1.016 / 1.026 / 1.037 ms at 1 / 2 / 3 / 4)
#define AQC_SWEEP_SPREAD 1
#endif
#ifndef AQC_APPLY_SPREAD
#define AQC_APPLY_SPREAD 1
#endif
// Attribution experiments (variant builds only, `make variant NAME=.. EXTRA=-DAQC_EXP_APPLY_SKIP=<bits>`, tools/apply_budget.sh): parts of
// the V / V^H sub-stage loop compiled OUT -- 1 LDS writes, 2 LDS reads, 8 MFMAs, 16 fp64 adds, 32 address XORs, 64 the tiles' HBM traffic, 128 the operand sums only, 256 the combining adds only
// (beware: the compiler then drops the third real product as dead code), 512 the barrier of every sub-stage.  Results are garbage by
// construction, only the launch times matter; the shipped library is built with 0.
#ifndef AQC_EXP_APPLY_SKIP
#define AQC_EXP_APPLY_SKIP 0
#endif
constexpr int kApplySkip = AQC_EXP_APPLY_SKIP;
#ifndef AQC_EXP_SWEEP_SKIP   // the same for the sweep: 64 = no HBM traffic of the w / z tiles (loads, prefetch, stores)
#define AQC_EXP_SWEEP_SKIP 0
#endif
constexpr int kSweepSkip = AQC_EXP_SWEEP_SKIP;

constexpr int kSweepSpread = AQC_SWEEP_SPREAD, kApplySpread = AQC_APPLY_SPREAD;   // MFMAs between two LDS writes inside a matrix run (sweep / V, V^H)
template <int K, bool SWEEP = false> struct TileShape {   // compile-time shape of a 2^K-amplitude tile
    // 4 waves from 2^10 amplitudes up.  (8 waves on the sweep's 2^12 tiles -- two per SIMD -- were measured: the matrix
    // work itself runs at 64 cycles per MFMA either way, and the per-wave cost of a sub-stage (operand prefetch, address
    // set-up, R hand-over, barriers) doubles: 14.2k vs 13.0k cycles per sub-stage.  That was before the prefetch moved into
    // accumulation registers: two waves per SIMD leave 256 registers per wave for the pipeline AND the prefetch, and the round-3
    // kernel no longer builds with kSweepWaves12 = 8.)
    static constexpr int kSweepWaves12 = 4;
    static constexpr int kWaves = (SWEEP && K == 12) ? kSweepWaves12 : (K >= 10 ? 4 : (1 << (K - 8)));
    static constexpr int kGroups = 1 << (K - 8);                       // groups of 16 chunks x 16 amplitudes
    static constexpr int kGpw = kGroups / kWaves;                      // groups per wave: 1, 2 or 4
    static constexpr int kLoads = (1 << (K - 6)) / kWaves;             // 16-byte HBM accesses per thread and vector
};

struct SubRegs {   // what a lane needs for one sub-stage from vector memory, fetched one sub-stage ahead (no load on the critical path)
    double u0[4], u1[4], u2[4];
    unsigned lane12;              // L1 slot of this lane (low half) and L2 slot (high half), DevSub3::lane12
};

template <int GPW>
__device__ __forceinline__ void fetch_sub(SubRegs& x, const DevSub3* subs, const double* umat, int sub_index, int lane, int wave,
                                          int nwaves) {
    const DevSub3* sub = subs + sub_index;
    const double* up = umat + (size_t)sub_index * 12 * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        x.u0[s] = up[(0 * 4 + s) * 64 + lane];
        x.u1[s] = up[(1 * 4 + s) * 64 + lane];
        x.u2[s] = up[(2 * 4 + s) * 64 + lane];
    }
    x.lane12 = sub->lane12[lane];
}

// LDS byte addresses of one sub-stage: one per-lane base for each layout and wave-uniform XOR terms kept in SGPRs
// (every access is then a single v_xor_b32 with a scalar operand).  The uniform terms come precomputed from the plan
// (DevSub3::kk) by SCALAR loads -- the plan tables are never written by a kernel, hence the constant address space --
// issued right after the previous sub-stage's matrix work, straight into the registers that work has just released.
template <int GPW>
struct SubAddr {
    unsigned a1, a2;                 // per lane: L1 / L2 byte address inside a tile
    unsigned k1[GPW][4], k2[GPW][4];  // uniform: ((step term ^ group term) << 4) for L1 (K-step s) and L2 (register r)
};
typedef unsigned uint8v_t __attribute__((ext_vector_type(8)));
template <int GPW>
__device__ __forceinline__ void fetch_k(SubAddr<GPW>& x, const DevSub3* subs, int sub_index, int wave, int nwaves) {
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        const uint8v_t kv = *reinterpret_cast<const __attribute__((address_space(4))) uint8v_t*>(
            (const __attribute__((address_space(4))) void*)(const void*)&subs[sub_index].kk[wave + j * nwaves][0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) { x.k1[j][s] = kv[s]; x.k2[j][s] = kv[4 + s]; }
    }
}
template <int GPW>
__device__ __forceinline__ void sub_addr(SubAddr<GPW>& x, const SubRegs& c, unsigned lds_base) {
    x.a1 = ((c.lane12 & 0xffffu) << 4) + lds_base;
    x.a2 = ((c.lane12 >> 16) << 4) + lds_base;
}
// LDS accesses by absolute 32-bit LDS address (the tile region starts at LDS address `lds_base`, checked once per
// kernel to be a multiple of the XOR span, so base + (a ^ k) == (base + a) ^ k and no add is left on the access path).
typedef double dbl2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) dbl2_t* lds_cplx_ptr;
__device__ __forceinline__ cplx lds_get(unsigned addr) {
    const dbl2_t v = *reinterpret_cast<lds_cplx_ptr>(addr);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void lds_put(unsigned addr, cplx v) { *reinterpret_cast<lds_cplx_ptr>(addr) = dbl2_t{v.x, v.y}; }
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)p);
}

// a wave-uniform pointer as a scalar register pair (an "s" operand of inline assembly must not end up in vector registers)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}

// HBM <-> LDS: 16 B per lane, runs of the stage's low local bits are contiguous in HBM.  l & 63 == lane for
// every element a thread touches and l >> 6 is wave-uniform; all loads of a thread are in flight together.
template <int K, int NV>
__device__ __forceinline__ void load_tiles3(cplx* t0, cplx* t1, const cplx* s0, const cplx* s1, const DevStage& st, unsigned lo, int wave) {
    constexpr int NL = TileShape<K, NV == 2>::kLoads, NW = TileShape<K, NV == 2>::kWaves;
    cplx v0[NL], v1[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const unsigned off = lo + st.dhi[wave + i * NW];
        v0[i] = s0[off];
        if (NV == 2) v1[i] = s1[off];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const unsigned slot = swz3((unsigned)((wave + i * NW) << 6) | (threadIdx.x & 63u));
        t0[slot] = v0[i];
        if (NV == 2) t1[slot] = v1[i];
    }
}
template <int K, bool SWEEP>
__device__ __forceinline__ void store_tile3(const cplx* tile, cplx* dst, const DevStage& st, unsigned lo, int wave) {
    constexpr int NL = TileShape<K, SWEEP>::kLoads, NW = TileShape<K, SWEEP>::kWaves;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const unsigned slot = swz3((unsigned)((wave + i * NW) << 6) | (threadIdx.x & 63u));
        dst[lo + st.dhi[wave + i * NW]] = tile[slot];
    }
}
__device__ __forceinline__ size_t tile_base3(const DevStage& st, unsigned tile) {
    size_t base = 0;
    for (int i0 = 0; i0 < st.nub; i0 += 4) {   // four bit positions per scalar load (ubits[32], nub <= 32): one load latency per four
        int ub[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ub[u] = st.ubits[i0 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u < st.nub) base |= (size_t)((tile >> (i0 + u)) & 1u) << ub[u];
    }
    return base;
}

// ---- V / V^H -------------------------------------------------------------------------------------------------
// 2^12 tiles (AQC_APPLY_PERSIST, the default there): a persistent grid of two workgroups per CU, each walking over its
// (tile, lane) items with the NEXT item's tile prefetched into registers behind sub-stages 0..3 -- the scheme of the sweep.
// In-kernel stamps of the one-tile-per-workgroup form showed the sub-stage loop at the matrix pipe's rate, and a fifth to
// a third of a workgroup's life in the tile load before and the store after it (every workgroup of a wave of tiles loads at
// the same moment: an HBM burst with idle matrix cores).
// LIST: the work items come from a device table (Stage3Args::items / nitems: a subset of the (tile, lane) pairs, see the
// sparse-lhs sweep in aqc_ws_sweep.cpp) instead of being all of them; the grid is sized for the largest possible list and
// workgroups beyond its length leave at once.
__device__ __forceinline__ int uniform_load(const int* p) { return __builtin_amdgcn_readfirstlane(*p); }
struct ItemAt { int bl, tile, slot; };
template <bool LIST>
__device__ __forceinline__ ItemAt item_at(const Stage3Args& a, int wi) {
    if (LIST) {
        const TileItem* it = a.items + wi;
        return ItemAt{uniform_load(&it->lane), uniform_load(&it->tile), uniform_load(&it->slot)};
    }
    const int bl = wi / a.ntiles;
    return ItemAt{bl, wi - bl * a.ntiles, 0};
}
template <int K, bool LIST = false>
__global__ __launch_bounds__(TileShape<K>::kWaves * 64, 2) void apply_mfma_kernel(const Stage3Args a) {
    using TS = TileShape<K>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage& st = a.stage;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const unsigned lds_base = lds_address(smem);
    if (lds_base & ((16u << K) - 1)) __builtin_trap();   // XOR addressing needs the tile aligned to its own size
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NL = TS::kLoads, NW = TS::kWaves;
    constexpr bool kPersist = K >= 12;      // grid < items only there (launch_apply3)
    const int nwork = LIST ? uniform_load(a.nitems) : a.ntiles * a.batch;   // items = (tile, lane of the batch), item wi on workgroup wi mod gridDim.x
    const unsigned lo = st.dlo[lane];
    const unsigned lo16 = lo << 4;
    SubRegs cur, nxt;
    SubAddr<TS::kGpw> ad;   // clustered software pipeline, see sweep_mfma_kernel
    dbl2_t pw[NL];          // the prefetched tile: accumulation registers, inline-assembly loads (see sweep_mfma_kernel)
    int wi = blockIdx.x + ((kPersist || LIST) ? 0 : (int)blockIdx.y * a.ntiles);
    if (LIST && wi >= nwork) return;   // (a whole workgroup, before any barrier)
    ItemAt it = item_at<LIST>(a, wi);
    {
        const int bl = it.bl;
        if (st.nsubs > 0) {
            fetch_sub<TS::kGpw>(cur, a.subs, a.umat + (size_t)bl * a.nsubs_total * 12 * 64, st.sub_begin, lane, wave, NW);
            fetch_k<TS::kGpw>(ad, a.subs, st.sub_begin, wave, NW);
        }
        AQC_STAMP(0);
        if (!(kApplySkip & 64)) load_tiles3<K, 1>(tw, nullptr, a.in0 + (size_t)bl * a.lane_stride + tile_base3(st, it.tile), nullptr, st, lo, wave);
    }
    // Every load so far has landed before the loop: otherwise the compiler's wait-count analysis, merging the loop
    // entry with the back edge, makes the first use of `cur` inside the loop wait for the prefetch of `nxt` as well.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    AQC_STAMP(1);
    for (;;) {
    const int bl = it.bl;
    const size_t lane_off = (size_t)bl * a.lane_stride + tile_base3(st, it.tile);
    const double* umat = a.umat + (size_t)bl * a.nsubs_total * 12 * 64;
    const int nwi = wi + (int)gridDim.x;
    const bool more = kPersist && st.nsubs > 0 && nwi < nwork;
    const ItemAt nit = more ? item_at<LIST>(a, nwi) : ItemAt{0, 0, 0};
    const int nbl = nit.bl;
    const size_t next_off = more ? (size_t)nbl * a.lane_stride + tile_base3(st, nit.tile) : 0;
    for (int si = 0; si < st.nsubs; ++si) {
        AQC_STAMP(4 + si);
        if (!(kApplySkip & 512)) __syncthreads();
        if (si + 1 < st.nsubs) fetch_sub<TS::kGpw>(nxt, a.subs, umat, st.sub_begin + si + 1, lane, wave, NW);
        else if (more) fetch_sub<TS::kGpw>(nxt, a.subs, a.umat + (size_t)nbl * a.nsubs_total * 12 * 64, st.sub_begin, lane, wave, NW);
        if (kPersist && more && !(kApplySkip & 64)) {   // a quarter of the next item's tile, issued BEHIND the operand fetch (loads retire in order)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (si == (c < st.nsubs ? c : st.nsubs - 1)) {
#pragma unroll
                    for (int i = c * (NL / 4); i < (c + 1) * (NL / 4); ++i) {
                        const size_t ub = next_off + st.dhi[wave + i * NW];
                        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=a"(pw[i]) : "v"(lo16), "s"(uniform_ptr(a.in0 + ub)) : "memory");
                    }
                }
            }
        }
        sub_addr(ad, cur, lds_base);
        cplx v[2][4];
        Acc3 acc;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[0][s] = lds_get(ad.a1 ^ ad.k1[0][s]);
#pragma unroll
        for (int j = 0; j <= TS::kGpw; ++j) {
            if (j + 1 < TS::kGpw && !(kApplySkip & 2)) {
#pragma unroll
                for (int s = 0; s < 4; ++s) v[(j + 1) & 1][s] = lds_get((kApplySkip & 32) ? ad.a1 : (ad.a1 ^ ad.k1[j + 1][s]));
            }
            __builtin_amdgcn_sched_barrier(0);
            double sv[4];
            cplx o[4];
            if (j < TS::kGpw) {
#pragma unroll
                for (int s = 0; s < 4; ++s) sv[s] = (kApplySkip & (16 | 128)) ? v[j & 1][s].x : v[j & 1][s].x + v[j & 1][s].y;
            }
            if (j > 0) {
                if (kApplySkip & (16 | 256)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = make_double2(acc.k1[r], acc.k2[r]);
                } else u_combine(acc, o);
            }
            unsigned wa[4];   // LDS write addresses of group j - 1, taken here in the vector-ALU bunch
            if (j < TS::kGpw) {   // (pinned: IR-level sinking otherwise moves the sums in between the MFMAs below)
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(sv[s]));
            }
            if (j > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { wa[r] = (kApplySkip & 32) ? ad.a2 : (ad.a2 ^ ad.k2[j - 1][r]); asm volatile("" : "+v"(wa[r])); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j < TS::kGpw) {
                acc.k1 = double4_t{0.0, 0.0, 0.0, 0.0}; acc.k2 = acc.k1; acc.k3 = acc.k1;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (kApplySkip & 8) continue;
                    acc.k1 = mfma(sv[s], cur.u0[s], acc.k1);
                    acc.k2 = mfma(v[j & 1][s].x, cur.u1[s], acc.k2);
                    acc.k3 = mfma(v[j & 1][s].y, cur.u2[s], acc.k3);
                }
            }
            if (j > 0 && !(kApplySkip & 1)) {   // the LDS writes of group j - 1 ride inside the MFMA run (see sweep_mfma_kernel)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds_put(wa[r], o[r]);
                if (j < TS::kGpw) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, kApplySpread, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (si + 1 < st.nsubs || more) {
            cur = nxt;
            fetch_k<TS::kGpw>(ad, a.subs, si + 1 < st.nsubs ? st.sub_begin + si + 1 : st.sub_begin, wave, NW);
        }
    }
    AQC_STAMP(4 + st.nsubs);
    __syncthreads();
    AQC_STAMP(2);
    if (!(kApplySkip & 64)) store_tile3<K, false>(tw, a.out0 + lane_off, st, lo, wave);
    AQC_STAMP(3);
    if (!more) break;
    __syncthreads();   // every wave has finished with the LDS tile (the stores' LDS reads included)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the prefetched tile has arrived
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const unsigned slot = swz3((unsigned)((wave + i * NW) << 6) | (threadIdx.x & 63u));
        asm volatile("ds_write_b128 %0, %1" : : "v"(lds_base + (slot << 4)), "a"(pw[i]) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the LDS writes above are not tracked by the compiler
    wi = nwi;
    it = nit;
    }
}

// ---- forward w / z sweep with R = Z W^H per sub-stage ------------------------------------------------------
// One barrier per sub-stage: the per-wave R of sub-stage s goes to an LDS scratch before it, the fixed-order sum over
// the waves (64 entries per wave) right after it.  The scratch is DOUBLE-BUFFERED by sub-stage parity: the buffer of
// sub-stage s is written again in sub-stage s + 2, and every wave has finished its reads of it before it arrives at
// the barrier of sub-stage s + 1, which every writer has passed -- no second barrier, no counter.  2^10 and 2^11 tiles
// are the exception (sweep_scratch_double): a second buffer would cost them a resident workgroup per CU (2^11: two
// workgroups share a CU only if each stays within exactly 80 KiB), so there a second barrier per sub-stage takes its place.
// vmcnt(0) for loads issued by inline assembly into accumulation registers: the "+a" operands make every later read of the
// registers depend on the wait (the compiler does not know that the loads are outstanding)
template <int N>
__device__ __forceinline__ void wait_prefetched(dbl2_t (&pw)[N], dbl2_t (&pz)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+a"(pw[i]), "+a"(pz[i]));
}
__device__ __forceinline__ unsigned tile_offset3(const DevStage& st, unsigned local) { return st.dlo[local & 63u] | st.dhi[local >> 6]; }
template <int K> struct SweepShape : TileShape<K, true> {};   // (no comma inside the __launch_bounds__ macro arguments)
// MEASURED SLOWER, hence opt-in (AQC_SKIP_ZERO_W=1): the branches cut the 36-MFMA runs of a group iteration into three and keep the
// compiler from giving the first sub-stage of an item its own copy of the loop; at the headline the last sweep stage takes 1.54 ms
// with the skips (19 % fewer MFMAs) against 1.46 ms without, 12-qubit 2-layer Trotter 0.164 against 0.161 ms, 12 layers 0.81 against
// 0.70 (gpurun_out/r5g).  Kept because it is exact (bit-identical gradients, tests/test_hip_round5.py) and states what a faster
// formulation has to beat: a static z pass over all groups plus a run-time loop over the non-zero groups only.
// SKIPW (sweep from basis states, Stage3Args::supp): per item and sub-stage two wave-uniform masks say which of the wave's groups
// can hold a non-zero w at all and which K-steps of the W product can (see skip_masks); the W product of the other groups / K-steps
// and the R product of the other groups are not issued -- they would multiply exact zeros.
__device__ __forceinline__ unsigned elem_bit(const DevStage& st, long long e, unsigned local_pos) {   // local bit `local_pos` of element e
    const unsigned m = local_pos < 6 ? st.dlo[1u << local_pos] : st.dhi[1u << (local_pos - 6)];
    return ((unsigned long long)e & m) ? 1u : 0u;
}
template <int GPW, int NW>
__device__ __forceinline__ void skip_masks(const DevStage& st, unsigned info, long long e0, long long e1, int wave, unsigned& nzmask, unsigned& kneed) {
    nzmask = 0; kneed = 0;
    const unsigned gfresh = info & 15u, kfresh = (info >> 6) & 3u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const long long e = k ? e1 : e0;
        if (e < 0) continue;
        unsigned gb = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (gfresh >> i & 1) gb |= elem_bit(st, e, (info >> (8 + 4 * i)) & 15u) << i;
        unsigned kb = 0;
        if (kfresh & 1) kb |= elem_bit(st, e, (info >> 24) & 15u);
        if (kfresh & 2) kb |= elem_bit(st, e, (info >> 28) & 15u) << 1;
#pragma unroll
        for (int j = 0; j < GPW; ++j)
            if ((((unsigned)(wave + j * NW) ^ gb) & gfresh) == 0) nzmask |= 1u << j;
#pragma unroll
        for (unsigned s2 = 0; s2 < 4; ++s2)
            if (((s2 ^ kb) & kfresh) == 0) kneed |= 1u << s2;
    }
}
// RLAST: the launch of the LAST stage when its last sub-stage is taken from its inputs alone (Stage3Args::r_only_last); a variant of
// its own because the explicit copies it needs cost the plain loop 3 % (see `substage` below).
template <int K, bool LIST = false, bool SKIPW = false, bool RLAST = false>
__global__ __launch_bounds__(SweepShape<K>::kWaves * 64, K >= 12 ? 1 : 2) void sweep_mfma_kernel(const Stage3Args a) {
    using TS = TileShape<K, true>;
    constexpr int kSlots = TS::kWaves > 4 ? 4 : TS::kWaves;   // scratch slots; 8 waves reduce in pairs first
    constexpr unsigned tsize = 1u << K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevStage& st = a.stage;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    cplx* tz = tw + tsize;
    cplx* scratch = tz + tsize;   // [2 if kDouble][kSlots][4][64]: per-wave R of the current sub-stage
    constexpr bool kDouble = sweep_scratch_double(K);
    const unsigned lds_base = lds_address(smem);
    if (lds_base & ((32u << K) - 1)) __builtin_trap();   // XOR addressing needs the two tiles aligned to their joint size
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Work items = (tile, lane of the batch), item wi on workgroup wi mod gridDim.x.  2^12 tiles leave room for ONE
    // workgroup per CU, so nothing else would hide its HBM traffic: there the grid is one workgroup per CU, each walks
    // over its items and fetches the NEXT item's w and z tiles into registers (2 x 16 x 16 B per lane, in four chunks
    // behind the operand fetch of sub-stages 1..4, in the operand layout of the item's FIRST sub-stage) while the current
    // one computes; the stores of a finished tile drain under the next one as well.  Smaller tiles run two workgroups per
    // CU, which overlap by themselves (grid = items).
    constexpr bool kPersist = K >= 12;
    constexpr int NL = TS::kLoads, NW = TS::kWaves;
    const int nwork = LIST ? uniform_load(a.nitems) : a.ntiles * a.batch;
    const unsigned lo = st.dlo[lane];
    SubRegs cur, nxt;
    SubAddr<TS::kGpw> ad;
    // The prefetched tiles live in ACCUMULATION registers (the sub-stage pipeline fills all 256 architectural VGPRs; left
    // to itself the allocator spills a prefetch array to scratch memory): global_load with an AGPR destination as inline
    // assembly; the compiler does not track these loads, so the wait before their first use is explicit (wait_prefetched).
    dbl2_t pw[NL], pz[NL];
    unsigned parity = 0;   // scratch buffer of the running sub-stage (kDouble)
    AQC_STAMP(0);
    // Persistent form: workgroup g walks over the CONTIGUOUS items [g chunk, (g + 1) chunk) in lane-major order, i.e. over
    // consecutive tiles of one lane (a.chunk > 0, launch_sweep3).  The R of a sub-stage is a sum over the tiles of a lane, so
    // the workgroup accumulates it over its run of tiles of that lane (a "segment") instead of leaving one partial per
    // tile: the per-(lane, sub-stage) partials in HBM shrink from `ntiles` to at most ntiles / chunk + 2 -- 16x less for the
    // gradient walk to read at the headline, 40x less memory at 20 qubits -- by a read-modify-write of the segment's own 4 KB
    // that stays in L2: the old value is requested at the top of the sub-stage (accumulation register, L1 bypassed: the
    // line was written by this workgroup one item earlier) and added where the sum over the waves is formed.
    // LIST (items from a device table, see apply_mfma_kernel): every item writes the partial slot the table names and nothing
    // is accumulated over a segment; the chunk follows from the length of the list.
    const int chunk = kPersist ? (LIST ? (nwork + (int)gridDim.x - 1) / (int)gridDim.x : a.chunk) : 0;
    const int wi_first = kPersist ? (int)blockIdx.x * chunk : (int)blockIdx.x;
    const int wi_end = kPersist ? (wi_first + chunk < nwork ? wi_first + chunk : nwork) : wi_first + 1;
    int wi = wi_first;
    if (wi >= nwork) return;   // (a whole workgroup: no barrier has been reached yet)
    const unsigned e16 = threadIdx.x << 4;
    dbl2_t racc;               // this thread's entry of the segment's accumulated R (kPersist; 256 threads = 256 entries)
    // Persistent form: the prefetched registers ARE the operands of an item's first sub-stage -- the loads fetch, for every
    // lane, exactly the amplitudes that sub-stage's MFMAs take from it (L1 layout: chunk l % 16, amplitude 4 s + l / 16 of
    // the wave's groups), so a tile never passes through LDS on its way in.  (Moving a prefetched tile from accumulation
    // registers to LDS between two items cost 4.2-4.7k cycles per item with nothing else running on the CU.)
    static_assert(!kPersist || NL == 4 * TS::kGpw, "the prefetch registers are the first sub-stage's operands");
    const bool reg_first = kPersist && st.nsubs > 0;
    unsigned flo16 = 0;        // first sub-stage: byte offset of this lane's operand position inside a tile (+ a.first_hi[group][K-step])
    // element offset of the running item's tile; the next item's is worked out once (a chain of dependent scalar loads)
    // and handed on
    ItemAt it = item_at<LIST>(a, wi);
    size_t item_off = (size_t)it.bl * a.lane_stride + tile_base3(st, it.tile);
    {
        const size_t off0 = item_off;
        if (st.nsubs > 0) {
            fetch_sub<TS::kGpw>(cur, a.subs, a.umat + (size_t)it.bl * a.nsubs_total * 12 * 64, st.sub_begin, lane, wave, NW);
            fetch_k<TS::kGpw>(ad, a.subs, st.sub_begin, wave, NW);
        }
        if (reg_first) {
            // slot tables hold swz3(local index); swz3 is an involution and GF(2)-linear, lane part and uniform part use
            // disjoint bits of the local index, so their tile offsets simply add
            flo16 = tile_offset3(st, swz3(cur.lane12 & 0xffffu)) << 4;
#pragma unroll
            for (int i = 0; i < ((kSweepSkip & 64) ? 0 : NL); ++i) {
                const size_t ub = off0 + a.first_hi[wave + (i / 4) * NW][i % 4];
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=a"(pw[i]) : "v"(flo16), "s"(uniform_ptr(a.in0 + ub)) : "memory");
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=a"(pz[i]) : "v"(flo16), "s"(uniform_ptr(a.in1 + ub)) : "memory");
            }
        } else {
            load_tiles3<K, 2>(tw, tz, a.in0 + off0, a.in1 + off0, st, lo, wave);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see apply_mfma_kernel
    if (reg_first) wait_prefetched<NL>(pw, pz);
    __syncthreads();
    AQC_STAMP(1);
    for (;;) {
    const int bl = it.bl, tile = it.tile;
    const size_t lane_off = item_off;
    long long se0 = -1, se1 = -1;   // SKIPW: the lane's basis elements that can reach this tile (fresh non-local bits agree with the tile's)
    if (SKIPW) {
        const unsigned long long toff = (unsigned long long)tile_base3(st, tile);
        const long long c0 = a.supp[2 * (size_t)bl], c1 = a.supp[2 * (size_t)bl + 1];
        const unsigned lo0 = __builtin_amdgcn_readfirstlane((unsigned)c0), hi0 = __builtin_amdgcn_readfirstlane((unsigned)(c0 >> 32));
        const unsigned lo1 = __builtin_amdgcn_readfirstlane((unsigned)c1), hi1 = __builtin_amdgcn_readfirstlane((unsigned)(c1 >> 32));
        const long long u0 = (long long)(((unsigned long long)hi0 << 32) | lo0), u1 = (long long)(((unsigned long long)hi1 << 32) | lo1);
        if (u0 >= 0 && (((unsigned long long)u0 ^ toff) & st.fresh_nonlocal) == 0) se0 = u0;
        if (u1 >= 0 && (((unsigned long long)u1 ^ toff) & st.fresh_nonlocal) == 0) se1 = u1;
    }
    const double* umat = a.umat + (size_t)bl * a.nsubs_total * 12 * 64;
    // slot of this item's partial: its tile (one partial per tile), or its segment = number of workgroups that hold earlier
    // tiles of the lane
    const int part = LIST ? it.slot : (kPersist ? (int)blockIdx.x - (bl * a.ntiles) / chunk : tile);
    const bool seg_first = LIST || !kPersist || wi == wi_first || tile == 0;   // nothing accumulated yet in this segment
    cplx* rpart = a.rpart + (((size_t)bl * a.nsubs_total + st.sub_begin) * a.nparts + part) * 256;
    const int nwi = wi + 1;
    const bool more = reg_first && nwi < wi_end;   // (launch_sweep3 refuses a persistent stage without sub-stages)
    const ItemAt nit = more ? item_at<LIST>(a, nwi) : ItemAt{0, 0, 0};
    const int nbl = nit.bl;
    const size_t next_off = more ? (size_t)nbl * a.lane_stride + tile_base3(st, nit.tile) : 0;
    constexpr unsigned ZOFF = tsize * 16;   // byte offset of the z tile (a power of two above every tile address)
    cplx vw[2][4], vz[2][4];   // operands of group j (slot j & 1) and j + 1; group 0 of a sub-stage is requested right after the
                               // barrier that ends the previous one, ahead of the R reduction (its latency hides there)
    constexpr bool kEarly = K >= 11;   // (smaller tiles: the ten extra live registers would cost a resident wave per SIMD)
    if (reg_first) {
        sub_addr(ad, cur, lds_base);
#pragma unroll
        for (int s = 0; s < 4; ++s) { vw[0][s] = make_double2(pw[s].x, pw[s].y); vz[0][s] = make_double2(pz[s].x, pz[s].y); }
    } else if (kEarly && st.nsubs > 0) {
        sub_addr(ad, cur, lds_base);
#pragma unroll
        for (int s = 0; s < 4; ++s) { vw[0][s] = lds_get(ad.a1 ^ ad.k1[0][s]); vz[0][s] = lds_get((ad.a1 | ZOFF) ^ ad.k1[0][s]); }
    }
    auto prefetch = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ((kSweepSkip & 64) ? 0 : NL); ++i)
            if (i / (NL / 4) == c) {   // scalar base + 32-bit lane offset: no vector address arithmetic
                const size_t ub = next_off + a.first_hi[wave + (i / 4) * NW][i % 4];
                // (s_nop: the hazard recogniser does not look inside inline assembly -- a scalar base that was written by
                // v_readfirstlane needs 5 wait states before a vector-memory instruction may read it)
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=a"(pw[i]) : "v"(flo16), "s"(uniform_ptr(a.in0 + ub)) : "memory");
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=a"(pz[i]) : "v"(flo16), "s"(uniform_ptr(a.in1 + ub)) : "memory");
            }
    };
    // The sub-stage body lives in aqc_sweep_substage.inc (see there).  RLAST: the very last sub-stage is taken from its inputs alone
    // (R = U (Z W^H) U^H), which needs explicit copies: first sub-stage of an item / the others / the R-only one.
    const int nmain = RLAST && a.r_only_last && st.nsubs >= 1 && !(kPersist && st.nsubs == 1) ? st.nsubs - 1 : st.nsubs;
    if (!RLAST) {
        for (int si = 0; si < st.nsubs; ++si) {
#define AQC_SS_FROM_REGS (reg_first && si == 0)
#define AQC_SS_RONLY false
#include "aqc_sweep_substage.inc"
#undef AQC_SS_FROM_REGS
#undef AQC_SS_RONLY
        }
    } else {
        if (nmain > 0) {
            const int si = 0;
#define AQC_SS_FROM_REGS reg_first
#define AQC_SS_RONLY false
#include "aqc_sweep_substage.inc"
#undef AQC_SS_FROM_REGS
#undef AQC_SS_RONLY
        }
        for (int si = 1; si < nmain; ++si) {
#define AQC_SS_FROM_REGS false
#define AQC_SS_RONLY false
#include "aqc_sweep_substage.inc"
#undef AQC_SS_FROM_REGS
#undef AQC_SS_RONLY
        }
        if (nmain < st.nsubs) {
            const int si = nmain;
#define AQC_SS_FROM_REGS false
#define AQC_SS_RONLY true
#include "aqc_sweep_substage.inc"
#undef AQC_SS_FROM_REGS
#undef AQC_SS_RONLY
        }
    }
    AQC_STAMP(kStampSlots - 2);
    // The next item's operands are waited for BEFORE this item's stores are issued: vmcnt counts stores as well, and a wait
    // placed after them would sit out their whole write latency; the loads went out sub-stages ago.  (The loads are inline
    // assembly: the register operands tie every later read of the prefetched values to this wait.)
    if (more) wait_prefetched<NL>(pw, pz);
    if (!(kSweepSkip & 64)) {   // store_out: bit 0 w, bit 1 z.  The last stage's w and z are never read again (only the gradient
                                // entries are results); a stage whose successor takes z from a checkpoint of V^H stores w alone
        if (a.store_out & 1) store_tile3<K, true>(tw, a.out0 + lane_off, st, lo, wave);
        if (a.store_out & 2) store_tile3<K, true>(tz, a.out1 + lane_off, st, lo, wave);
    }
    AQC_STAMP(kStampSlots - 1);
    if (!more) break;
    AQC_STAMP(kStampSlots - 6);
    __syncthreads();   // every wave has finished with the LDS tiles (sub-stage reads, scratch, the stores' LDS reads) before the
                       // next item's first sub-stage writes its results there; the stores drain under that sub-stage
    AQC_STAMP(kStampSlots - 5);
    wi = nwi;
    it = nit;
    item_off = next_off;
    }
    AQC_STAMP(kStampSlots - 4);
}

// ---- small kernels: U of every sub-stage, gradient entries from R -----------------------------------------
// Both work on ONE sub-stage's 16 x 16 complex matrix in LDS with one wave, gate GROUP by gate group (not gate by
// gate: the serial chain is what these kernels cost).  A group is (C (x) T) ENT on register bits (pc, pt): C, T are
// 2 x 2 products of the group's rotations (Trotter Rz(-+pi/2) folded in -- Rz on the control commutes with every
// entangler), built by one lane per group straight from the thetas (no coefficient kernel on this path).
constexpr int kGrpChunk = 16;   // groups decoded per batch: keeps the LDS footprint of a workgroup (one wave) near 12 KB, so
                                // that a CU holds a dozen of them instead of four
struct Gm {            // one gate group, decoded
    cplx c[4], t[4];   // C', T' row-major
    double ec, es;     // CP phase e^{i theta4}
    double rc[4], rs[4];   // (cos, sin) of the half angles of the group's rotations (gradient walk)
    int type, pc, pt, flags, slot0, jblock, o0, o1;   // o0 < o1: the two register bits the group does not touch
};
__device__ __forceinline__ void m2_mul(const cplx (&a)[4], const cplx (&b)[4], cplx (&o)[4]) {   // o = a b
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const cplx x = a[2 * i], y = b[j], u = a[2 * i + 1], v = b[2 + j];
            o[2 * i + j] = make_double2(x.x * y.x - x.y * y.y + u.x * v.x - u.y * v.y, x.x * y.y + x.y * y.x + u.x * v.y + u.y * v.x);
        }
}
__device__ __forceinline__ void m2_ry(double c, double s, cplx (&o)[4]) {
    o[0] = make_double2(c, 0.0); o[1] = make_double2(-s, 0.0); o[2] = make_double2(s, 0.0); o[3] = make_double2(c, 0.0);
}
__device__ __forceinline__ void m2_rz(double c, double s, cplx (&o)[4]) {
    o[0] = make_double2(c, -s); o[1] = make_double2(0.0, 0.0); o[2] = make_double2(0.0, 0.0); o[3] = make_double2(c, s);
}
__device__ __forceinline__ void m2_rx(double c, double s, cplx (&o)[4]) {
    o[0] = make_double2(c, 0.0); o[1] = make_double2(0.0, -s); o[2] = make_double2(0.0, -s); o[3] = make_double2(c, 0.0);
}
// (a0, a1) <- M (a0, a1);  DAG: M^H, TR: M^T
template <bool DAG, bool TR>
__device__ __forceinline__ void m2_apply(const cplx (&m)[4], cplx& a0, cplx& a1) {
    const cplx m00 = m[0], m01 = (DAG || TR) ? m[2] : m[1], m10 = (DAG || TR) ? m[1] : m[2], m11 = m[3];
    const double sg = DAG ? -1.0 : 1.0;   // conjugate the entries
    const cplx x = a0, y = a1;
    a0 = make_double2(m00.x * x.x - sg * m00.y * x.y + m01.x * y.x - sg * m01.y * y.y, m00.x * x.y + sg * m00.y * x.x + m01.x * y.y + sg * m01.y * y.x);
    a1 = make_double2(m10.x * x.x - sg * m10.y * x.y + m11.x * y.x - sg * m11.y * y.y, m10.x * x.y + sg * m10.y * x.x + m11.x * y.y + sg * m11.y * y.x);
}
// decode the groups [begin, begin + count) of a sub-stage (count <= kGrpChunk = 16).  The half-angle sincos of the up to four
// rotations of a group -- the long part -- are taken by four lanes per group in parallel (lane = 4 group + rotation), the
// 2 x 2 products by one lane per group afterwards.  Called by ONE whole wave (its LDS instructions execute in order, so
// the hand-over between the two phases needs no workgroup barrier).
__device__ __forceinline__ void stage_groups(Gm* gm, const DevGrp* grps, int begin, int count, const double* th, int ent, int lane) {
    {
        const int gi = lane >> 2, k = lane & 3;
        if (gi < count) {
            const DevGrp d = grps[begin + gi];
            double c = 1.0, s = 0.0;
            if (k < (d.type == 0 ? 3 : 4)) sincos(0.5 * th[d.theta0 + k], &s, &c);
            gm[gi].rc[k] = c; gm[gi].rs[k] = s;
            if (k == 0) {
                double ec = 1.0, es = 0.0;
                if (ent == 2 && d.type != 0) sincos(th[d.theta0 + 4], &es, &ec);
                gm[gi].ec = ec; gm[gi].es = es;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane >= count) return;
    const DevGrp d = grps[begin + lane];
    Gm g;
    g.type = d.type; g.pc = d.pc; g.pt = d.pt; g.flags = d.flags; g.slot0 = d.slot0; g.jblock = d.jblock;
    const int mask = 0xF & ~(1 << d.pc) & ~(1 << d.pt);
    g.o0 = __ffs(mask) - 1; g.o1 = __ffs(mask & (mask - 1)) - 1;
    g.ec = gm[lane].ec; g.es = gm[lane].es;
#pragma unroll
    for (int k = 0; k < 4; ++k) { g.rc[k] = gm[lane].rc[k]; g.rs[k] = gm[lane].rs[k]; }
    cplx a[4], b[4], m[4];
    if (d.type == 0) {   // Rz(t0) Ry(t1) Rz(t2), rightmost first (core_operations.py:671-677)
        m2_rz(g.rc[0], g.rs[0], a); m2_ry(g.rc[1], g.rs[1], b); m2_mul(a, b, m);
        m2_rz(g.rc[2], g.rs[2], b); m2_mul(m, b, g.c);
        g.t[0] = make_double2(1.0, 0.0); g.t[1] = make_double2(0.0, 0.0); g.t[2] = g.t[1]; g.t[3] = g.t[0];
    } else {             // control Rz(t1) Ry(t0), target Rs(t3) Ry(t2) (core_operations.py:686-705)
        m2_rz(g.rc[1], g.rs[1], a); m2_ry(g.rc[0], g.rs[0], b); m2_mul(a, b, g.c);
        if (d.flags & 1) { m2_rz(kR, -kR, b); m2_mul(g.c, b, m); g.c[0] = m[0]; g.c[1] = m[1]; g.c[2] = m[2]; g.c[3] = m[3]; }
        if (ent == 0) m2_rx(g.rc[3], g.rs[3], a); else m2_rz(g.rc[3], g.rs[3], a);
        m2_ry(g.rc[2], g.rs[2], b); m2_mul(a, b, g.t);
        if (d.flags & 2) { m2_rz(kR, kR, a); m2_mul(a, g.t, m); g.t[0] = m[0]; g.t[1] = m[1]; g.t[2] = m[2]; g.t[3] = m[3]; }
    }
    gm[lane] = g;
}
// index inside the 16 register-bit values: local pair value l = 2 cb + tb on (pc, pt), `rest` on the other two bits
__device__ __forceinline__ int grp_index(const Gm& g, int l, int rest) {
    return ((l >> 1) << g.pc) | ((l & 1) << g.pt) | ((rest & 1) << g.o0) | ((rest >> 1) << g.o1);
}
// x <- B x (or B^H x, or B^T x) for B = (C (x) T) ENT on the four values x[2 cb + tb]
template <int MODE>   // 0: B, 1: B^H, 2: B^T
__device__ __forceinline__ void grp_apply(const Gm& g, int ent, cplx (&x)[4]) {
    auto entangle = [&](double es) {
        if (g.type == 0) return;
        if (ent == 0) { const cplx t = x[2]; x[2] = x[3]; x[3] = t; }
        else if (ent == 1) x[3] = make_double2(-x[3].x, -x[3].y);
        else x[3] = cmul(x[3], g.ec, es);
    };
    if (MODE == 0) {
        entangle(g.es);
        m2_apply<false, false>(g.t, x[0], x[1]); m2_apply<false, false>(g.t, x[2], x[3]);
        m2_apply<false, false>(g.c, x[0], x[2]); m2_apply<false, false>(g.c, x[1], x[3]);
    } else {
        m2_apply<MODE == 1, MODE == 2>(g.c, x[0], x[2]); m2_apply<MODE == 1, MODE == 2>(g.c, x[1], x[3]);
        m2_apply<MODE == 1, MODE == 2>(g.t, x[0], x[1]); m2_apply<MODE == 1, MODE == 2>(g.t, x[2], x[3]);
        entangle(MODE == 1 ? -g.es : g.es);
    }
}

// One unitary per (job, lane): U = product of the sub-stage's gate groups, written as MFMA B operands
// umat[lane][sub][plane (re, im - re, re + im)][K-step s][l] = U[l % 16][4 s + l / 16].  Jobs cover the sub-stages of
// several plans (V^H and the sweep are built by one launch).
// `thetas` may be pinned HOST memory (the one-call evaluation path hands the parameters over without a copy node);
// `thetas_copy` (may be null) then receives them in HBM for the kernels that follow: job 0 of every lane copies its lane's.
__global__ __launch_bounds__(64) void ubuild_kernel(const UJob* jobs, const double* thetas, int T, double* thetas_copy) {
    __shared__ cplx u[256];   // [row = output amplitude][col = input amplitude]
    __shared__ Gm gm[kGrpChunk];
    const int lane = threadIdx.x, b = blockIdx.y;
    const UJob job = jobs[blockIdx.x];
    const DevSub3 sub = *job.sub;
    const double* th = thetas + (size_t)b * T;
    if (thetas_copy && blockIdx.x == 0)
        for (int i = lane; i < T; i += 64) thetas_copy[(size_t)b * T + i] = th[i];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int e = lane + 64 * m;
        u[e] = make_double2((e >> 4) == (e & 15) ? 1.0 : 0.0, 0.0);
    }
    const int col = lane & 15, rest = lane >> 4;
    for (int base = 0; base < sub.ngrp; base += kGrpChunk) {
        const int count = min(kGrpChunk, sub.ngrp - base);
        __syncthreads();
        stage_groups(gm, job.grps, sub.grp_begin + base, count, th, job.entangler, lane);
        __syncthreads();
        for (int i = 0; i < count; ++i) {   // U <- B U (or B^H U for V^H plans): 4 rows x 16 columns per (col, rest) lane
            const Gm& g = gm[i];
            cplx x[4];
            int idx[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) { idx[l] = grp_index(g, l, rest) * 16 + col; x[l] = u[idx[l]]; }
            if (job.inverse) grp_apply<1>(g, job.entangler, x); else grp_apply<0>(g, job.entangler, x);
#pragma unroll
            for (int l = 0; l < 4; ++l) u[idx[l]] = x[l];
            __syncthreads();
        }
    }
    __syncthreads();
    double* out = job.umat + ((size_t)b * job.nsubs + job.index) * 12 * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const cplx v = u[(lane & 15) * 16 + 4 * s + (lane >> 4)];
        out[(0 * 4 + s) * 64 + lane] = v.x;         // u0 = Re U
        out[(1 * 4 + s) * 64 + lane] = v.y - v.x;   // u1 = Im U - Re U
        out[(2 * 4 + s) * 64 + lane] = v.x + v.y;   // u2 = Re U + Im U
    }
    if (job.umat_mirror) {   // the mirrored V^H plan runs this sub-stage's U^H: the same matrix conjugate-transposed, no second chain
        double* om = job.umat_mirror + ((size_t)b * job.mirror_nsubs + job.mirror_index) * 12 * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const cplx t = u[(4 * s + (lane >> 4)) * 16 + (lane & 15)];
            const double re = t.x, im = -t.y;       // U^H[l % 16][4 s + l / 16] = conj(U[4 s + l / 16][l % 16])
            om[(0 * 4 + s) * 64 + lane] = re;
            om[(1 * 4 + s) * 64 + lane] = im - re;
            om[(2 * 4 + s) * 64 + lane] = re + im;
        }
    }
}

// rho <- G^H rho G for a rotation G (kind, c, s) on local bit LB of a 4 x 4 matrix rho[a * 4 + b] (a: z side, b: w side)
template <int LB>
__device__ __forceinline__ void rho_unapply(cplx (&rho)[16], int kind, double c, double s) {
    constexpr int st = LB == 0 ? 1 : 2, other = LB == 0 ? 2 : 1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) {   // rows: G^H
            const int a0 = q * other;
            if (kind == MOP_RY) ry2(rho[a0 * 4 + b], rho[(a0 + st) * 4 + b], c, -s);
            else if (kind == MOP_RZ) rz2(rho[a0 * 4 + b], rho[(a0 + st) * 4 + b], c, -s);
            else rx2(rho[a0 * 4 + b], rho[(a0 + st) * 4 + b], c, -s);
        }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int a = 0; a < 4; ++a) {   // columns: G^T (Ry^T = Ry(-theta); Rz, Rx symmetric)
            const int b0 = q * other;
            if (kind == MOP_RY) ry2(rho[a * 4 + b0], rho[a * 4 + b0 + st], c, -s);
            else if (kind == MOP_RZ) rz2(rho[a * 4 + b0], rho[a * 4 + b0 + st], c, s);
            else rx2(rho[a * 4 + b0], rho[a * 4 + b0 + st], c, s);
        }
}
// 0.5j <P w|z> (Ry: 0.5 <Y w|z> / i) of a rotation on local bit LB from rho (core_operations.py:267-351)
template <int LB>
__device__ __forceinline__ cplx rho_dot(const cplx (&rho)[16], int kind) {
    constexpr int st = LB == 0 ? 1 : 2, other = LB == 0 ? 2 : 1;
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i0 = q * other, i1 = i0 + st;
        if (kind == MOP_RY) { re += rho[i1 * 4 + i0].x - rho[i0 * 4 + i1].x; im += rho[i1 * 4 + i0].y - rho[i0 * 4 + i1].y; }
        else if (kind == MOP_RZ) { re += rho[i0 * 4 + i0].x - rho[i1 * 4 + i1].x; im += rho[i0 * 4 + i0].y - rho[i1 * 4 + i1].y; }
        else { re += rho[i0 * 4 + i1].x + rho[i1 * 4 + i0].x; im += rho[i0 * 4 + i1].y + rho[i1 * 4 + i0].y; }
    }
    return kind == MOP_RY ? make_double2(0.5 * re, 0.5 * im) : make_double2(-0.5 * im, 0.5 * re);
}

// Gradient entries of one sub-stage from its R = Z W^H (summed over tiles in a fixed order).  R belongs to the END of
// the sub-stage.  Walking the gate groups backwards: the group's inner products only involve its two qubits, so they
// are functions of the 4 x 4 partial trace rho_g of the current R over the other two register bits; rho_g is set aside
// and the whole group is peeled off R at once, R <- B^H R B (two LDS round trips per group -- this chain is the serial
// part).  Afterwards one lane per group walks its rho_g through the group's rotations (value right after each
// rotation, core_operations.py:921-935) and stores the slots.  One wave per (sub-stage, lane).
// WAVES = 4 (many tiles per lane of the batch, i.e. few lanes: the single-evaluation regime): three helper waves share
// the sum over the tiles -- the dependent-load chain that otherwise dominates this kernel -- and retire; wave 0 adds the
// four partial sums in a fixed order and walks the groups alone.
#ifdef AQC_TUNING
__device__ unsigned long long g_rgrad_stamps[32 * 8];
#define RG_STAMP(slot) do { if (blockIdx.y == 0 && blockIdx.x < 32 && threadIdx.x == 0) g_rgrad_stamps[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_STAMP(slot) do { } while (0)
#endif
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rgrad_kernel(const DevSub3* subs, const DevGrp* grps, int ent, const double* thetas, int T,
                                                           const cplx* rpart, int ntiles, int nsubs_total, cplx* partial, int nslots,
                                                           int from, int to, int front, const int* slot_theta, cplx* grads, cplx* mirror,
                                                           const GatherJob gj, int tiles_per_lane, int chunk, int sparse_subs,
                                                           const int* lane_parts, int conj_sub, const double* umat, int nsubs_run,
                                                           const RgradSecond second) {   // ntiles: partial slots per (lane, sub-stage)
    if ((int)blockIdx.x == nsubs_run + second.count) {   // (nsubs_run: the launch walks the first nsubs_run sub-stages of the plan, then `second`'s)   // the passenger (see GatherJob): one extra workgroup per lane of the batch
        const cplx* src = static_cast<const cplx*>(gj.buf) + (size_t)blockIdx.y * gj.lane_stride;
        for (int i = threadIdx.x; i < gj.count; i += 64 * WAVES) {
            const cplx v = src[(size_t)gj.elem[i]];
            static_cast<cplx*>(gj.out)[(size_t)blockIdx.y * gj.count + i] = v;
            if (gj.mirror) static_cast<cplx*>(gj.mirror)[(size_t)blockIdx.y * gj.count + i] = v;
        }
        return;
    }
    __shared__ cplx R[16 * 17];    // R[j * 17 + i] = sum_c z_c[j] conj(w_c[i]); rows padded: column walks hit 16 different banks
    __shared__ Gm gm[kGrpChunk];
    __shared__ cplx rho_s[kGrpChunk][16];
    __shared__ cplx psum[WAVES > 1 ? WAVES - 1 : 1][WAVES > 1 ? 256 : 1];   // (one wave: no partial sums to hand over -- 4 KB less, 13 workgroups per CU instead of 10)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    int si = blockIdx.x;
    if (si >= nsubs_run) {   // a sub-stage of the second plan of the launch (projected route: the virtual plan's walk rides along)
        si -= nsubs_run;
        subs = second.subs; grps = second.grps; rpart = second.rpart; ntiles = second.nparts; nsubs_total = second.nsubs_total;
        tiles_per_lane = second.nparts; chunk = 0; sparse_subs = second.nsubs_total; lane_parts = second.lane_parts; conj_sub = -1; umat = second.umat;
    }
    RG_STAMP(0);
    const DevSub3 sub = subs[si];
    const double* th = thetas + (size_t)b * T;
    // partials of this (lane, sub-stage): one per tile, or one per segment of the persistent sweep (chunk > 0: the workgroups
    // that hold tiles of lane b are (b ntiles) / chunk .. ((b + 1) ntiles - 1) / chunk; summed in that fixed order)
    const int nparts_stride = ntiles;
    if (chunk > 0) ntiles = ((b + 1) * tiles_per_lane - 1) / chunk - (b * tiles_per_lane) / chunk + 1;
    if (si < sparse_subs) ntiles = lane_parts[b];   // sub-stages of a stage that ran over an item list: one partial per item of the lane
    const cplx* rp = rpart + ((size_t)b * nsubs_total + si) * nparts_stride * 256;
    {   // fixed-order sum over the tiles (wave w: tiles w, w + WAVES, ...), 8 tiles (32 loads per lane) in flight at a time;
        // in the few-lane variant wave 0 decodes the last chunk of groups (the first one the walk needs) while its first batch
        // of loads is in flight
        cplx acc[4] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
        constexpr int TB = 8;
        for (int t0 = wave; t0 < ntiles; t0 += TB * WAVES) {
            cplx v[TB][4];
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[u][r] = t0 + u * WAVES < ntiles ? rp[(size_t)(t0 + u * WAVES) * 256 + 64 * r + lane] : make_double2(0.0, 0.0);
            if (WAVES > 1 && t0 == wave && wave == 0 && sub.ngrp > 0)   // latency regime only: the live loads cost occupancy
                stage_groups(gm, grps, sub.grp_begin + max(0, sub.ngrp - kGrpChunk), min(sub.ngrp, kGrpChunk), th, ent, lane);
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[r].x += v[u][r].x; acc[r].y += v[u][r].y; }
        }
        if (WAVES > 1) {
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) psum[wave - 1][64 * r + lane] = acc[r];
            }
            __syncthreads();
            if (wave > 0) return;   // s_barrier only waits for the surviving waves of a workgroup
#pragma unroll
            for (int w = 0; w < WAVES - 1; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const cplx p = psum[w][64 * r + lane]; acc[r].x += p.x; acc[r].y += p.y; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)   // MFMA D layout: entry 64 r + l is R[4 r + l / 16][l % 16]
            R[(4 * r + (lane >> 4)) * 17 + (lane & 15)] = acc[r];
    }
    if (si == conj_sub) {   // this sub-stage's R was taken from its INPUTS (sweep_mfma_kernel, r_only): R_end = U R U^H with the sub-stage's U,
                            // read back from the operand planes the stage kernels use (u0 = Re U, u1 = Im U - Re U; entry [s][l] = U[l % 16][4 s + l / 16])
        cplx* const Us = &rho_s[0][0];   // (16 x 16, in an array the walk has not started to use; T = U R passes through registers back into
                                         // R's own storage: no extra LDS -- a dozen of these workgroups share a CU)
        const double* up = umat + ((size_t)b * nsubs_total + si) * 12 * 64;
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const double re = up[(0 * 4 + s2) * 64 + lane], im = up[(1 * 4 + s2) * 64 + lane] + re;
            Us[(lane & 15) * 16 + 4 * s2 + (lane >> 4)] = make_double2(re, im);
        }
        __syncthreads();
        const int i0 = lane & 15, j0 = lane >> 4;   // this lane: entries (j0 + 4 m, i0)
        cplx acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {               // T = U R
            const int j = j0 + 4 * m;
            double re = 0.0, im = 0.0;
            for (int k = 0; k < 16; ++k) {
                const cplx u = Us[j * 16 + k], r = R[k * 17 + i0];
                re += u.x * r.x - u.y * r.y; im += u.x * r.y + u.y * r.x;
            }
            acc[m] = make_double2(re, im);
        }
        __syncthreads();                            // (every entry of R has been read)
#pragma unroll
        for (int m = 0; m < 4; ++m) R[(j0 + 4 * m) * 17 + i0] = acc[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) {               // R = T U^H
            const int j = j0 + 4 * m;
            double re = 0.0, im = 0.0;
            for (int k = 0; k < 16; ++k) {
                const cplx t = R[j * 17 + k], u = Us[i0 * 16 + k];
                re += t.x * u.x + t.y * u.y; im += t.y * u.x - t.x * u.y;
            }
            acc[m] = make_double2(re, im);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) R[(j0 + 4 * m) * 17 + i0] = acc[m];
        __syncthreads();
    }
    RG_STAMP(1);
    cplx* out = partial + (size_t)b * nslots;
    // slot_theta != null: every theta is fed by exactly one slot, so the entry is final here -- it goes straight into the
    // gradient (and its pinned host copy) and finalize_kernel is not launched; entries outside block_range / front_layer
    // are written as zeros
    auto put = [&](int slot, cplx v) {
        out[slot] = v;
        if (slot_theta) {
            const int t = slot_theta[slot];
            if (t >= 0) {
                grads[(size_t)b * T + t] = v;
                if (mirror) mirror[(size_t)b * T + t] = v;
            }
        }
    };
    const int lo = lane & 15, hi = lane >> 4;
    for (int top = sub.ngrp; top > 0; top -= kGrpChunk) {
        const int base = max(0, top - kGrpChunk), count = top - base;
        __syncthreads();
        if (WAVES == 1 || top != sub.ngrp) stage_groups(gm, grps, sub.grp_begin + base, count, th, ent, lane);   // WAVES > 1: the first chunk was decoded above
        __syncthreads();
        RG_STAMP(2);
        for (int i = count - 1; i >= 0; --i) {
            const Gm& g = gm[i];
            {   // rho_g[a][b] = sum_o R[(a, o)][(b, o)]: lane = (a, b, o), quad sum over o
                const int a = lane >> 4, bb = (lane >> 2) & 3, o = lane & 3;
                cplx v = R[grp_index(g, a, o) * 17 + grp_index(g, bb, o)];
                v.x += dpp_mov<0xB1, 0xf>(v.x); v.y += dpp_mov<0xB1, 0xf>(v.y);   // quad_perm [1,0,3,2]
                v.x += dpp_mov<0x4E, 0xf>(v.x); v.y += dpp_mov<0x4E, 0xf>(v.y);   // quad_perm [2,3,0,1]
                if (o == 0) rho_s[i][a * 4 + bb] = v;
            }
            cplx x[4];
            int idx[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) { idx[l] = grp_index(g, l, hi) * 17 + lo; x[l] = R[idx[l]]; }   // rows: B^H on the z side
            grp_apply<1>(g, ent, x);
#pragma unroll
            for (int l = 0; l < 4; ++l) R[idx[l]] = x[l];
            __syncthreads();
#pragma unroll
            for (int l = 0; l < 4; ++l) { idx[l] = lo * 17 + grp_index(g, l, hi); x[l] = R[idx[l]]; }   // columns: B^T on the w side
            grp_apply<2>(g, ent, x);
#pragma unroll
            for (int l = 0; l < 4; ++l) R[idx[l]] = x[l];
            __syncthreads();
        }
        RG_STAMP(3);
        if (lane < count) {   // one group per lane: walk rho backwards through the group's rotations
            const Gm& g = gm[lane];
            const bool on = g.jblock < 0 ? (front != 0) : (g.jblock >= from && g.jblock < to);
            if (on && g.slot0 >= 0) {
                cplx rho[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) rho[e] = rho_s[lane][e];
                if (g.type == 0) {   // Rz(t2) [slot 0], Ry(t1) [slot 1], Rz(t0) [slot 2] on pc (local bit 1)
                    put(g.slot0 + 2, rho_dot<1>(rho, MOP_RZ));
                    rho_unapply<1>(rho, MOP_RZ, g.rc[0], g.rs[0]);
                    put(g.slot0 + 1, rho_dot<1>(rho, MOP_RY));
                    rho_unapply<1>(rho, MOP_RY, g.rc[1], g.rs[1]);
                    put(g.slot0 + 0, rho_dot<1>(rho, MOP_RZ));
                } else {             // Ry(c,t0) [0], Rz(c,t1) [1], Ry(t,t2) [2], Rs(t,t3) [3], CP [4]; rotations on c and t commute
                    const int ks = ent == 0 ? MOP_RX : MOP_RZ;
                    if (g.flags & 2) rho_unapply<0>(rho, MOP_RZ, kR, kR);
                    put(g.slot0 + 3, rho_dot<0>(rho, ks));
                    rho_unapply<0>(rho, ks, g.rc[3], g.rs[3]);
                    put(g.slot0 + 2, rho_dot<0>(rho, MOP_RY));
                    put(g.slot0 + 1, rho_dot<1>(rho, MOP_RZ));
                    rho_unapply<1>(rho, MOP_RZ, g.rc[1], g.rs[1]);
                    put(g.slot0 + 0, rho_dot<1>(rho, MOP_RY));
                    if (ent == 2) {   // -i <P11 w|z> at the entangler (core_op_matrix.py:430-477): peel the two Ry first
                        rho_unapply<0>(rho, MOP_RY, g.rc[2], g.rs[2]);
                        rho_unapply<1>(rho, MOP_RY, g.rc[0], g.rs[0]);
                        put(g.slot0 + 4, make_double2(rho[15].y, -rho[15].x));
                    }
                }
            } else if (slot_theta && g.slot0 >= 0) {
                for (int k = 0; k < 5; ++k) put(g.slot0 + k, make_double2(0.0, 0.0));
            }
        }
        RG_STAMP(4);
    }
}
#ifdef AQC_TUNING
void rgrad_print_stamps(int nsubs) {
    unsigned long long h[32 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rgrad_stamps), sizeof h) != hipSuccess) return;
    double a = 0, b = 0, c = 0, d = 0;
    const int n = nsubs < 32 ? nsubs : 32;
    for (int i = 0; i < n; ++i) { a += h[i * 8 + 1] - h[i * 8]; b += h[i * 8 + 2] - h[i * 8 + 1]; c += h[i * 8 + 3] - h[i * 8 + 2]; d += h[i * 8 + 4] - h[i * 8 + 3]; }
    fprintf(stderr, "aqc_hip stamps: gradient walk (lane 0, %d sub-stages): tile sum %.0f + group decode %.0f + R walk %.0f + rho walk %.0f cycles\n", n, a / n, b / n, c / n, d / n);
}
#endif

// ---- launchers -----------------------------------------------------------------------------------------------
// first_hi[g][s]: tile offset of (amplitude 4 s, chunk 16 g) of the stage's first sub-stage.  The slot tables hold swz3(local
// index); swz3 is an involution, and the uniform part uses other bits of the local index than a lane's own part.
void stage3_first_offsets(Stage3Args& a, const DevSub3& first_sub) {
    for (int g = 0; g < 16; ++g)
        for (int s = 0; s < 4; ++s) {
            const unsigned local = swz3(first_sub.kk[g][s] >> 4);
            a.first_hi[g][s] = a.stage.dlo[local & 63u] | a.stage.dhi[local >> 6];
        }
}

int mfma_threads(int k, bool sweep) { return (sweep && k == 12) ? 64 * SweepShape<12>::kWaves : 64 * std::min(4, 1 << std::max(0, k - 8)); }
size_t apply3_lds_bytes(int k) { return (size_t)16 << k; }
size_t sweep3_lds_bytes(int k) {   // two tiles + R scratch of up to 4 slots, double-buffered where sweep_scratch_double says so
    return ((size_t)32 << k) + (size_t)(sweep_scratch_double(k) ? 2 : 1) * std::min(4, mfma_threads(k, true) / 64) * 256 * sizeof(cplx);
}

template <typename F>
static hipError_t big_lds(F kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t init_kernels3() {
    hipError_t e;
#define AQC_TRY(x) if ((e = (x)) != hipSuccess) return e
    AQC_TRY(big_lds(apply_mfma_kernel<8>)); AQC_TRY(big_lds(apply_mfma_kernel<9>)); AQC_TRY(big_lds(apply_mfma_kernel<10>));
    AQC_TRY(big_lds(apply_mfma_kernel<11>)); AQC_TRY(big_lds(apply_mfma_kernel<12>));
    AQC_TRY(big_lds(sweep_mfma_kernel<8>)); AQC_TRY(big_lds(sweep_mfma_kernel<9>)); AQC_TRY(big_lds(sweep_mfma_kernel<10>));
    AQC_TRY(big_lds(sweep_mfma_kernel<11>)); AQC_TRY(big_lds(sweep_mfma_kernel<12>));
    AQC_TRY(big_lds((apply_mfma_kernel<8, true>))); AQC_TRY(big_lds((apply_mfma_kernel<9, true>))); AQC_TRY(big_lds((apply_mfma_kernel<10, true>)));
    AQC_TRY(big_lds((apply_mfma_kernel<11, true>))); AQC_TRY(big_lds((apply_mfma_kernel<12, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<8, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<9, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<10, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<11, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<12, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<8, false, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<9, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<10, false, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<11, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<12, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<8, false, false, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<9, false, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<10, false, false, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<11, false, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<12, false, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<8, true, false, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<9, true, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<10, true, false, true>))); AQC_TRY(big_lds((sweep_mfma_kernel<11, true, false, true>)));
    AQC_TRY(big_lds((sweep_mfma_kernel<12, true, false, true>)));
#undef AQC_TRY
    return hipSuccess;
}
int mfma_occupancy(int k, bool sweep) {   // resident workgroups per CU for the tile size (diagnostics)
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    const int t = mfma_threads(k, sweep);
    const size_t l = sweep ? sweep3_lds_bytes(k) : apply3_lds_bytes(k);
#define AQC_OCC(KK) case KK: e = sweep ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sweep_mfma_kernel<KK>, t, l) \
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, apply_mfma_kernel<KK>, t, l); break
    switch (k) { AQC_OCC(8); AQC_OCC(9); AQC_OCC(10); AQC_OCC(11); AQC_OCC(12); default: break; }
#undef AQC_OCC
    return e == hipSuccess ? n : -1;
}
// workgroups of the persistent 2^12 sweep: one per CU of the current device (the occupancy its 144 KiB of LDS allows)
static long persistent_sweep_grid() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    if (const char* e = getenv("AQC_SWEEP_GRID")) { const long v = atol(e); if (v > 0) return v; }   // experiments
    return cus[dev];
}
hipError_t launch_apply3(int ntiles, int batch, int k, hipStream_t s, const Stage3Args& a) {
    if (a.ntiles != ntiles || a.batch != batch) return hipErrorInvalidValue;
    for (int l = 0; l < 64; ++l)   // 32-bit byte offset per lane in the prefetch (see launch_sweep3)
        if (k >= 12 && a.stage.dlo[l] >= (1u << 28)) return hipErrorInvalidValue;
    const bool list = a.items != nullptr;
    if (list && (!a.nitems || a.max_items < 1)) return hipErrorInvalidValue;
    // 2^12 tiles: persistent workgroups, two per CU, walking over (tile, lane) items; smaller tiles: one item per workgroup
    const long nwork = list ? (long)a.max_items : (long)ntiles * batch;
    static const long persist = []() { const char* e = getenv("AQC_APPLY_PERSIST"); return e ? atol(e) : 2L; }();   // workgroups per CU, 0 = off
    const dim3 grid = (k >= 12 && persist > 0 && a.stage.nsubs > 0) ? dim3((unsigned)std::min<long>(nwork, persist * persistent_sweep_grid()))
                                                                     : ((k >= 12 || list) ? dim3((unsigned)nwork) : dim3(ntiles, batch));
    if (list && k >= 12 && !(persist > 0 && a.stage.nsubs > 0)) return hipErrorInvalidValue;   // (a list launch of 2^12 tiles is persistent)
    const int t = mfma_threads(k, false);
    const size_t l = apply3_lds_bytes(k);
#define AQC_LAUNCH(KK) case KK: if (list) apply_mfma_kernel<KK, true><<<grid, t, l, s>>>(a); else apply_mfma_kernel<KK, false><<<grid, t, l, s>>>(a); break
    switch (k) {
        AQC_LAUNCH(8); AQC_LAUNCH(9); AQC_LAUNCH(10); AQC_LAUNCH(11); AQC_LAUNCH(12);
        default: return hipErrorInvalidValue;
    }
#undef AQC_LAUNCH
    return hipGetLastError();
}
// persistent sweep: items per workgroup (contiguous, lane-major) and partial slots per (lane, sub-stage); 0 / ntiles otherwise
int sweep3_chunk(int ntiles, int batch, int k) {
    if (k < 12) return 0;
    const long nwork = (long)ntiles * batch, g = std::min<long>(nwork, persistent_sweep_grid());
    return (int)((nwork + g - 1) / g);
}
int sweep3_nparts(int ntiles, int batch, int k) {
    const int chunk = sweep3_chunk(ntiles, batch, k);
    return chunk > 0 ? std::min(ntiles, (ntiles + chunk - 1) / chunk + 1) : ntiles;
}
hipError_t launch_sweep3(int ntiles, int batch, int k, hipStream_t s, const Stage3Args& a) {
    if (a.ntiles != ntiles || a.batch != batch) return hipErrorInvalidValue;
    const bool list = a.items != nullptr;
    if (list && (!a.nitems || a.max_items < 1)) return hipErrorInvalidValue;
    if (a.nparts < 1 || (!list && (a.nparts != sweep3_nparts(ntiles, batch, k) || a.chunk != sweep3_chunk(ntiles, batch, k)))) return hipErrorInvalidValue;   // (a list names its slots)
    if (k >= 12 && a.stage.nsubs <= 0) return hipErrorInvalidValue;   // the persistent form feeds the first sub-stage from registers
    for (int l = 0; l < 64; ++l)   // the persistent sweep addresses its prefetch with a 32-bit byte offset per lane
        if (k >= 12 && a.stage.dlo[l] >= (1u << 28)) return hipErrorInvalidValue;
    // 2^12 tiles: one persistent workgroup per CU walking over its items (see the kernel); smaller tiles: one item each
    const long nwork = list ? (long)a.max_items : (long)ntiles * batch;
    const dim3 grid((unsigned)(k >= 12 ? (list ? std::min<long>(nwork, persistent_sweep_grid()) : (nwork + a.chunk - 1) / a.chunk) : nwork));
    const int t = mfma_threads(k, true);
    const size_t l = sweep3_lds_bytes(k);
    const bool skipw = !list && a.supp != nullptr;
    const bool rlast = !skipw && a.r_only_last != 0;
#define AQC_LAUNCH(KK) case KK: if (list && rlast) sweep_mfma_kernel<KK, true, false, true><<<grid, t, l, s>>>(a); \
                                else if (list) sweep_mfma_kernel<KK, true, false><<<grid, t, l, s>>>(a); \
                                else if (skipw) sweep_mfma_kernel<KK, false, true><<<grid, t, l, s>>>(a); \
                                else if (rlast) sweep_mfma_kernel<KK, false, false, true><<<grid, t, l, s>>>(a); \
                                else sweep_mfma_kernel<KK, false, false><<<grid, t, l, s>>>(a); break
    switch (k) {
        AQC_LAUNCH(8); AQC_LAUNCH(9); AQC_LAUNCH(10); AQC_LAUNCH(11); AQC_LAUNCH(12);
        default: return hipErrorInvalidValue;
    }
#undef AQC_LAUNCH
    return hipGetLastError();
}

// ---- tile lists ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_of_elem(const DevStage& st, long long e) {
    int t = 0;
    for (int i = 0; i < st.nub; ++i) t |= (int)((e >> st.ubits[i]) & 1) << i;
    return t;
}
// unique tiles of lane b in first-occurrence order (support first); returns their number
__device__ __forceinline__ int lane_tiles(const DevStage& st, const long long* supp, int per_lane, const long long* extra, int nextra, int b,
                                          int (&tiles)[kMaxTileCands]) {
    int n = 0;
    auto add = [&](long long e) {
        if (e < 0) return;
        const int t = tile_of_elem(st, e);
        for (int i = 0; i < n; ++i)
            if (tiles[i] == t) return;
        tiles[n++] = t;
    };
    for (int k = 0; supp && k < per_lane; ++k) add(supp[(size_t)b * per_lane + k]);
    for (int k = 0; extra && k < nextra; ++k) add(extra[k]);
    return n;
}
__global__ __launch_bounds__(1024) void tile_items_kernel(const DevStage st, const long long* supp, int per_lane, const long long* extra, int nextra,
                                                          int batch, TileItem* items, int* nitems, int* lane_parts, int* prev_tiles,
                                                          TileItem* clear_items, int* nclear) {
    __shared__ int scan_a[1024], scan_b[1024];
    const int tid = threadIdx.x, per = (batch + 1023) / 1024;
    const int b0 = tid * per, b1 = min(batch, b0 + per);
    int tiles[kMaxTileCands];
    int mine = 0, mine_clear = 0;
    for (int b = b0; b < b1; ++b) {
        const int n = lane_tiles(st, supp, per_lane, extra, nextra, b, tiles);
        mine += n;
        if (prev_tiles)
            for (int k = 0; k < 2; ++k) {
                const int pt = prev_tiles[2 * b + k];
                bool stale = pt >= 0;
                for (int i = 0; i < n; ++i) stale = stale && tiles[i] != pt;
                mine_clear += stale ? 1 : 0;
            }
    }
    scan_a[tid] = mine; scan_b[tid] = mine_clear;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // inclusive scans
        const int va = tid >= d ? scan_a[tid - d] : 0, vb = tid >= d ? scan_b[tid - d] : 0;
        __syncthreads();
        scan_a[tid] += va; scan_b[tid] += vb;
        __syncthreads();
    }
    int at = scan_a[tid] - mine, at_clear = scan_b[tid] - mine_clear;
    if (tid == 1023) { *nitems = scan_a[tid]; if (nclear) *nclear = scan_b[tid]; }
    for (int b = b0; b < b1; ++b) {
        const int n = lane_tiles(st, supp, per_lane, extra, nextra, b, tiles);
        if (lane_parts) lane_parts[b] = n;
        if (prev_tiles) {
            for (int k = 0; k < 2; ++k) {
                const int pt = prev_tiles[2 * b + k];
                bool stale = pt >= 0;
                for (int i = 0; i < n; ++i) stale = stale && tiles[i] != pt;
                if (stale) clear_items[at_clear++] = TileItem{b, pt, 0, 0};
            }
            prev_tiles[2 * b] = n > 0 ? tiles[0] : -1;
            prev_tiles[2 * b + 1] = n > 1 ? tiles[1] : -1;
        }
        for (int i = 0; i < n; ++i) items[at++] = TileItem{b, tiles[i], i, 0};
    }
}
hipError_t launch_tile_items(const DevStage& stage, const long long* supp, int per_lane, const long long* extra, int nextra, int batch,
                             TileItem* items, int* nitems, int* lane_parts, int* prev_tiles, TileItem* clear_items, int* nclear, hipStream_t s) {
    if ((supp ? per_lane : 0) + (extra ? nextra : 0) > kMaxTileCands || batch < 1) return hipErrorInvalidValue;
    if (prev_tiles && ((supp ? per_lane : 0) > 2 || (extra && nextra > 0) || !clear_items || !nclear)) return hipErrorInvalidValue;
    tile_items_kernel<<<1, 1024, 0, s>>>(stage, supp, per_lane, extra, nextra, batch, items, nitems, lane_parts, prev_tiles, clear_items, nclear);
    return hipGetLastError();
}
// buf[item.lane][tile item.tile of the stage] <- 0 for the first *nclear items of the list
__global__ __launch_bounds__(256) void clear_tiles_kernel(const DevStage st, cplx* buf, size_t lane_stride, const TileItem* clear_items, const int* nclear) {
    if ((int)blockIdx.x >= *nclear) return;
    const TileItem it = clear_items[blockIdx.x];
    cplx* dst = buf + (size_t)it.lane * lane_stride + tile_base3(st, (unsigned)it.tile);
    for (unsigned l = threadIdx.x; l < (1u << st.k); l += 256) dst[tile_offset3(st, l)] = make_double2(0.0, 0.0);
}
hipError_t launch_clear_tiles(const DevStage& stage, void* buf, size_t lane_stride, const TileItem* clear_items, const int* nclear, int max_items,
                              hipStream_t s) {
    if (max_items < 1) return hipSuccess;
    clear_tiles_kernel<<<max_items, 256, 0, s>>>(stage, static_cast<cplx*>(buf), lane_stride, clear_items, nclear);
    return hipGetLastError();
}
hipError_t launch_ubuild(const UJob* jobs, int njobs, const double* thetas, int T, int batch, hipStream_t s, double* thetas_copy) {
    if (njobs < 1) return hipSuccess;
    ubuild_kernel<<<dim3(njobs, batch), 64, 0, s>>>(jobs, thetas, T, thetas_copy);
    return hipGetLastError();
}
hipError_t launch_rgrad(const DevSub3* subs, const DevGrp* grps, int entangler, const double* thetas, int T, const void* rpart,
                        int ntiles, int nsubs_total, void* partial, int nslots, int from, int to, int front, int batch, hipStream_t s,
                        const int* slot_theta, void* grads, void* mirror, GatherJob gather, int nparts, int chunk, int sparse_subs,
                        const int* lane_parts, int conj_sub, const double* umat, int nsubs_run, const RgradSecond* second_plan) {
    if (nsubs_total < 1) return hipSuccess;
    if (nsubs_run < 0 || nsubs_run > nsubs_total) nsubs_run = nsubs_total;
    RgradSecond second;
    memset(&second, 0, sizeof second);
    if (second_plan) second = *second_plan;
    const int extra = (gather.count > 0 && gather.buf ? 1 : 0) + second.count;
    if (nsubs_run + extra < 1) return hipSuccess;
    const int tiles_per_lane = ntiles;
    if (nparts <= 0) nparts = ntiles;
    ntiles = nparts;   // slots per (lane, sub-stage); the kernels derive the number in use from (tiles_per_lane, chunk)
    if (nparts >= 32 && second.count == 0)
        rgrad_kernel<4><<<dim3(nsubs_run + extra, batch), 256, 0, s>>>(subs, grps, entangler, thetas, T, static_cast<const cplx*>(rpart), ntiles,
                                                                  nsubs_total, static_cast<cplx*>(partial), nslots, from, to, front, slot_theta,
                                                                  static_cast<cplx*>(grads), static_cast<cplx*>(mirror), gather, tiles_per_lane, chunk, sparse_subs, lane_parts, conj_sub, umat, nsubs_run, second);
    else
        rgrad_kernel<1><<<dim3(nsubs_run + extra, batch), 64, 0, s>>>(subs, grps, entangler, thetas, T, static_cast<const cplx*>(rpart), ntiles,
                                                                 nsubs_total, static_cast<cplx*>(partial), nslots, from, to, front, slot_theta,
                                                                 static_cast<cplx*>(grads), static_cast<cplx*>(mirror), gather, tiles_per_lane, chunk, sparse_subs, lane_parts, conj_sub, umat, nsubs_run, second);
    return hipGetLastError();
}

}  // namespace aqc

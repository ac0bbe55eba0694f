// Host-visible launch interface of aqc_kernels.hip.
#pragma once
#include <string>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>

#include "aqc_device.h"

namespace aqc {

struct StageArgs {
    const DevStage* stage;  // this stage's descriptor (device)
    const DevOp* ops;       // all ops of the plan (device)
    const DevSub* subs;     // register-blocked kernels: sub-stages of the plan
    const DevMop* mops;     // ... and their micro-ops
    const double* coef;     // [batch][ncoef][kCoefStride]
    int ncoef;
    const double2* in0;     // apply: src; sweep: w in
    const double2* in1;     // sweep: z in
    double2* out0;          // apply: dst; sweep: w out
    double2* out1;          // sweep: z out
    size_t lane_stride;     // elements between consecutive batch lanes
    double2* partial;       // [batch][nslots][ntiles_max] inner-product partials
    int nslots, ntiles_max;
    int from, to, front;    // block_range / front_layer (core_operations.py:829-830)
    int final_stage;        // register-blocked V / V^H: last launch applies the lane's overall sign
    int debug;              // AQC_DEBUG_SKIP bits (timing experiments only): 1 skip micro-op bodies, 2 skip reductions
};

size_t apply_lds_bytes(int k);
size_t sweep_lds_bytes(int k, int threads);
hipError_t init_kernels();
hipError_t launch_apply(int ent, bool inverse, int ntiles, int batch, int threads, int k, hipStream_t s, const StageArgs& a);
hipError_t launch_sweep(int ent, int ntiles, int batch, int threads, int k, hipStream_t s, const StageArgs& a);
hipError_t launch_coef(const double* thetas, double* coef, int n, int nblocks, int tpb, int tail, int batch, hipStream_t s);
hipError_t launch_finalize(const void* partial, const int* theta_slots, const int* slot_ntiles, void* grads, int T,
                           int nslots, int ntiles_max, int n, int tpb, int from, int to, int front, int batch,
                           hipStream_t s, void* mirror = nullptr);   // mirror: optional pinned host copy of the result
hipError_t launch_scatter_one(void* buf, size_t lane_stride, int batch, const long long* elem, hipStream_t s);
hipError_t launch_scatter_two(void* buf, size_t lane_stride, int batch, const long long* elem, const void* coef, long long* prev, hipStream_t s);
hipError_t launch_set_identity(void* buf, size_t lane_stride, int dim, int pitch, int batch, hipStream_t s);
hipError_t launch_gather(const void* buf, size_t lane_stride, const long long* elem, int count, int batch, void* out,
                         hipStream_t s, void* mirror = nullptr);
hipError_t launch_vdot(const void* a, const void* b, size_t lane_stride, size_t count, int batch, void* part, int nparts,
                       void* out, hipStream_t s);

// aqc_kernels2.hip (register-blocked kernels)
hipError_t init_kernels2();
hipError_t launch_apply2(int ent, int ntiles, int batch, int k, hipStream_t s, const StageArgs& a);
hipError_t launch_sweep2(int ent, int ntiles, int batch, int k, int reg_bits, hipStream_t s, const StageArgs& a);

// aqc_kernels3.hip (matrix-core kernels)
struct TileItem { int lane, tile, slot, pad; };   // one (tile, lane) work item of a stage launch over a SUBSET of the tiles; slot: its partial-R slot
struct Stage3Args {
    DevStage stage;          // by value: kernel arguments live in the constant address space, so the offset tables are
                             // scalar loads that no store of the kernel can invalidate
    const DevSub3* subs;     // all sub-stages of the plan
    const double* umat;      // [batch][nsubs_total][3][4][64]: MFMA B operands of every sub-stage's 16 x 16 unitary
    int nsubs_total;
    const double2* in0;      // apply: src; sweep: w in
    const double2* in1;      // sweep: z in
    double2* out0;
    double2* out1;
    size_t lane_stride;
    double2* rpart;          // sweep: [batch][nsubs_total][ntiles][256] per-tile R = Z W^H of every sub-stage
    int ntiles;
    int batch;               // lanes of the batch (sweep: work items = ntiles x batch)
    int chunk, nparts;       // sweep: items per persistent workgroup (0: one item per workgroup) and partial-R slots per
                             // (lane, sub-stage) -- sweep3_chunk / sweep3_nparts
    int store_out;           // sweep: bit 0 store w, bit 1 store z; 0 for the last stage (its w and z are never read again)
    int r_only_last;         // sweep, last stage: its last sub-stage computes R from its INPUTS alone (no U products); the gradient walk
                             // conjugates that R by the sub-stage's U (launch_rgrad: conj_sub)
    const long long* supp;   // sweep from basis states: supp[lane][2] = element index of the lane's (<= 2) basis states (-1: none), or null.
                             // The kernel then skips the W and R products of 16-chunk groups (and K-steps of the W product) where w is
                             // zero by construction (DevSub3::skipinfo, DevStage::fresh_nonlocal) -- exact zeros, so the results do not change
    const TileItem* items;   // launch over a subset of the (tile, lane) pairs: device table + its length on the device (null: all pairs);
    const int* nitems;       // the grid is sized for max_items, the kernels read the actual length
    int max_items;
    unsigned first_hi[16][4];   // persistent sweep: tile offset (elements) of the FIRST sub-stage's operand of (group, K-step);
                                // a lane adds the offset of its own (chunk l % 16, amplitude l / 16) position (stage3_first_offsets)
    int debug;               // tuning builds: work-skipping bits for timing experiments (1 LDS writes, 2 LDS reads, 4 R MFMAs, 8 U MFMAs)
    unsigned long long* stamps;   // tuning builds (-DAQC_TUNING): [workgroup][kStampSlots] s_memtime stamps of wave 0, else null
};
constexpr int kStampSlots = 40;
void stage3_first_offsets(Stage3Args& a, const DevSub3& first_sub);   // host: fills first_hi from the stage and its first sub-stage
hipError_t init_kernels3();
int mfma_threads(int k, bool sweep);
int mfma_occupancy(int k, bool sweep);
hipError_t launch_apply3(int ntiles, int batch, int k, hipStream_t s, const Stage3Args& a);
hipError_t launch_sweep3(int ntiles, int batch, int k, hipStream_t s, const Stage3Args& a);
void rgrad_print_stamps(int nsubs);   // tuning builds only
int sweep3_chunk(int ntiles, int batch, int k);
int sweep3_nparts(int ntiles, int batch, int k);
// Projected dense stages of the sparse-lhs sweep (aqc_project.hip, aqc_ws_project.cpp): the state that enters the stages after the
// first is psi (x) |e> on (first stage's local bits) x (the other bits), and those stages touch the qubits T only, so everything the
// gradient needs of z there is its projection y0[i_T, c] = sum_u conj(psi[u, c]) z[u, i_T] (u: local bits of the first stage outside T,
// c: those inside T) -- a register of |T| + |c| virtual qubits per lane instead of n.
struct ProjMap { const unsigned* tab; int shift; };   // index -> element offset: a table, or the index shifted (virtual register)
struct ProjArgs {
    // the product  out[keep, c] = sum_k conj(S[k, c]) Y[k, keep]  per item
    const double2* y;        // full-size operand, [batch][lane_stride]; read at (lane) + (the item's tile bits & ff_mask) + y_keep[keep] + y_k[k]
    const double2* s;        // small operand: real register (lane + the item's tile + s_k[k] + s_c[c]) or virtual ((2 lane + slot) << nvp + ...)
    double2* out;            // the same choice
    int s_virtual, out_virtual;
    int staged;              // k runs along contiguous memory of y and s (low four bits of k = address bits 0..3): fetch 256-byte runs through LDS
    int keep_bits, k_bits, cb;
    ProjMap y_keep, y_k, s_k, s_c, o_keep, o_c;
    size_t lane_stride;
    const TileItem* items;   // first-stage items of the sparse sweep (lane, tile of the first stage, slot) and their number
    const int* nitems;
    const int* lane_parts;   // items per lane
    int nub0, ubits0[32];    // non-local address bits of the first stage (tile index -> element offset)
    unsigned ff_mask, cb_mask, tf_mask;   // address bits outside the first stage's local set and outside T (fixed to the item's) / in both / in T only
    int nvp;
    // project_init_kernel: the virtual lhs pattern and the bookkeeping of the virtual stage launches
    const unsigned* off_t;   // [2^t]: element offset of the value i of the T bits;  [2^cb]: of the value c of the bits shared with the first stage
    const unsigned* off_cb;
    const unsigned* it_of_c; // [2^cb]: index on the T bits whose shared bits hold c (the others 0)
    const unsigned* off_us;  // [2^us]: element offset of the value u of the first stage's local bits outside T
    int us_bits;
    size_t part_stride;      // fused pass with more than 256 summed values: elements between the partial copies of the virtual z
    double2* cpart;          // fused pass with few items: [shares][2 batch][2^us][16] partial tile products (null: never split)
    int cpart_shares;
    int t, ntiles_v;
    double2* vm;             // [batch][2][2^nvp]
    TileItem* vitems;        // items of the virtual stage launches: (lane, slot ntiles_v + tile, the same as partial slot)
    int* vcount;
    int* vlane_parts;
    int batch;
};
hipError_t launch_project_init(const ProjArgs& a, hipStream_t s);
hipError_t launch_project_fused(const ProjArgs& a, const void* mend, void* ctile, void* yout, hipStream_t s);   // both products of the objective by projection, one fetch of y
hipError_t launch_project_amps(const ProjArgs& a, const long long* gather, int ngather, const long long* supp, void* small, const void* vy, const void* z,
                               hipStream_t s);   // the whole gather of an evaluation by projection: inside the lhs tile from z, outside from the virtual z
hipError_t launch_project(const ProjArgs& a, hipStream_t s);
struct UJob {               // one 16 x 16 unitary to build: sub-stage `index` of a plan with `nsubs` sub-stages
    const DevSub3* sub;
    const DevGrp* grps;      // the plan's gate groups
    double* umat;            // the plan's operand buffer [batch][nsubs][12][64]
    int index, nsubs;
    int inverse, entangler;  // V^H plans apply every group conjugate-transposed; 0 cx, 1 cz, 2 cp
    double* umat_mirror;     // mirrored V^H plan (or null): this sub-stage's U^H is sub-stage `mirror_index` of that plan's operand buffer
    int mirror_index, mirror_nsubs;
};
hipError_t launch_ubuild(const UJob* jobs, int njobs, const double* thetas, int T, int batch, hipStream_t s, double* thetas_copy = nullptr);
struct GatherJob {           // optional passenger of the gradient walk: out[lane][i] = buf[lane][elem[i]] (+ pinned host copy)
    const void* buf; size_t lane_stride; const long long* elem; int count; void* out; void* mirror;
};
struct RgradSecond {         // a second plan whose gradient walk rides in the same launch (projected route: the virtual sweep plan); all of its
    const DevSub3* subs;     // sub-stages hold lane_parts[lane] partials (item-list launches), none is R-only
    const DevGrp* grps;
    const double2* rpart;
    const int* lane_parts;
    const double* umat;
    int nparts, nsubs_total, count;   // partial slots per (lane, sub-stage); sub-stages of the plan; how many of them walk here (0: none)
};
hipError_t launch_rgrad(const DevSub3* subs, const DevGrp* grps, int entangler, const double* thetas, int T, const void* rpart,
                        int ntiles, int nsubs_total, void* partial, int nslots, int from, int to, int front, int batch, hipStream_t s,
                        const int* slot_theta = nullptr, void* grads = nullptr, void* mirror = nullptr,
                        GatherJob gather = GatherJob{nullptr, 0, nullptr, 0, nullptr, nullptr},   // slot_theta: direct mode, see rgrad_kernel
                        int nparts = 0, int chunk = 0,   // partial-R slots per (lane, sub-stage) and the persistent sweep's chunk
                        int sparse_subs = 0, const int* lane_parts = nullptr,   // the first sparse_subs sub-stages hold lane_parts[lane] partials (item-list launch)
                        int conj_sub = -1, const double* umat = nullptr,        // sub-stage whose R arrives as Z W^H of its INPUTS: R <- U R U^H first (umat: the plan's operands)
                        int nsubs_run = -1,                                     // walk the first nsubs_run sub-stages only (-1: all)
                        const RgradSecond* second_plan = nullptr);
// Tile lists on the device (aqc_ws_sweep.cpp).  Per lane, the tiles of `stage` that hold the elements supp[lane][0 .. per_lane)
// (the support of the lane's sparse lhs state; -1 = none; may be null) and extra[0 .. nextra) (the same for every lane: the
// registered gather set; may be null), each tile once, in that order -> item list (lane-major, slot = position inside the lane),
// its length, items per lane.  prev_tiles (may be null; [lane][2], per_lane <= 2, no extras): tiles named by the previous list
// of this table that the new one drops -> clear list (they hold stale amplitudes in a buffer that is zero elsewhere); updated.
// One workgroup.
constexpr int kMaxTileCands = 72;   // per_lane + nextra
hipError_t launch_tile_items(const DevStage& stage, const long long* supp, int per_lane, const long long* extra, int nextra, int batch,
                             TileItem* items, int* nitems, int* lane_parts, int* prev_tiles, TileItem* clear_items, int* nclear, hipStream_t s);
hipError_t launch_clear_tiles(const DevStage& stage, void* buf, size_t lane_stride, const TileItem* clear_items, const int* nclear, int max_items,
                              hipStream_t s);

// aqc_lbfgs.hip (device-resident multi-start L-BFGS on the lane-batched surrogate objective)
struct LbState {
    int B, T, S, memory;          // lanes, parameters, gathered flip states, history length
    double *x, *g, *f;            // current point, gradient, value (under the current objective state)
    double *d, *slope, *step;     // search direction, g.d, step length per lane
    double *x_new;                // accepted trial points of the running line search
    double *Smem, *Ymem, *rho;    // history [memory][B][T], [memory][B]
    int *active, *done;
    long long* nit;
    double *weight, *fidelity;   // objective state: smoothed weight, |h_0|^2
    int* max_no;
    double2 *cur_hs, *cur_g0;     // raw device results of the current point: amplitudes, complex gradient of its one sweep
    double2 *acc_hs, *acc_g0;     // ... of the accepted trial points
};
hipError_t lb_prepare(const LbState& st, const void* hs, int update, double* f_out, void* raw_hs, void* x2, size_t lane_stride,
                      const long long* index, long long* prev, hipStream_t s);
hipError_t lb_take(const LbState& st, const void* grads, double* g_out, void* raw_g, hipStream_t s);
hipError_t lb_probe(const LbState& st, const void* hs, int* flags, hipStream_t s);
hipError_t lb_commit0(const LbState& st, const void* hs, const void* raw_g, double* f_out, double* g_out, hipStream_t s);
hipError_t lb_active(const LbState& st, double gtol, double fid_thr, int* flags, hipStream_t s);
hipError_t lb_direction(const LbState& st, int count, hipStream_t s);
hipError_t lb_trial(const LbState& st, double* thetas, hipStream_t s);
hipError_t lb_armijo(const LbState& st, double c1, const double* thetas, const double* ft, const void* raw_hs_t, const void* raw_g0_t,
                     int* flags, hipStream_t s);
hipError_t lb_copy_raw(const LbState& st, hipStream_t s);
hipError_t lb_history(const LbState& st, int count, double ftol, const double* f_acc, const double* g_acc, hipStream_t s);

// aqc_mps.hip
hipError_t launch_mps_scale(void* g, const double* lam, int rows, int cols, hipStream_t s);
hipError_t launch_zgemm(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                        void* C, int ldc, hipStream_t s);
hipError_t launch_zgemm_bh(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                           void* C, int ldc, hipStream_t s);   // C = op(A) . B^H, B stored (N x K)
hipError_t launch_mps_env_dot(const void* e, const void* rc, size_t count, void* out, hipStream_t s);
hipError_t launch_zgemm_batched(bool conj_t, bool accum, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                void* C, int ldc, size_t stride_a, size_t stride_b, size_t stride_c, int nbatch, hipStream_t s);
hipError_t launch_zgemm_tables(int M, int N, int K, const void* const* a_tab, int lda, const void* const* b_tab, int ldb, void* const* c_tab, int ldc,
                               size_t stride_a, size_t stride_b, size_t stride_c, int outer, int inner, hipStream_t s, int b_transposed = 0);
hipError_t launch_mps_product(const void* const* t_tab, void* const* out_tab, int n, int count, hipStream_t s);
struct MpsSites {            // site table of one MPS (by value in the kernel arguments)
    int n;
    size_t total;            // complex elements of all site tensors
    size_t offset[65];       // element offset of site q
    int lam_offset[64];      // offset of lambda_q inside the packed Schmidt vectors
    int cols[64];            // right bond dimension of site q
};
hipError_t launch_mps_scale_all(void* t, const double* lam, const MpsSites& sites, hipStream_t s);

// aqc_cd.hip
int cd_num_parts(size_t npairs);
hipError_t launch_cd_dot(const void* w, const void* z, size_t npairs, int hbit, int kind, void* part, hipStream_t s);
hipError_t launch_cd_update(void* w, void* z, size_t npairs, int hbit, int kind, const void* part, int nparts,
                            const double* theta_in, double* theta_out, int tindex, double dim, hipStream_t s);
hipError_t launch_cd_entangle(void* w, void* z, size_t ngroups, int cbit, int tbit, int ent, hipStream_t s);
// the whole walk as one persistent launch (a workgroup per lane, operands in LDS): segment list entry = CdSeg of aqc_cd.hip
struct CdSegHost { int32_t ha, hb, ent, nrot, kind[4], on_b[4], tindex[4]; };
size_t cd_persistent_lds_bytes(int nbits, int T);
hipError_t launch_cd_persistent(const void* prog, int nsegs, int nbits, int col_bits, const void* target, size_t lane_stride, double* thetas,
                                int T, double* fobj, int nsweeps, int max_steps, int batch, hipStream_t s);


// aqc_gate.hip (gate-level building blocks, one pass per call)
hipError_t launch_gate1q(const void* src, void* dst, int n, size_t ncols, int q, const double* g, hipStream_t s);
hipError_t launch_gate2q(const void* src, void* dst, int n, size_t ncols, int qc, int qt, const double* g, hipStream_t s);
int gate_dot_parts(int n, size_t ncols, int kind);
hipError_t launch_gate_dot(const void* w, const void* z, int n, size_t ncols, int kind, int q0, int q1, void* partial, void* out,
                           hipStream_t s);


// aqc_api.cpp: thread-local message behind aqc_last_error(); returns 1
int set_error(const std::string& msg);

// aqc_svd.hip (one-sided Jacobi SVD + the pieces of a truncated 2-qubit MPS gate)
hipError_t launch_svd_identity(void* V, int cols, hipStream_t s);
hipError_t launch_jacobi_round(void* W, int rows, void* V, int cols, const void* pairs, int npairs, double tol, const double* fro2, int* rotations,
                               hipStream_t s);
hipError_t launch_svd_fro2(const void* W, size_t n, double* out, hipStream_t s);   // out[0] = Frobenius norm squared (fixed-order sum)
hipError_t launch_svd_load(const void* a, int m, int n, int mode, void* work, hipStream_t s);
hipError_t launch_svd_assemble(const void* W, const void* V, const int* ord, const double* sigma, int m, int n, int k, int mode, void* u, void* vh,
                               double* s_sorted, hipStream_t s);
bool svd_fits_small(int rows, int cols);
hipError_t launch_jacobi_small(void* W, int rows, void* V, int cols, double tol, int max_sweeps, int* sweeps_out, double* sigma_out,
                               hipStream_t s);   // sigma_out (may be null): the column norms at the end
bool svd_fits_block(int rows, int cols);
int svd_block_size();
hipError_t launch_jacobi_block(void* W, int rows, void* V, int cols, const void* bpairs, int rounds, int per_round, double tol, int max_sweeps,
                               const double* fro2, int* rot, unsigned* bar, int* status, hipStream_t s);
hipError_t launch_svd_norms(const void* W, int rows, int cols, double* sigma, hipStream_t s);
hipError_t launch_mps_theta(const void* theta0, const double* lam_left, int chil, int chir, const double* g16, int mode, void* work, hipStream_t s);
// ---- device-resident lanes (aqc_mps_batch.cpp): L MPS in flat storage whose bond dimensions live on the device.  A launch takes only
// lane-independent arguments (site, gate kind, parameter indices); a lane's workgroup reads its own bond dimensions and thetas, so the
// host never waits for a rank decision and the whole walk of an evaluation is one uninterrupted sequence of launches.
constexpr int kLaneCap = 32;                        // largest bond dimension of a lane
constexpr int kLaneSite = 2 * kLaneCap * kLaneCap;  // complex elements reserved per site tensor
constexpr int kLaneEnv = kLaneCap * kLaneCap;       // ... per environment
constexpr int kLaneNoConv = 1, kLaneOverflow = 2, kLaneZero = 4, kLaneLdsShort = 8;   // status bits of a lane
struct LaneMps { void* T; double* lam; int* dims; double* discarded; int n, pad; };   // T[L][n][kLaneSite], lam[L][max(n-1,1)][kLaneCap], dims[L][n+1]
struct LaneRot { int kind /* 0 none, 1 rz, 2 ry, 3 rx */, idx; double scale; };        // angle = scale * thetas[lane][idx], or = scale when idx < 0
struct LaneGate1 { LaneRot r[3]; };                                                    // the product r[0] r[1] r[2]
struct LaneGate2 { int kind /* 0 swap, 1 cx, 2 cz, 3 cp */, idx, flip, pad; double scale; };   // cp angle = scale * thetas[lane][idx]; flip: control on site q + 1
struct LaneOp1 { int q, pad; LaneGate1 g; };   // a 1-qubit gate on site q
struct LaneOp2 { int q, pad; LaneGate2 g; };   // a 2-qubit gate on the sites (q, q + 1)
// One gate (`one`), or -- ops != null -- `nops` gates on pairwise disjoint sites from a table in device memory, all in one launch (gates of
// one layer of the circuit touch disjoint tensors and bond dimensions: any order, and so also "at once", gives the same bits).
// b / m2 (may be null): a second state that takes the same gates.
hipError_t launch_lanes_gate1(const LaneMps& a, const LaneMps* b, const LaneOp1* ops, int nops, const LaneOp1& one, const double* thetas, int T, int lanes,
                              int bond_hint, hipStream_t s);
// jstats (may be null): [0] += fp64 flops of the Jacobi rotations actually run (per column pair and sweep: 3 inner products and the
// rotation of the pair over the rows of the work matrix, 36 flop a row, + the rotation of V, 20 flop a row), [1] += SVDs,
// [2] += sweeps, [3] += rotations
hipError_t launch_lanes_gate2(const LaneMps& m, const LaneMps* m2, const LaneOp2* ops, int nops, const LaneOp2& one, const double* thetas, int T,
                              double trunc_thr, int max_bond, int* status, int* peak, int lanes, int bond_hint, hipStream_t s,
                              unsigned long long* jstats = nullptr);
hipError_t launch_lanes_env_left(const LaneMps& w, const LaneMps& z, int p, const void* in, size_t in_stride, void* out, size_t out_stride,
                                 const double* gh8, int lanes, hipStream_t s);
hipError_t launch_lanes_env_right(const LaneMps& w, const LaneMps& z, int p, const void* in, size_t in_stride, void* out, size_t out_stride, int lanes,
                                  hipStream_t s);
hipError_t launch_lanes_env_dot(const LaneMps& w, const LaneMps& z, int hi, const void* e, size_t e_stride, const void* rc, size_t rc_stride, void* vals,
                                int nvals, int slot, int lanes, hipStream_t s);
// up to three parameters on the SAME site: per parameter k the rotation g[k] on site q of both operands, then vals[lane][slot + k] = <P_k w|z>
// (gh[k] = P_k^H, 2 x 2 row-major c128) from the environments L[q] and R[q]
struct LaneSteps { LaneGate1 g[3]; double gh[3][8]; int count, pad; };
hipError_t launch_lanes_grad_step(const LaneMps& w, const LaneMps& z, int q, const LaneSteps& steps, const double* thetas, int T, const void* env_l,
                                  size_t l_stride, const void* env_r, size_t r_stride, void* scratch, void* vals, int nvals, int slot, int lanes, hipStream_t s);
hipError_t launch_lanes_basis(const LaneMps& m, const unsigned char* bits /* [lanes][n] */, int lanes, hipStream_t s);   // product basis states
hipError_t launch_lanes_env_init(void* env_l, size_t l_stride, void* env_r_last, size_t r_stride, int lanes, hipStream_t s);
// environment steps of <(ops) w|z> for small bonds, one launch per site (aqc_svd.hip); gh8: 2x2 (row-major, 4 c128) applied to z's site or null
bool mps_env_fits_small(int xa, int ua, int yb, int vb);
hipError_t launch_mps_env_left(const void* in, const void* A, const void* B, int xa, int ua, int yb, int vb, const double* gh8, void* out, hipStream_t s);
hipError_t launch_mps_env_right(const void* rc, const void* A, const void* B, int xa, int ua, int yb, int vb, void* out, hipStream_t s);
hipError_t launch_mps_theta_fused(const void* tq, const void* tq1, const double* lam_left, int chil, int chim, int chir, const double* g16, int mode,
                                  void* work, hipStream_t s);   // the same from the two site tensors (small bonds: no zgemm launches)
hipError_t launch_mps_split(const void* W, const void* V, const int* ord, const double* sigma, const double* lam_left, int chil, int chir,
                            int k, int mode, double rescale, void* tq, void* tq1, const double* lam_new, double* lam_dst, hipStream_t s);   // lam_new (device, may be null) -> lam_dst[0..k)
hipError_t launch_mps_colscale(void* t, const double* lam, size_t rows, int cols, int mul, hipStream_t s);

}  // namespace aqc

/*
 * aqc_hip.h -- C ABI of the MI355X-native fidelity/gradient path of aqc-research.
 *
 * The reference (qiskit-community/aqc-research v0.1.0) has no FFI: the boundary
 * of this path is a set of Python functions and duck-typed objective objects.
 * Each entry point below names the reference interface it replaces (file:line
 * relative to the reference tree); INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - complex128 arrays travel as interleaved (re, im) doubles;
 *   - qubit q is bit q of the amplitude index (Qiskit order,
 *     core_operations.py:34-43); a (d, k) matrix is row-major
 *     (core_op_matrix.py:56);
 *   - blocks is int32[2][L] row-major: row 0 = control, row 1 = target
 *     (parametric_circuit.py:24-70);
 *   - every function returns 0 on success; on failure a non-zero code is
 *     returned and aqc_last_error() gives the message (the ABI never throws
 *     and never calls back into the host language);
 *   - a context is immutable after creation and may be shared; a workspace
 *     owns one HIP device + one stream and is NOT thread-safe (one workspace
 *     per thread / process, exactly like the reference's one-objective-per-
 *     process model, job_executor.py:141).
 */
#ifndef AQC_HIP_H
#define AQC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aqc_ctx aqc_ctx; /* ansatz description + gate program (host only) */
typedef struct aqc_ws aqc_ws;   /* device-resident batch workspace            */

enum { AQC_CX = 0, AQC_CZ = 1, AQC_CP = 2 };
/* device buffers of a workspace, each [batch][2^n][ncols] complex128: Y target, Z = V^H Y, X / X2 lhs states of the sweep,
 * W / ZW scratch of the sweep.  ZW doubles as the checkpoint of V^H: aqc_ws_apply(inverse, Y -> Z) leaves there the state before
 * its last stage, which is what the sweep's second stage takes as z when the lhs state is sparse (aqc_ws_set_basis /
 * aqc_ws_set_combo; grad_of_dot_product, core_operations.py:892-935, copies x into w: a one-hot x leaves w zero outside one tile
 * during the first stage).  One-call evaluations (aqc_ws_eval, aqc_ws_objective_launch, aqc_ws_surrogate_eval, aqc_ws_lbfgs) may
 * leave Z computed only on the tiles they read; any reader of AQC_BUF_Z through this interface completes it first. */
enum { AQC_BUF_Y = 0, AQC_BUF_Z = 1, AQC_BUF_X = 2, AQC_BUF_W = 3, AQC_BUF_ZW = 4, AQC_BUF_X2 = 5, AQC_NUM_BUFS = 6 };
/* kernel families for aqc_ws_profile_get */
enum { AQC_K_APPLY = 0, AQC_K_SWEEP = 1, AQC_K_COEF = 2, AQC_K_FINALIZE = 3, AQC_K_MISC = 4,
       AQC_K_SWEEP_LIST = 5, AQC_K_APPLY_LIST = 6,   /* stage launches over a subset of the tiles (sparse lhs state / objective V^H) */
       AQC_K_PROJECT = 7, AQC_K_SWEEP_VIRTUAL = 8,   /* projected route: a pass over z / the target, the sweep's later stages on the virtual register, */
       AQC_K_APPLY_VIRTUAL = 9,                      /* their gates (or the inverses) applied to one virtual state (objective by projection) */
       AQC_NUM_KINDS = 10 };

const char* aqc_version(void);
const char* aqc_last_error(void);
/* number of visible HIP devices (0 when HIP is unusable); the reference's counterpart is
 * joblib's worker count, job_executor.py:136-143 */
int aqc_device_count(void);

/* ---- ansatz description: ParametricCircuit / TrotterAnsatz
 *      (parametric_circuit.py:24-70,267-320; validity rules :234-254,391-423) */
int aqc_create(int num_qubits, int entangler, const int32_t* blocks, int num_blocks,
               int trotter, int second_order, aqc_ctx** out);
int aqc_destroy(aqc_ctx* ctx);
int aqc_num_thetas(const aqc_ctx* ctx);          /* parametric_circuit.py:108-112 */
int aqc_num_gate_groups(const aqc_ctx* ctx);     /* G = n + L_eff     */

/* ---- one-shot, host-pointer entry points (function-level drop-ins).
 * vec/out/x/vh_y: complex128[2^n]; grad: complex128[num_thetas]. */
/* core_operations.py:606  v_mul_vec(circ, thetas, vec, out, workspace) */
int aqc_v_mul_vec(aqc_ctx* ctx, const double* thetas, const double* vec, double* out);
/* core_operations.py:713  v_dagger_mul_vec(circ, thetas, vec, out, workspace) */
int aqc_vdag_mul_vec(aqc_ctx* ctx, const double* thetas, const double* vec, double* out);
/* core_operations.py:823  grad_of_dot_product(circ, thetas, x_vec, vh_y_vec, workspace,
 *                          block_range, front_layer); block_from<0 => full range */
int aqc_grad_dot_vec(aqc_ctx* ctx, const double* thetas, const double* x, const double* vh_y,
                     int block_from, int block_to, int front_layer, double* grad);
/* core_op_matrix.py:480 / :562  v_mul_mat / v_dagger_mul_mat(circ, thetas, mat, workspace);
 * mat: complex128[2^n][ncols], updated in place */
int aqc_v_mul_mat(aqc_ctx* ctx, const double* thetas, double* mat, int ncols);
int aqc_vdag_mul_mat(aqc_ctx* ctx, const double* thetas, double* mat, int ncols);
/* core_op_matrix.py:645  grad_of_matrix_dot_product(circ, thetas, x_mat, vh_y_mat, workspace);
 * unlike the reference the inputs are left intact (the Python shim reproduces the clobbering
 * contract where callers rely on it) */
int aqc_grad_dot_mat(aqc_ctx* ctx, const double* thetas, const double* x_mat, const double* vh_y_mat,
                     int ncols, double* grad);

/* ---- device-resident batch workspace: `batch` independent evaluations
 * (own thetas, own target) advance together in every launch.  This is what the
 * objective objects (objective_lhs_sur_max.py:82-191, sk_core.py:167-222) and
 * the job executor (job_executor.py:96) sit on.
 * ncols = 1 for state vectors, k for (2^n x k) matrices.
 * tile_bits_* = 0 selects the default LDS tile size. */
int aqc_ws_create(aqc_ctx* ctx, int device, int batch, int ncols, int tile_bits_apply,
                  int tile_bits_sweep, aqc_ws** out);
int aqc_ws_destroy(aqc_ws* ws);
int aqc_ws_set_thetas(aqc_ws* ws, const double* thetas /* [batch][T] */);
int aqc_ws_upload(aqc_ws* ws, int buf, const double* src /* [batch][2^n][ncols] c128 */);
int aqc_ws_upload_lane(aqc_ws* ws, int buf, int lane, const double* src);
int aqc_ws_broadcast(aqc_ws* ws, int buf, const double* src /* [2^n][ncols] c128, same for all lanes */);
int aqc_ws_download(aqc_ws* ws, int buf, double* dst);
int aqc_ws_download_lane(aqc_ws* ws, int buf, int lane, double* dst);
/* X_lane <- one-hot basis state |index[lane]>  (ThinStateHandler.init_state, objective_base.py:99-116) */
/* dst_ws.buf[dst_lane] <- src_ws.buf[src_lane] on the device (same device, same lane size; asynchronous on dst's stream) */
int aqc_ws_copy_lane(aqc_ws* dst_ws, int dst_buf, int dst_lane, aqc_ws* src_ws, int src_buf, int src_lane);
int aqc_ws_set_basis(aqc_ws* ws, int buf, const int64_t* index /* [batch] */);
/* buffer[lane] = coef[lane][0] |index[lane][0]> + coef[lane][1] |index[lane][1]> (index[lane][1] < 0: one term).  The
 * gradient of <V x|y> is conjugate-linear in x, so the two sweeps of the surrogate objective (|state_0> and the leading
 * flip state, objective_lhs_sur_max.py:147-155,167-175) combined as c_0 g_0 + c_max g_max are ONE sweep from
 * x = conj(c_0) |state_0> + conj(c_max) |state_max>. */
int aqc_ws_set_combo(aqc_ws* ws, int buf, const int64_t* index /* [batch][2] */, const double* coef /* [batch][2] c128 */);
/* X <- identity matrix (FullRangeSketchingVectors.generate, sk_core.py:317-326); needs ncols == 2^n */
int aqc_ws_set_identity(aqc_ws* ws, int buf);
/* dst <- V src (inverse=0) or V^H src (inverse=1), per lane, with the lane's thetas */
int aqc_ws_apply(aqc_ws* ws, int inverse, int src_buf, int dst_buf);
/* grads <- complex gradient of <V X|Y> from X and Z = V^H Y (kept intact); W/ZW are scratch */
int aqc_ws_grad(aqc_ws* ws, int block_from, int block_to, int front_layer);
int aqc_ws_get_grads(aqc_ws* ws, double* grads /* [batch][T] c128 */);
/* out[lane][i] = buf[lane][index[i]]  (ThinStateHandler.state_dot_vector, objective_base.py:166-177) */
int aqc_ws_gather(aqc_ws* ws, int buf, const int64_t* index, int count, double* out);
/* out[lane] = <a_lane|b_lane> (np.vdot; GenericStateHandler.state_dot_vector, sk_core.py:192) */
int aqc_ws_vdot(aqc_ws* ws, int buf_a, int buf_b, double* out /* [batch] c128 */);
int aqc_ws_sync(aqc_ws* ws);

/* ---- one-call evaluation: what objective(thetas) + gradient(thetas) of the objective objects need
 * (objective_lhs_sur_max.py:82-191), with a single host synchronisation:
 *   thetas -> device, coefficients, [Z = V^H Y], [gather of the amplitudes selected by
 *   aqc_ws_gather_setup from Z], [sweep with lhs buffer x_buf -> grads], results -> host.
 * Any of gathered / grads may be NULL to skip that part.  Host buffers are staged through pinned
 * memory owned by the workspace. */
int aqc_ws_eval(aqc_ws* ws, const double* thetas /* [batch][T] or NULL = keep */, int do_vdag,
                double* gathered /* [batch][count] c128 */, int x_buf, int block_from, int block_to,
                int front_layer, double* grads /* [batch][T] c128 */);
/* sweep with an explicit lhs buffer (aqc_ws_grad uses AQC_BUF_X) */
int aqc_ws_grad_from(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer);

/* ---- fully asynchronous variants: inputs stay resident in HBM, nothing below synchronises.
 * A theta bank holds `nsets` parameter sets [nsets][batch][T]; selecting one only launches the
 * coefficient kernel.  Gathered amplitudes stay on the device until fetched. */
int aqc_ws_theta_bank(aqc_ws* ws, const double* thetas, int nsets);
int aqc_ws_use_theta_set(aqc_ws* ws, int set_index);
int aqc_ws_gather_setup(aqc_ws* ws, const int64_t* index, int count);
int aqc_ws_gather_launch(aqc_ws* ws, int buf);
int aqc_ws_gather_fetch(aqc_ws* ws, double* out /* [batch][count] c128 */);
int aqc_ws_vdot_launch(aqc_ws* ws, int buf_a, int buf_b);
int aqc_ws_vdot_fetch(aqc_ws* ws, double* out /* [batch] c128 */);
/* what the optimizer on the host reads after every evaluation (optimizer.py:579-590 consumes objective(theta) and
 * gradient(theta)): aqc_ws_results_async enqueues the copies of the gradients and of the gathered amplitudes (or of <A|B>
 * after aqc_ws_vdot_launch) into pinned host memory behind the kernels already on the stream, without synchronising;
 * aqc_ws_results_fetch waits for the stream and hands them out (either pointer may be NULL). */
int aqc_ws_results_async(aqc_ws* ws);
/* one evaluation of an objective with the thetas in use, enqueued and not waited for: Z = V^H Y (where the evaluation reads it),
 * the amplitudes registered with aqc_ws_gather_setup (if any; objective_lhs_sur_max.py:99-106), the sweep from x_buf
 * (:147-175).  Same results as aqc_ws_apply(1, Y, Z); aqc_ws_gather_launch(Z); aqc_ws_grad_from(x_buf, ...). */
int aqc_ws_objective_launch(aqc_ws* ws, int x_buf, int block_from, int block_to, int front_layer);
int aqc_ws_results_fetch(aqc_ws* ws, double* small_out /* [batch][count] c128 */, double* grads_out /* [batch][T] c128 */);

/* ---- MPS helpers (state-vector workspaces only).  An MPS arrives in the reference's QiskitMPS
 * layout (mps_operations.py:33,87-123): site q has Gamma^0, Gamma^1 of shape (dims[q], dims[q+1]),
 * dims[0] = dims[n] = 1, and lambda_q (float64[dims[q+1]]) for q < n-1.
 *   gammas : per site [2][dims[q]][dims[q+1]] complex128, sites concatenated
 *   lambdas: per site [dims[q+1]] float64, sites 0..n-2 concatenated
 * Upload folds lambda into the right bond of Gamma on the device (_preprocess_mps, :126-156). */
enum { AQC_MPS_SLOTS = 64 };   /* 0..3: explicit calls; the rest: resident copies managed by the host side (engine.py slot cache) */
int aqc_ws_mps_upload(aqc_ws* ws, int slot, const int32_t* dims /* [n+1] */, const double* gammas,
                      const double* lambdas);
/* buf[lane] <- dense state of the MPS; index bit q <-> site q   (mps_to_vector, :159-189) */
int aqc_ws_mps_to_vec(aqc_ws* ws, int slot, int buf, int lane);
/* the same for `count` (slot, lane) pairs; slots of equal bond dimensions (the lanes of a batched objective,
 * mps_dot_objective.py:41 called once per lane in the reference) share every launch of the contraction chain */
int aqc_ws_mps_to_vec_batch(aqc_ws* ws, int count, const int32_t* slots, int buf, const int32_t* lanes);
/* out <- <mps_a|mps_b> by transfer matrices                      (mps_dot, :192-213) */
int aqc_ws_mps_dot(aqc_ws* ws, int slot_a, int slot_b, double* out /* 1 c128 */);

/* ---- dense complex GEMM on the device: C (M x N) = op(A) B with op(A) = A (M x K) or A^H (A stored K x M),
 * row-major, host pointers.  Used by the sketching-vector generators (Y = U X, sk_core.py:357,463) and by
 * the MPS contractions internally. */
int aqc_zgemm(int device, int conj_trans_a, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
              double* C, int ldc);

/* ---- gate-level building blocks, one pass per call over a (2^n x ncols) row-major c128 array in HOST memory
 * (ncols = 1: state vector).  Qubit numbers are bit indices of the row (Qiskit order); the Python shim converts
 * the big-endian `pos` of the state-vector functions (core_operations.py:34-43).  dst may equal src.
 * aqc_gate_1q: dst = (I x g x I) src, g = row-major 2x2 (4 c128)
 *   replaces gate2x2_mul_vec :46, proj00/11_mul_vec :122,:143, rx/ry/rz_mul_vec :164,:200,:236 of core_operations.py
 *   and rx/ry/rz_mul_mat :32,:66,:100, gate2x2_mul_mat :392 of core_op_matrix.py.
 * aqc_gate_2q: dst = (4x4 on the pair, index 2*bit_ctrl + bit_targ; 16 c128 row-major) src
 *   replaces cx/cz/cp_mul_vec :422,:468,:514, derv_cphase_mul_vec :561, block_mul_vec :354 (core_operations.py)
 *   and cx/cz/cp_mul_mat :130,:181,:232 (core_op_matrix.py).
 * aqc_gate_dot: kind 0 / 1 / 2 = 0.5j<Xw|z>, 0.5j<Yw|z>, 0.5j<Zw|z> on qubit q0 (dot_x/y/z :267,:296,:325;
 *   x/y/z_dot_mat :284,:320,:356); kind 3 = -1j<P11(q0,q1) w|z> (derv_cphase, core_op_matrix.py:430). */
int aqc_gate_1q(int device, int n, int64_t ncols, int qubit, const double* gate, const double* src, double* dst);
int aqc_gate_2q(int device, int n, int64_t ncols, int ctrl, int targ, const double* gate, const double* src, double* dst);
int aqc_gate_dot(int device, int n, int64_t ncols, int kind, int q0, int q1, const double* w, const double* z,
                 double* out /* 1 c128 */);

/* ---- device-resident MPS with truncated 2-qubit gates: the arithmetic the reference hands to qiskit-aer's
 * matrix_product_state simulator (mps_operations.py:216-298 mps_from_circuit / qcircuit_mul_mps; gate level:
 * mps_dot_objective.py:245-468).  Qiskit MPS format in and out (mps_operations.py:33): `dims` = n+1 bond
 * dimensions (dims[0] = dims[n] = 1), `gammas` = per site [2][dims[q]][dims[q+1]] c128 packed back to back,
 * `lambdas` = the n-1 Schmidt vectors packed back to back.  trunc_thr: the smallest singular values are dropped
 * while the sum of their squares stays below it (Aer's matrix_product_state_truncation_threshold); max_bond <= 0
 * means unlimited.  Aer's own arithmetic is third party: truncated results are parity unpinned. */
typedef struct aqc_mps aqc_mps;
int aqc_mps_create(int device, int n, const int32_t* dims, const double* gammas, const double* lambdas, aqc_mps** out);
int aqc_mps_destroy(aqc_mps* mps);
int aqc_mps_clone(const aqc_mps* src, aqc_mps** out);
int aqc_mps_num_qubits(const aqc_mps* mps);
int aqc_mps_device(const aqc_mps* mps);   /* the HIP device the state lives on */
int aqc_mps_dims(const aqc_mps* mps, int32_t* dims /* n+1 */);
double aqc_mps_discarded_weight(const aqc_mps* mps);
int aqc_mps_export(aqc_mps* mps, double* gammas, double* lambdas);
/* {x,y,z,rx,ry,rz}_mul_mps (mps_dot_objective.py:245-377): 2x2 gate, row-major 4 c128 */
int aqc_mps_gate1(aqc_mps* mps, int qubit, const double* gate);
/* {cx,cp,cz}_mul_mps (:380-468) and any other 4x4 (index 2*bit_ctrl + bit_targ, 16 c128) on any qubit pair */
int aqc_mps_gate2(aqc_mps* mps, int ctrl, int targ, const double* gate, double trunc_thr, int max_bond);
/* <a|b>  (mps_dot, mps_operations.py:192-213) */
int aqc_mps_dot(aqc_mps* a, aqc_mps* b, double* out /* 1 c128 */);
/* <(prod_i G_i on qubits[i]) a|b> without forming G.a: dot_{x,y,z} (mps_dot_objective.py:471-516) up to the factor
 * 0.5j with one Pauli; two projectors |1><1| give the CPhase derivative term.  gates: nops 2x2 matrices (4 c128 each) */
int aqc_mps_dot_ops(aqc_mps* a, aqc_mps* b, int nops, const int32_t* qubits, const double* gates, double* out);
/* ansatz description for the MPS entry points (ParametricCircuit / TrotterAnsatz, parametric_circuit.py:24-70,267-333);
 * unlike aqc_ctx it is not limited to dense-reachable registers.  blocks: int32[2][num_blocks], row 0 = control */
typedef struct aqc_circuit {
    int32_t num_qubits, entangler /* AQC_CX / AQC_CZ / AQC_CP */, num_blocks, trotter, second_order;
    const int32_t* blocks;
} aqc_circuit;
/* v_mul_mps / v_dagger_mul_mps(circ, thetas, mps, trunc_thr) (mps_operations.py:326-371): the whole ansatz (inverse = 1:
 * its conjugate transpose) applied in place, one call instead of one per gate */
int aqc_mps_apply_circuit(aqc_mps* mps, const aqc_circuit* circ, const double* thetas, int inverse, double trunc_thr, int max_bond);
/* fast_dot_gradient(circ, thetas, lvec, vh_phi, trunc_thr, block_range, front_layer) (mps_dot_objective.py:41-242):
 * complex gradient (c128[num_thetas]) of <V lvec|phi> from vh_phi = V^H|phi>.  One call walks all gates on device copies
 * of both operands; the ~num_thetas inner products 0.5j<P w|z> reuse cached left / right environments of the pair and
 * come back in a single transfer.  block_from < 0: all blocks.  lvec and vh_phi are left intact. */
int aqc_mps_fast_dot_gradient(const aqc_circuit* circ, const aqc_mps* lvec, const aqc_mps* vh_phi, const double* thetas,
                              double trunc_thr, int max_bond, int block_from, int block_to, int front_layer, double* grad);
/* ---- lockstep lanes of the MPS objective: `lanes` independent problems that share one ansatz (the seeds / restarts / targets
 * the reference evaluates one per job, job_executor.py:141 around mps_dot_objective.py:41) evaluated TOGETHER and device-resident:
 * every step of the gate walk is one launch for all lanes (grid dimension = lane); a truncated 2-qubit gate is one workgroup per
 * lane (two-site tensor, Jacobi SVD, rank / truncation decision, new tensors -- the lane's bond dimensions never leave the device);
 * gate matrices are formed by the kernels from thetas[lane][index].  The host enqueues the whole evaluation without waiting and
 * reads all results in one transfer.  Bonds up to 32 per lane; a lane whose bond would grow beyond that makes the call fail (never
 * a silent truncation) and the caller falls back to aqc_mps_fast_dot_gradient lane by lane.  Truncation rule and outputs per lane
 * are those of aqc_mps_apply_circuit + aqc_mps_dot + aqc_mps_fast_dot_gradient (sums of the rule are taken in a different order:
 * last-bit differences). */
typedef struct aqc_mpsb aqc_mpsb;
int aqc_mpsb_create(int device, int num_qubits, int lanes, aqc_mpsb** out);
int aqc_mpsb_destroy(aqc_mpsb* b);
/* work and time of the truncated 2-qubit gates of the lanes (the SVDs that qiskit-aer runs per gate, mps_operations.py:252-257):
 * enable 1 / 0 / -1 (leave); out[6] = fp64 flops of the Jacobi rotations that ran, SVDs, sweeps, rotations, launches timed, their ms */
int aqc_mpsb_gate2_stats(aqc_mpsb* b, int enable, double* out, int reset);
/* |phi_l> of every lane (the objective's target, objective_base.py:112 set_target) and <lhs_l| (its left-hand state |0> or the
 * surrogate's low-entangled state, objective_lhs_sur_fast_mps_trotter.py:99); shared != 0: handles[0] serves every lane */
int aqc_mpsb_set_targets(aqc_mpsb* b, aqc_mps* const* handles, int shared);
int aqc_mpsb_set_lhs(aqc_mpsb* b, aqc_mps* const* handles, int shared);
/* lhs states of all lanes = computational-basis states built on the device: bits[lane][n] (0 / 1), bit q = qubit q (the |state_i> of
 * objective_base.py:42-255: |0>, X_q|0> on top of a basis preparation such as the Neel pattern) */
int aqc_mpsb_set_lhs_basis(aqc_mpsb* b, const uint8_t* bits);
/* per lane l with thetas[l][T]: vh = V(theta_l)^H|phi_l> (v_dagger_mul_mps, mps_operations.py:350), h[l] = <lhs_l|vh> (c128) and
 * grad[l][T] (c128) = fast_dot_gradient(circ, theta_l, lhs_l, vh, trunc_thr, block_range, front_layer) (mps_dot_objective.py:41).
 * discarded[l] (optional) = weight truncated while forming vh, max_bond_out[l] (optional) = its largest bond. */
int aqc_mpsb_eval(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int block_from,
                  int block_to, int front_layer, double* h /* [lanes] c128 */, double* grad /* [lanes][T] c128 */,
                  double* discarded /* [lanes] or NULL */, int32_t* max_bond_out /* [lanes] or NULL */);
/* v_mul_mps / v_dagger_mul_mps(circ, thetas, mps, trunc_thr) (mps_operations.py:326-371) for every lane: the batch's working state <-
 * V(theta_l)|phi_l> (inverse = 0) or V(theta_l)^H|phi_l> (inverse = 1), the gates of a circuit layer in one launch; aqc_mpsb_export hands
 * lane `lane` of it out as a single-lane MPS of its own (caller destroys it). */
int aqc_mpsb_apply_circuit(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, int inverse, double trunc_thr, int max_bond,
                           double* discarded /* [lanes] or NULL */, int32_t* max_bond_out /* [lanes] or NULL */);
int aqc_mpsb_export(aqc_mpsb* b, int lane, aqc_mps** out);
/* The same in two phases, for objectives that choose the lhs state after seeing amplitudes (objective_lhs_sur_max.py:82-191: the
 * leading flip state).  Phase 1: vh of every lane, kept in the batch, and amps[lane][0] = <lhs_l|vh_l>, amps[lane][1 + q] =
 * <X_q lhs_l|vh_l> (num_amps = 1 or 1 + n; with a basis state as lhs: the amplitudes of its single-flip states, :99-111).
 * half != 0: lanes [lanes/2, lanes) repeat targets and thetas of the first half (the same problems seen from a second lhs state):
 * V^H runs on the first half only and is copied.  Phase 2: the gradient walk from the CURRENT lhs states (they may have been
 * replaced since phase 1) and the vh, thetas, trunc_thr, max_bond of phase 1 (same circuit). */
int aqc_mpsb_vh(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int half, int num_amps,
                double* amps /* [lanes][num_amps] c128 */, double* discarded /* [lanes] or NULL */, int32_t* max_bond_out /* or NULL */);
int aqc_mpsb_grad(aqc_mpsb* b, const aqc_circuit* circ, int block_from, int block_to, int front_layer, double* grad /* [lanes][T] c128 */);
/* fast_dot_gradient with vh_phi formed by the caller, as the reference's function takes it (mps_dot_objective.py:41): the states given to
 * aqc_mpsb_set_targets ARE vh_phi_l = V^H|phi_l>, those given to aqc_mpsb_set_lhs the lvec_l */
int aqc_mpsb_gradient_of(aqc_mpsb* b, const aqc_circuit* circ, const double* thetas, double trunc_thr, int max_bond, int block_from,
                         int block_to, int front_layer, double* grad /* [lanes][T] c128 */);
/* one-sided Jacobi SVD on the device (the kernel behind aqc_mps_gate2): A (m x n row-major) = U diag(S) Vh,
 * k = min(m, n), S descending, U (m x k), Vh (k x n); *sweeps (optional) = Jacobi sweeps used */
int aqc_svd(int device, int m, int n, const double* a, double* u, double* s, double* vh, int* sweeps);

/* ---- coordinate descent (core_op_matrix.py:765  coord_descent_single_sweep(circ, thetas, target,
 * workspace)).  Square workspace (ncols == 2^n) with the target unitary in AQC_BUF_Y.  One
 * Gauss-Seidel sweep over all parameters of 1 - |<V,U>|^2/d^2; thetas are updated in place and
 * the objective at the end of the sweep is returned.  cx / cz entanglers only (:818-827). */
int aqc_ws_cd_sweep(aqc_ws* ws, double* thetas_io /* [T] */, double* fobj);
/* The same walk for EVERY lane of the workspace (lane = an independent problem: a random restart of the ansatz and / or its
 * own target in AQC_BUF_Y) and for `nsweeps` consecutive sweeps, as ONE persistent launch: a workgroup per lane keeps the two
 * d x d operands in LDS, re-derives z = V(theta)^H U at the start of every sweep (:806-810) and walks all parameters (:852-912)
 * without leaving the kernel.  Needs 2 d^2 16 B + 24 T B <= 160 KiB (up to 6 qubits; aqc_ws_cd_fits_one_launch); larger
 * problems keep the launch chain of aqc_ws_cd_sweep (one lane).  fobj[lane][s] = the objective at the end of sweep s (:917).
 * max_steps >= 0 stops every sweep's walk after that many parameters (tests pin single steps with it); -1 = all. */
int aqc_ws_cd_sweeps(aqc_ws* ws, double* thetas_io /* [batch][T] */, double* fobj /* [batch][nsweeps] */, int nsweeps, int max_steps);
int aqc_ws_cd_fits_one_launch(const aqc_ws* ws);

/* ---- measurement hooks (bench.py): HIP events on the workspace's own stream */
int aqc_ws_timer_start(aqc_ws* ws);
int aqc_ws_timer_stop(aqc_ws* ws, float* elapsed_ms); /* synchronises */
/* per-kernel-family timing: when enabled every launch is bracketed by events */
int aqc_ws_profile_enable(aqc_ws* ws, int on);
int aqc_ws_profile_get(aqc_ws* ws, int kind, int64_t* launches, double* total_ms);
int aqc_ws_profile_reset(aqc_ws* ws);
/* the launches profiled since the last reset, in order: kinds[i], ms[i] for i < min(*count, cap); *count = their number */
int aqc_ws_profile_log(aqc_ws* ws, int32_t* kinds, double* ms, int cap, int* count);
/* stage `stage` of plan `which` as this workspace runs it: sub-stages, gate groups, local address bits (bits_out: [tile_bits]) */
int aqc_ws_plan_stage(aqc_ws* ws, int which, int stage, int* num_subs, int* num_groups, int* bits_out);
/* per sub-stage of that stage, what a sweep from ONE basis state per lane leaves out (out[sub][2]): log2 of the share of 16-chunk
 * groups whose W and R products are issued, log2 of the share of K-steps of the W product (both <= 0); zeros when nothing is skipped */
int aqc_ws_plan_skips(aqc_ws* ws, int which, int stage, int* out, int max_subs);
/* index (over all stages) of the sweep's sub-stage that is taken from its inputs alone -- the last one of the last stage,
 * R = U (Z W^H) U^H: a third of a sub-stage's matrix work -- or -1 when the sweep runs it like the others */
int aqc_ws_sweep_r_only_sub(aqc_ws* ws);
/* item lists of the last sparse evaluation (synchronises): counts[0] first-stage items of the sweep, [1] tiles it cleared in W,
 * [2] last-stage items of V^H; -1 where that list has never been built */
int aqc_ws_sparse_counts(aqc_ws* ws, int64_t* counts);
/* projected route of the sparse-lhs sweep (the stages after the first on a virtual register, csrc/aqc_ws_project.cpp): info[0] 1 if
 * the workspace has it, [1] virtual qubits, [2] qubits the later stages touch, [3] of them local to the first stage, [4] stages and
 * [5] sub-stages of the virtual plan, [6] its tile bits, [7] sub-stages left on the real register, [8] bits of the first stage the
 * pass over z sums over, [9] virtual qubits after padding (>= 8), [10..15] sub-stages of the virtual stages; info holds 16 entries */
int aqc_ws_projected_info(aqc_ws* ws, int32_t* info);
/* host-only (no GPU needed): the same for the state-vector workspace a context would get at this tiling */
int aqc_plan_projected(aqc_ctx* ctx, int tile_bits, int low_bits, int32_t* info);
/* plan introspection: number of fused stages (kernel launches) of V^H and of the sweep */
int aqc_ws_plan_info(aqc_ws* ws, int which /*0 apply-inverse, 1 sweep, 2 apply-forward*/,
                     int* num_stages, int* tile_bits, int* num_tiles);
/* number of sub-stages (16 x 16 unitaries per lane on the matrix-core path) of plan `which`; 0 for the per-group kernels */
int aqc_ws_plan_substages(aqc_ws* ws, int which);
/* kernel family that runs plan `which`: 1 per-gate-group (VALU), 2 register-blocked (VALU), 3 matrix-core (MFMA) */
int aqc_ws_kernel_family(aqc_ws* ws, int which);
/* host-only planner introspection (no GPU needed): stage s of plan `which` for the given tiling (which: 0 V^H planned on its
 * own, 1 sweep, 2 V, 3 V^H as the sweep's plan walked backwards -- what matrix-core workspaces run, see AQC_BUF_ZW above);
 * ops_out receives gate-group indices (forward program order), bits_out the local address bits */
int aqc_plan_query(aqc_ctx* ctx, int ncols, int which, int tile_bits, int low_bits, int stage,
                   int* num_stages, int* bits_out, int* num_bits, int* ops_out, int* num_ops);

/* host-only: the sub-stages of stage `stage` as the matrix-core kernels run it: subs_out[i] = {4 register bits as local
 * positions inside the stage's tile (-1: unused), number of gate groups}; at most max_subs entries are written */
int aqc_plan_substages(aqc_ctx* ctx, int ncols, int which, int tile_bits, int low_bits, int stage, int* num_subs,
                       int* subs_out /* [max_subs][5] */, int max_subs);

/* ---- device-resident multi-start L-BFGS on the lane-batched surrogate objective: one optimisation per lane, thetas /
 * gradients / history stay in HBM (stand-in for the scipy L-BFGS-B behind AqcOptimizer.optimize, optimizer.py:579-590,
 * on objective_lhs_sur_max.py:82-191).  Preconditions: targets in buffer Y, flip-state indices registered with
 * aqc_ws_gather_setup (state 0 first); buffer X2 is used for the lhs states.  An evaluation is V^H, the amplitudes and ONE
 * sweep from conj(c_0)|state_0> + conj(c_max)|state_max> (see aqc_ws_set_combo).  block_from / block_to / front_layer as in
 * aqc_ws_grad (block_from < 0: all blocks).  x0 / x_out: [batch][T]; f / fidelity / nit / weight / max_no: [batch]; the
 * last two are the objective's state at exit (smoothed weight, index of the leading state), either may be NULL. */
/* One evaluation of that objective as a call of its own (the evaluate step of aqc_ws_lbfgs; stand-in for one
 * objective(theta) + gradient(theta) pair of objective_lhs_sur_max.py:82-191 on every lane, one host synchronisation):
 * thetas [batch][T]; weight_io / max_no_io [batch]: the objective state of every lane (smoothed weight, index of the leading
 * state), updated in place when update_state != 0 -- 1: hysteresis :113-117 and smoothing :186 FIRST, then value and gradient
 * under the new state (one accepted optimizer step); 2: hysteresis only, the weight stays (what objective(theta) does before
 * gradient(theta) moves the weight) -- and left alone when it is 0 (line-search trials); f_out [batch]; fidelity_out [batch] (|h_0|^2, written on an
 * update; may be NULL); hs_out [batch][states] complex amplitudes (may be NULL); grads_out [batch][T] COMPLEX gradient of the
 * lane's one sweep from conj(c_0)|state_0> + conj(c_max)|state_max> (may be NULL); grad_real_out [batch][T] its real part,
 * which is the surrogate's gradient (may be NULL; with grads_out NULL only the real parts cross the bus). */
int aqc_ws_surrogate_eval(aqc_ws* ws, const double* thetas, int update_state, double* weight_io, int64_t* max_no_io,
                          int block_from, int block_to, int front_layer, double* f_out, double* fidelity_out, double* hs_out,
                          double* grads_out, double* grad_real_out);
int aqc_ws_lbfgs(aqc_ws* ws, const double* x0, int maxiter, int memory, double gtol, double ftol, double fid_thr,
                 int max_backtracks, int block_from, int block_to, int front_layer, double* x_out, double* f_out,
                 double* fidelity_out, int64_t* nit_out, int64_t* nfev_out, double* weight_out, int64_t* max_no_out);

/* ---- multi-GPU: the one collective layer of the path, bound straight to librccl (RCCL over xGMI; loaded lazily).
 * One process per GPU; jobs are sharded over the ranks (job_executor.py:136-143: joblib processes in the reference) and
 * only fixed-size result records cross GPUs.  All buffers are HOST pointers (the records are a few KB).
 * Bootstrap: rank 0 calls aqc_comm_unique_id and passes the 128 bytes to the other ranks (file / environment),
 * every rank calls aqc_comm_create. */
typedef struct aqc_comm aqc_comm;
int aqc_comm_unique_id(char* out128);
int aqc_comm_create(const char* id128, int nranks, int rank, int device, aqc_comm** out);
int aqc_comm_destroy(aqc_comm* comm);
int aqc_comm_rank(const aqc_comm* comm);
int aqc_comm_size(const aqc_comm* comm);
/* recv[r * count + i] = send[i] of rank r (the final gather of run_jobs: job_executor.py:141-161) */
int aqc_comm_allgather(aqc_comm* comm, const double* send, double* recv, size_t count);
/* in place, op 0 = sum (column-sharded AQC objective: (trace, gradient) record), 2 = max (timing) */
int aqc_comm_allreduce(aqc_comm* comm, double* data, size_t count, int op);
int aqc_comm_barrier(aqc_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* AQC_HIP_H */

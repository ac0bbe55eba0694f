"""Native MPS engine at large bonds: V^H|phi> and the gate-by-gate gradient at n = 16 against the dense route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd import mps_engine as eng
from oracle import aqc_oracle as orc
from oracle import aqc_ref as cref

n = 16
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(1)
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
th = orc.rand_thetas(circ.num_thetas, rng)
def canonical_mps(vec, cap):
    """Vidal form (Gamma, lambda) of a dense state by successive SVDs, bonds capped at `cap` (then renormalised)."""
    nq = int(np.log2(vec.size))
    rest = vec.reshape([2] * nq).transpose(list(range(nq - 1, -1, -1))).reshape(1, -1)   # axes (b_0, ..., b_{n-1})
    gam, lam, prev = [], [], np.ones(1)
    for q in range(nq - 1):
        chi_l = rest.shape[0]
        u, s_, vh_ = np.linalg.svd(rest.reshape(chi_l * 2, -1), full_matrices=False)
        k = min(cap, int((s_ > 1e-14 * s_[0]).sum()))
        u, s_, vh_ = u[:, :k], s_[:k] / np.linalg.norm(s_[:k]), vh_[:k]
        a = u.reshape(chi_l, 2, k)
        gam.append((a[:, 0, :] / prev[:, None], a[:, 1, :] / prev[:, None]))
        lam.append(s_.copy())
        prev = s_
        rest = s_[:, None] * vh_
    a = rest.reshape(rest.shape[0], 2, 1)
    gam.append((a[:, 0, :] / prev[:, None], a[:, 1, :] / prev[:, None]))
    return gam, lam


raw = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
phi = canonical_mps(raw / np.linalg.norm(raw), chi)
dense = orc.mps_to_vector(phi)
print("canonical MPS bonds", max(l.size for l in phi[1]), "norm", np.linalg.norm(dense), flush=True)
m = eng.DeviceMPS.from_qiskit(phi)
vh = eng.v_dagger_mul_mps(circ, th, m, trunc_thr=1e-16)   # warm-up (module load, buffer growth)
t0 = time.perf_counter()
vh = eng.v_dagger_mul_mps(circ, th, m, trunc_thr=1e-16)
t1 = time.perf_counter()
ref = cref.v_dagger_mul_vec(circ, th, dense)
from aqc_research_amd.mps_operations import mps_to_vector
got = mps_to_vector(vh.to_qiskit())
print(f"chi={chi} L={L}: V^H on the engine {t1 - t0:.2f} s, bonds {vh.bond_dims.max()}, max |diff| vs dense {np.abs(got - ref).max():.2e}", flush=True)
zero = eng.DeviceMPS.basis_state(n, 0)
g = eng.fast_dot_gradient_mps(circ, th, zero, vh, trunc_thr=1e-16)
t0 = time.perf_counter()
g = eng.fast_dot_gradient_mps(circ, th, zero, vh, trunc_thr=1e-16)
t1 = time.perf_counter()
if os.environ.get("AQC_PROBE_GATEWISE"):
    t2 = time.perf_counter()
    g2 = eng.fast_dot_gradient_mps_gatewise(circ, th, zero, vh, trunc_thr=1e-16)
    print(f"   gate-per-call walk {time.perf_counter() - t2:.2f} s, max |diff| {np.abs(g - g2).max():.2e}")
x = np.zeros(1 << n, complex); x[0] = 1
gref = cref.grad_of_dot_product(circ, th, x, ref)
print(f"   gradient {t1 - t0:.2f} s, max |diff| {np.abs(g - gref).max():.2e}", flush=True)

import sys; sys.path.insert(0, ".")
import numpy as np, time
from oracle import aqc_oracle as orc
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.mps_engine import DeviceMPS, v_mul_mps, v_dagger_mul_mps, fast_dot_gradient_mps
rng = np.random.default_rng(3)
n = 10
for L, sc, thr in ((54, 0.25, 1e-5), (54, 0.35, 1e-6), (72, 0.2, 1e-6), (54, 1.0, 1e-4)):
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
    th = sc * orc.rand_thetas(circ.num_thetas, rng)
    zero = DeviceMPS.basis_state(n)
    t = time.time(); exact = v_mul_mps(circ, th, zero); te = time.time() - t
    cut = v_mul_mps(circ, th, zero, trunc_thr=thr)
    print(L, sc, thr, "exact", exact.bond_dims, "cut", cut.bond_dims, "disc", cut.discarded_weight, "infid", 1 - abs(exact.dot(cut)) ** 2, "norm", abs(cut.dot(cut)), f"{te:.2f}s")
n = 40
circ = ParametricCircuit(n, "cz", create_ansatz_structure(n, "spin", "full", n - 1))
th = 0.5 * orc.rand_thetas(circ.num_thetas, rng)
zero = DeviceMPS.basis_state(n)
t = time.time(); target = v_mul_mps(circ, th + 0.05, zero, trunc_thr=1e-14); print("n=40 apply", time.time() - t, target.bond_dims.max())
t = time.time(); vh = v_dagger_mul_mps(circ, th, target, trunc_thr=1e-14); print("vdag", time.time() - t, vh.bond_dims.max(), abs(vh.dot(vh)))
t = time.time(); g = fast_dot_gradient_mps(circ, th, zero, vh, trunc_thr=1e-14); print("grad", time.time() - t, np.abs(g).max())

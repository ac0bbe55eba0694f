import sys; sys.path.insert(0, ".")
import numpy as np, time
from oracle import aqc_oracle as orc
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.mps_engine import DeviceMPS, v_mul_mps, svd
rng = np.random.default_rng(3)
for m in (128, 256, 512):
    a = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
    svd(a[:8, :8])
    t = time.time(); u, s, vh, sw = svd(a); dt = time.time() - t
    print("svd", m, f"{dt*1e3:.1f} ms", "sweeps", sw, "err", np.abs((u * s) @ vh - a).max())
for n, L, cap in ((14, 78, 0), (16, 90, 64), (20, 114, 32)):
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
    th = orc.rand_thetas(circ.num_thetas, rng)
    zero = DeviceMPS.basis_state(n)
    t = time.time(); out = v_mul_mps(circ, th, zero, max_bond=cap); dt = time.time() - t
    print("n", n, "L", L, "cap", cap, "bonds", out.bond_dims.max(), f"{dt:.2f} s", f"{dt / L * 1e3:.1f} ms/gate", "norm", abs(out.dot(out)))

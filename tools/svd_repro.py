#!/usr/bin/env python3
"""Reproduces a non-converging two-site SVD of the MPS engine: walks the 32-qubit Trotter target gate by gate (one ABI call per
gate); when a gate fails, the MPS is still in its pre-gate state -- its tensors go to gpurun_out/svd_fail.npz."""
import sys

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import TrotterAnsatz, gates                              # noqa: E402
from aqc_research_amd.circuit_structures import make_trotter_like_circuit      # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index   # noqa: E402
from aqc_research_amd import mps_engine as me                                   # noqa: E402

n, layers = 32, 2
tcirc = TrotterAnsatz(n, make_trotter_like_circuit(n, 3 * layers), second_order=True)
tth = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=0.6 * layers, delta=1.0)
m = me.DeviceMPS.basis_state(n, neel_state_index(n))
orig = me.DeviceMPS.gate2
count = [0]


def gate2(self, g, c, t, thr=0.0, mb=0):
    count[0] += 1
    try:
        return orig(self, g, c, t, thr, mb)
    except RuntimeError as exc:
        gam, lam = self.to_qiskit()
        q = min(c, t)
        np.savez("gpurun_out/svd_fail.npz", g=np.asarray(g), c=c, t=t, q=q, g0=gam[q][0], g1=gam[q][1], h0=gam[q + 1][0], h1=gam[q + 1][1],
                 lam_l=lam[q - 1] if q > 0 else np.ones(1), lam_m=lam[q], lam_r=lam[q + 1] if q + 1 < n - 1 else np.ones(1))
        print(f"gate2 #{count[0]} on ({c},{t}) failed: {exc}; dims {self.bond_dims}", flush=True)
        raise


me.DeviceMPS.gate2 = gate2
try:
    me._apply_circuit_gatewise(tcirc, tth, m, False, 1e-12, 0)
    print("no failure; gates:", count[0], "dims", m.bond_dims)
except RuntimeError:
    pass

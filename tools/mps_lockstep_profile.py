#!/usr/bin/env python3
"""A few lockstep evaluations of the 32-qubit engine workload for a kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lockstep -- python3 tools/mps_lockstep_profile.py [lanes] [reps] [distinct targets]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import TrotterAnsatz                                    # noqa: E402
from aqc_research_amd.circuit_structures import make_trotter_like_circuit     # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index   # noqa: E402
from aqc_research_amd.mps_engine import DeviceMPS, LockstepLanes, v_mul_mps   # noqa: E402

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, layers, thr = 32, 2, 1e-6
circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
th0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.6 * layers, delta=1.0)
tcirc = TrotterAnsatz(n, make_trotter_like_circuit(n, 3 * layers), second_order=True)
tth = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=0.6 * layers, delta=1.0)
basis = DeviceMPS.basis_state(n, neel_state_index(n))
distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # number of different targets shared round-robin by the lanes
targets = [v_mul_mps(tcirc, init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=0.6 * layers * (1.0 + 0.01 * b), delta=1.0), basis,
                     trunc_thr=1e-12) for b in range(distinct)]
rng = np.random.default_rng(3)
ths = np.stack([th0 + 0.02 * rng.standard_normal(th0.size) for _ in range(lanes)])
lk = LockstepLanes(n, lanes).set_targets(targets[0] if distinct == 1 else [targets[b % distinct] for b in range(lanes)]).set_lhs(basis)
def run():
    try:
        lk.evaluate(circ, ths, trunc_thr=thr)
    except RuntimeError as err:   # (measurement hooks that break convergence still run every kernel)
        print("evaluate:", err, flush=True)


run()
run()
t0 = time.perf_counter()
for _ in range(reps):
    run()
dt = (time.perf_counter() - t0) / reps
h, g, disc, bonds = lk.evaluate(circ, ths, trunc_thr=thr, details=True)
print(f"lockstep lanes {lanes} ({distinct} targets): {lanes / dt:.1f} evals/s, {dt * 1e3:.1f} ms per round; largest bond of V^H|target> {bonds.max()}, "
      f"mean |h|^2 {np.mean(np.abs(h) ** 2):.4f}", flush=True)
from aqc_research_amd import _lib   # noqa: E402

if hasattr(_lib.lib(), "aqc_dbg_gate2_stamps"):   # tuning builds (AQC_HIP_LIB=.../libaqc_hip_tuning.so): in-kernel stamps
    _lib.lib().aqc_dbg_gate2_stamps()

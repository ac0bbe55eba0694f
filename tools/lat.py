import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from aqc_research_amd import TrotterAnsatz, ParametricCircuit
from aqc_research_amd.circuit_structures import make_trotter_like_circuit, create_ansatz_structure
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace
rng = np.random.default_rng(0)
for name, circ in (("n12 trotter2", TrotterAnsatz(12, make_trotter_like_circuit(12, 2), True)), ("n16 L40", ParametricCircuit(16, "cx", create_ansatz_structure(16, "spin", "full", 40)))):
    n, T = circ.num_qubits, circ.num_thetas
    ws = Workspace(HipContext.of(circ), batch=1)
    y = rng.random(1 << n) + 1j * rng.random(1 << n)
    ws.upload(BUF_Y, y / np.linalg.norm(y)); ws.set_basis(BUF_X, 0); ws.gather_setup(np.arange(n + 1))
    ths = np.pi * (2 * rng.random((60, T)) - 1)
    for mode in ("eval", "apply-only", "grad-only", "thetas-only"):
        def one(i):
            if mode == "eval": ws.eval(ths[i], vdag=True, gather=True, grad=True)
            elif mode == "apply-only": ws.eval(ths[i], vdag=True, gather=True, grad=False)
            elif mode == "grad-only": ws.eval(None, vdag=False, gather=False, grad=True)
            else: ws.eval(ths[i], vdag=False, gather=False, grad=False)
        for i in range(10): one(i)
        t = time.perf_counter()
        for i in range(10, 60): one(i)
        print(name, mode, f"{(time.perf_counter() - t) / 50 * 1e3:.3f} ms", ws.plan_info(0), ws.plan_info(1), flush=True)
    ws.close()

"""Single-evaluation latency (batch 1, thetas from host, amplitudes + gradient back) for several tile sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aqc_research_amd import ParametricCircuit, TrotterAnsatz
from aqc_research_amd.circuit_structures import create_ansatz_structure, make_trotter_like_circuit
from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace
from oracle import aqc_oracle as orc

cases = {"sv16_l40": lambda: ParametricCircuit(16, "cx", create_ansatz_structure(16, "spin", "full", 40)),
         "sv12_trotter2": lambda: TrotterAnsatz(12, make_trotter_like_circuit(12, 2), second_order=True),
         "sv20_l40": lambda: ParametricCircuit(20, "cx", create_ansatz_structure(20, "spin", "full", 40))}
rng = np.random.default_rng(0)
for name, mk in cases.items():
    circ = mk()
    n, T = circ.num_qubits, circ.num_thetas
    tgt = orc.rand_state(n, rng)
    for ka, ks in ((12, 12), (11, 11), (10, 10), (9, 9), (8, 8), (10, 11), (11, 10)):
        if ka > n or ks > n:
            continue
        ws = Workspace(HipContext.of(circ), batch=1, tile_bits_apply=ka, tile_bits_sweep=ks)
        ws.upload(BUF_Y, tgt); ws.set_basis(BUF_X, 0); ws.gather_setup(orc.flip_state_indices(n, 1))
        ths = np.pi * (2 * rng.random((60, T)) - 1)
        for i in range(10):
            ws.eval(ths[i], vdag=True, gather=True, grad=True)
        best = 1e9
        for r in range(5):
            t0 = time.perf_counter()
            for i in range(10, 60):
                ws.eval(ths[i], vdag=True, gather=True, grad=True)
            best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
        print(f"{name} ka={ka} ks={ks}: {best:.4f} ms  launches vdag/sweep {ws.plan_info(0)[0]}/{ws.plan_info(1)[0]} subs {ws.plan_substages(0)}/{ws.plan_substages(1)}", flush=True)
        ws.close()

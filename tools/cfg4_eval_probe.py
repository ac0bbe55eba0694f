"""20-qubit, 8 lanes: where one value_and_grad of the batched surrogate objective spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz
from oracle import aqc_oracle as orc

n, h, B = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 8
circ = trotter_ansatz(n, 2 * h, True)
base = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=1.2 * h, delta=1.0)
rng = np.random.default_rng(0)
t = orc.rand_state(n, rng)
targets = np.stack([np.roll(t, 17 * b) for b in range(B)])
bo = BatchedSurrogateObjective(circ, targets, base_index=neel_state_index(n))
th = np.tile(base, (B, 1)) + 0.01 * rng.standard_normal((B, base.size))
for _ in range(3):
    f, g = bo.value_and_grad(th)
ws = bo.ws if hasattr(bo, "ws") else bo._ws
t0 = time.perf_counter()
for _ in range(10):
    f, g = bo.value_and_grad(th)
t1 = time.perf_counter()
for _ in range(10):
    ws.eval(th, vdag=True, gather=True, grad=True)
t2 = time.perf_counter()
print(f"h={h}: value_and_grad {(t1 - t0) * 100:.3f} ms, raw eval (V^H + gather + one sweep) {(t2 - t1) * 100:.3f} ms; plan {ws.plan_info(0)} {ws.plan_info(1)} subs {ws.plan_substages(0)}/{ws.plan_substages(1)}")

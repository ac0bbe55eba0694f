"""Where the time of one config-4 job goes (host set-up vs evaluations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz
from oracle import aqc_oracle as orc

n = 20
for h in (1, 4, 8):
    for rep in range(2):
        t0 = time.perf_counter(); circ = trotter_ansatz(n, 2 * h, True)
        t1 = time.perf_counter(); base = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=1.2 * h, delta=1.0)
        t2 = time.perf_counter()
        rng = np.random.default_rng(rep); targets = np.stack([orc.rand_state(n, rng) for _ in range(8)])
        t3 = time.perf_counter(); bo = BatchedSurrogateObjective(circ, targets, base_index=neel_state_index(n))
        t4 = time.perf_counter(); th = np.tile(base, (8, 1))
        for _ in range(10):
            f, g = bo.value_and_grad(th); th = th - 0.05 * g
        t5 = time.perf_counter(); bo.close(); t6 = time.perf_counter()
        print(f"h={h} rep={rep}: ansatz {t1-t0:.3f} trotter-init {t2-t1:.3f} targets {t3-t2:.3f} objective+workspace {t4-t3:.3f} 10 evals {t5-t4:.3f} close {t6-t5:.3f}", flush=True)

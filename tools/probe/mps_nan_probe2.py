import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from oracle import aqc_oracle as orc
from aqc_research_amd import ParametricCircuit
import aqc_research_amd.mps_operations as mpsop
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.engine import BUF_Y, BUF_Z, HipContext
n, chi = 16, 16
for fam in ("3", "2", "1"):
    os.environ["AQC_KERNEL_FAMILY"] = fam
    HipContext._cache.clear()
    rng = np.random.default_rng(3000 + chi)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 40))
    th = orc.rand_thetas(circ.num_thetas, rng)
    phi = orc.random_mps(n, chi, rng)
    dense = orc.mps_to_vector(phi)
    print("fam", fam, "mps_to_vector err", np.abs(mpsop.mps_to_vector(phi) - dense).max(), flush=True)
    ws = HipContext.of(circ).workspace(1, 1)
    ws.set_thetas(th)
    ws.mps_to_vec_batch([phi], BUF_Y)
    y = ws.download(BUF_Y, lane=0)
    print("  Y nan", np.isnan(y).any(), "err", np.abs(y - dense).max(), "family", ws.kernel_family(0), flush=True)
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z, lane=0)
    print("  Z nan", np.isnan(z).any(), np.linalg.norm(z), flush=True)

import os, sys
os.environ["AQC_SWEEP_REG_BITS"] = sys.argv[1]
os.environ["AQC_KERNEL_V2"] = "1"
if len(sys.argv) > 2: os.environ["AQC_DEBUG_SKIP"] = sys.argv[2]
sys.path.insert(0, ".")
import numpy as np
from oracle import aqc_oracle as orc
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace
n, L, B = int(os.environ.get("PN", 12)), 20, 2
rng = np.random.default_rng(3)
a = orc.Ansatz(n, "cx", orc.spin_blocks(n, L))
ws = Workspace(HipContext.of(ParametricCircuit(n, "cx", a.blocks)), batch=B, tile_bits_apply=int(os.environ.get("PK", 10)), tile_bits_sweep=int(os.environ.get("PK", 10)))
print("plan", ws.plan_info(1), flush=True)
th = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(B)])
y = orc.rand_state(n, rng); x = orc.rand_state(n, rng)
ws.set_thetas(th); ws.broadcast(BUF_Y, y); ws.apply(True, BUF_Y, BUF_Z); ws.broadcast(BUF_X, x)
print("apply done", flush=True)
ws.grad(); ws.sync()
print("grad done", flush=True)
g = ws.get_grads()
for b in range(B):
    ref = orc.grad_of_dot_product(a, th[b], x, orc.v_dagger_mul_vec(a, th[b], y))
    print("lane", b, "max err", np.abs(g[b] - ref).max(), flush=True)

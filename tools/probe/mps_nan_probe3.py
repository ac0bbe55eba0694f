import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import aqc_research_amd.mps_operations as mpsop
from aqc_research_amd import engine
from aqc_research_amd.engine import BUF_Y, BUF_Z

orig_batch = engine.Workspace.mps_to_vec_batch
def traced(self, mps_list, buf, lanes=None):
    orig_batch(self, mps_list, buf, lanes)
    y = self.download(buf)
    dims = [g[0].shape for g in mps_list[0][0]]
    print("  mps_to_vec_batch: lanes", len(mps_list), "buf", buf, "nan", np.isnan(y).any(), "norm", np.linalg.norm(y[0]), "dims", dims[:3], "...", dims[-2:], "family", self.kernel_family(0), flush=True)
engine.Workspace.mps_to_vec_batch = traced
orig_apply = engine.Workspace.apply
def traced_apply(self, inverse, src, dst):
    orig_apply(self, inverse, src, dst)
    z = self.download(dst)
    print("  apply: nan", np.isnan(z).any(), "norm", np.linalg.norm(z[0]), flush=True)
engine.Workspace.apply = traced_apply

class MP:
    def setenv(self, k, v): os.environ[k] = v
import tests.test_hip_full_size as t
for fam in ("register-blocked",):
    print("family", fam, flush=True)
    try:
        t.test_config3_mps_front_door_n16_l40(16, fam, MP())
        print("  passed")
    except Exception as e:
        print("  FAILED", type(e).__name__, str(e)[:100])

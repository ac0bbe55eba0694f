import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
import tune
tune.run(B=256, configs=[{"ka": 12, "ks": 12}, {"ka": 11, "ks": 12}, {"ka": 10, "ks": 12}, {"ka": 9, "ks": 12}], steps=20, rounds=3)

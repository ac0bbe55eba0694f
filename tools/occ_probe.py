import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.engine import HipContext, Workspace
circ = ParametricCircuit(16, "cx", create_ansatz_structure(16, "spin", "full", 40))
for k in (12, 11, 10):
    ws = Workspace(HipContext.of(circ), batch=64, tile_bits_apply=k, tile_bits_sweep=k)
    ws.close()

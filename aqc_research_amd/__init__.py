"""
aqc_research_amd -- MI355X-native fidelity/gradient path of aqc-research.

Hand-written gfx950 HIP kernels behind a C ABI (include/aqc_hip.h), loaded with
ctypes; this package mirrors the reference's Python interface for that path.
"""
from .parametric_circuit import ParametricCircuit, TrotterAnsatz, first_layer_included, layer_to_block_range  # noqa: F401

__version__ = "0.1.0"

#!/usr/bin/env python3
"""A/B timing of kernel experiment builds (AQC_HIP_LIB=... python tools/variant_time.py [lanes ...]): the headline shape
(16 qubits, 40 blocks) through tools/tune.py's interleaved timer; one line per lane count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tune  # noqa: E402

print("lib:", os.environ.get("AQC_HIP_LIB", "(shipped)"), flush=True)
for B in [int(x) for x in sys.argv[1:]] or [256, 64]:
    tune.run(B=B, configs=[{}], steps=20, rounds=3)

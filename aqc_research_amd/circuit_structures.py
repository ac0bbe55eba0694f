"""
Block-layout generators (pure index arithmetic) needed to build inputs of the
path: mirror of circuit_structures.py:31-349 for the layouts the reference offers.
"""
import itertools
from typing import List

import numpy as np


def lower_limit(num_qubits: int) -> int:
    """Blocks needed for exact compiling: ceil((4^n - 3n - 1)/4) (circuit_structures.py:31-43)."""
    return int(-(-(4**num_qubits - 3 * num_qubits - 1) // 4))


def circuit_layout_list() -> List[str]:
    return ["spin", "line", "cyclic_spin", "cyclic_line"]


def num_blocks_per_layer(num_qubits: int, circuit_layout: str) -> int:
    return num_qubits if circuit_layout.startswith("cyclic_") else num_qubits - 1


def _pairs(layout: str, n: int):
    """Infinite generator of (control, target) pairs of a layout."""
    if layout == "spin":  # even bonds then odd bonds
        return itertools.cycle([(i, i + 1) for s in (0, 1) for i in range(s, n - 1, 2)])
    if layout == "line":  # 0-1, 1-2, ..., never (n-1)-0
        return itertools.cycle([(i, i + 1) for i in range(n - 1)])
    if layout == "cyclic_line":
        return itertools.cycle([(i, (i + 1) % n) for i in range(n)])
    if layout == "cyclic_spin":

        def gen():
            for i in itertools.count():
                off = (i // (n // 2)) % 2 if n % 2 == 0 else 0
                yield (2 * i + off) % n, (2 * i + off + 1) % n

        return gen()
    raise ValueError(f"Unknown type of circuit layout, expects one of {circuit_layout_list()}, got {layout}")


def create_ansatz_structure(
    num_qubits: int, layout: str = "spin", connectivity: str = "full", depth: int = 0, block_repeat: int = 1, logger=None
) -> np.ndarray:
    """(2, depth) array of control/target positions (circuit_structures.py:46-130)."""
    if num_qubits < 2:
        raise ValueError("Number of qubits must be greater or equal to 2")
    if connectivity not in ("full", "line"):
        raise ValueError(f"layout '{layout}' assumes 'line' or 'full' connectivity, got {connectivity}")
    if not 1 <= block_repeat <= 3:
        raise ValueError("'block_repeat' argument must be equal 1, 2 or 3")
    if depth <= 0:
        depth = lower_limit(num_qubits)
    gen = _pairs(layout, num_qubits)
    blocks = np.array([next(gen) for _ in range(depth)], dtype=np.int64).reshape(depth, 2).T.copy()
    return np.repeat(blocks, block_repeat, axis=1) if block_repeat > 1 else blocks


def make_trotter_like_circuit(num_qubits: int, num_layers: int, *, connectivity: str = "full", verbose: bool = False) -> np.ndarray:
    """Spin pattern with every bond expanded to the triplet (t,c),(c,t),(t,c)
    (circuit_structures.py:133-178)."""
    if num_qubits < 2:
        raise ValueError("number of qubits must be greater or equal to 2")
    if connectivity not in ("full", "line"):
        raise ValueError("expects 'full' or 'line' connectivity")
    if num_layers < 0:
        raise ValueError("expects non-negative number of layers")
    if num_layers == 0:
        return np.zeros((2, 0), dtype=np.int64)
    base = create_ansatz_structure(num_qubits, "spin", "full", num_layers * (num_qubits - 1))
    out = np.repeat(base, 3, axis=1)
    out[:, 0::3] = base[::-1]
    out[:, 2::3] = base[::-1]
    return out


def fraction_of_lower_bound(depth_fraction: float, num_qubits: int, circuit_layout: str) -> int:
    """Number of layers for a fraction of the exact-compiling depth (circuit_structures.py:210-251)."""
    if circuit_layout not in circuit_layout_list():
        raise ValueError(f"'circuit_layout' must be one of {circuit_layout_list()}")
    if not 0 < depth_fraction <= 1:
        raise ValueError("expects: 0 < depth_fraction <= 1")
    bpl = num_blocks_per_layer(num_qubits, circuit_layout)
    depth = int(round(depth_fraction * lower_limit(num_qubits)))
    return int(max(1, (depth + bpl - 1) // bpl))

"""
Ansatz descriptors consumed by the HIP path: host-side mirror of the reference's
``ParametricCircuit`` / ``TrotterAnsatz`` (parametric_circuit.py:24-187,267-466).
Plain data + validation only; same attribute names so reference drivers can pass
either their own objects or these.
"""
from typing import Optional, Tuple, Union

import numpy as np

_ENTANGLERS = ("cx", "cz", "cp")


def _valid_blocks(num_qubits, blocks) -> bool:
    return (
        isinstance(num_qubits, (int, np.integer))
        and num_qubits >= 2
        and isinstance(blocks, np.ndarray)
        and np.issubdtype(blocks.dtype, np.integer)
        and blocks.ndim == 2
        and blocks.shape[0] == 2
        and bool(np.all((0 <= blocks) & (blocks < num_qubits)))
        and bool(np.all(blocks[0] != blocks[1]))
    )


class ParametricCircuit:
    """n front gates Rz Ry Rz followed by L unit-blocks (C (x) T) CG; thetas laid
    out as [3n | tpb*L] (parametric_circuit.py:108-187)."""

    def __init__(self, num_qubits: int, entangler: str, blocks: np.ndarray, name: Optional[str] = None, power: int = 1):
        if entangler not in _ENTANGLERS:
            raise ValueError(f"entangler must be one of {_ENTANGLERS}")
        if not (isinstance(power, (int, np.integer)) and power == 1):
            raise ValueError("expects circuit power (V^p) to be integer and p == 1")
        self.check_block_layout(num_qubits, blocks)
        self._n = int(num_qubits)
        self._entangler = entangler
        self._blocks = np.array(blocks, dtype=np.int64)
        self._name = name if isinstance(name, str) else ""

    # -- structure -----------------------------------------------------------
    def check_block_layout(self, num_qubits: int, blocks: np.ndarray) -> None:
        if not _valid_blocks(num_qubits, blocks):
            raise ValueError("not a valid structure of unit-blocks")

    def update_structure(self, blocks: np.ndarray) -> None:
        self.check_block_layout(self._n, blocks)
        self._blocks = np.array(blocks, dtype=np.int64)

    def insert_unit_blocks(self, pos: int, extra_blocks: np.ndarray, thetas: Optional[np.ndarray] = None):
        """Inserts blocks at block position ``pos``; thetas (if given) get zeros at
        the new positions.  Returns (thetas, new_indices) (parametric_circuit.py:189-232)."""
        self.check_block_layout(self._n, extra_blocks)
        if not 0 <= pos <= self.num_blocks:
            raise ValueError("block position out of range")
        if thetas is not None and np.size(thetas) != self.num_thetas:
            raise ValueError("thetas do not match the current circuit")
        tpos = 3 * self._n + pos * self.tpb
        count = self.tpb * extra_blocks.shape[1]
        self._blocks = np.concatenate((self._blocks[:, :pos], extra_blocks.astype(np.int64), self._blocks[:, pos:]), axis=1)
        if thetas is None:
            return None, None
        thetas = np.concatenate((thetas[:tpos], np.zeros(count, dtype=thetas.dtype), thetas[tpos:]))
        return thetas, np.arange(tpos, tpos + count, dtype=int)

    # -- properties ----------------------------------------------------------
    name = property(lambda self: self._name)
    num_qubits = property(lambda self: self._n)
    dimension = property(lambda self: 1 << self._n)
    num_blocks = property(lambda self: int(self._blocks.shape[1]))
    blocks = property(lambda self: self._blocks)
    entangler = property(lambda self: self._entangler)
    tpb = property(lambda self: 5 if self._entangler == "cp" else 4)
    num_thetas = property(lambda self: 3 * self._n + self.tpb * self.num_blocks)
    circuit_power = property(lambda self: 1)

    def subset1q(self, vec: np.ndarray) -> np.ndarray:
        if vec.shape != (self.num_thetas,):
            raise ValueError("vector length must equal num_thetas")
        return vec[: 3 * self._n].reshape(-1, 3)

    def subset2q(self, vec: np.ndarray) -> np.ndarray:
        if vec.shape != (self.num_thetas,):
            raise ValueError("vector length must equal num_thetas")
        return vec[3 * self._n :].reshape(-1, self.tpb)

    @property
    def num_layers(self) -> int:
        raise NotImplementedError("there are no layers in generic ansatz")

    @property
    def bpl(self) -> int:
        raise NotImplementedError("there are no layers in generic ansatz")


class TrotterAnsatz(ParametricCircuit):
    """Layers of (n-1) block triplets (t,c),(c,t),(t,c) on adjacent qubits, cx only;
    2nd order implies a virtual trailing half-layer sharing the leading half-layer's
    thetas (parametric_circuit.py:267-423)."""

    def __init__(self, num_qubits: int, blocks: np.ndarray, second_order: bool, name: Optional[str] = None):
        if not isinstance(second_order, (bool, np.bool_)):
            raise TypeError("second_order must be bool")
        self._second_order = bool(second_order)
        super().__init__(num_qubits, "cx", blocks, name)

    is_second_order = property(lambda self: self._second_order)
    half_layer_num_blocks = property(lambda self: 3 * (self._n // 2) if self._second_order else 0)
    bpl = property(lambda self: 3 * (self._n - 1))
    num_layers = property(lambda self: self.num_blocks // self.bpl)

    def insert_unit_blocks(self, pos, extra_blocks, thetas=None):
        if pos % (3 * (self._n - 1)) != 0:
            raise ValueError("position of blocks insertion must be aligned at layer boundary")
        return super().insert_unit_blocks(pos, extra_blocks, thetas)

    def check_block_layout(self, num_qubits: int, blocks: np.ndarray) -> None:
        super().check_block_layout(num_qubits, blocks)
        nb = blocks.shape[1]
        if nb == 0:
            return
        if nb % (3 * (num_qubits - 1)) != 0:
            raise ValueError("not a valid Trotterized block layout")
        t = blocks.reshape(2, -1, 3)
        ok = (
            np.all(t[:, :, 0] == t[:, :, 2])
            and np.all(t[0, :, 0] == t[1, :, 1])
            and np.all(t[1, :, 0] == t[0, :, 1])
            and np.all(t[0, :, 0] == t[1, :, 0] + 1)
        )
        if not ok:
            raise ValueError("not a valid Trotterized block layout")
        if self._second_order:
            lead = t[:, : num_qubits // 2, 1]
            want = 2 * np.arange(num_qubits // 2)
            if not (np.array_equal(lead[0], want) and np.array_equal(lead[1], want + 1)):
                raise ValueError("unexpected layout of the leading half-layer")


def layer_to_block_range(circ: ParametricCircuit, layer_range: Union[Tuple[int, int], None]) -> Tuple[int, int]:
    """parametric_circuit.py:426-451."""
    if layer_range is None:
        return 0, circ.num_blocks
    lo, hi = layer_range
    if not 0 <= lo < hi <= circ.num_layers:
        raise ValueError("invalid layer range")
    return lo * circ.bpl, hi * circ.bpl


def first_layer_included(circ: ParametricCircuit, layer_range: Union[Tuple[int, int], None]) -> bool:
    """parametric_circuit.py:454-466."""
    if layer_range is None:
        return True
    lo, hi = layer_range
    if not 0 <= lo < hi <= circ.num_layers:
        raise ValueError("invalid layer range")
    return lo == 0

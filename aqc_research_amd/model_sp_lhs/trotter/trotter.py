"""
Trotterised XXZ-Heisenberg evolution expressed through the ansatz itself: the Trotter
circuit of the reference (trotter.py:317-378) is a ``TrotterAnsatz`` with the angles produced by
``init_ansatz_to_trotter`` (trotter.py:478-537; equivalence proven by the reference's
test_trotter_initial_point.py:54-98), so target states are synthesised on the GPU with the path's own
``V |neel>`` kernel instead of a Qiskit simulation.  Host code here is index/angle bookkeeping only.
"""
from typing import Optional, Tuple

import numpy as np

from ...circuit_structures import make_trotter_like_circuit
from ...parametric_circuit import TrotterAnsatz, first_layer_included


def trotter_alphas(dt: float, delta: float) -> np.ndarray:
    """Three angles of one 2-qubit Trotter building block (trotter.py:269-283)."""
    if not (dt > 0 and delta > 0):
        raise ValueError("dt and delta must be positive")
    return np.asarray([np.pi / 2 - 0.5 * delta * dt, 0.5 * dt - np.pi / 2, np.pi / 2 - 0.5 * dt])


def trotter_global_phase(num_qubits: int, num_steps: int, second_order: bool) -> float:
    """Phase phi such that exp(i phi) * (Trotter ansatz)|psi> approximates exp(-iHt)|psi>: pi/4 per 2-qubit
    Trotter block, i.e. (n-1) blocks per step plus n//2 blocks of the trailing half-layer.  (The reference
    ignores the global phase, trotter.py:331-332; its helper at trotter.py:286-314 counts n resp. n-1 blocks
    for the half-layer -- the n//2 form, kept there as a comment, is the one that matches exp(-iHt), see
    tests/test_host_logic.py.)"""
    phs = 0.25 * np.pi * (num_qubits - 1) * num_steps
    if second_order:
        phs += 0.25 * np.pi * (num_qubits // 2)
    return phs


def neel_state_index(num_qubits: int) -> int:
    """Basis index of X on every even qubit applied to |0> (neel_init_state, trotter.py:389-398)."""
    return sum(1 << q for q in range(0, num_qubits, 2))


class _Op:
    def __init__(self, name):
        self.name, self.params = name, []


class _Instruction:
    """One entry of ``BasisPrepCircuit.data``: ``operation.name`` and ``qubits`` (integers), the attribute names of
    Qiskit's ``CircuitInstruction``."""

    def __init__(self, name, qubit):
        self.operation, self.qubits, self.clbits = _Op(name), (int(qubit),), ()


class BasisPrepCircuit:
    """Stand-in for the ``QuantumCircuit`` a reference ``state_prep_func`` returns (trotter.py:381-410) when the
    circuit only flips qubits: ``num_qubits``, ``x(q)`` and ``data`` with Qiskit's attribute names -- what
    ``objective_base.basis_mask_of_circuit`` walks.  Qiskit itself is not needed (and not imported)."""

    def __init__(self, num_qubits: int):
        if not (isinstance(num_qubits, (int, np.integer)) and num_qubits >= 2):
            raise ValueError("num_qubits must be an integer >= 2")
        self.num_qubits, self.data = int(num_qubits), []

    def x(self, qubit: int):
        if not 0 <= int(qubit) < self.num_qubits:
            raise ValueError("qubit out of range")
        self.data.append(_Instruction("x", qubit))
        return self

    @property
    def basis_index(self) -> int:
        mask = 0
        for ins in self.data:
            mask ^= 1 << ins.qubits[0]
        return mask


def identity_circuit(num_qubits: int) -> BasisPrepCircuit:
    """The empty preparation circuit (trotter.py:381-386)."""
    return BasisPrepCircuit(num_qubits)


def neel_init_state(num_qubits: int) -> BasisPrepCircuit:
    """|0> -> the Neel state: X on every even qubit (trotter.py:389-398)."""
    qc = BasisPrepCircuit(num_qubits)
    for k in range(0, num_qubits, 2):
        qc.x(k)
    return qc


def half_zero_circuit(num_qubits: int) -> BasisPrepCircuit:
    """|0> -> half zero / half unit bits: X on the upper half of the qubits (trotter.py:401-410)."""
    qc = BasisPrepCircuit(num_qubits)
    for k in range(num_qubits // 2, num_qubits):
        qc.x(k)
    return qc


def slice2q(circ, vec: np.ndarray, *, layer_range: Optional[Tuple[int, int]] = None):
    """View of the block parameters as (layers, triplets, 12) (trotter.py:431-475)."""
    if not hasattr(circ, "is_second_order"):
        raise ValueError("expects Trotterized ansatz")
    if vec.shape != (circ.num_thetas,):
        raise ValueError("vector length must equal num_thetas")
    nl = circ.num_layers
    layer_range = (0, nl) if layer_range is None else layer_range
    if not 0 <= layer_range[0] < layer_range[1] <= nl:
        raise ValueError("invalid layer range")
    v = circ.subset2q(vec).reshape((nl, circ.num_qubits - 1, 12))
    return v[layer_range[0] : layer_range[1]], layer_range


def init_ansatz_to_trotter(circ, thetas: np.ndarray, *, evol_time: float, delta: float,
                           layer_range: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """Sets ``thetas`` (in place) so that the layers in ``layer_range`` equal a Trotter circuit for
    ``evol_time`` (trotter.py:478-537): per triplet only theta[5], theta[0], theta[6] are non-zero; the
    leading (and implied trailing) half-layer of a 2nd-order ansatz uses dt/2."""
    th2q, layer_range = slice2q(circ, thetas, layer_range=layer_range)
    dt = evol_time / float(layer_range[1] - layer_range[0])
    a = trotter_alphas(dt, delta)
    layer_0 = first_layer_included(circ, layer_range)
    if layer_0:
        circ.subset1q(thetas).fill(0)
    th2q.fill(0)
    th2q[:, :, 5], th2q[:, :, 0], th2q[:, :, 6] = a[0], a[1], a[2]
    if circ.is_second_order and layer_0:
        b = trotter_alphas(0.5 * dt, delta)
        half = circ.half_layer_num_blocks // 3
        th2q[0, :half, 5], th2q[0, :half, 0], th2q[0, :half, 6] = b[0], b[1], b[2]
    return thetas


def trotter_ansatz(num_qubits: int, num_steps: int, second_order: bool) -> TrotterAnsatz:
    return TrotterAnsatz(num_qubits, make_trotter_like_circuit(num_qubits, num_steps), second_order=second_order)


def trotter_state(num_qubits: int, *, evol_time: float, num_steps: int, delta: float = 1.0, second_order: bool = True,
                  ini_index: Optional[int] = None, with_global_phase: bool = False) -> np.ndarray:
    """Trotter-evolved basis state, computed by the HIP ``v_mul_vec`` on the Trotter-initialised ansatz
    (Trotter.as_vector, trotter.py:97-127; target_states.py:373-455 uses 10x the steps for the
    "ground truth")."""
    from ...core_operations import v_mul_vec

    circ = trotter_ansatz(num_qubits, num_steps, second_order)
    thetas = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=evol_time, delta=delta)
    vec = np.zeros(circ.dimension, dtype=np.complex128)
    vec[neel_state_index(num_qubits) if ini_index is None else int(ini_index)] = 1
    out = v_mul_vec(circ, thetas, vec, np.zeros_like(vec), None)
    if with_global_phase:
        out *= np.exp(1j * trotter_global_phase(num_qubits, num_steps, second_order))
    return out


def make_hamiltonian(num_qubits: int, delta: float) -> np.ndarray:
    """H = -1/4 sum_i (X_i X_{i+1} + Y_i Y_{i+1} + delta Z_i Z_{i+1}) (trotter.py:183-230); dense, for
    tests and small-n ground truth only."""
    sx = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    sy = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
    sz = np.array([[1, 0], [0, -1]], dtype=np.complex128)

    def two(s, i):
        m = np.eye(1, dtype=np.complex128)
        for q in range(num_qubits):
            m = np.kron(m, s if q in (i, i + 1) else np.eye(2))
        return m

    h = np.zeros((1 << num_qubits, 1 << num_qubits), dtype=np.complex128)
    for i in range(num_qubits - 1):
        h += two(sx, i) + two(sy, i) + delta * two(sz, i)
    return -0.25 * h

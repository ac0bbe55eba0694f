"""
Host-side glue of the state-preparation (ASP) objectives: flip-state bookkeeping,
statistics / early-stop service and the common base class.  Mirrors the public surface of
model_sp_lhs/objective_base.py:42-255,437-834; all vector arithmetic is delegated to the
HIP workspace, only scalars live here.
"""
import itertools
from typing import List, Optional, Tuple

import numpy as np

from ..engine import BUF_Y, HipContext, Workspace
from ..parametric_circuit import ParametricCircuit


class ThinStateHandler:
    """|0>, X_i|0>, X_i X_j|0>, ... kept as the index of their single non-zero amplitude
    (objective_base.py:42-255).  ``base_index`` XORs a computational-basis preparation
    (e.g. the Neel pattern, trotter.py:389-398) on top of every state."""

    def __init__(self, num_qubits: int, max_flips: int, verbose: bool = False, base_index: int = 0):
        if not (isinstance(num_qubits, (int, np.integer)) and num_qubits >= 2):
            raise ValueError("num_qubits must be an integer >= 2")
        if not (isinstance(max_flips, (int, np.integer)) and 0 <= max_flips <= num_qubits):
            raise ValueError("max_flips must be in [0, num_qubits]")
        self._n = int(num_qubits)
        self._comb_labels: List[List[Tuple]] = [
            list(itertools.combinations(range(num_qubits), f)) for f in range(1, max_flips + 1)
        ]
        idx = [0]
        for combos in self._comb_labels:
            for sub in combos:
                v = 0
                for q in sub:
                    v ^= 1 << q  # Qiskit bit order (core_operations.py:34-43)
                idx.append(v)
        # int64 indices up to 62 qubits (what the dense kernels gather with); Python integers beyond, where only
        # the MPS engine can run and consumes them bit by bit
        if self._n <= 62:
            self._state_idx = np.asarray(idx, dtype=np.int64) ^ np.int64(base_index)
        else:
            self._state_idx = np.array([v ^ int(base_index) for v in idx], dtype=object)

    @property
    def num_states(self) -> int:
        return int(self._state_idx.size)

    @property
    def state_indices(self) -> np.ndarray:
        return self._state_idx

    @property
    def flip_qubit_positions(self) -> List[List[Tuple]]:
        return self._comb_labels

    def init_state(self, state_no: int) -> np.ndarray:
        """Dense copy of a state (host convenience / tests only)."""
        s = np.zeros(1 << self._n, dtype=np.complex128)
        s[self._state_idx[state_no]] = 1
        return s

    @property
    def state0(self) -> np.ndarray:
        return self.init_state(0)

    def state_dot_vector(self, state_no: int, vec: np.ndarray) -> np.complex128:
        return vec[self._state_idx[state_no]]


def basis_mask_of_circuit(qc, num_qubits: int) -> int:
    """Bit mask of a state-preparation circuit that only flips qubits -- what the reference's ``state_prep_func``s return
    (``neel_init_state``, ``half_zero_circuit``, ``identity_circuit``, trotter.py:381-410; consumed at
    objective_base.py:298-303).  Duck-typed walk over ``qc.data``: entries with ``.operation`` / ``.qubits`` (Qiskit's
    ``CircuitInstruction``) or the older ``(operation, qargs, cargs)`` triples; X toggles a bit, identities / barriers are
    skipped, anything else raises.  Qiskit is never imported: a real ``QuantumCircuit`` works through the same attributes."""
    if getattr(qc, "num_qubits", num_qubits) != num_qubits:
        raise ValueError("state_prep_func returned a circuit on a different number of qubits")
    mask = 0
    for ins in qc.data:
        op = getattr(ins, "operation", None)
        qubits = getattr(ins, "qubits", None)
        if op is None:
            op, qubits = ins[0], ins[1]
        name = str(getattr(op, "name", "")).lower()
        if name in ("id", "i", "barrier", "delay"):
            continue
        if name != "x" or len(qubits) != 1:
            raise NotImplementedError(
                f"state_prep_func returned a circuit with a '{name}' gate: the HIP path prepares computational-basis states "
                "only (X / identity circuits, a bit mask, or a dense array of states); Qiskit circuits are not simulated")
        q = qubits[0]
        if not isinstance(q, (int, np.integer)):
            find = getattr(qc, "find_bit", None)
            q = find(q).index if find is not None else getattr(q, "index", getattr(q, "_index", None))
        if q is None or not 0 <= int(q) < num_qubits:
            raise ValueError("state_prep_func: cannot resolve a qubit index of the circuit")
        mask ^= 1 << int(q)
    return mask


class DenseStateHandler:
    """States given explicitly as rows of a (num_states, 2^n) array -- the HIP-side
    counterpart of GenericStateHandler (objective_base.py:258-342), whose Qiskit circuit
    simulation is outside this path."""

    def __init__(self, states: np.ndarray):
        st = np.ascontiguousarray(states, dtype=np.complex128)
        if st.ndim != 2 or st.shape[1] & (st.shape[1] - 1):
            raise ValueError("states must have shape (num_states, 2^n)")
        self._states = st

    num_states = property(lambda self: int(self._states.shape[0]))
    state0 = property(lambda self: self._states[0])

    def init_state(self, state_no: int) -> np.ndarray:
        return self._states[state_no]


def _circuit_gate_matrix(name: str, params) -> np.ndarray:
    """Matrix of a gate of the small standard set ``GenericStateHandler`` simulates (Qiskit's conventions; 2-qubit matrices
    are indexed 2 * bit(first qubit) + bit(second qubit), the first qubit of cx / cy / cz / cp being the control)."""
    p = [float(v) for v in params]
    c, sn = (np.cos(0.5 * p[0]), np.sin(0.5 * p[0])) if p else (1.0, 0.0)
    one = {
        "x": [[0, 1], [1, 0]], "y": [[0, -1j], [1j, 0]], "z": [[1, 0], [0, -1]], "h": np.array([[1, 1], [1, -1]]) / np.sqrt(2.0),
        "s": [[1, 0], [0, 1j]], "sdg": [[1, 0], [0, -1j]], "t": [[1, 0], [0, np.exp(0.25j * np.pi)]],
        "tdg": [[1, 0], [0, np.exp(-0.25j * np.pi)]], "sx": 0.5 * np.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]]),
        "rx": [[c, -1j * sn], [-1j * sn, c]], "ry": [[c, -sn], [sn, c]],
    }
    if name in one:
        return np.asarray(one[name], dtype=np.complex128)
    if name == "rz":
        return np.diag([np.exp(-0.5j * p[0]), np.exp(0.5j * p[0])]).astype(np.complex128)
    if name in ("p", "u1"):
        return np.diag([1.0, np.exp(1j * p[0])]).astype(np.complex128)
    if name in ("u", "u3"):
        th, ph, lam = p
        return np.array([[np.cos(0.5 * th), -np.exp(1j * lam) * np.sin(0.5 * th)],
                         [np.exp(1j * ph) * np.sin(0.5 * th), np.exp(1j * (ph + lam)) * np.cos(0.5 * th)]], dtype=np.complex128)
    two = np.eye(4, dtype=np.complex128)
    if name == "cx":
        two[2:, 2:] = [[0, 1], [1, 0]]
    elif name == "cy":
        two[2:, 2:] = [[0, -1j], [1j, 0]]
    elif name == "cz":
        two[3, 3] = -1
    elif name in ("cp", "cu1"):
        two[3, 3] = np.exp(1j * p[0])
    elif name == "swap":
        two = two[[0, 2, 1, 3]]
    else:
        raise NotImplementedError(
            f"state_prep_func returned a circuit with a '{name}' gate: GenericStateHandler simulates x, y, z, h, s, sdg, t, tdg, sx, "
            "rx, ry, rz, p, u, cx, cy, cz, cp, swap (pass a dense (num_states, 2^n) array of prepared states for anything else)")
    return two


class GenericStateHandler(DenseStateHandler):
    """The reference's ``GenericStateHandler`` (objective_base.py:258-342): the states ``S|0>`` and ``S X_i|0>`` of a GENERAL
    state-preparation circuit S, built explicitly.  Qiskit is absent here, so S is any duck-typed circuit -- ``num_qubits``
    and ``data`` with entries ``.operation.name / .operation.params / .qubits`` (a real ``QuantumCircuit`` has exactly these) --
    over a small standard gate set, simulated gate by gate on the device for all num_qubits + 1 states at once (they are the
    columns of one 2^n x (n + 1) matrix: one ``aqc_gate_1q`` / ``aqc_gate_2q`` pass per gate).  X_i is applied BEFORE S (:298-303)."""

    def __init__(self, num_qubits: int, max_flips: int, state_prep_func=None, verbose: bool = False):
        from .. import gates

        if max_flips > 1:
            raise ValueError("expects 'max_flips <= 1' to save memory")          # objective_base.py:294-295
        n = int(num_qubits)
        nstates = n + 1 if max_flips == 1 else 1
        cols = np.zeros((1 << n, nstates), dtype=np.complex128)
        cols[0, 0] = 1.0
        for i in range(1, nstates):
            cols[1 << (i - 1), i] = 1.0
        qc = state_prep_func(n) if callable(state_prep_func) else state_prep_func
        if qc is not None:
            if getattr(qc, "num_qubits", n) != n:
                raise ValueError("state_prep_func returned a circuit on a different number of qubits")
            for ins in qc.data:
                op, qubits = getattr(ins, "operation", None), getattr(ins, "qubits", None)
                if op is None:
                    op, qubits = ins[0], ins[1]
                name = str(getattr(op, "name", "")).lower()
                if name in ("id", "i", "barrier", "delay"):
                    continue
                idx = []
                for q in qubits:
                    if not isinstance(q, (int, np.integer)):
                        find = getattr(qc, "find_bit", None)
                        q = find(q).index if find is not None else getattr(q, "index", getattr(q, "_index", None))
                    if q is None or not 0 <= int(q) < n:
                        raise ValueError("state_prep_func: cannot resolve a qubit index of the circuit")
                    idx.append(int(q))
                g = _circuit_gate_matrix(name, getattr(op, "params", ()))
                if g.shape == (2, 2) and len(idx) == 1:
                    gates.apply_1q(g, idx[0], cols, cols)
                elif g.shape == (4, 4) and len(idx) == 2:
                    gates.apply_2q(g, idx[0], idx[1], cols, cols)
                else:
                    raise ValueError(f"gate '{name}' on {len(idx)} qubit(s)")
            phase = float(getattr(qc, "global_phase", 0.0) or 0.0)
            if phase:
                cols *= np.exp(1j * phase)
        super().__init__(np.ascontiguousarray(cols.T))


class SpService:
    """Counters, statistics and early-termination hooks (objective_base.py:437-622).
    Stoppers are duck-typed (TimeoutChecker / EarlyStopper of the reference's optimizer.py
    work unchanged) and always run on the host, outside native calls, so the exceptions
    they raise propagate to the optimizer."""

    def __init__(self, user_parameters: dict, circuit, num_states: int, verbose: bool = False):
        self._params, self._circuit, self._num_states, self._verbose = user_parameters, circuit, num_states, verbose
        self._num_fun_ev = 0
        self._num_grad_ev = 0
        self._timeout_checker = None
        self._early_stopper = None
        self._stats = {}
        if user_parameters.get("enable_optim_stats", False):
            self._stats = {
                "hs2": np.empty((0, num_states), dtype=np.float16),
                "weight": np.empty(0, dtype=np.float16),
                "fobj": np.empty(0, dtype=np.float32),
                "grad": np.empty(0, dtype=np.float32),
                "num_fun_ev": 0,
                "num_grad_ev": 0,
            }

    def set_status_trackers(self, timeout=None, stopper=None):
        self._timeout_checker, self._early_stopper = timeout, stopper

    statistics = property(lambda self: self._stats)

    def _on_stop(self, fobj: float, thetas: np.ndarray) -> dict:
        return {
            "cost": fobj,
            "num_fun_ev": self._num_fun_ev,
            "num_grad_ev": self._num_grad_ev,
            "num_iters": self._num_grad_ev,
            "thetas": thetas.copy(),
            "blocks": self._circuit.blocks.copy(),
        }

    def on_begin_gradient(self, fobj: float, thetas: np.ndarray, fidelity: Optional[float] = None):
        if self._timeout_checker:
            self._timeout_checker.check(fobj, thetas, self._on_stop)
        if self._early_stopper:
            self._early_stopper.check(fobj=fobj, fidelity=fidelity, thetas=thetas, iter_no=self._num_grad_ev, on_stop=self._on_stop)

    def on_end_gradient(self, fobj: float, fidelity: float, grad: np.ndarray, hs2: np.ndarray, weight: float):
        self._num_grad_ev += 1
        if "hs2" in self._stats:   # enabled by enable_optim_stats (callers may add keys of their own to the dictionary)
            s = self._stats
            s["hs2"] = np.vstack((s["hs2"], hs2.astype(np.float16)[None, :]))
            s["weight"] = np.append(s["weight"], np.float16(weight))
            s["fobj"] = np.append(s["fobj"], np.float32(fobj))
            s["grad"] = np.append(s["grad"], np.float32(np.linalg.norm(grad)))
            s["num_fun_ev"], s["num_grad_ev"], s["num_iters"] = self._num_fun_ev, self._num_grad_ev, self._num_grad_ev
        if self._params.get("verbose", 0) and self._num_grad_ev % max(1, self._params.get("maxiter", 50) // 50) == 0:
            print(".", end="", flush=True)

    def on_end_objective(self):
        self._num_fun_ev += 1

    def on_epoch_end(self):
        if self._stats:
            s = self._stats
            s["hs2"] = np.vstack((s["hs2"], np.full((1, self._num_states), np.nan, dtype=np.float16)))
            s["weight"] = np.append(s["weight"], np.float16(np.nan))
            s["fobj"] = np.append(s["fobj"], np.float32(np.nan))
            s["grad"] = np.append(s["grad"], np.float32(np.nan))


class SpLHSObjectiveBase:
    """Base of the local-Hilbert-Schmidt state-preparation objectives
    (objective_base.py:630-834).  ``user_parameters`` keys: num_qubits, max_flips, optional
    state_prep_func, enable_optim_stats, verbose, maxiter, device.

    ``state_prep_func(num_qubits)`` may return an ``int`` (bit mask of a computational-basis
    preparation, e.g. the Neel state), a circuit that only flips qubits (the reference's ``neel_init_state`` & co.,
    duck-typed: ``basis_mask_of_circuit``), a dense (num_states, 2^n) array of prepared states, or a general (duck-typed)
    circuit over the gate set of ``GenericStateHandler``, whose states are then built explicitly on the device."""

    def __init__(self, user_parameters: dict, circuit: ParametricCircuit, use_mps: bool = False, verbose: bool = False):
        if not isinstance(user_parameters, dict):
            raise TypeError("user_parameters must be a dict")
        self._params, self._circuit, self._verbose = user_parameters, circuit, verbose
        self._use_mps = bool(use_mps) or bool(user_parameters.get("_use_mps", False))
        self._target = None
        self._last_thetas = np.empty(0)
        n = int(user_parameters["num_qubits"])
        if n != circuit.num_qubits:
            raise ValueError("user_parameters['num_qubits'] differs from the circuit")
        max_flips = int(user_parameters["max_flips"])
        prep = user_parameters.get("state_prep_func", None)
        prepared = prep(n) if callable(prep) else prep
        if prepared is None:
            self._state_handler = ThinStateHandler(n, max_flips, verbose)
        elif isinstance(prepared, (int, np.integer)):
            self._state_handler = ThinStateHandler(n, max_flips, verbose, base_index=int(prepared))
        elif isinstance(prepared, np.ndarray):
            self._state_handler = DenseStateHandler(prepared)
        elif hasattr(prepared, "data") and hasattr(prepared, "num_qubits"):   # a (duck-typed) circuit
            try:            # X gates only: the flip states stay one-hot (ThinStateHandler, objective_base.py:42-255)
                self._state_handler = ThinStateHandler(n, max_flips, verbose, base_index=basis_mask_of_circuit(prepared, n))
            except NotImplementedError:   # a general preparation: explicit states (GenericStateHandler, :258-342)
                if self._use_mps:
                    from ..mps_dot_objective import use_dense

                    if not use_dense(n, float(user_parameters.get("trunc_thr", 1e-16))):
                        raise NotImplementedError(
                            "a general state-preparation circuit on the native MPS route (registers beyond dense reach: the reference's "
                            "MpsStateHandler, objective_base.py:345-435) is not built; circuits of X gates (basis preparations) are")
                self._state_handler = GenericStateHandler(n, max_flips, prepared, verbose)
        else:
            raise NotImplementedError(
                "state_prep_func must return a basis-state bit mask (int) or a dense array of states; "
                "Qiskit circuits are not simulated by the HIP path"
            )
        self._num_states = self._state_handler.num_states
        self._service = SpService(user_parameters, circuit, self._num_states, verbose=verbose)
        self._hs2 = np.zeros(self._num_states)
        self._fobj = 1.0
        self._weight = 1.0
        # device side: a private one-lane workspace; Y = target, Z = V^H target, X / X2 = lhs states.
        # user_parameters["workspace"] substitutes a lane of a lockstep batch (lockstep.LaneView).
        self._ws = user_parameters.get("workspace", None)
        self._native_mps = False
        if self._use_mps and self._ws is None:
            from ..mps_dot_objective import use_dense

            # registers beyond dense reach (or a real truncation threshold): no 2^n buffers at all, the objective
            # runs on the device MPS engine (objective_lhs_sur_fast_mps_trotter.py)
            self._native_mps = not use_dense(n, float(user_parameters.get("trunc_thr", 1e-16)))
        if self._native_mps:
            pass
        elif self._ws is None:
            self._ws = Workspace(HipContext.of(circuit), batch=1, ncols=1, device=user_parameters.get("device"))
        elif self._ws.T != circuit.num_thetas or self._ws.dim != circuit.dimension:
            raise ValueError("user_parameters['workspace'] belongs to a different ansatz")

    def _store_latest_thetas(self, thetas: np.ndarray):
        self._last_thetas = np.array(thetas, dtype=np.float64)

    def _calc_objective_before_gradient(self, thetas: np.ndarray):
        """objective_base.py:715-734: optimizers may ask for the gradient first (ADAM)."""
        tol = float(np.sqrt(np.finfo(np.float64).eps))
        last = self._last_thetas
        if last.size != 0 and (thetas is last or np.array_equal(thetas, last)):
            return   # the usual case (objective(theta) just ran): identical arrays are close, without allclose's 15 us
        if last.size == 0 or not np.allclose(thetas, last, atol=tol, rtol=tol):
            self.objective(thetas)

    def objective(self, thetas: np.ndarray) -> float:
        raise NotImplementedError()

    def gradient(self, thetas: np.ndarray) -> np.ndarray:
        raise NotImplementedError()

    def set_status_trackers(self, timeout=None, stopper=None):
        self._service.set_status_trackers(timeout, stopper)

    num_thetas = property(lambda self: self._circuit.num_thetas)
    num_states = property(lambda self: self._num_states)
    target = property(lambda self: self._target)
    statistics = property(lambda self: self._service.statistics)

    def set_target(self, target) -> None:
        """Uploads the state to approximate (objective_base.py:804-819)."""
        if isinstance(target, np.ndarray):
            if target.dtype != np.complex128 or target.shape != (self._circuit.dimension,):
                raise ValueError("target: expects a complex128 vector of size 2^n")
            self._target = target
            self._ws.upload(BUF_Y, target)
        else:
            self._set_mps_target(target)
        self._last_thetas = np.empty(0)

    def _set_mps_target(self, target) -> None:
        raise ValueError("this objective expects a dense target vector")

    def on_epoch_end(self):
        self._service.on_epoch_end()

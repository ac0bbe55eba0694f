"""
Surrogate state-preparation objective with the target given as an MPS: drop-in for
SpSurrogateObjectiveFastMpsTrotter (objective_lhs_sur_fast_mps_trotter.py:42-232).  The
target MPS is contracted to a dense state on the GPU once (set_target); after that the
math and the state machine are exactly those of the state-vector objective, which is the
semantics the reference's tests pin for trunc_thr -> 0.
"""
from typing import Optional, Tuple

from ..engine import BUF_Y
from ..mps_operations import DenseBackedMPS, check_mps, mps_num_qubits
from ..parametric_circuit import TrotterAnsatz, first_layer_included, layer_to_block_range
from .objective_lhs_sur_max import SpSurrogateObjectiveMax


class SpSurrogateObjectiveFastMpsTrotter(SpSurrogateObjectiveMax):
    def __init__(
        self,
        *,
        user_parameters: dict,
        circ,
        layer_range: Optional[Tuple[int, int]] = None,
        alt_layers: bool = False,
        verbose: bool = False,
        grad_scaler=None,
    ):
        if not (isinstance(circ, TrotterAnsatz) or hasattr(circ, "is_second_order")):
            raise TypeError("expects a TrotterAnsatz (objective_lhs_sur_fast_mps_trotter.py:82)")
        if int(user_parameters["max_flips"]) != 1:
            raise ValueError("expects max_flips=1 in case of using MPS")
        user_parameters = dict(user_parameters)
        user_parameters["_use_mps"] = True
        super().__init__(
            user_parameters=user_parameters,
            circ=circ,
            block_range=layer_to_block_range(circ, layer_range),
            front_layer=first_layer_included(circ, layer_range),
            verbose=verbose,
            grad_scaler=grad_scaler,
        )
        self._use_mps = True
        self._trunc_thr = float(user_parameters.get("trunc_thr", 1e-16))
        self._layer_range = (0, circ.num_layers) if layer_range is None else layer_range

    def _set_mps_target(self, target) -> None:
        if not check_mps(target) or mps_num_qubits(target) != self._circuit.num_qubits:
            raise ValueError("target must be an MPS in Qiskit format matching the circuit")
        self._target = target
        if not self._native_mps:
            if isinstance(target, DenseBackedMPS):      # its dense state is at hand: no tensors, no contraction
                self._ws.upload(BUF_Y, target.dense_state)
            else:
                self._ws.mps_upload(0, target)
                self._ws.mps_to_vec(0, BUF_Y, 0)
            return
        from ..mps_engine import DeviceMPS

        self._target_dev = DeviceMPS.from_qiskit(target, device=self._mps_device(), trunc_thr=self._trunc_thr)
        self._vh = None
        self._basis_dev = {}
        if getattr(self, "_lk", None) is not None:
            self._lk.close()
        self._lk, self._lk_refused, self._lk_live = None, False, False

    def _mps_device(self) -> int:
        from ..engine import default_device

        d = self._params.get("device")
        return default_device() if d is None else int(d)

    # ---- native MPS mode: no dense state anywhere (objective_lhs_sur_fast_mps_trotter.py:114-227) ----------
    # Two lockstep lanes of the engine (mps_engine.LockstepLanes; bonds <= 32): lane 0 sees the problem from |state_0>, lane 1 from the
    # leading flip state.  V^H|target> runs once and stays on the device with its bond dimensions, the amplitudes of all n + 1 states
    # come back with it, the two sweeps of :162-227 run together.  Larger bonds: the single-lane engine (one ABI call per piece).
    def _lanes(self):
        if self._lk is None and not self._lk_refused:
            from ..mps_engine import LOCKSTEP_MAX_BOND, LockstepLanes

            if int(self._target_dev.bond_dims.max()) > LOCKSTEP_MAX_BOND or self._num_states != self._circuit.num_qubits + 1:
                self._lk_refused = True
            else:
                self._lk = LockstepLanes(self._circuit.num_qubits, 2, self._mps_device()).set_targets(self._target_dev)
        return self._lk

    def _lane_bits(self, state_a: int, state_b: int):
        import numpy as np

        idx = self._state_handler.state_indices
        return np.array([[(int(idx[s]) >> q) & 1 for q in range(self._circuit.num_qubits)] for s in (state_a, state_b)], dtype=np.uint8)

    def _basis(self, state_no: int):
        from ..mps_engine import DeviceMPS

        if state_no not in self._basis_dev:
            self._basis_dev[state_no] = DeviceMPS.basis_state(self._circuit.num_qubits, int(self._state_handler.state_indices[state_no]),
                                                              device=self._mps_device())
        return self._basis_dev[state_no]

    def _evaluate(self, thetas) -> None:
        if not self._native_mps:
            return super()._evaluate(thetas)
        import numpy as np

        from ..mps_engine import v_dagger_mul_mps

        self._grad0 = None
        self._lk_live = False
        lk = self._lanes()
        if lk is not None:
            try:
                lk.set_lhs_basis(self._lane_bits(0, self._max_no))
                th = np.ascontiguousarray(thetas, dtype=np.float64)
                self._hs[:] = lk.apply_vh(self._circuit, np.stack([th, th]), trunc_thr=self._trunc_thr, flips=True, half=True)[0]
                self._lk_live = True
                return
            except RuntimeError as err:
                if "lockstep lanes" not in str(err):
                    raise
                lk.close()          # a bond outgrew the lanes: this objective stays on the single-lane engine
                self._lk, self._lk_refused = None, True
        if self._vh is not None:
            self._vh.close()
        self._vh = v_dagger_mul_mps(self._circuit, thetas, self._target_dev, trunc_thr=self._trunc_thr, method="single")   # V^H|target>
        for i in range(self._num_states):
            self._hs[i] = self._basis(i).dot(self._vh)                                                   # <state_i|V^H|target>

    def _sweep(self, state_no: int, front: bool):
        if not self._native_mps:
            return super()._sweep(state_no, front)
        if self._lk_live:
            self._lk.set_lhs_basis(self._lane_bits(state_no, state_no))
            return self._lk.gradient(self._circuit, block_range=self._block_range, front_layer=front)[0]
        from ..mps_engine import fast_dot_gradient_mps

        return fast_dot_gradient_mps(self._circuit, self._last_thetas, self._basis(state_no), self._vh, trunc_thr=self._trunc_thr,
                                     block_range=self._block_range, front_layer=front, method="single")

    def _sweep_combined(self, c_0: complex, c_max: complex, front: bool):
        if not self._native_mps:
            return super()._sweep_combined(c_0, c_max, front)
        if self._lk_live:   # both sweeps of the reference together, one lane each
            self._lk.set_lhs_basis(self._lane_bits(0, self._max_no))
            g = self._lk.gradient(self._circuit, block_range=self._block_range, front_layer=front)
            return c_0 * g[0] + c_max * g[1]
        # single-lane engine: the two product-state sweeps of the reference (a combined lhs would be a bond-2 MPS)
        return c_0 * self._sweep(0, front) + c_max * self._sweep(self._max_no, front)

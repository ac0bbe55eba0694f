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

    def _mps_device(self) -> int:
        from ..engine import default_device

        d = self._params.get("device")
        return default_device() if d is None else int(d)

    # ---- native MPS mode: no dense state anywhere (objective_lhs_sur_fast_mps_trotter.py:114-227) ----------
    def _basis(self, state_no: int):
        from ..mps_engine import DeviceMPS

        if state_no not in self._basis_dev:
            self._basis_dev[state_no] = DeviceMPS.basis_state(self._circuit.num_qubits, int(self._state_handler.state_indices[state_no]),
                                                              device=self._mps_device())
        return self._basis_dev[state_no]

    def _evaluate(self, thetas) -> None:
        if not self._native_mps:
            return super()._evaluate(thetas)
        from ..mps_engine import v_dagger_mul_mps

        if self._vh is not None:
            self._vh.close()
        self._vh = v_dagger_mul_mps(self._circuit, thetas, self._target_dev, trunc_thr=self._trunc_thr)   # V^H|target>
        for i in range(self._num_states):
            self._hs[i] = self._basis(i).dot(self._vh)                                                   # <state_i|V^H|target>
        self._grad0 = None

    def _sweep(self, state_no: int, front: bool):
        if not self._native_mps:
            return super()._sweep(state_no, front)
        from ..mps_engine import fast_dot_gradient_mps

        return fast_dot_gradient_mps(self._circuit, self._last_thetas, self._basis(state_no), self._vh, trunc_thr=self._trunc_thr,
                                     block_range=self._block_range, front_layer=front)

    def _sweep_combined(self, c_0: complex, c_max: complex, front: bool):
        if not self._native_mps:
            return super()._sweep_combined(c_0, c_max, front)
        # native MPS engine: the two product-state sweeps of the reference (a combined lhs would be a bond-2 MPS)
        return c_0 * self._sweep(0, front) + c_max * self._sweep(self._max_no, front)

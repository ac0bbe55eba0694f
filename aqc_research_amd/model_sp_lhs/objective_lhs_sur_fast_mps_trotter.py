"""
Surrogate state-preparation objective with the target given as an MPS: drop-in for
SpSurrogateObjectiveFastMpsTrotter (objective_lhs_sur_fast_mps_trotter.py:42-232).  The
target MPS is contracted to a dense state on the GPU once (set_target); after that the
math and the state machine are exactly those of the state-vector objective, which is the
semantics the reference's tests pin for trunc_thr -> 0.
"""
from typing import Optional, Tuple

from ..engine import BUF_Y
from ..mps_operations import check_mps
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
        if not check_mps(target) or len(target[0]) != self._circuit.num_qubits:
            raise ValueError("target must be an MPS in Qiskit format matching the circuit")
        self._target = target
        self._ws.mps_upload(0, target)
        self._ws.mps_to_vec(0, BUF_Y, 0)

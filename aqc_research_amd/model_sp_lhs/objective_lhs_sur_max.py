"""
Surrogate state-preparation objective ``1 - (1-w)|h_0|^2 - w|h_max|^2`` on state vectors:
drop-in for SpSurrogateObjectiveMax (objective_lhs_sur_max.py:32-196).  ``objective`` runs
V^H|target> on the GPU and reads back the n+1 flip-state amplitudes; ``gradient`` runs one
(or two) w/z sweeps; the hysteresis / weight-smoothing state machine stays on the host.
"""
from typing import Optional, Tuple

import numpy as np

from ..engine import BUF_X, BUF_X2, BUF_Y, BUF_Z
from ..parametric_circuit import ParametricCircuit
from .objective_base import DenseStateHandler, SpLHSObjectiveBase


class SpSurrogateObjectiveMax(SpLHSObjectiveBase):
    _gamma = 0.1  # exponential smoothing of the weight (objective_lhs_sur_max.py:40)

    def __init__(
        self,
        *,
        user_parameters: dict,
        circ: ParametricCircuit,
        block_range: Optional[Tuple[int, int]] = None,
        front_layer: bool = False,
        verbose: bool = False,
        grad_scaler=None,
    ):
        super().__init__(user_parameters, circ, verbose=verbose)
        block_range = (0, circ.num_blocks) if block_range is None else block_range
        if not (isinstance(block_range, tuple) and len(block_range) == 2
                and 0 <= block_range[0] < block_range[1] <= circ.num_blocks):
            raise ValueError("block_range must be a tuple (from, to) with 0 <= from < to <= num_blocks")
        if not isinstance(front_layer, (bool, np.bool_)):
            raise TypeError("front_layer must be bool")
        self._block_range = (int(block_range[0]), int(block_range[1]))
        self._front_layer = bool(front_layer)
        self._fidelity = float(-1)
        self._grad_scaler = grad_scaler
        self._hs = np.zeros(self._num_states, dtype=np.complex128)
        self._max_no = 0
        self._dense = isinstance(self._state_handler, DenseStateHandler)
        # One native call per objective(): V^H|target>, the flip-state amplitudes AND (speculatively) the sweep
        # for |state_0>, so that the gradient() call that optimizers issue right after costs no extra GPU
        # round trip.  Pure caching: values are exactly those of separate calls.
        self._speculative = bool(user_parameters.get("speculative_gradient", True)) and not self._dense
        self._grad0 = None       # cached complex gradient of the |state_0> term at self._last_thetas
        self._gradc = None       # cached complex gradient of the combined sweep at self._last_thetas (a flip state leads)
        self._gradc_max_no = -1  # leading state that sweep was taken for
        self._x2_state = -1      # state currently held in BUF_X2
        if not self._dense and self._ws is not None:
            self._ws.set_basis(BUF_X, int(self._state_handler.state_indices[0]))
            self._ws.gather_setup(self._state_handler.state_indices)

    def _projections(self) -> np.ndarray:
        """hs[i] = <state_i|V^H|target> for every state."""
        ws = self._ws
        if not self._dense:
            return ws.gather(BUF_Z, self._state_handler.state_indices)[0]
        out = np.empty(self._num_states, dtype=np.complex128)
        for i in range(self._num_states):
            ws.upload(BUF_X, self._state_handler.init_state(i))
            out[i] = ws.vdot(BUF_X, BUF_Z)[0]
        return out

    def _load_lhs(self, state_no: int) -> None:
        if self._dense:
            self._ws.upload(BUF_X, self._state_handler.init_state(state_no))
        else:
            self._ws.set_basis(BUF_X, int(self._state_handler.state_indices[state_no]))

    def _evaluate(self, thetas: np.ndarray) -> None:
        """Z = V^H|target> and hs[i] = <state_i|Z> on the device (objective_lhs_sur_max.py:96-108)."""
        ws = self._ws
        front = bool(self._front_layer or self._block_range == (0, self._circuit.num_blocks))
        self._gradc = None
        if self._dense:
            ws.set_thetas(thetas)
            ws.apply(True, BUF_Y, BUF_Z)
            self._hs[:] = self._projections()
            self._grad0 = None
        elif self._speculative and (self._max_no != 0 or getattr(ws, "prefers_surrogate_eval", False)):
            # a flip state leads (or the workspace is a lane of a lockstep batch, where every lane files this one kind of
            # request): ONE native call does V^H, the amplitudes, the hysteresis (the weight stays: it moves in gradient())
            # and the sweep from the combination of |state_0> and the leading state under that state
            # (aqc_ws_surrogate_eval, mode 2), so the gradient() call that follows costs no GPU round trip either
            w = np.array([self._weight], dtype=np.float64)
            mx = np.array([self._max_no], dtype=np.int64)
            _, _, hs, gc = ws.surrogate_eval(thetas, w, mx, 2, self._block_range, front)
            self._hs[:] = hs[0]
            self._grad0, self._gradc, self._gradc_max_no = None, gc[0], int(mx[0])
            self._x2_state = -1
        else:
            # the sweep from |state_0> rides along only while |state_0> leads: otherwise gradient() runs ONE sweep from the
            # combination of |state_0> and the leading state (see gradient), and a speculative one would be wasted
            spec = self._speculative and self._max_no == 0
            hs, g = ws.eval(thetas, vdag=True, gather=True, grad=spec, x_buf=BUF_X,
                            block_range=self._block_range, front_layer=front)
            self._hs[:] = hs[0]
            self._grad0 = g[0] if spec else None

    def objective(self, thetas: np.ndarray) -> float:
        if self._target is None:
            raise RuntimeError("set_target() has not been called")
        self._store_latest_thetas(thetas)
        self._evaluate(thetas)
        np.copyto(self._hs2, np.absolute(self._hs) ** 2)
        # hysteresis: the leading state changes only for a 10 % better candidate (:113-117)
        max_proj = self._hs2[self._max_no]
        if 1.1 * max_proj < self._hs2.max():   # (otherwise no candidate passes the test below: skip the Python loop)
            for i in range(self._num_states):
                if 1.1 * max_proj < self._hs2[i]:
                    max_proj = self._hs2[i]
                    self._max_no = i
        if self._gradc is not None and self._gradc_max_no != self._max_no:
            self._gradc = None   # (a tie broken differently on the device: gradient() sweeps again)
        wgh = self._weight
        self._fobj = float(1.0 - (1.0 - wgh) * self._hs2[0] - wgh * self._hs2[self._max_no])
        self._fidelity = float(self._hs2[0])
        self._service.on_end_objective()
        return self._fobj

    def _sweep(self, state_no: int, front: bool) -> np.ndarray:
        ws = self._ws
        if self._dense:
            self._load_lhs(state_no)
            ws.grad(self._block_range, front)
            return ws.get_grads()[0]
        if state_no == 0:
            if self._grad0 is not None:
                g, self._grad0 = self._grad0, None
                return g
            return ws.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X, block_range=self._block_range, front_layer=front)[1][0]
        if self._x2_state != state_no:
            ws.set_basis(BUF_X2, int(self._state_handler.state_indices[state_no]))
            self._x2_state = state_no
        return ws.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2, block_range=self._block_range, front_layer=front)[1][0]

    def _sweep_combined(self, c_0: complex, c_max: complex, front: bool) -> np.ndarray:
        """c_0 g(|state_0>) + c_max g(|state_max>) as one sweep from the combined lhs state."""
        ws = self._ws
        self._grad0 = None
        if self._dense:
            x = np.conj(c_0) * self._state_handler.init_state(0) + np.conj(c_max) * self._state_handler.init_state(self._max_no)
            ws.upload(BUF_X, x)
            ws.grad(self._block_range, front)
            return ws.get_grads()[0]
        idx = self._state_handler.state_indices
        ws.set_combo(BUF_X2, [[int(idx[0]), int(idx[self._max_no])]], [[np.conj(c_0), np.conj(c_max)]])
        self._x2_state = -1
        return ws.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2, block_range=self._block_range, front_layer=front)[1][0]

    def gradient(self, thetas: np.ndarray) -> np.ndarray:
        self._service.on_begin_gradient(self._fobj, thetas, self._fidelity)  # may raise (stoppers)
        self._calc_objective_before_gradient(thetas)
        front = bool(self._front_layer or self._block_range == (0, self._circuit.num_blocks))
        if self._gradc is not None:
            # taken together with the objective: c_0 g_0 + c_max g_max under the weight and the leading state of this pair
            # (c_max = 0 and c_0 = -2 conj(h_0) when the hysteresis went back to |state_0>)
            full_grad, self._gradc = self._gradc.real.copy(), None
        elif self._max_no == 0:
            grad_0 = self._sweep(0, front)
            full_grad = (grad_0 * (-2 * np.conj(self._hs[0]))).real.copy()
        else:
            # objective_lhs_sur_max.py:147-175 runs two sweeps and combines them as c_0 g_0 + c_max g_max.  The gradient of
            # <V x|y> is conjugate-linear in x, so that sum IS the gradient from x = conj(c_0)|state_0> + conj(c_max)|state_max>:
            # one sweep, the same number up to rounding in the last place
            c_0 = -2 * (1 - self._weight) * np.conj(self._hs[0])
            c_max = -2 * self._weight * np.conj(self._hs[self._max_no])
            full_grad = self._sweep_combined(c_0, c_max, front).real.copy()
        if self._grad_scaler:
            full_grad *= self._grad_scaler.estimate(self._fobj)
        self._weight += self._gamma * (float(np.sqrt(abs(self._fobj))) - self._weight)
        self._service.on_end_gradient(self._fobj, self._fidelity, full_grad, self._hs2, self._weight)
        return full_grad

    @property
    def fidelity(self) -> float:
        return self._fidelity

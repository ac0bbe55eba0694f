"""
Multi-start optimisation at kernel speed: B independent state-preparation problems that share one ansatz
(the seeds / restarts / targets of one time horizon -- config 4 of BASELINE.json) are evaluated as the B lanes of
one workspace AND optimised by one vectorised L-BFGS, so the host does O(1) NumPy work per iteration instead of
B Python optimizer loops (``lockstep.py`` keeps the per-job scipy optimizers and is host-bound at ~6 k evals/s).

``BatchedSurrogateObjective`` is the lane-vectorised form of ``SpSurrogateObjectiveMax`` (objective_lhs_sur_max.py:
82-191: surrogate ``1 - (1-w)|h_0|^2 - w|h_max|^2`` over the flip states, 10 % hysteresis of the leading state,
exponentially smoothed weight); ``batched_lbfgs`` is a standard L-BFGS (two-loop recursion, memory m, Armijo
backtracking) written over arrays of shape (B, T).  The optimizer trajectory is not the reference's (that one is
scipy's L-BFGS-B behind a Qiskit wrapper and is not pinned by anything); what is checked is that every
lane reaches the fidelity scipy reaches on the same problem.
"""
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .engine import BUF_X, BUF_X2, BUF_Y, HipContext, Workspace
from .model_sp_lhs.objective_base import ThinStateHandler

__all__ = ["BatchedSurrogateObjective", "BatchedMpsSurrogateObjective", "BatchedSketchingObjective", "batched_lbfgs"]


class _RawResults:
    """(hs, g0, None, max_no) of the last evaluation, as ``commit`` and ``batched_lbfgs`` index it.  g0 -- the raw complex
    gradient of the sweep from |state_0>, available while |state_0> leads on every lane -- is the device's combined gradient
    divided by c_0; the division is only done when somebody asks for it."""

    def __init__(self, hs, gc, c0, max_no):
        self._hs, self._gc, self._c0, self._max_no, self._g0 = hs, gc, c0, max_no.copy(), None

    def __getitem__(self, i):
        if i == 0:
            return self._hs
        if i == 1:
            if self._c0 is None:
                return None
            if self._g0 is None:
                self._g0 = self._gc / self._c0[:, None]
            return self._g0
        if i == 2:
            return None
        if i == 3:
            return self._max_no
        raise IndexError(i)


class BatchedSurrogateObjective:
    """B lanes of the surrogate state-preparation objective on one workspace.  ``targets``: (B, 2^n) complex128;
    ``base_index``: computational-basis preparation (e.g. the Neel pattern) shared by all lanes."""

    _gamma = 0.1  # objective_lhs_sur_max.py:40

    def __init__(self, circ, targets: Optional[np.ndarray], *, max_flips: int = 1, base_index: int = 0,
                 block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True, device: Optional[int] = None,
                 lanes: Optional[int] = None):
        """``targets`` may be None when ``lanes`` is given: the lanes' targets are then handed over on the device
        (``target_from``) -- how a driver re-uses one objective for the jobs of a horizon."""
        if targets is not None:
            targets = np.ascontiguousarray(targets, dtype=np.complex128)
            if targets.ndim != 2 or targets.shape[1] != circ.dimension:
                raise ValueError("targets must have shape (lanes, 2^n)")
            lanes = targets.shape[0]
        elif lanes is None or lanes < 1:
            raise ValueError("give the targets or the number of lanes")
        self.circ, self.batch, self.T = circ, int(lanes), circ.num_thetas
        self._states = ThinStateHandler(circ.num_qubits, max_flips, base_index=base_index)
        self._idx = np.asarray(self._states.state_indices, dtype=np.int64)
        self._block_range = None if block_range is None else (int(block_range[0]), int(block_range[1]))
        self._front = bool(front_layer or block_range is None or tuple(block_range) == (0, circ.num_blocks))
        self.ws = Workspace(HipContext.of(circ), batch=self.batch, ncols=1, device=device)
        if targets is not None:
            self.ws.upload(BUF_Y, targets)
        self.ws.set_basis(BUF_X, int(self._idx[0]))
        self.ws.gather_setup(self._idx)
        self.num_evals = 0
        self.reset_state()

    def reset_state(self) -> None:
        """A fresh objective (weight 1, |state_0> leads) on the same workspace: the next job of a driver."""
        self.weight = np.ones(self.batch)
        self.max_no = np.zeros(self.batch, dtype=np.int64)
        self.fidelity = np.full(self.batch, -1.0)
        self.last_raw = None

    def target_from(self, lane: int, src: Workspace, src_buf: int, src_lane: int) -> None:
        """Lane ``lane``'s target <- lane ``src_lane`` of buffer ``src_buf`` of another workspace, on the device."""
        self.ws.copy_lane_from(src, src_buf, src_lane, BUF_Y, lane)

    def set_mps_targets(self, mps_list) -> None:
        """Targets given as QiskitMPS tuples, one per lane (mps_dot_objective.py:41 hands them over like that): resident
        device copies through the workspace's slot cache, all lanes contracted to dense states by one launch chain."""
        if len(mps_list) != self.batch:
            raise ValueError("one MPS per lane")
        self.ws.mps_to_vec_batch(list(mps_list), BUF_Y)

    def value_and_grad(self, thetas: np.ndarray, update_state: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """f[B], g[B][T] at thetas[B][T].  ``update_state=False`` evaluates with the current weights / leading states
        without touching them (line-search trials); ``True`` first applies the hysteresis and the weight smoothing --
        once per accepted step, as an objective()/gradient() pair does -- and evaluates under the new state.

        One native call (``aqc_ws_surrogate_eval``): V^H, the amplitudes, the state update and the lhs state of every lane on
        the device, then ONE sweep -- its lhs is the combination conj(c_0)|state_0> + conj(c_max)|state_max> per lane, whose
        gradient IS c_0 g_0 + c_max g_max (the gradient of <V x|y> is conjugate-linear in x) where
        objective_lhs_sur_max.py:147-175 runs two sweeps.  (``_update`` / ``_assemble`` below are the same arithmetic on the
        host: ``commit`` uses them on results that are already known.)"""
        th = np.ascontiguousarray(thetas, dtype=np.float64).reshape(self.batch, self.T)
        self.num_evals += self.batch
        w = np.array(self.weight, dtype=np.float64)
        mx = np.array(self.max_no, dtype=np.int64)
        # the complex gradient is only of use to commit(), and only while |state_0> leads everywhere: with a flip state in the
        # lead before the call the real parts alone cross the bus
        real_only = bool((self.max_no != 0).any())
        f, fid, hs, gc = self.ws.surrogate_eval(th, w, mx, update_state, self._block_range, self._front, real_only=real_only)
        if update_state:
            self.max_no, self.weight, self.fidelity = mx, w, fid
        if real_only:
            self.last_raw = _RawResults(hs, None, None, mx)
            return f, gc
        c0 = None
        if not (mx != 0).any():   # |state_0> leads everywhere: the sweep ran from conj(c_0)|state_0>, c_0 = -2 conj(h_0)
            c0 = -2.0 * np.conj(hs[:, 0])
            if not (np.abs(c0) > 1e-150).all():
                c0 = None
        self.last_raw = _RawResults(hs, gc, c0, mx)
        return f, gc.real.copy()

    def _update(self, hs2: np.ndarray):
        """10 % hysteresis (objective_lhs_sur_max.py:113-117) and weight smoothing (:186), lane-wise; returns (max_no, w)."""
        lanes = np.arange(self.batch)
        max_no = self.max_no.copy()
        best = hs2[lanes, max_no]
        for i in range(hs2.shape[1]):
            better = 1.1 * best < hs2[:, i]
            best = np.where(better, hs2[:, i], best)
            max_no = np.where(better, i, max_no)
        w = self.weight
        f_old = 1.0 - (1.0 - w) * hs2[:, 0] - w * hs2[lanes, max_no]      # the value under the OLD weight feeds the smoothing
        return max_no, w + self._gamma * (np.sqrt(np.abs(f_old)) - w)

    def _assemble(self, hs, g0, gm_in, update_state: bool):
        """Value and gradient from the device results under the current / the updated state.  ``g0`` (complex gradient of the
        sweep from |state_0>) may be None: then -- and whenever some lane leads with a flip state -- the gradient comes from
        one sweep over the combined lhs states.  Keeps the raw results so that ``commit`` can re-assemble accepted trial
        points without another device evaluation while |state_0> leads everywhere."""
        hs2 = np.abs(hs) ** 2
        lanes = np.arange(self.batch)
        max_no, w = self.max_no, self.weight
        if update_state:   # the value and gradient returned below use the NEW state, so that all trial points of the next line
            max_no, w = self._update(hs2)                                          # search see one function
            self.max_no, self.weight, self.fidelity = max_no, w, hs2[:, 0].copy()
        f = 1.0 - (1.0 - w) * hs2[:, 0] - w * hs2[lanes, max_no]
        h0 = hs[:, 0]
        lead = max_no != 0
        if g0 is not None and not lead.any():
            grad = (g0 * (-2.0 * np.conj(h0))[:, None]).real
            self.last_raw = (hs, g0, None, max_no.copy())
            return f, grad
        hm = hs[lanes, max_no]
        c0 = np.where(lead, -2.0 * (1.0 - w), -2.0) * np.conj(h0)
        cm = np.where(lead, -2.0 * w, 0.0) * np.conj(hm)
        idx = np.stack([np.full(self.batch, self._idx[0]), np.where(lead, self._idx[max_no], -1)], axis=1)
        self.ws.set_combo(BUF_X2, idx, np.stack([np.conj(c0), np.conj(cm)], axis=1))
        _, gc = self.ws.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2, block_range=self._block_range,
                             front_layer=self._front)
        self.last_raw = (hs, None, None, max_no.copy())
        return f, gc.real.copy()

    def commit(self, rows_hs: np.ndarray, rows_g0: Optional[np.ndarray]) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """State update (hysteresis + weight smoothing) at points whose device results are already known -- the trial
        points a line search accepted, lane by lane -- and the value / gradient under the new state, assembled on the
        host.  Returns None when some lane leads (or would lead) with a flip state other than |state_0>: the sweep then
        depends on the state chosen now, so the caller evaluates on the device instead."""
        if rows_g0 is None or (self.max_no != 0).any():
            return None
        probe, _ = self._update(np.abs(rows_hs) ** 2)
        if (probe != 0).any():
            return None            # caller re-evaluates on the device
        return self._assemble(rows_hs, rows_g0, None, True)

    def minimize_on_device(self, x0: np.ndarray, *, maxiter: int = 100, memory: int = 10, gtol: float = 1e-7, ftol: float = 1e-12,
                           fidelity_thr: float = 0.0, max_backtracks: int = 12) -> Dict:
        """All lanes minimised by the device-resident L-BFGS (``aqc_ws_lbfgs``): the same algorithm as ``batched_lbfgs``
        on this objective (two-loop recursion, Armijo backtracking, state update once per accepted step) with thetas,
        gradients and history resident in HBM -- the host only reads a few flags per step.  Starts from a fresh objective
        state (weight 1, leading state |state_0>), like a new objective object."""
        import ctypes

        from . import _lib

        x = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.batch, self.T)
        xo = np.empty_like(x)
        f = np.empty(self.batch)
        fid = np.empty(self.batch)
        w = np.empty(self.batch)
        nit = np.zeros(self.batch, dtype=np.int64)
        max_no = np.zeros(self.batch, dtype=np.int64)
        nfev = ctypes.c_int64()
        lo, hi = (-1, -1) if self._block_range is None else self._block_range
        i64 = ctypes.POINTER(ctypes.c_int64)
        self.ws._touch(_lib.BUF_Z, _lib.BUF_W, _lib.BUF_ZW, _lib.BUF_X2)   # rewritten by the driver
        _lib.check(self.ws._L.aqc_ws_lbfgs(self.ws.handle, _lib.dptr(x), int(maxiter), int(memory), float(gtol), float(ftol),
                                          float(fidelity_thr), int(max_backtracks), int(lo), int(hi), int(self._front),
                                          _lib.dptr(xo), _lib.dptr(f), _lib.dptr(fid), nit.ctypes.data_as(i64), ctypes.byref(nfev),
                                          _lib.dptr(w), max_no.ctypes.data_as(i64)))
        # the objective's state follows the device's: later host-side evaluations continue from where the optimizer stopped
        self.fidelity, self.weight, self.max_no = fid, w, max_no
        self.last_raw = None
        self.num_evals += int(nfev.value) * self.batch
        return {"x": xo, "fun": f, "nit": nit, "nfev": int(nfev.value), "fidelity": fid}

    def close(self) -> None:
        self.ws.close()


class BatchedMpsSurrogateObjective:
    """``BatchedSurrogateObjective`` for registers beyond dense reach: B lanes of the surrogate state-preparation objective with
    the targets as MPS (``SpSurrogateObjectiveFastMpsTrotter``, objective_lhs_sur_fast_mps_trotter.py:99-227, one object per job in
    the reference) on the lockstep lanes of the native MPS engine (``mps_engine.LockstepLanes``, bonds <= 32).

    One evaluation = two phases on 2B lanes (lane B + l is problem l seen from its leading flip state): V^H|target_l> once per
    problem plus the amplitudes of |state_0> and of its n single-flip states; the state machine of objective_lhs_sur_max.py:99-191
    (10 % hysteresis, smoothed weight) on the host; then both gradient walks -- from |state_0> and from the leading state -- together,
    combined as c_0 g_0 + c_max g_max (:147-175).  ``targets``: one ``DeviceMPS`` per lane, or one for all lanes."""

    _gamma = 0.1  # objective_lhs_sur_max.py:40

    def __init__(self, circ, targets, *, base_index: int = 0, trunc_thr: float = 1e-6, max_bond: int = 0,
                 block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True, device: Optional[int] = None, lanes: Optional[int] = None):
        from .mps_engine import LockstepLanes

        tg = list(targets) if isinstance(targets, (list, tuple)) else None
        if tg is None and (lanes is None or lanes < 1):
            raise ValueError("one target for all lanes needs the number of lanes")
        self.circ, self.batch, self.T, self.n = circ, int(len(tg) if tg is not None else lanes), circ.num_thetas, circ.num_qubits
        self._trunc, self._max_bond = float(trunc_thr), int(max_bond)
        self._block_range = None if block_range is None else (int(block_range[0]), int(block_range[1]))
        self._front = bool(front_layer or block_range is None or tuple(block_range) == (0, circ.num_blocks))
        base = int(base_index)
        self._base_bits = np.array([(base >> q) & 1 for q in range(self.n)], dtype=np.uint8)
        self.lanes = LockstepLanes(self.n, 2 * self.batch, device)
        self.lanes.set_targets(tg + tg if tg is not None else targets)
        self.num_evals = 0
        self._lhs_of = None
        self.reset_state()

    def reset_state(self) -> None:
        """A fresh objective (weight 1, |state_0> leads): the next job of a driver."""
        self.weight = np.ones(self.batch)
        self.max_no = np.zeros(self.batch, dtype=np.int64)
        self.fidelity = np.full(self.batch, -1.0)
        self.last_hs = None

    def _lhs_bits(self, max_no: np.ndarray) -> np.ndarray:
        bits = np.tile(self._base_bits, (2 * self.batch, 1))
        lead = np.nonzero(max_no)[0]
        bits[self.batch + lead, max_no[lead] - 1] ^= 1            # state i >= 1 flips qubit i - 1 (objective_base.py:42-255)
        return bits

    def _upload_lhs(self, max_no: np.ndarray) -> None:
        if self._lhs_of is None or (self._lhs_of != max_no).any():
            self.lanes.set_lhs_basis(self._lhs_bits(max_no))
            self._lhs_of = max_no.copy()

    def value_and_grad(self, thetas: np.ndarray, update_state: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """f[B], g[B][T] at thetas[B][T]; ``update_state`` as in ``BatchedSurrogateObjective.value_and_grad``."""
        th = np.ascontiguousarray(thetas, dtype=np.float64).reshape(self.batch, self.T)
        B = self.batch
        self.num_evals += B
        self._upload_lhs(self.max_no)
        amps = self.lanes.apply_vh(self.circ, np.concatenate([th, th]), trunc_thr=self._trunc, max_bond=self._max_bond, flips=True, half=True)
        hs = amps[:B]                                              # hs[l][i] = <state_i|V^H|target_l>
        hs2 = np.abs(hs) ** 2
        rows = np.arange(B)
        max_no, w = self.max_no, self.weight
        if update_state:                                           # hysteresis (:113-117) and weight smoothing (:186)
            max_no = max_no.copy()
            best = hs2[rows, max_no]
            for i in range(hs2.shape[1]):
                better = 1.1 * best < hs2[:, i]
                best = np.where(better, hs2[:, i], best)
                max_no = np.where(better, i, max_no)
            f_old = 1.0 - (1.0 - w) * hs2[:, 0] - w * hs2[rows, max_no]
            w = w + self._gamma * (np.sqrt(np.abs(f_old)) - w)
            self._upload_lhs(max_no)
            self.max_no, self.weight, self.fidelity = max_no, w, hs2[:, 0].copy()
        f = 1.0 - (1.0 - w) * hs2[:, 0] - w * hs2[rows, max_no]
        lead = max_no != 0
        c0 = np.where(lead, -2.0 * (1.0 - w), -2.0) * np.conj(hs[:, 0])
        cm = np.where(lead, -2.0 * w, 0.0) * np.conj(hs[rows, max_no])
        g = self.lanes.gradient(self.circ, block_range=self._block_range, front_layer=self._front)
        self.last_hs = hs
        return f, (c0[:, None] * g[:B] + cm[:, None] * g[B:]).real

    def close(self) -> None:
        self.lanes.close()


def batched_lbfgs(fun: Callable[[np.ndarray, bool], Tuple[np.ndarray, np.ndarray]], x0: np.ndarray, *, maxiter: int = 100,
                  memory: int = 10, gtol: float = 1e-7, ftol: float = 1e-12, c1: float = 1e-4, max_backtracks: int = 12,
                  stop: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None) -> Dict:
    """Minimises B functions at once: ``fun(x[B][T], update_state) -> (f[B], g[B][T])``.  Lanes that converged keep
    being evaluated at their final point (the batch is evaluated as a whole anyway) but no longer move.
    ``stop(f, x) -> bool[B]`` is an optional extra per-lane stopping rule (e.g. a fidelity threshold)."""
    x = np.array(x0, dtype=np.float64)
    B, T = x.shape
    f, g = fun(x, True)
    S = np.zeros((memory, B, T))
    Y = np.zeros((memory, B, T))
    rho = np.zeros((memory, B))
    count = 0
    active = np.ones(B, dtype=bool)
    nit = np.zeros(B, dtype=np.int64)
    nfev = 1
    lanes = np.arange(B)
    for _ in range(maxiter):
        active &= np.max(np.abs(g), axis=1) > gtol
        if stop is not None:
            active &= ~stop(f, x)
        if not active.any():
            break
        # two-loop recursion, all lanes at once
        q = g.copy()
        k = min(count, memory)
        alpha = np.zeros((k, B))
        for j in range(k - 1, -1, -1):
            slot = (count - k + j) % memory
            alpha[j] = rho[slot] * np.einsum("bt,bt->b", S[slot], q)
            q -= alpha[j][:, None] * Y[slot]
        if k:
            last = (count - 1) % memory
            yy = np.einsum("bt,bt->b", Y[last], Y[last])
            gamma = np.where((yy > 0) & (rho[last] > 0), 1.0 / np.where(rho[last] * yy > 0, rho[last] * yy, 1.0), 1.0)
            q *= gamma[:, None]
        else:
            q /= np.maximum(np.linalg.norm(g, axis=1), 1.0)[:, None]      # first step: at most unit length
        for j in range(k):
            slot = (count - k + j) % memory
            beta = rho[slot] * np.einsum("bt,bt->b", Y[slot], q)
            q += (alpha[j] - beta)[:, None] * S[slot]
        d = -q
        slope = np.einsum("bt,bt->b", g, d)
        bad = slope >= 0                                                  # not a descent direction: steepest descent
        d[bad] = -g[bad]
        slope[bad] = -np.einsum("bt,bt->b", g[bad], g[bad])
        # Armijo backtracking, lane-wise step lengths; trial evaluations do not touch the objective's state
        step = np.where(active, 1.0, 0.0)
        done = ~active
        f_new, g_new, x_new = f.copy(), g.copy(), x.copy()
        owner = getattr(fun, "__self__", None)          # objectives that expose their raw device results can commit
        raw_hs = raw_g0 = None                          # accepted trial points on the host (no second evaluation)
        can_commit = (owner is not None and hasattr(owner, "commit") and getattr(owner, "last_raw", None) is not None
                      and owner.last_raw[1] is not None)
        if can_commit:
            raw_hs, raw_g0 = owner.last_raw[0].copy(), owner.last_raw[1].copy()   # rows of lanes that do not move
        for _bt in range(max_backtracks):
            trial = x + step[:, None] * d
            ft, gt = fun(trial, False)
            nfev += 1
            ok = (~done) & (ft <= f + c1 * step * slope)
            f_new[ok], g_new[ok], x_new[ok] = ft[ok], gt[ok], trial[ok]
            if can_commit and owner.last_raw[1] is None:
                can_commit = False    # (cannot happen while the state is frozen during the trials; stay safe)
            if can_commit:
                raw_hs[ok], raw_g0[ok] = owner.last_raw[0][ok], owner.last_raw[1][ok]
            done |= ok
            if done.all():
                break
            step = np.where(done, step, 0.5 * step)
        moved = done & active & (x_new != x).any(axis=1)
        # accepted points: the state update (hysteresis / weight smoothing) happens here, once per step
        acc = owner.commit(raw_hs, raw_g0) if can_commit else None
        if acc is None:
            acc = fun(x_new, True)
            nfev += 1
        f_acc, g_acc = acc
        s = x_new - x
        y = g_acc - g
        sy = np.einsum("bt,bt->b", s, y)
        good = moved & (sy > 1e-12 * np.einsum("bt,bt->b", y, y))
        slot = count % memory
        S[slot] = np.where(good[:, None], s, 0.0)
        Y[slot] = np.where(good[:, None], y, 0.0)
        rho[slot] = np.where(good, 1.0 / np.where(good, sy, 1.0), 0.0)
        count += 1
        small = np.abs(f - f_acc) <= ftol * np.maximum(1.0, np.abs(f))
        nit += active
        active &= moved & ~small
        x, f, g = x_new, f_acc, g_acc
    return {"x": x, "fun": f, "jac": g, "nit": nit, "nfev": nfev, "lanes_converged": ~active, "lanes": lanes}


class BatchedSketchingObjective:
    """B lanes of the full-range AQC objective ``1 - Re<V, U_b>/d`` (``SketchingObjectiveEx`` +
    ``FullRangeSketchingVectors``, sk_core.py:167-326): random restarts and / or different target unitaries on one
    workspace.  ``targets``: (B, d, d) complex128, or (d, d) shared by all ``lanes``."""

    def __init__(self, circ, targets: np.ndarray, lanes: Optional[int] = None, device: Optional[int] = None):
        t = np.ascontiguousarray(targets, dtype=np.complex128)
        d = circ.dimension
        if t.shape == (d, d):
            if not lanes:
                raise ValueError("give the number of lanes for a shared target")
            t = np.ascontiguousarray(np.broadcast_to(t, (int(lanes), d, d)))
        if t.ndim != 3 or t.shape[1:] != (d, d):
            raise ValueError("targets must have shape (lanes, 2^n, 2^n)")
        self.circ, self.batch, self.T, self._d = circ, t.shape[0], circ.num_thetas, d
        self.ws = Workspace(HipContext.of(circ), batch=self.batch, ncols=d, device=device)
        self.ws.upload(BUF_Y, t)
        self.ws.set_identity(BUF_X)
        self.num_evals = 0

    def value_and_grad(self, thetas: np.ndarray, update_state: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        from .engine import BUF_Z

        th = np.ascontiguousarray(thetas, dtype=np.float64).reshape(self.batch, self.T)
        self.ws.set_thetas(th)
        self.ws.apply(True, BUF_Y, BUF_Z)                       # V^H U            (sk_core.py:191)
        trace = self.ws.vdot(BUF_X, BUF_Z)                      # <I|V^H U>        (:192)
        self.ws.grad(None, True)                                # sweep            (:193)
        self.num_evals += self.batch
        return 1.0 - trace.real / self._d, -self.ws.get_grads().real / self._d

    def close(self) -> None:
        self.ws.close()

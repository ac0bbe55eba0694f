"""
Optimizer wrapper + stoppers around the objective objects (contract of optimizer.py:36-628).

The reference hands ``objv.objective`` / ``objv.gradient`` to Qiskit's L_BFGS_B / ADAM wrappers;
Qiskit is not a dependency here, so L-BFGS-B goes straight to ``scipy.optimize.minimize`` (which
is what Qiskit's class wraps) and ADAM is a few lines of NumPy on the parameter vector.  These
are host-side scalar/parameter-vector operations around the path, not part of it.
"""
from time import perf_counter
from typing import Any, Callable, Optional, Union

import numpy as np
from scipy.optimize import minimize


class StagnantOptimizationWarning(UserWarning):
    """No progress of the objective for too many iterations (optimizer.py:30-33)."""


class _Deadline:
    """Wall-clock limit shared by the two timeout classes: whole seconds, rounded up, counted from ``arm()``; a limit <= 0
    never expires (the reference's convention for "no limit", optimizer.py:52-60,177-186)."""

    def __init__(self, seconds: int):
        self._seconds, self._at = int(seconds), None

    def arm(self) -> None:
        self._at = None if self._seconds <= 0 else int(round(perf_counter() + 0.5)) + self._seconds

    def expired(self) -> bool:
        return self._at is not None and perf_counter() > self._at


class _BestSoFar:
    """Smallest objective seen and the iteration it was seen at: the "no improvement for N iterations" rule of
    NotImproveStopper (optimizer.py:95-116) and EarlyStopper (:305-318)."""

    def __init__(self, window: int):
        self.window = int(window)
        self.clear()

    def clear(self) -> None:
        self.fobj, self.iteration = np.inf, 0

    def stale(self, fobj: float, iter_no: int) -> bool:
        """Records ``fobj``; True when nothing better has come for more than ``window`` iterations."""
        if fobj < self.fobj:
            self.fobj, self.iteration = fobj, iter_no
            return False
        return iter_no - self.iteration > self.window


class TimeoutStopper:
    """Raises TimeoutError once ``time_limit`` seconds have passed since construction (optimizer.py:36-65)."""

    def __init__(self, *, time_limit: int):
        self._deadline = _Deadline(time_limit)
        self._deadline.arm()

    def check(self):
        if self._deadline.expired():
            raise TimeoutError("Early termination: timeout")


class NotImproveStopper:
    """Flags (or raises StagnantOptimizationWarning on) ``num_iters`` iterations without a new minimum (optimizer.py:68-118)."""

    def __init__(self, *, num_iters: int, raise_ex: bool = True):
        if not num_iters > 1:
            raise ValueError("num_iters must be > 1")
        self._best, self._raise_ex, self._enabled = _BestSoFar(num_iters), bool(raise_ex), True

    def reset(self):
        self._best.clear()
        self._enabled = True

    def disable(self):
        self._enabled = False

    def check(self, fobj: float, iter_no: int) -> bool:
        if not (self._enabled and self._best.stale(fobj, iter_no)):
            return False
        if self._raise_ex:
            raise StagnantOptimizationWarning("Early termination, no improvement")
        return True


class SmallObjectiveStopper:
    """Raises StopIteration when the objective falls below ``fobj_thr`` (optimizer.py:121-155)."""

    def __init__(self, *, fobj_thr: float):
        self._fobj_thr = float(fobj_thr)

    def check(self, fobj: float):
        if fobj < self._fobj_thr:
            raise StopIteration(f"Early termination, objective fobj={fobj:0.5f} fell below the threshold={self._fobj_thr:0.5f}")


class TimeoutChecker:
    """Timeout that stores the best result through ``on_stop`` before raising (optimizer.py:158-225); ``time_limit`` may be
    the user-parameter dictionary with a "timeout" entry."""

    def __init__(self, *, time_limit: Union[int, dict], start_immediately: bool = True):
        seconds = time_limit.get("timeout", -1) if isinstance(time_limit, dict) else time_limit
        self._deadline, self._results = _Deadline(seconds), {}
        if start_immediately:
            self.start()

    def start(self):
        self._deadline.arm()

    def check(self, fobj: float, thetas: np.ndarray, on_stop: Optional[Callable] = None):
        if not self._deadline.expired():
            return
        if on_stop is not None:
            self._results = on_stop(fobj, thetas)
        raise TimeoutError("early termination: timeout")

    optim_results = property(lambda self: self._results)


class EarlyStopper:
    """Objective / fidelity thresholds and no-improvement window (optimizer.py:228-336).  The three rules are tried in the
    reference's order: objective threshold, stagnation (reports the BEST point seen, not the current one), fidelity threshold."""

    def __init__(self, fobj_thr: Optional[float] = None, fidelity_thr: Optional[float] = None, num_iters: Optional[int] = None):
        if fidelity_thr is not None and not 0 < fidelity_thr <= 1:
            raise ValueError("fidelity_thr must be in (0, 1]")
        self._fobj_thr, self._fidelity_thr = fobj_thr, fidelity_thr
        self._best = _BestSoFar(num_iters) if num_iters else None
        self._best_thetas, self._results = None, {}

    def _stop(self, on_stop: Callable, fobj, thetas, why: str):
        self._results = on_stop(fobj, thetas)
        raise StopIteration(why)

    def check(self, fobj, fidelity, thetas: np.ndarray, iter_no: int, on_stop: Callable):
        if fobj is not None and self._fobj_thr is not None and fobj < self._fobj_thr:
            self._stop(on_stop, fobj, thetas, f"early termination, objective fobj={fobj:0.5f} fell below the threshold={self._fobj_thr:0.5f}")
        if fobj is not None and self._best is not None:
            improved_before = self._best.fobj
            if self._best.stale(fobj, iter_no):
                self._stop(on_stop, self._best.fobj, thetas if self._best_thetas is None else self._best_thetas,
                           "Early termination, no improvement")
            if self._best.fobj < improved_before or self._best_thetas is None:
                self._best_thetas = thetas.copy()
        if fidelity is not None and self._fidelity_thr is not None and fidelity >= self._fidelity_thr:
            self._stop(on_stop, fobj, thetas, f"early termination, fidelity={fidelity:0.3f} exceeded the threshold={self._fidelity_thr:0.3f}")

    optim_results = property(lambda self: self._results)


class GradientAmplifier:
    """Logarithmic gradient boost on a barren plateau (optimizer.py:339-398)."""

    def __init__(self, history: int = 5, strong: bool = False, verbose: bool = False):
        if history < 3:
            raise ValueError("history must be >= 3")
        self._history, self._counter = np.zeros(history), 0
        self._log, self._scale = (np.log if strong else np.log10), 1.0

    def estimate(self, fobj: float) -> float:
        self._history[self._counter % self._history.size] = fobj
        self._counter += 1
        if self._counter < self._history.size:
            return 1.0
        dev = float(np.ptp(self._history))
        new_scale = max(-float(self._log(max(dev, 1e-8))), 1.0)
        self._scale += 0.3 * (new_scale - self._scale)
        return self._scale


class _Result:
    def __init__(self, x, fun, nit, nfev, njev):
        self.x, self.fun, self.nit, self.nfev, self.njev = x, fun, nit, nfev, njev


def _adam(fun, jac, x0, maxiter, lr, beta1=0.9, beta2=0.99, eps=1e-8, tol=1e-6):  # Qiskit ADAM: noise_factor 1e-8
    x, m, v = np.array(x0, dtype=float), np.zeros_like(x0, dtype=float), np.zeros_like(x0, dtype=float)
    t = 0
    for t in range(1, maxiter + 1):
        g = jac(x)
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        step = lr * np.sqrt(1 - beta2**t) / (1 - beta1**t) * m / (np.sqrt(v) + eps)
        x = x - step
        if np.linalg.norm(step) < tol:
            break
    return _Result(x, float(fun(x)), t, t, t)   # Qiskit reports nfev = t


class AqcOptimizer:
    """Runs the optimisation and assembles the reference's result dictionary
    (optimizer.py:479-628: cost, num_iters, num_fun_ev, num_grad_ev, ini_thetas, thetas, blocks,
    entangler, stats, is_timeout, fidelity)."""

    _optimizers = ["adam", "lbfgs", "cobyla"]   # bobyqa (scikit-quant, absent here) is not offered

    def __init__(self, *, optimizer_name: str = "lbfgs", maxiter: int = 1000, learn_rate: float = 0.1,
                 lbfgs_maxcor: Optional[int] = None, verbose: bool = False):
        if optimizer_name not in self._optimizers:
            raise ValueError(f"unsupported optimizer: {optimizer_name}, expects one of: {self._optimizers}")
        if not maxiter > 0 or not 0 < learn_rate < 1:
            raise ValueError("maxiter must be positive and 0 < learn_rate < 1")
        self._name, self._maxiter, self._lr, self._maxcor, self._verbose = optimizer_name, int(maxiter), learn_rate, lbfgs_maxcor, verbose

    def optimize(self, objv: Any, circ, thetas_0: np.ndarray, *, stopper=None, timeout=None) -> dict:
        for attr in ("objective", "gradient", "set_status_trackers"):
            if not hasattr(objv, attr):
                raise TypeError(f"objective object lacks '{attr}'")
        result = {
            "cost": float(1e30), "num_iters": 0, "num_fun_ev": 0, "num_grad_ev": 0,
            "ini_thetas": thetas_0.copy(), "thetas": thetas_0.copy(), "blocks": circ.blocks.copy(),
            "entangler": circ.entangler, "stats": {},
        }
        is_timeout = False
        try:
            objv.set_status_trackers(timeout=timeout, stopper=stopper)
            if self._name == "adam":
                res = _adam(objv.objective, objv.gradient, thetas_0, self._maxiter, self._lr)
            elif self._name == "cobyla":   # gradient-free choice of optimizer.py:593 (Qiskit COBYLA wraps scipy's)
                r = minimize(objv.objective, np.array(thetas_0, dtype=float), method="COBYLA", tol=1e-3,
                             options={"maxiter": self._maxiter})
                res = _Result(r.x, float(r.fun), getattr(r, "nit", None) or r.nfev, r.nfev, 0)
            else:
                opts = {"maxfun": 5 * self._maxiter, "maxiter": self._maxiter}
                if self._maxcor:
                    opts["maxcor"] = self._maxcor
                r = minimize(objv.objective, np.array(thetas_0, dtype=float), jac=objv.gradient, method="L-BFGS-B", options=opts)
                res = _Result(r.x, float(r.fun), r.nit, r.nfev, getattr(r, "njev", r.nfev))
            result.update(cost=res.fun, thetas=res.x.copy(), blocks=circ.blocks.copy())
            result["num_iters"] += res.nit or 0
            result["num_fun_ev"] += res.nfev or 0
            result["num_grad_ev"] += res.njev or 0
        except StopIteration:
            result.update(objv.optim_results if hasattr(objv, "optim_results") else stopper.optim_results)
        except TimeoutError:
            is_timeout = True
            result.update(objv.optim_results if hasattr(objv, "optim_results") else timeout.optim_results)
        finally:
            result["is_timeout"] = is_timeout
            if hasattr(objv, "fidelity"):
                result["fidelity"] = objv.fidelity
        if hasattr(objv, "statistics"):
            result["stats"] = objv.statistics
            result["stats"]["is_timeout"] = is_timeout
        return result

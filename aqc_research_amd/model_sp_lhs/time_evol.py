"""
ASP time-evolution driver on top of the HIP objectives -- the horizon logic of
``time_evol_best_init.py`` without its pickling / plotting / Qiskit parts:

* per time horizon a pair of targets (``target_states.py:373-455``): the *ground truth* ``|t1_gt>`` (Trotter circuit with
  ``precise_multiplier()`` = 10x the steps) and the *reference* ``|t1>`` (the plain number of steps), both synthesised on the
  device with the path's own ``V|ini>`` kernel (``trotter.trotter_state``) or, beyond dense reach, on the native MPS engine;
* the fidelity threshold derived from them (``_calc_fidelity_threshold``, ``time_evol_best_init.py:118-140``);
* ``_model_function`` (``:143-218``): TrotterAnsatz of ``num_layers`` layers, Trotter initial point, objective picked by
  ``opts.objective`` in {``sur_max``, ``sur_fast_mps_trotter``} (``_create_objective``, ``:64-115``), L-BFGS under
  ``EarlyStopper(fidelity_thr)`` / ``TimeoutChecker``;
* ``_time_evolution`` (``:221-334``): the circuit-expansion retry loop (``num_expansions``; the reference's ``run_simulation``
  passes 0) and the three fidelity figures ``fid_a1_vs_gt / fid_t1_vs_gt / fid_a1_vs_t1`` of the result record.

Horizons are independent jobs (``:232-233``) and go through ``run_jobs`` (rank-sharded).  On top of the reference's loop:
random restarts of a horizon as lanes of one batched workspace (``num_seeds``; lockstep / vectorised / device-resident L-BFGS).
"""
from time import perf_counter
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from ..job_executor import run_jobs
from ..mps_operations import DenseBackedMPS, check_mps, mps_dot, mps_to_vector, no_truncation_threshold
from ..optimizer import AqcOptimizer, EarlyStopper, GradientAmplifier, TimeoutChecker
from ..parametric_circuit import first_layer_included, layer_to_block_range
from .objective_base import basis_mask_of_circuit
from .objective_lhs_sur_fast_mps_trotter import SpSurrogateObjectiveFastMpsTrotter
from .objective_lhs_sur_max import SpSurrogateObjectiveMax
from .trotter import init_ansatz_to_trotter, neel_init_state, trotter_ansatz, trotter_state

_DENSE_MAX_QUBITS = 24


def precise_multiplier() -> int:
    """Ground-truth targets use this many times the Trotter steps of the reference ones (target_states.py:30-36)."""
    return 10


class UserOptions:
    """The tunables of the driver that concern the path, with the reference's names and defaults
    (user_options.py:25-129).  ``num_horizons`` / ``evol_time_step`` / ``trotter_steps_per_horizon`` are shorthands that
    fill ``trotter_steps`` and ``evol_times`` (the reference hard-codes 6 horizons of 1.2 with 3 steps each)."""

    def __init__(self, **kw):
        self.num_qubits = 12
        self.second_order_trotter = True
        self.delta = 1.0
        self.trunc_thr = 1e-6                          # user_options.py:55
        self.trunc_thr_target = no_truncation_threshold()
        self.evol_time_step = 1.2                      # big step: evolution time per horizon
        self.num_horizons = 6
        self.trotter_steps_per_horizon = 3             # round(big_step / small_step) = round(1.2 / 0.4)
        self.trotter_steps = None                      # per horizon; None: (1..num_horizons) * trotter_steps_per_horizon
        self.evol_times = None                         # per horizon; None: (1..num_horizons) * evol_time_step
        self.num_layers_inc = 2                        # ansatz layers added per horizon
        self.manual_num_layers = None                  # or an explicit schedule, e.g. [2, 4, 6, 7, 8]
        self.num_expansions = 0                        # circuit expansions per horizon (run_simulation of the reference: 0)
        self.objective = "sur_max"                     # or "sur_fast_mps_trotter" (the reference's default)
        self.ini_state_func = (neel_init_state,)
        self.maxiter = 40
        self.fidelity_thr = 0.995                      # None: automatic, 1.03 x fidelity(|t1>, |t1_gt>)
        self.time_limit = -1
        self.enable_grad_scaling = True
        self.verbose = False
        self.seed = 1234
        self.device = None                  # None: this rank's GPU (LOCAL_RANK under a one-process-per-GPU launcher)
        self.num_seeds = 1                 # random restarts per horizon (lockstep lanes of one workspace)
        self.theta_jitter = 0.1            # restart s > 0 starts from Trotter angles + jitter * pi * U(-1, 1)
        self.vectorised_lbfgs = False      # restarts driven by ONE vectorised L-BFGS (batched_optimizer.py) instead of scipy per lane
        self.device_lbfgs = False          # ... and that L-BFGS resident on the device (aqc_ws_lbfgs), thetas never leave HBM
        self.__dict__.update(kw)
        if self.trotter_steps is None:
            self.trotter_steps = (1 + np.arange(self.num_horizons)) * int(self.trotter_steps_per_horizon)
        if self.evol_times is None:
            self.evol_times = np.round((1 + np.arange(self.num_horizons)) * float(self.evol_time_step), 3)
        if len(self.trotter_steps) != len(self.evol_times):
            raise ValueError("trotter_steps and evol_times must have one entry per horizon")

    @property
    def use_mps(self) -> bool:
        """MPS or full vectors, depending on the objective (user_options.py:126-129)."""
        return self.objective.find("mps") >= 0

    def ini_state_index(self) -> int:
        """Basis index the preparation circuit ``ini_state_func[0]`` produces from |0> (X gates only)."""
        prep = self.ini_state_func[0](self.num_qubits)
        return int(prep) if isinstance(prep, (int, np.integer)) else basis_mask_of_circuit(prep, self.num_qubits)


State = Union[np.ndarray, tuple]


class TargetState:
    """Ground-truth and reference target of one horizon (TargetClassicState / TargetMpsState,
    target_states.py:39-168): ``t1_gt``, ``t1`` are dense vectors, or QiskitMPS tuples when ``opts.use_mps``."""

    def __init__(self, *, num_qubits, num_trot_steps, evol_time, my_id, delta, second_order, t1_gt, t1):
        self.num_qubits, self.num_trot_steps, self.evol_time, self.my_id = int(num_qubits), int(num_trot_steps), float(evol_time), int(my_id)
        self.precise_multiplier, self.delta, self.second_order = precise_multiplier(), float(delta), bool(second_order)
        self.t1_gt, self.t1 = t1_gt, t1


def _evolved_state(opts: UserOptions, circ, thetas: np.ndarray, trunc_thr: float) -> State:
    """V(thetas)|ini_state> as the kind of state the objective works on: dense vector, or -- ``use_mps`` -- a QiskitMPS
    (a ``DenseBackedMPS`` up to 24 qubits: computed by the fused kernels, canonical tensors on demand; the native MPS
    engine beyond).  trot_utils.get_solution_from_optim_result, trotter_evol_utils.py:79-125."""
    n = opts.num_qubits
    ini = opts.ini_state_index()
    if opts.use_mps and n > _DENSE_MAX_QUBITS:
        from ..mps_engine import DeviceMPS, v_mul_mps

        m0 = DeviceMPS.basis_state(n, ini)
        m = v_mul_mps(circ, thetas, m0, trunc_thr=float(trunc_thr))
        try:
            return m.to_qiskit()
        finally:
            m.close()
            m0.close()
    from ..core_operations import v_mul_vec

    vec = np.zeros(circ.dimension, dtype=np.complex128)
    vec[ini] = 1
    out = v_mul_vec(circ, thetas, vec, np.zeros_like(vec), None)
    return DenseBackedMPS(out, float(trunc_thr)) if opts.use_mps else out


def generate_target(opts: UserOptions, my_id: int) -> TargetState:
    """|t1_gt> = precise_Trotter(t)|ini>, |t1> = reference_Trotter(t)|ini> (generate_classic_target /
    generate_mps_target, target_states.py:373-455,458-540): the Trotter circuit is the ansatz itself at its Trotter angles."""
    n, steps, t = opts.num_qubits, int(opts.trotter_steps[my_id]), float(opts.evol_times[my_id])
    states = []
    for num_steps in (steps * precise_multiplier(), steps):
        if opts.use_mps:
            circ = trotter_ansatz(n, num_steps, opts.second_order_trotter)
            th = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=t, delta=opts.delta)
            states.append(_evolved_state(opts, circ, th, opts.trunc_thr_target))
        else:
            states.append(trotter_state(n, evol_time=t, num_steps=num_steps, delta=opts.delta, second_order=opts.second_order_trotter,
                                        ini_index=opts.ini_state_index()))
    return TargetState(num_qubits=n, num_trot_steps=steps, evol_time=t, my_id=my_id, delta=opts.delta,
                       second_order=opts.second_order_trotter, t1_gt=states[0], t1=states[1])


def fidelity(state1: State, state2: State) -> float:
    """|<s1|s2>|^2 for two states of the same kind (trotter.py:413-422)."""
    if isinstance(state1, np.ndarray) and isinstance(state2, np.ndarray):
        return float(np.abs(np.vdot(state1, state2)) ** 2)
    if isinstance(state1, DenseBackedMPS) and isinstance(state2, DenseBackedMPS):   # exact: no tensors needed
        return float(np.abs(np.vdot(state1.dense_state, state2.dense_state)) ** 2)
    if not (check_mps(state1) and check_mps(state2)):
        raise ValueError("fidelity: expects two vectors or two MPS")
    return float(np.abs(mps_dot(state1, state2)) ** 2)


def _create_objective(*, opts: UserOptions, circ, target: State, layer_range: Optional[Tuple[int, int]]):
    """The objective selector of time_evol_best_init.py:64-115 (same parameter dictionary, same two choices)."""
    params = {
        "job_index": 0, "num_qubits": circ.num_qubits, "max_flips": 1, "maxiter": opts.maxiter, "verbose": opts.verbose,
        "enable_optim_stats": True, "num_simulations": 1, "trunc_thr": opts.trunc_thr,
        "state_prep_func": opts.ini_state_func[0], "device": opts.device,
    }
    grad_scaler = GradientAmplifier(history=5, strong=False, verbose=opts.verbose) if opts.enable_grad_scaling else None
    if opts.objective == "sur_max":
        objv = SpSurrogateObjectiveMax(user_parameters=params, circ=circ, block_range=layer_to_block_range(circ, layer_range),
                                       front_layer=first_layer_included(circ, layer_range), verbose=opts.verbose, grad_scaler=grad_scaler)
    elif opts.objective == "sur_fast_mps_trotter":
        objv = SpSurrogateObjectiveFastMpsTrotter(user_parameters=params, circ=circ, layer_range=layer_range, alt_layers=False,
                                                  verbose=opts.verbose, grad_scaler=grad_scaler)
    else:
        raise ValueError(f"unknown objective function: {opts.objective}")
    objv.set_target(target)
    return objv


def _calc_fidelity_threshold(target: TargetState, fidelity_thr: Optional[float] = None) -> Tuple[float, float]:
    """A bit above the fidelity of the reference state but not too high (time_evol_best_init.py:118-140): the larger of
    fidelity(|t1>, |t1_gt>) and the desired least fidelity, or 1.03 x the former when none is given."""
    fid_t1_vs_gt = fidelity(target.t1, target.t1_gt)
    if fidelity_thr is not None:
        if not 0 < fidelity_thr <= 1:
            raise ValueError("fidelity_thr must be in (0, 1]")
        fid_thr = max(fid_t1_vs_gt, float(fidelity_thr))
    else:
        fid_thr = 1.03 * fid_t1_vs_gt
    return fid_thr, fid_t1_vs_gt


def _model_function(*, opts: UserOptions, num_layers: int, evol_time: float, target: State, fid_thr: float) -> dict:
    """One optimisation from the 'perfect' Trotter initial point (time_evol_best_init.py:143-218)."""
    tic = perf_counter()
    if num_layers < 1 or not 0 < fid_thr <= 1:
        raise ValueError("num_layers >= 1 and 0 < fid_thr <= 1 expected")
    layer_range = (0, num_layers)
    circ = trotter_ansatz(opts.num_qubits, num_layers, opts.second_order_trotter)
    thetas_0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=evol_time, delta=opts.delta, layer_range=layer_range)
    objv = _create_objective(opts=opts, circ=circ, target=target, layer_range=layer_range)
    result = AqcOptimizer(optimizer_name="lbfgs", maxiter=int(opts.maxiter), verbose=opts.verbose).optimize(
        objv, circ, thetas_0, stopper=EarlyStopper(fidelity_thr=fid_thr), timeout=TimeoutChecker(time_limit=opts.time_limit))
    result.update({"num_qubits": circ.num_qubits, "num_layers": num_layers, "entangler": circ.entangler, "time": perf_counter() - tic})
    return result


def _time_evolution(*, opts: UserOptions, num_layers: int, num_expansions: int, target: TargetState) -> dict:
    """One horizon from scratch, possibly with several expansions of the ansatz (time_evol_best_init.py:221-334)."""
    if num_layers < 1 or num_expansions < 0:
        raise ValueError("num_layers >= 1 and num_expansions >= 0 expected")
    if target.num_trot_steps != opts.trotter_steps[target.my_id]:
        raise ValueError("target does not belong to this horizon")
    fidelity_thr, fid_t1_vs_gt = _calc_fidelity_threshold(target, opts.fidelity_thr)
    attempt = 0
    while True:
        a_state_result = _model_function(opts=opts, num_layers=num_layers, evol_time=target.evol_time, target=target.t1_gt,
                                         fid_thr=fidelity_thr)
        circ = trotter_ansatz(opts.num_qubits, num_layers, opts.second_order_trotter)
        a1 = _evolved_state(opts, circ, a_state_result["thetas"], opts.trunc_thr)
        # computed directly the fidelity may come out slightly below the objective's own figure (truncated MPS)
        fid_a1_vs_gt = fidelity(a1, target.t1_gt)
        if max(fid_a1_vs_gt, a_state_result.get("fidelity", 0.0)) > fidelity_thr:
            break                      # high enough: next horizon
        if attempt >= num_expansions:
            break                      # no more expansions allowed
        attempt += 1
        num_layers += 1                # one more layer, optimised again from its own Trotter point
    if opts.use_mps:                   # the final figures are recomputed without truncation (:301-310)
        a1 = _evolved_state(opts, circ, a_state_result["thetas"], no_truncation_threshold())
        fid_a1_vs_gt = fidelity(a1, target.t1_gt)
    return {
        "fid_a1_vs_gt": fid_a1_vs_gt, "fid_t1_vs_gt": fid_t1_vs_gt, "fid_a1_vs_t1": fidelity(a1, target.t1),
        "num_qubits": opts.num_qubits, "num_layers": num_layers, "block_reps": 3, "entangler": str(a_state_result["entangler"]),
        "num_trotter_steps": target.num_trot_steps, "evol_time1": target.evol_time, "thetas": a_state_result["thetas"].copy(),
        "blocks": a_state_result["blocks"].copy(), "use_mps": bool(opts.use_mps), "second_order_trotter": bool(opts.second_order_trotter),
        "ini_state_func": getattr(opts.ini_state_func[0], "__name__", str(opts.ini_state_func[0])), "stats": a_state_result.get("stats"),
        # (extras of this driver)
        "fidelity_thr": fidelity_thr, "expansions": attempt, "fidelity": float(a_state_result["fidelity"]), "cost": float(a_state_result["cost"]),
        "num_iters": int(a_state_result["num_iters"]), "num_fun_ev": int(a_state_result["num_fun_ev"]),
        "num_grad_ev": int(a_state_result.get("num_grad_ev", a_state_result["num_iters"])), "num_thetas": int(a_state_result["thetas"].size),
        "optim_time": float(a_state_result["time"]),
    }


def _initial_layers(opts: UserOptions, idx: int) -> int:
    """Manual schedule if given and long enough, the constant increment otherwise (time_evol_best_init.py:366-371)."""
    if isinstance(opts.manual_num_layers, (list, tuple)) and len(opts.manual_num_layers) > idx:
        return int(opts.manual_num_layers[idx])
    return int(opts.num_layers_inc * (idx + 1))


def _horizon_job(job_index: int, cfg: Dict) -> Dict:
    opts: UserOptions = cfg["opts"]
    idx = cfg["horizon"] - 1                # horizons are 1-based in the records
    tic = perf_counter()
    target = generate_target(opts, idx)
    t_target = perf_counter() - tic
    num_layers = _initial_layers(opts, idx)
    # fidelity of the plain Trotter initial point against the ground truth (what the optimisation starts from)
    circ0 = trotter_ansatz(opts.num_qubits, num_layers, opts.second_order_trotter)
    th0 = init_ansatz_to_trotter(circ0, np.zeros(circ0.num_thetas), evol_time=target.evol_time, delta=opts.delta)
    fid0 = fidelity(_evolved_state(opts, circ0, th0, opts.trunc_thr_target), target.t1_gt)
    res = _time_evolution(opts=opts, num_layers=num_layers, num_expansions=int(opts.num_expansions), target=target)
    res.update({"horizon": cfg["horizon"], "evol_time": target.evol_time, "fidelity_trotter_init": float(fid0), "target_time": t_target})
    return res


def _seeded_horizon_job(job_index: int, cfg: Dict) -> Dict:
    """All restarts of one horizon: they share the ansatz, so they run as lanes of one batched workspace
    (lockstep.py); the record of the best restart is returned with the fidelities of all of them."""
    from ..lockstep import run_jobs_lockstep

    opts: UserOptions = cfg["opts"]
    h = cfg["horizon"]
    n = opts.num_qubits
    if opts.use_mps and not opts.vectorised_lbfgs:
        raise ValueError("random restarts (num_seeds > 1) of an MPS objective run under the vectorised L-BFGS (vectorised_lbfgs=True); "
                         "the per-restart scipy optimizers work on the state-vector objective 'sur_max'")
    tgt = generate_target(opts, h - 1)
    evol_time, target = tgt.evol_time, tgt.t1_gt
    fid_thr, fid_t1_vs_gt = _calc_fidelity_threshold(tgt, opts.fidelity_thr)
    circ = trotter_ansatz(n, _initial_layers(opts, h - 1), opts.second_order_trotter)
    trotter_thetas = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=evol_time, delta=opts.delta)
    ini = opts.ini_state_index()

    def restart(s: int, c: Dict, workspace) -> Dict:
        thetas0 = trotter_thetas.copy()
        if s > 0:   # restart 0 is the plain Trotter initial point
            thetas0 += opts.theta_jitter * np.pi * (2.0 * c["rng"].random(thetas0.size) - 1.0)
        user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: ini, enable_optim_stats=False, verbose=0,
                    maxiter=opts.maxiter, workspace=workspace)
        scaler = GradientAmplifier(history=5, strong=False) if opts.enable_grad_scaling else None   # as _create_objective
        objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True, grad_scaler=scaler)
        objv.set_target(target)
        res = AqcOptimizer(optimizer_name="lbfgs", maxiter=opts.maxiter).optimize(
            objv, circ, thetas0, stopper=EarlyStopper(fidelity_thr=fid_thr),
            timeout=TimeoutChecker(time_limit=opts.time_limit))
        return {"restart": s, "fidelity": float(res["fidelity"]), "cost": float(res["cost"]),
                "num_iters": int(res["num_iters"]), "num_fun_ev": int(res["num_fun_ev"]), "thetas": res["thetas"]}

    common = {"horizon": h, "evol_time": evol_time, "num_layers": circ.num_layers, "num_thetas": circ.num_thetas,
              "fid_t1_vs_gt": fid_t1_vs_gt, "fidelity_thr": fid_thr}
    if opts.vectorised_lbfgs or opts.device_lbfgs:   # all restarts as lanes of one batched objective, one optimizer for all of them
        from ..batched_optimizer import BatchedSurrogateObjective, batched_lbfgs

        starts = np.tile(trotter_thetas, (opts.num_seeds, 1))
        for s in range(1, opts.num_seeds):
            rng = np.random.default_rng(opts.seed + 1000 * h + 7 * (s + 1))
            starts[s] += opts.theta_jitter * np.pi * (2.0 * rng.random(trotter_thetas.size) - 1.0)
        target_dev = bo = None
        if opts.use_mps and n > _DENSE_MAX_QUBITS:   # beyond dense reach: the restarts are lockstep lanes of the native MPS engine (bonds <= 32)
            from ..batched_optimizer import BatchedMpsSurrogateObjective
            from ..mps_engine import DeviceMPS

            target_dev = DeviceMPS.from_qiskit(target, device=opts.device, trunc_thr=float(opts.trunc_thr), assume_canonical=True)
            try:
                bo = BatchedMpsSurrogateObjective(circ, target_dev, lanes=opts.num_seeds, base_index=ini, trunc_thr=float(opts.trunc_thr), device=opts.device)
                res = batched_lbfgs(bo.value_and_grad, starts, maxiter=opts.maxiter, stop=lambda f, x: bo.fidelity >= fid_thr)
            except RuntimeError as err:
                if "lockstep lanes" not in str(err):
                    raise
                if bo is not None:
                    bo.close()
                bo = None            # a bond beyond the lanes' 32 (long horizons: the untruncated target alone can exceed it)
            finally:
                target_dev.close()
            if bo is None:           # ... the restarts one after the other on the objective's single-lane route, as the reference runs a job
                recs = []
                for s_no in range(opts.num_seeds):
                    objv = _create_objective(opts=opts, circ=circ, target=target, layer_range=(0, circ.num_layers))
                    r = AqcOptimizer(optimizer_name="lbfgs", maxiter=int(opts.maxiter), verbose=opts.verbose).optimize(
                        objv, circ, starts[s_no], stopper=EarlyStopper(fidelity_thr=fid_thr), timeout=TimeoutChecker(time_limit=opts.time_limit))
                    recs.append(r)
                best = int(np.argmax([r["fidelity"] for r in recs]))
                return dict(common, fidelity=float(recs[best]["fidelity"]), cost=float(recs[best]["cost"]), thetas=recs[best]["thetas"].copy(),
                            best_restart=best, fidelities=[float(r["fidelity"]) for r in recs], num_fun_ev=int(sum(r["num_fun_ev"] for r in recs)),
                            route="single-lane engine, one restart after the other")
        else:
            dense = (target.dense_state if isinstance(target, DenseBackedMPS) else mps_to_vector(target)) if opts.use_mps else target
            bo = BatchedSurrogateObjective(circ, np.tile(dense, (opts.num_seeds, 1)), base_index=ini, device=opts.device)
            if opts.device_lbfgs:
                res = bo.minimize_on_device(starts, maxiter=opts.maxiter, fidelity_thr=fid_thr)
            else:
                res = batched_lbfgs(bo.value_and_grad, starts, maxiter=opts.maxiter, stop=lambda f, x: bo.fidelity >= fid_thr)
        fids = bo.fidelity.copy()
        evals = bo.num_evals
        bo.close()
        best = int(np.argmax(fids))
        return dict(common, fidelity=float(fids[best]), cost=float(res["fun"][best]), thetas=res["x"][best].copy(), best_restart=best,
                    fidelities=[float(v) for v in fids], num_fun_ev=int(evals))
    recs = run_jobs_lockstep(circ, [{} for _ in range(opts.num_seeds)], opts.seed + 1000 * h, restart,
                             nlanes=min(64, opts.num_seeds), device=opts.device)
    ok = [r for r in recs if r["status"] == "ok"]
    best = max(ok, key=lambda r: r["fidelity"])
    return dict(common, fidelity=best["fidelity"], cost=best["cost"], thetas=best["thetas"], best_restart=best["restart"],
                fidelities=[r["fidelity"] for r in ok], num_fun_ev=int(sum(r["num_fun_ev"] for r in ok)))


def run_simulation(opts: Optional[UserOptions] = None) -> List[Dict]:
    """One optimisation per time horizon; returns the list of result records (run_simulation,
    time_evol_best_init.py:337-395: the reference walks the horizons in a loop -- they are independent -- and pickles the
    list; here they are jobs of ``run_jobs``, sharded over the ranks, and the list is returned)."""
    opts = opts or UserOptions()
    configs = [{"opts": opts, "horizon": h} for h in range(1, len(opts.trotter_steps) + 1)]
    job = _seeded_horizon_job if opts.num_seeds > 1 else _horizon_job
    return run_jobs(configs, opts.seed, job, tolerate_failure=False)

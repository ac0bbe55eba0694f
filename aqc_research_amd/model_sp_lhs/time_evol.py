"""
ASP time-evolution driver on top of the HIP objective: per horizon build the
TrotterAnsatz, start from the Trotter angles, minimise the surrogate objective with L-BFGS and report
fidelities -- the loop of time_evol_best_init.py:143-334 without its pickling / plotting / Qiskit parts.
Horizons are independent jobs (time_evol_best_init.py:232-233) and go through ``run_jobs``.
"""
from time import perf_counter
from typing import Dict, List, Optional

import numpy as np

from ..job_executor import run_jobs
from ..optimizer import AqcOptimizer, EarlyStopper, TimeoutChecker
from .objective_lhs_sur_max import SpSurrogateObjectiveMax
from .trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz, trotter_state


class UserOptions:
    """The tunables of the driver that concern the path (user_options.py:25-129 defaults)."""

    def __init__(self, **kw):
        self.num_qubits = 12
        self.second_order_trotter = True
        self.delta = 1.0
        self.evol_time_step = 1.2          # evolution time per horizon
        self.num_horizons = 6
        self.num_layers_inc = 2            # ansatz layers added per horizon
        self.trotter_steps_per_horizon = 6  # steps of the target Trotter circuit per horizon
        self.maxiter = 40
        self.fidelity_thr = 0.9999
        self.time_limit = -1
        self.seed = 1234
        self.device = None                  # None: this rank's GPU (LOCAL_RANK under a one-process-per-GPU launcher)
        self.num_seeds = 1                 # random restarts per horizon (lockstep lanes of one workspace)
        self.theta_jitter = 0.1            # restart s > 0 starts from Trotter angles + jitter * pi * U(-1, 1)
        self.vectorised_lbfgs = False      # restarts driven by ONE vectorised L-BFGS (batched_optimizer.py) instead of scipy per lane
        self.device_lbfgs = False          # ... and that L-BFGS resident on the device (aqc_ws_lbfgs), thetas never leave HBM
        self.__dict__.update(kw)


def _horizon_job(job_index: int, cfg: Dict) -> Dict:
    opts: UserOptions = cfg["opts"]
    h = cfg["horizon"]                      # 1-based
    n = opts.num_qubits
    evol_time = opts.evol_time_step * h
    tic = perf_counter()
    target = trotter_state(n, evol_time=evol_time, num_steps=opts.trotter_steps_per_horizon * h, delta=opts.delta,
                           second_order=opts.second_order_trotter)
    t_target = perf_counter() - tic
    circ = trotter_ansatz(n, opts.num_layers_inc * h, opts.second_order_trotter)
    thetas0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=evol_time, delta=opts.delta)
    neel = neel_state_index(n)
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=True, verbose=0,
                maxiter=opts.maxiter, device=opts.device)
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(target)
    fid0 = 1.0 - objv.objective(thetas0) if True else 0.0   # weight = 1, max_no = 0 at the start => 1 - fobj = |h0|^2
    res = AqcOptimizer(optimizer_name="lbfgs", maxiter=opts.maxiter).optimize(
        objv, circ, thetas0, stopper=EarlyStopper(fidelity_thr=opts.fidelity_thr),
        timeout=TimeoutChecker(time_limit=opts.time_limit))
    return {
        "horizon": h, "evol_time": evol_time, "num_layers": circ.num_layers, "num_thetas": circ.num_thetas,
        "fidelity_trotter_init": float(fid0), "fidelity": float(res["fidelity"]), "cost": float(res["cost"]),
        "num_iters": int(res["num_iters"]), "num_fun_ev": int(res["num_fun_ev"]), "thetas": res["thetas"],
        "blocks": res["blocks"], "target_time": t_target,
    }


def _seeded_horizon_job(job_index: int, cfg: Dict) -> Dict:
    """All restarts of one horizon: they share the ansatz, so they run as lanes of one batched workspace
    (lockstep.py); the record of the best restart is returned with the fidelities of all of them."""
    from ..lockstep import run_jobs_lockstep

    opts: UserOptions = cfg["opts"]
    h = cfg["horizon"]
    n = opts.num_qubits
    evol_time = opts.evol_time_step * h
    target = trotter_state(n, evol_time=evol_time, num_steps=opts.trotter_steps_per_horizon * h, delta=opts.delta,
                           second_order=opts.second_order_trotter)
    circ = trotter_ansatz(n, opts.num_layers_inc * h, opts.second_order_trotter)
    trotter_thetas = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=evol_time, delta=opts.delta)
    neel = neel_state_index(n)

    def restart(s: int, c: Dict, workspace) -> Dict:
        thetas0 = trotter_thetas.copy()
        if s > 0:   # restart 0 is the plain Trotter initial point
            thetas0 += opts.theta_jitter * np.pi * (2.0 * c["rng"].random(thetas0.size) - 1.0)
        user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0,
                    maxiter=opts.maxiter, workspace=workspace)
        objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
        objv.set_target(target)
        res = AqcOptimizer(optimizer_name="lbfgs", maxiter=opts.maxiter).optimize(
            objv, circ, thetas0, stopper=EarlyStopper(fidelity_thr=opts.fidelity_thr),
            timeout=TimeoutChecker(time_limit=opts.time_limit))
        return {"restart": s, "fidelity": float(res["fidelity"]), "cost": float(res["cost"]),
                "num_iters": int(res["num_iters"]), "num_fun_ev": int(res["num_fun_ev"]), "thetas": res["thetas"]}

    if opts.vectorised_lbfgs or opts.device_lbfgs:   # all restarts as lanes of one batched objective, one optimizer for all of them
        from ..batched_optimizer import BatchedSurrogateObjective, batched_lbfgs

        starts = np.tile(trotter_thetas, (opts.num_seeds, 1))
        for s in range(1, opts.num_seeds):
            rng = np.random.default_rng(opts.seed + 1000 * h + 7 * (s + 1))
            starts[s] += opts.theta_jitter * np.pi * (2.0 * rng.random(trotter_thetas.size) - 1.0)
        bo = BatchedSurrogateObjective(circ, np.tile(target, (opts.num_seeds, 1)), base_index=neel, device=opts.device)
        if opts.device_lbfgs:
            res = bo.minimize_on_device(starts, maxiter=opts.maxiter, fidelity_thr=opts.fidelity_thr)
        else:
            res = batched_lbfgs(bo.value_and_grad, starts, maxiter=opts.maxiter, stop=lambda f, x: bo.fidelity >= opts.fidelity_thr)
        fids = bo.fidelity.copy()
        evals = bo.num_evals
        bo.close()
        best = int(np.argmax(fids))
        return {"horizon": h, "evol_time": evol_time, "num_layers": circ.num_layers, "num_thetas": circ.num_thetas,
                "fidelity": float(fids[best]), "cost": float(res["fun"][best]), "thetas": res["x"][best].copy(), "best_restart": best,
                "fidelities": [float(v) for v in fids], "num_fun_ev": int(evals)}
    recs = run_jobs_lockstep(circ, [{} for _ in range(opts.num_seeds)], opts.seed + 1000 * h, restart,
                             nlanes=min(64, opts.num_seeds), device=opts.device)
    ok = [r for r in recs if r["status"] == "ok"]
    best = max(ok, key=lambda r: r["fidelity"])
    return {"horizon": h, "evol_time": evol_time, "num_layers": circ.num_layers, "num_thetas": circ.num_thetas,
            "fidelity": best["fidelity"], "cost": best["cost"], "thetas": best["thetas"], "best_restart": best["restart"],
            "fidelities": [r["fidelity"] for r in ok], "num_fun_ev": int(sum(r["num_fun_ev"] for r in ok))}


def run_simulation(opts: Optional[UserOptions] = None) -> List[Dict]:
    """One optimisation per time horizon; returns the list of result records (run_simulation,
    time_evol_best_init.py:337-395)."""
    opts = opts or UserOptions()
    configs = [{"opts": opts, "horizon": h} for h in range(1, opts.num_horizons + 1)]
    job = _seeded_horizon_job if opts.num_seeds > 1 else _horizon_job
    return run_jobs(configs, opts.seed, job, tolerate_failure=False)

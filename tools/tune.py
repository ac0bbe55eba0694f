#!/usr/bin/env python3
"""Interleaved A/B timing of workspace configurations (threads, tile bits, low bits) in ONE process."""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit, TrotterAnsatz  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure, make_trotter_like_circuit  # noqa: E402
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, K_APPLY, K_SWEEP, HipContext, Workspace  # noqa: E402


def run(n=16, L=40, B=64, configs=None, steps=20, rounds=3, trotter_layers=0, grad_args=(None, True)):
    if trotter_layers:
        circ = TrotterAnsatz(n, make_trotter_like_circuit(n, trotter_layers), second_order=True)
    else:
        circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
    ctx = HipContext.of(circ)
    rng = np.random.default_rng(0)
    T = circ.num_thetas
    wss = []
    for cfg in configs:
        for k, v in cfg.get("env", {}).items():
            os.environ[k] = str(v)
        ws = Workspace(ctx, batch=B, tile_bits_apply=cfg.get("ka", 0), tile_bits_sweep=cfg.get("ks", 0))
        tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
        ws.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True))
        ws.set_basis(BUF_X, 0)
        ws.gather_setup(np.arange(n + 1))
        ws.theta_bank(np.pi * (2 * rng.random((4, B, T)) - 1))
        wss.append(ws)
    res = [[] for _ in configs]
    for r in range(rounds):
        for ci, ws in enumerate(wss):
            for i in range(3):
                ws.use_theta_set(i % 4); ws.apply(True, BUF_Y, BUF_Z); ws.gather_launch(BUF_Z); ws.grad(*grad_args)
            ws.sync()
            ws.timer_start()
            for i in range(steps):
                ws.use_theta_set(i % 4); ws.apply(True, BUF_Y, BUF_Z); ws.gather_launch(BUF_Z); ws.grad(*grad_args)
            res[ci].append(ws.timer_stop() / steps)
    for ci, (cfg, ws) in enumerate(zip(configs, wss)):
        ws.profile(True)
        for i in range(5):
            ws.use_theta_set(i % 4); ws.apply(True, BUF_Y, BUF_Z); ws.gather_launch(BUF_Z); ws.grad(*grad_args)
        ws.sync()
        a, s = ws.profile_get(K_APPLY), ws.profile_get(K_SWEEP)
        ws.profile(False)
        ms = min(res[ci])
        print(f"{cfg}: step {ms:.4f} ms (med {np.median(res[ci]):.4f}) -> {B / ms * 1e3:,.0f} evals/s | "
              f"apply {a[1] / 5:.3f} ms/{a[0] // 5} launches, sweep {s[1] / 5:.3f} ms/{s[0] // 5} launches | "
              f"plan inv{ws.plan_info(0)} sweep{ws.plan_info(1)}", flush=True)
        ws.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "a"
    if which == "a":
        cfgs = []
        for thr, ks, low in itertools.product((256, 512), (10, 11, 12), (2, 3)):
            cfgs.append({"env": {"AQC_THREADS": thr, "AQC_LOW_BITS": low}, "ks": ks, "ka": min(ks + 1, 13)})
        run(configs=cfgs)
    elif which == "nodots":
        cfgs = [{"env": {"AQC_THREADS": 256, "AQC_LOW_BITS": 3}, "ks": 12, "ka": 13}]
        print("with dots"); run(configs=cfgs)
        print("dots only for block 0"); run(configs=cfgs, grad_args=((0, 1), False))
    elif which == "k":
        cfgs = [{"env": {"AQC_LOW_BITS": low}, "ks": ks, "ka": ka} for low in (2, 3) for ks, ka in ((10, 11), (11, 12), (11, 13), (12, 13), (12, 12))]
        run(configs=cfgs)
    elif which == "n12":
        cfgs = []
        for v2, ks in ((0, 6), (0, 8), (0, 10), (0, 12), (1, 8), (1, 10), (1, 12)):
            cfgs.append({"env": {"AQC_KERNEL_V2": v2, "AQC_THREADS": 0}, "ks": ks, "ka": ks})
        run(n=12, B=int(sys.argv[2]) if len(sys.argv) > 2 else 1, configs=cfgs, steps=50, trotter_layers=2)
    elif which == "r3":
        run(configs=[{"env": {"AQC_SWEEP_REG_BITS": r}, "ks": 12, "ka": 13} for r in (4, 3)])
    elif which == "one":
        run(configs=[{"env": {"AQC_LOW_BITS": 3}, "ks": 12, "ka": 13}])
    elif which == "skip":
        cfgs = [{"env": {"AQC_THREADS": 256, "AQC_LOW_BITS": 3}, "ks": 12, "ka": 13}]
        for dbg in (0, 3, 4):
            os.environ["AQC_DEBUG_SKIP"] = str(dbg)
            print("AQC_DEBUG_SKIP =", dbg); run(configs=cfgs)
    elif which == "b1":
        cfgs = []
        for v2, thr, ks in itertools.product((0, 1), (64, 128, 256), (7, 8, 9, 10, 11, 12)):
            if v2 and thr != 256:
                continue
            cfgs.append({"env": {"AQC_KERNEL_V2": v2, "AQC_THREADS": thr, "AQC_LOW_BITS": 2}, "ks": ks, "ka": ks})
        run(B=int(sys.argv[2]) if len(sys.argv) > 2 else 1, configs=cfgs, steps=50)
    elif which == "b1v2":   # latency regime: per-group kernels vs register-blocked kernels on small tiles
        cfgs = [{"env": {"AQC_KERNEL_V2": 0, "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": 4}, "ks": 0, "ka": 0}]
        for r, ks, ka in ((3, 9, 10), (3, 10, 10), (3, 10, 11), (4, 10, 10), (4, 11, 11), (3, 9, 9), (3, 8, 8)):
            cfgs.append({"env": {"AQC_KERNEL_V2": 1, "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": r}, "ks": ks, "ka": ka})
        print("n=16 L=40 B=1"); run(B=1, configs=cfgs, steps=50)
        print("n=12 trotter2 B=1"); run(n=12, B=1, configs=cfgs, steps=50, trotter_layers=2)
    elif which == "mid":    # family / tile choice for middling batch sizes
        for B in (2, 4, 8, 16, 32):
            cfgs = [{"env": {"AQC_KERNEL_V2": "", "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": 4}, "ks": 0, "ka": 0}]   # automatic choice
            for v2, ks, ka in ((0, 8, 8), (0, 9, 9), (0, 10, 10), (0, 11, 11), (1, 10, 11), (1, 11, 12), (1, 12, 13)):
                cfgs.append({"env": {"AQC_KERNEL_V2": v2, "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": 4}, "ks": ks, "ka": ka})
            print(f"n=16 L=40 B={B}"); run(B=B, configs=cfgs, steps=30)
    elif which == "b1k":
        cfgs = [{"env": {"AQC_KERNEL_V2": "", "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": 4}, "ks": 0, "ka": 0}]
        for ks, ka in ((8, 8), (9, 9), (10, 10), (9, 10), (10, 9), (8, 9), (11, 11)):
            cfgs.append({"env": {"AQC_KERNEL_V2": 0, "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": 4}, "ks": ks, "ka": ka})
        print("n=16 L=40 B=1"); run(B=1, configs=cfgs, steps=50)
        print("n=12 trotter2 B=1"); run(n=12, B=1, configs=cfgs, steps=50, trotter_layers=2)
        print("n=20 L=40 B=1"); run(n=20, B=1, configs=cfgs[:1] + [{"env": {"AQC_KERNEL_V2": v2, "AQC_THREADS": 0, "AQC_SWEEP_REG_BITS": 4}, "ks": ks, "ka": ka}
                                                                   for v2, ks, ka in ((0, 10, 10), (0, 11, 11), (0, 12, 12), (1, 12, 13))], steps=20)

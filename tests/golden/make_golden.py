#!/usr/bin/env python3
"""
Generates the golden input/output vectors in this directory by importing and
running the *reference* (qiskit-community/aqc-research v0.1.0, mounted
read-only at /root/reference).  Runs only in the build container; the GPU box
never sees the reference, only the ``.npz`` files committed here.

Import accommodations (no reference code is modified or copied):
  * ``np.cfloat`` was removed in NumPy 2; the alias is restored before import.
  * qiskit / qiskit-aer are not installed.  Their names are registered as empty
    placeholder modules so that reference modules which import them *at module
    top level* can be loaded.  No placeholder is ever executed: every function
    exercised below is pure reference NumPy arithmetic.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""

import os
import sys
import types

import numpy as np

REF = os.environ.get("AQC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
np.cfloat = np.complex128  # NumPy >= 2


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Missing:  # any attempt to *use* a qiskit object fails loudly
    def __init__(self, *a, **k):
        raise RuntimeError("qiskit is not available in this container")


_placeholder("qiskit", QuantumCircuit=_Missing)
_placeholder("qiskit.quantum_info", Operator=_Missing, Statevector=_Missing)
_placeholder("qiskit.circuit")
_placeholder("qiskit.circuit.library", QFT=_Missing)
_placeholder("qiskit_aer", AerSimulator=_Missing)
_placeholder("qiskit.algorithms")
_placeholder("qiskit.algorithms.optimizers", L_BFGS_B=_Missing, ADAM=_Missing, COBYLA=_Missing, BOBYQA=_Missing)
_placeholder("qiskit.algorithms.optimizers.optimizer", OptimizerResult=_Missing)
sys.modules["qiskit"].quantum_info = sys.modules["qiskit.quantum_info"]

sys.path.insert(0, REF)
import aqc_research.core_operations as cop  # noqa: E402
import aqc_research.core_op_matrix as com  # noqa: E402
import aqc_research.mps_operations as mpsop  # noqa: E402
from aqc_research.circuit_structures import create_ansatz_structure, make_trotter_like_circuit  # noqa: E402
from aqc_research.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx  # noqa: E402
from aqc_research.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax  # noqa: E402
from aqc_research.parametric_circuit import ParametricCircuit, TrotterAnsatz  # noqa: E402

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.aqc_oracle import random_mps  # noqa: E402  (input generator only)


def rand_blocks(n, depth, rng):
    b = np.zeros((2, depth), dtype=np.int64)
    for i in range(depth):
        b[:, i] = rng.permutation(n)[:2]
    return b


def rand_vec(size, rng):
    v = rng.standard_normal(size) + 1j * rng.standard_normal(size)
    return v / np.linalg.norm(v)


def circuits():
    """(name, circuit) for every golden case."""
    rng = np.random.default_rng(0x0696969)
    out = []
    for ent in ("cx", "cz", "cp"):
        for n in (2, 3, 5, 6):
            depth = {2: 3, 3: 5, 5: 9, 6: 11}[n]
            out.append((f"{ent}_rand_n{n}", ParametricCircuit(n, ent, rand_blocks(n, depth, rng))))
            out.append((f"{ent}_spin_n{n}", ParametricCircuit(n, ent, create_ansatz_structure(n, "spin", "full", depth))))
    for n in (2, 3, 5, 6):
        for order2 in (False, True):
            nl = 2 if n < 6 else 1
            blocks = make_trotter_like_circuit(n, nl)
            out.append((f"trot{2 if order2 else 1}_n{n}", TrotterAnsatz(n, blocks, second_order=order2)))
    return out


def describe(circ):
    trot = isinstance(circ, TrotterAnsatz)
    return dict(
        n=np.int64(circ.num_qubits),
        ent=np.array(circ.entangler),
        blocks=circ.blocks.astype(np.int64),
        trotter=np.bool_(trot),
        second_order=np.bool_(trot and circ.is_second_order),
    )


def gen_state_vector():
    rng = np.random.default_rng(1234)
    data = {}
    names = []
    for name, circ in circuits():
        n, dim = circ.num_qubits, circ.dimension
        th = np.pi * (2 * rng.random(circ.num_thetas) - 1)
        x, y = rand_vec(dim, rng), rand_vec(dim, rng)
        ws = np.zeros((3, dim), dtype=np.complex128)
        vx = cop.v_mul_vec(circ, th, x, np.zeros(dim, np.complex128), ws[:2]).copy()
        vhy = cop.v_dagger_mul_vec(circ, th, y, np.zeros(dim, np.complex128), ws[:2]).copy()
        g_full = cop.grad_of_dot_product(circ, th, x, vhy, ws).copy()
        nb = circ.num_blocks
        if isinstance(circ, TrotterAnsatz):
            bpl = circ.bpl
            br = (bpl, 2 * bpl) if nb >= 2 * bpl else (0, bpl)
        else:
            br = (1, max(2, nb - 1))
        g_part = cop.grad_of_dot_product(circ, th, x, vhy, ws, block_range=br, front_layer=False).copy()
        for k, v in describe(circ).items():
            data[f"{name}/{k}"] = v
        data[f"{name}/thetas"] = th
        data[f"{name}/x"] = x
        data[f"{name}/y"] = y
        data[f"{name}/v_x"] = vx
        data[f"{name}/vh_y"] = vhy
        data[f"{name}/grad_full"] = g_full
        data[f"{name}/block_range"] = np.asarray(br, np.int64)
        data[f"{name}/grad_part"] = g_part
        names.append(name)
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "state_vector.npz"), **data)


def gen_gate2x2():
    """Degenerate 2x2 cases of test/test_core_operations.py:162-178 + random."""
    rng = np.random.default_rng(77)
    mats = [
        [[0, 0], [0, 0]], [[1, 0], [0, 0]], [[0, 1], [0, 0]], [[0, 0], [1, 0]],
        [[0, 0], [0, 1]], [[1, 1], [0, 0]], [[0, 0], [1, 1]], [[1, 0], [1, 0]],
    ]
    mats = [np.asarray(m, np.complex128) * (0.3 + 0.7j) for m in mats]
    mats.append(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))
    data, n = {}, 4
    vec = rand_vec(2**n, rng)
    data["vec"], data["n"] = vec, np.int64(n)
    data["gates"] = np.stack(mats)
    outs = np.zeros((len(mats), n, 2**n), np.complex128)
    for i, g in enumerate(mats):
        for q in range(n):
            out = np.zeros_like(vec)
            cop.gate2x2_mul_vec(n, cop.bit2bit_transform(n, q), g, vec.copy(), out, False)
            outs[i, q] = out
    data["outs"] = outs  # outs[i, q] = (gate i on qubit q) @ vec
    np.savez_compressed(os.path.join(HERE, "gate2x2.npz"), **data)


def gen_matrix():
    rng = np.random.default_rng(4321)
    data, names = {}, []
    for name, circ in circuits():
        if isinstance(circ, TrotterAnsatz):
            continue
        n, dim = circ.num_qubits, circ.dimension
        if n == 6:
            continue
        th = np.pi * (2 * rng.random(circ.num_thetas) - 1)
        for k in sorted({1, 3, dim}):
            if k > dim:
                continue
            key = f"{name}_k{k}"
            x = rng.standard_normal((dim, k)) + 1j * rng.standard_normal((dim, k))
            y = rng.standard_normal((dim, k)) + 1j * rng.standard_normal((dim, k))
            ws = np.zeros((dim, k), np.complex128)
            vx = com.v_mul_mat(circ, th, x.copy(), ws).copy()
            vhy = com.v_dagger_mul_mat(circ, th, y.copy(), ws).copy()
            g = com.grad_of_matrix_dot_product(circ, th, x.copy(), vhy.copy(), ws).copy()
            for kk, v in describe(circ).items():
                data[f"{key}/{kk}"] = v
            data[f"{key}/thetas"] = th
            data[f"{key}/x"], data[f"{key}/y"] = x, y
            data[f"{key}/v_x"], data[f"{key}/vh_y"], data[f"{key}/grad"] = vx, vhy, g
            names.append(key)
        if circ.entangler != "cp":
            # coordinate descent on a random unitary target
            key = f"{name}_cd"
            q, _ = np.linalg.qr(rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim)))
            th_io = th.copy()
            ws3 = np.zeros((3, dim, dim), np.complex128)
            f1 = com.coord_descent_single_sweep(circ, th_io, q, ws3)
            th1 = th_io.copy()
            f2 = com.coord_descent_single_sweep(circ, th_io, q, ws3)
            for kk, v in describe(circ).items():
                data[f"{key}/{kk}"] = v
            data[f"{key}/thetas"] = th
            data[f"{key}/target"] = q
            data[f"{key}/thetas_1"], data[f"{key}/fobj_1"] = th1, np.float64(f1)
            data[f"{key}/thetas_2"], data[f"{key}/fobj_2"] = th_io.copy(), np.float64(f2)
            names.append(key)
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "matrix.npz"), **data)


def gen_objectives():
    rng = np.random.default_rng(2468)
    data, names = {}, []
    # --- surrogate "sur_max" objective: 3-call sequence ----------------------
    for name, circ in circuits():
        n = circ.num_qubits
        if n not in (3, 5) or circ.entangler == "cz":
            continue
        user = dict(num_qubits=n, max_flips=1, enable_optim_stats=True, verbose=0, maxiter=10, num_simulations=2)
        target = rand_vec(circ.dimension, rng)
        # make flip states matter: mix some weight onto one-hot states
        target[1 << (n - 1)] += 0.8
        target /= np.linalg.norm(target)
        th = 0.3 * np.pi * (2 * rng.random(circ.num_thetas) - 1)
        dth = 0.05 * (2 * rng.random(circ.num_thetas) - 1)
        obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=False)
        obj.set_target(target)
        key = f"surmax_{name}"
        seq = []
        f0 = obj.objective(th)
        seq.append((f0, obj._max_no, obj._weight))
        g0 = obj.gradient(th)
        seq.append((obj._fobj, obj._max_no, obj._weight))
        hs0 = obj._hs.copy()
        f1 = obj.objective(th + dth)
        seq.append((f1, obj._max_no, obj._weight))
        g1 = obj.gradient(th + dth)  # forced re-evaluation not needed (same thetas)
        seq.append((obj._fobj, obj._max_no, obj._weight))
        g2 = obj.gradient(th)  # gradient before objective at changed thetas
        seq.append((obj._fobj, obj._max_no, obj._weight))
        for kk, v in describe(circ).items():
            data[f"{key}/{kk}"] = v
        data[f"{key}/target"], data[f"{key}/thetas"], data[f"{key}/dthetas"] = target, th, dth
        data[f"{key}/f0"], data[f"{key}/f1"] = np.float64(f0), np.float64(f1)
        data[f"{key}/g0"], data[f"{key}/g1"], data[f"{key}/g2"] = g0, g1, g2
        data[f"{key}/hs0"], data[f"{key}/hs_last"] = hs0, obj._hs.copy()
        data[f"{key}/seq"] = np.asarray(seq, float)
        data[f"{key}/stats_fobj"] = obj.statistics["fobj"]
        names.append(key)
    # --- full-range sketching objective (sk_core) ------------------------------
    for name, circ in circuits():
        if isinstance(circ, TrotterAnsatz) or circ.num_qubits not in (2, 3, 5):
            continue
        dim = circ.dimension
        u, _ = np.linalg.qr(rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim)))
        u = u / np.linalg.det(u) ** (1.0 / dim)
        th = np.pi * (2 * rng.random(circ.num_thetas) - 1)
        objv = SketchingObjectiveEx(circ, FullRangeSketchingVectors(u))
        f = objv.objective(th)
        g = objv.gradient(th).copy()
        key = f"sketch_{name}"
        for kk, v in describe(circ).items():
            data[f"{key}/{kk}"] = v
        data[f"{key}/target"], data[f"{key}/thetas"] = u, th
        data[f"{key}/fobj"], data[f"{key}/grad"] = np.float64(f), g
        names.append(key)
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "objectives.npz"), **data)


def gen_mps():
    """mps_dot / mps_to_vector on synthetic (gammas, lambdas) tuples."""
    rng = np.random.default_rng(9753)
    data, names = {}, []
    for n, chi in ((2, 2), (3, 2), (4, 4), (5, 3), (6, 4)):
        a, b = random_mps(n, chi, rng), random_mps(n, chi, rng)
        key = f"mps_n{n}_chi{chi}"
        for tag, m in (("a", a), ("b", b)):
            gam, lam = m
            for q in range(n):
                data[f"{key}/{tag}_g0_{q}"] = gam[q][0]
                data[f"{key}/{tag}_g1_{q}"] = gam[q][1]
            for q in range(n - 1):
                data[f"{key}/{tag}_lam_{q}"] = lam[q]
        data[f"{key}/n"] = np.int64(n)
        assert mpsop.check_mps(a) and mpsop.check_mps(b)
        data[f"{key}/vec_a"] = mpsop.mps_to_vector(a)
        data[f"{key}/vec_b"] = mpsop.mps_to_vector(b)
        data[f"{key}/dot_ab"] = np.complex128(mpsop.mps_dot(a, b))
        data[f"{key}/dot_aa"] = np.complex128(mpsop.mps_dot(a, a))
        names.append(key)
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "mps.npz"), **data)


def gen_primitives():
    """Gate-level building blocks on random data.  ``pos`` arguments of the
    state-vector functions are the reference's own (big-endian) positions, stored as given."""
    rng = np.random.default_rng(4242)
    data = {}
    n = 5
    N = 2**n
    vec, w, z = rand_vec(N, rng), rand_vec(N, rng), rand_vec(N, rng)
    data["n"], data["vec"], data["w"], data["z"] = np.int64(n), vec, w, z
    angles = np.array([0.37, -1.9, 2.6])
    data["angles"] = angles
    pairs = np.array([[0, 1], [3, 1], [4, 0], [2, 4]])
    data["pairs"] = pairs
    tmp = np.zeros(N, np.complex128)
    for name, fn in (("rx", cop.rx_mul_vec), ("ry", cop.ry_mul_vec), ("rz", cop.rz_mul_vec)):
        data[f"sv/{name}"] = np.stack([[fn(n, pos, float(a), vec.copy(), tmp.copy()) for pos in range(n)] for a in angles])
    for name, fn in (("dot_x", cop.dot_x), ("dot_y", cop.dot_y), ("dot_z", cop.dot_z)):
        data[f"sv/{name}"] = np.array([fn(n, pos, w.copy(), z.copy(), tmp.copy()) for pos in range(n)])
    for name, fn in (("cx", cop.cx_mul_vec), ("cz", cop.cz_mul_vec), ("cp", cop.cp_mul_vec)):
        data[f"sv/{name}"] = np.stack([fn(n, int(c), int(t), 0.83, vec.copy(), tmp.copy()) for c, t in pairs])
    data["sv/derv_cp"] = np.stack([cop.derv_cphase_mul_vec(n, int(c), int(t), 0.83, vec.copy(), tmp.copy()).copy() for c, t in pairs])
    data["sv/proj00"] = np.stack([cop.proj00_mul_vec(n, pos, vec.copy()) for pos in range(n)])
    data["sv/proj11"] = np.stack([cop.proj11_mul_vec(n, pos, vec.copy()) for pos in range(n)])
    mats = rng.standard_normal((3, 2, 2)) + 1j * rng.standard_normal((3, 2, 2))   # c_mat, t_mat, g_mat
    data["block_mats"] = mats
    ws2 = np.zeros((2, N), np.complex128)
    data["sv/block"] = np.stack([cop.block_mul_vec(n, int(c), int(t), mats[0], mats[1], mats[2], vec.copy(), ws2.copy(), False) for c, t in pairs])
    data["sv/block_dagger"] = np.stack([cop.block_mul_vec(n, int(c), int(t), mats[0], mats[1], mats[2], vec.copy(), ws2.copy(), True) for c, t in pairs])
    # matrices: (2^n x k), k = 3 and k = 2^n; qubit numbers are plain bit indices there
    for k in (3, N):
        m = rng.standard_normal((N, k)) + 1j * rng.standard_normal((N, k))
        wm = rng.standard_normal((N, k)) + 1j * rng.standard_normal((N, k))
        zm = rng.standard_normal((N, k)) + 1j * rng.standard_normal((N, k))
        data[f"mat{k}/m"], data[f"mat{k}/w"], data[f"mat{k}/z"] = m, wm, zm
        wsm = np.zeros((N, k), np.complex128)
        for name, fn in (("rx", com.rx_mul_mat), ("ry", com.ry_mul_mat), ("rz", com.rz_mul_mat)):
            data[f"mat{k}/{name}"] = np.stack([fn(0.37, q, m.copy(), wsm.copy()) for q in range(n)])
        data[f"mat{k}/gate2x2"] = np.stack([com.gate2x2_mul_mat(q, mats[0], m.copy(), wsm.copy()) for q in range(n)])
        for name, fn in (("cx", com.cx_mul_mat), ("cz", com.cz_mul_mat), ("cp", com.cp_mul_mat)):
            data[f"mat{k}/{name}"] = np.stack([fn(int(c), int(t), 0.83, m.copy(), wsm.copy()) for c, t in pairs])
        for name, fn in (("x_dot", com.x_dot_mat), ("y_dot", com.y_dot_mat), ("z_dot", com.z_dot_mat)):
            data[f"mat{k}/{name}"] = np.array([fn(q, wm.copy(), zm.copy(), wsm.copy()) for q in range(n)])
        data[f"mat{k}/derv_cphase"] = np.array([com.derv_cphase(int(c), int(t), wm.copy(), zm.copy(), wsm.copy()) for c, t in pairs])
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **data)


if __name__ == "__main__":
    if "--primitives-only" in sys.argv:   # added later: leaves the earlier fixture files byte-identical
        gen_primitives()
        sys.exit(0)
    gen_state_vector()
    gen_gate2x2()
    gen_matrix()
    gen_objectives()
    gen_mps()
    gen_primitives()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")

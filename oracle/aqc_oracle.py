"""
CPU oracle for the aqc-research fidelity/gradient hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain-NumPy restatement of the
reference algorithm (qiskit-community/aqc-research v0.1.0).  It exists so the
HIP path can be checked against an independent CPU implementation on a box
where the reference itself is absent.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``aqc_research_amd`` never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
below against ``tests/golden/*.npz``, which were produced by importing and
running the reference itself (``tests/golden/make_golden.py``).  The MPS
gradient (``fast_dot_gradient``) is pinned to *dense state-vector semantics
under no truncation* (the level at which the reference's own tests pin it,
``test/test_mps_fast_dot_gradient.py:126-153``); the truncated-SVD arithmetic
of qiskit-aer (third party, version unpinned, not installed) is parity
unpinned.

Conventions (reference ``core_operations.py:34-43,671-708``):
  * qubit ``q`` is bit ``q`` of the amplitude index (Qiskit order) => the two
    amplitudes a 1-qubit gate couples are ``h = 2**q`` elements apart;
  * a (d, k) row-major matrix is treated as a flat array in which qubit ``q``
    has stride ``h = k * 2**q`` (``core_op_matrix.py:56``);
  * thetas[0:3n].reshape(n,3): front gate of qubit q = Rz(t0) Ry(t1) Rz(t2);
  * thetas[3n:].reshape(L,tpb): block = (Rz(t1)Ry(t0) (x) Rs(t3)Ry(t2)) CG,
    Rs = Rx for "cx", Rz for "cz"/"cp"; CG in {CX, CZ, CP(t4)};
  * Trotter ansatz: Rz(-pi/2) on ctrl before block i when i%3==0, Rz(+pi/2) on
    targ after block i when i%3==2; 2nd order appends 3*(n//2) virtual blocks
    re-using the leading half-layer's thetas.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

C128 = np.complex128


# ---------------------------------------------------------------------------
# Ansatz description (restates parametric_circuit.py:24-187,267-347 as data).
# ---------------------------------------------------------------------------


@dataclass(frozen=True)
class Ansatz:
    n: int
    entangler: str  # "cx" | "cz" | "cp"
    blocks: np.ndarray  # int (2, L): row 0 = control, row 1 = target
    trotter: bool = False
    second_order: bool = False

    @property
    def num_blocks(self) -> int:
        return int(self.blocks.shape[1])

    @property
    def tpb(self) -> int:
        return 5 if self.entangler == "cp" else 4

    @property
    def num_thetas(self) -> int:
        return 3 * self.n + self.tpb * self.num_blocks

    @property
    def tail_blocks(self) -> int:
        """Virtual trailing half-layer (parametric_circuit.py:328-333)."""
        return 3 * (self.n // 2) if (self.trotter and self.second_order) else 0

    @property
    def dim(self) -> int:
        return 1 << self.n


def as_ansatz(circ) -> Ansatz:
    """Accepts an ``Ansatz`` or any duck-typed circuit object exposing
    num_qubits / entangler / blocks (+ optional is_second_order)."""
    if isinstance(circ, Ansatz):
        return circ
    trotter = hasattr(circ, "is_second_order")
    return Ansatz(
        int(circ.num_qubits),
        str(circ.entangler),
        np.asarray(circ.blocks, dtype=np.int64),
        trotter,
        bool(circ.is_second_order) if trotter else False,
    )


# ---------------------------------------------------------------------------
# Primitive gates on a flat array; ``h`` is the element stride of the qubit.
# ---------------------------------------------------------------------------


def _halves(flat: np.ndarray, h: int) -> Tuple[np.ndarray, np.ndarray]:
    v = flat.reshape(-1, 2, h)
    return v[:, 0, :], v[:, 1, :]


def rz(flat: np.ndarray, h: int, angle: float) -> None:
    """core_operations.py:236-264 / core_op_matrix.py:100-127."""
    a0, a1 = _halves(flat, h)
    a0 *= np.exp(-0.5j * angle)
    a1 *= np.exp(+0.5j * angle)


def ry(flat: np.ndarray, h: int, angle: float) -> None:
    """core_operations.py:200-233: [[c,-s],[s,c]]."""
    c, s = np.cos(0.5 * angle), np.sin(0.5 * angle)
    a0, a1 = _halves(flat, h)
    t0 = c * a0 - s * a1
    a1[...] = s * a0 + c * a1
    a0[...] = t0


def rx(flat: np.ndarray, h: int, angle: float) -> None:
    """core_operations.py:164-197: [[c,-is],[-is,c]]."""
    c, s = np.cos(0.5 * angle), -1j * np.sin(0.5 * angle)
    a0, a1 = _halves(flat, h)
    t0 = c * a0 + s * a1
    a1[...] = s * a0 + c * a1
    a0[...] = t0


def gate2x2(flat: np.ndarray, h: int, g: np.ndarray) -> None:
    """Arbitrary 2x2 gate (core_operations.py:46-119, core_op_matrix.py:392-427)."""
    a0, a1 = _halves(flat, h)
    t0 = g[0, 0] * a0 + g[0, 1] * a1
    a1[...] = g[1, 0] * a0 + g[1, 1] * a1
    a0[...] = t0


def _quad(flat: np.ndarray, hc: int, ht: int):
    """View of the four (ctrl,targ) sub-arrays; returns getter f(bc, bt)."""
    hi, lo = (hc, ht) if hc > ht else (ht, hc)
    v = flat.reshape(-1, 2, hi // (2 * lo), 2, lo)
    if hc > ht:
        return lambda bc, bt: v[:, bc, :, bt, :]
    return lambda bc, bt: v[:, bt, :, bc, :]


def cx(flat: np.ndarray, hc: int, ht: int) -> None:
    """core_operations.py:422-465."""
    q = _quad(flat, hc, ht)
    t = q(1, 0).copy()
    q(1, 0)[...] = q(1, 1)
    q(1, 1)[...] = t


def cz(flat: np.ndarray, hc: int, ht: int) -> None:
    """core_operations.py:468-511."""
    q = _quad(flat, hc, ht)
    np.negative(q(1, 1), out=q(1, 1))


def cp(flat: np.ndarray, hc: int, ht: int, angle: float) -> None:
    """core_operations.py:514-558: diag(1,1,1,e^{i angle})."""
    q = _quad(flat, hc, ht)
    q(1, 1)[...] *= np.exp(1j * angle)


def _entangle(flat, hc, ht, ent: str, angle: float) -> None:
    if ent == "cx":
        cx(flat, hc, ht)
    elif ent == "cz":
        cz(flat, hc, ht)
    elif ent == "cp":
        cp(flat, hc, ht, angle)
    else:
        raise ValueError(f"unknown entangler {ent!r}")


def dot_x(w: np.ndarray, z: np.ndarray, h: int) -> complex:
    """0.5j <X w|z> (core_operations.py:267-293)."""
    w0, w1 = _halves(w, h)
    z0, z1 = _halves(z, h)
    return 0.5j * (np.vdot(w1, z0) + np.vdot(w0, z1))


def dot_y(w: np.ndarray, z: np.ndarray, h: int) -> complex:
    """0.5j <Y w|z> = 0.5 (<w0|z1> - <w1|z0>) (core_operations.py:296-322)."""
    w0, w1 = _halves(w, h)
    z0, z1 = _halves(z, h)
    return 0.5 * (np.vdot(w0, z1) - np.vdot(w1, z0))


def dot_z(w: np.ndarray, z: np.ndarray, h: int) -> complex:
    """0.5j <Z w|z> (core_operations.py:325-351)."""
    w0, w1 = _halves(w, h)
    z0, z1 = _halves(z, h)
    return 0.5j * (np.vdot(w0, z0) - np.vdot(w1, z1))


def dot_cp11(w: np.ndarray, z: np.ndarray, hc: int, ht: int) -> complex:
    """-1j <P11 w|z> taken before the CP gate (core_op_matrix.py:430-477);
    equals core_operations.py:972-975 (vdot(dCP w, CP z))."""
    return -1j * np.vdot(_quad(w, hc, ht)(1, 1), _quad(z, hc, ht)(1, 1))


# ---------------------------------------------------------------------------
# V, V^H applied to a flat array whose qubit q has stride k * 2**q.
# ---------------------------------------------------------------------------


def _block_list(a: Ansatz):
    """Running block index i -> (i, i mod L, ctrl, targ)."""
    nb = a.num_blocks
    for i in range(nb + a.tail_blocks):
        j = i % nb
        yield i, j, int(a.blocks[0, j]), int(a.blocks[1, j])


def _rs(a: Ansatz):
    return rx if a.entangler == "cx" else rz


def _apply_v(a: Ansatz, thetas: np.ndarray, flat: np.ndarray, k: int) -> None:
    """flat <- V flat (core_operations.py:606-710, core_op_matrix.py:480-559)."""
    n = a.n
    t1 = thetas[: 3 * n].reshape(n, 3)
    t2 = thetas[3 * n :].reshape(-1, a.tpb)
    rs = _rs(a)
    for q in range(n):
        h = k << q
        rz(flat, h, t1[q, 2])
        ry(flat, h, t1[q, 1])
        rz(flat, h, t1[q, 0])
    for i, j, c, t in _block_list(a):
        hc, ht = k << c, k << t
        th = t2[j]
        if a.trotter and i % 3 == 0:
            rz(flat, hc, -np.pi / 2)
        _entangle(flat, hc, ht, a.entangler, float(th[4]) if a.tpb == 5 else 0.0)
        ry(flat, hc, th[0])
        rz(flat, hc, th[1])
        ry(flat, ht, th[2])
        rs(flat, ht, th[3])
        if a.trotter and i % 3 == 2:
            rz(flat, ht, np.pi / 2)


def _apply_vh(a: Ansatz, thetas: np.ndarray, flat: np.ndarray, k: int) -> None:
    """flat <- V^H flat (core_operations.py:713-820, core_op_matrix.py:562-642)."""
    n = a.n
    t1 = thetas[: 3 * n].reshape(n, 3)
    t2 = thetas[3 * n :].reshape(-1, a.tpb)
    rs = _rs(a)
    for i, j, c, t in reversed(list(_block_list(a))):
        hc, ht = k << c, k << t
        th = t2[j]
        if a.trotter and i % 3 == 2:
            rz(flat, ht, -np.pi / 2)
        rs(flat, ht, -th[3])
        ry(flat, ht, -th[2])
        rz(flat, hc, -th[1])
        ry(flat, hc, -th[0])
        _entangle(flat, hc, ht, a.entangler, -float(th[4]) if a.tpb == 5 else 0.0)
        if a.trotter and i % 3 == 0:
            rz(flat, hc, np.pi / 2)
    for q in range(n):
        h = k << q
        rz(flat, h, -t1[q, 0])
        ry(flat, h, -t1[q, 1])
        rz(flat, h, -t1[q, 2])


def _sweep(
    a: Ansatz,
    thetas: np.ndarray,
    w: np.ndarray,
    z: np.ndarray,
    k: int,
    block_range: Tuple[int, int],
    front_layer: bool,
) -> np.ndarray:
    """Forward w/z sweep with one inner product per parameter
    (core_operations.py:918-1019, core_op_matrix.py:713-762); w, z are
    overwritten.  Inner products are taken after the rotation (SV convention);
    the matrix code takes them before, which is the same number because each
    Pauli commutes with its own rotation."""
    n = a.n
    grad = np.zeros(a.num_thetas, dtype=C128)
    g1 = grad[: 3 * n].reshape(n, 3)
    g2 = grad[3 * n :].reshape(-1, a.tpb)
    t1 = thetas[: 3 * n].reshape(n, 3)
    t2 = thetas[3 * n :].reshape(-1, a.tpb)
    is_cx = a.entangler == "cx"
    rs = _rs(a)
    dot_s = dot_x if is_cx else dot_z

    for q in range(n):
        h = k << q
        for slot, rot, dot in ((2, rz, dot_z), (1, ry, dot_y), (0, rz, dot_z)):
            rot(w, h, t1[q, slot])
            rot(z, h, t1[q, slot])
            if front_layer:
                g1[q, slot] = dot(w, z, h)

    for i, j, c, t in _block_list(a):
        hc, ht = k << c, k << t
        th = t2[j]
        live = block_range[0] <= j < block_range[1]
        if a.trotter and i % 3 == 0:
            rz(w, hc, -np.pi / 2)
            rz(z, hc, -np.pi / 2)
        angle = float(th[4]) if a.tpb == 5 else 0.0
        if live and a.tpb == 5:
            g2[j, 4] += dot_cp11(w, z, hc, ht)
        _entangle(z, hc, ht, a.entangler, angle)
        _entangle(w, hc, ht, a.entangler, angle)
        for slot, rot, dot, h in (
            (0, ry, dot_y, hc),
            (1, rz, dot_z, hc),
            (2, ry, dot_y, ht),
            (3, rs, dot_s, ht),
        ):
            rot(w, h, th[slot])
            rot(z, h, th[slot])
            if live:
                g2[j, slot] += dot(w, z, h)
        if a.trotter and i % 3 == 2:
            rz(w, ht, np.pi / 2)
            rz(z, ht, np.pi / 2)
    return grad


def _check_range(a: Ansatz, block_range) -> Tuple[int, int]:
    br = (0, a.num_blocks) if block_range is None else (int(block_range[0]), int(block_range[1]))
    if a.num_blocks > 0 and not (0 <= br[0] < br[1] <= a.num_blocks):
        raise ValueError("invalid block_range")
    return br


# ---------------------------------------------------------------------------
# State-vector API (core_operations.py).
# ---------------------------------------------------------------------------


def v_mul_vec(circ, thetas: np.ndarray, vec: np.ndarray) -> np.ndarray:
    """out = V vec (core_operations.py:606)."""
    a = as_ansatz(circ)
    out = np.array(vec, dtype=C128).ravel().copy()
    _apply_v(a, np.asarray(thetas, float), out, 1)
    return out


def v_dagger_mul_vec(circ, thetas: np.ndarray, vec: np.ndarray) -> np.ndarray:
    """out = V^H vec (core_operations.py:713)."""
    a = as_ansatz(circ)
    out = np.array(vec, dtype=C128).ravel().copy()
    _apply_vh(a, np.asarray(thetas, float), out, 1)
    return out


def grad_of_dot_product(
    circ,
    thetas: np.ndarray,
    x_vec: np.ndarray,
    vh_y_vec: np.ndarray,
    block_range: Optional[Tuple[int, int]] = None,
    front_layer: bool = True,
) -> np.ndarray:
    """Complex gradient of <V x|y> given vh_y = V^H y (core_operations.py:823)."""
    a = as_ansatz(circ)
    w = np.array(x_vec, dtype=C128).ravel().copy()
    z = np.array(vh_y_vec, dtype=C128).ravel().copy()
    return _sweep(a, np.asarray(thetas, float), w, z, 1, _check_range(a, block_range), bool(front_layer))


# ---------------------------------------------------------------------------
# Matrix API (core_op_matrix.py); no Trotter decorations there.
# ---------------------------------------------------------------------------


def _no_trotter(a: Ansatz) -> Ansatz:
    return Ansatz(a.n, a.entangler, a.blocks, False, False)


def v_mul_mat(circ, thetas: np.ndarray, mat: np.ndarray) -> np.ndarray:
    """V mat (core_op_matrix.py:480)."""
    a = _no_trotter(as_ansatz(circ))
    m = np.array(mat, dtype=C128, order="C")
    _apply_v(a, np.asarray(thetas, float), m.reshape(-1), m.shape[1])
    return m


def v_dagger_mul_mat(circ, thetas: np.ndarray, mat: np.ndarray) -> np.ndarray:
    """V^H mat (core_op_matrix.py:562)."""
    a = _no_trotter(as_ansatz(circ))
    m = np.array(mat, dtype=C128, order="C")
    _apply_vh(a, np.asarray(thetas, float), m.reshape(-1), m.shape[1])
    return m


def grad_of_matrix_dot_product(circ, thetas: np.ndarray, x_mat: np.ndarray, vh_y_mat: np.ndarray) -> np.ndarray:
    """Complex gradient of <V X|Y>_F given vh_y = V^H Y (core_op_matrix.py:645)."""
    a = _no_trotter(as_ansatz(circ))
    w = np.array(x_mat, dtype=C128, order="C")
    z = np.array(vh_y_mat, dtype=C128, order="C")
    return _sweep(a, np.asarray(thetas, float), w.reshape(-1), z.reshape(-1), w.shape[1], (0, a.num_blocks), True)


def coord_descent_single_sweep(circ, thetas: np.ndarray, target: np.ndarray, max_steps: int = -1) -> Tuple[np.ndarray, float]:
    """One Gauss-Seidel sweep for 1 - |<V,U>|^2/d^2 (core_op_matrix.py:765-917).
    Returns (updated thetas, fobj); the input ``thetas`` is not modified.  ``max_steps`` >= 0 stops the walk after that
    many parameter updates (the entangler of a block is applied before the block's first update): single steps of the walk
    can then be pinned without the amplification of rounding by the ~T sequential Newton steps that follow."""
    a = _no_trotter(as_ansatz(circ))
    if a.entangler == "cp":
        raise NotImplementedError("CPhase entangler is not supported yet")
    th = np.array(thetas, dtype=float).copy()
    n, d = a.n, target.shape[0]
    tol = float(np.sqrt(np.finfo(np.float64).eps))
    learn_rate, max_dt = np.pi / 16, np.pi / 4
    w = np.eye(d, dtype=C128).reshape(-1)
    z = v_dagger_mul_mat(a, th, target).reshape(-1)
    t1 = th[: 3 * n].reshape(n, 3)
    t2 = th[3 * n :].reshape(-1, 4)
    rs = _rs(a)
    dot_s = dot_x if a.entangler == "cx" else dot_z

    def delta(prod: complex, grad: complex) -> float:
        d1 = (-2.0 * np.real(np.conj(prod) * grad)) / (d**2)
        d2 = (-2.0 * abs(grad) ** 2 + 0.5 * abs(prod) ** 2) / (d**2)
        if d2 < tol:
            d1 /= max(abs(d1), 1.0)
            dt = -learn_rate * d1
        else:
            dt = -d1 / d2
        r = abs(dt / max_dt)
        return dt if r <= 1 else dt / r

    done = [0]

    def step(tht: np.ndarray, slot: int, rot, dot, h: int) -> None:
        if 0 <= max_steps <= done[0]:
            return
        done[0] += 1
        grad = dot(w, z, h)
        prod = np.vdot(w, z)
        rot(z, h, tht[slot])
        tht[slot] += delta(prod, grad)
        rot(w, h, tht[slot])

    for q in range(n):
        h = d << q
        step(t1[q], 2, rz, dot_z, h)
        step(t1[q], 1, ry, dot_y, h)
        step(t1[q], 0, rz, dot_z, h)
    for _, j, c, t in _block_list(a):
        hc, ht = d << c, d << t
        if 0 <= max_steps <= done[0]:
            break
        _entangle(z, hc, ht, a.entangler, 0.0)
        _entangle(w, hc, ht, a.entangler, 0.0)
        step(t2[j], 0, ry, dot_y, hc)
        step(t2[j], 1, rz, dot_z, hc)
        step(t2[j], 2, ry, dot_y, ht)
        step(t2[j], 3, rs, dot_s, ht)
    return th, float(1 - np.abs(np.vdot(w, z) / d) ** 2)


# ---------------------------------------------------------------------------
# MPS helpers (mps_operations.py) and the MPS-dot gradient.
# ---------------------------------------------------------------------------


def preprocess_mps(mps, conjugate: bool = False) -> List[np.ndarray]:
    """(gammas, lambdas) -> list of (2, chi_l, chi_r) tensors with lambda_q
    folded into the right bond (mps_operations.py:126-156)."""
    gam, lam = mps
    out = []
    for q, (g0, g1) in enumerate(gam):
        t = np.stack((np.asarray(g0, C128), np.asarray(g1, C128)))
        if q < len(gam) - 1:
            t = t * np.asarray(lam[q], float).reshape(1, 1, -1)
        out.append(np.conj(t) if conjugate else t)
    return out


def mps_to_vector(mps) -> np.ndarray:
    """Dense state; index bit q <-> site q (mps_operations.py:159-189)."""
    ts = preprocess_mps(mps)
    acc = ts[0][:, 0, :]  # (2, chi0): rows = bit 0
    for t in ts[1:]:
        # new index = old index + 2^q * b  => b is the slow (row-major outer) axis
        acc = np.einsum("ia,bac->bic", acc, t).reshape(-1, t.shape[2])
    return acc.reshape(-1).astype(C128)


def mps_dot(mps1, mps2) -> complex:
    """<mps1|mps2> by transfer matrices (mps_operations.py:192-213)."""
    a, b = preprocess_mps(mps1), preprocess_mps(mps2)
    e = np.einsum("bx,by->xy", np.conj(a[0][:, 0, :]), b[0][:, 0, :])
    for ta, tb in zip(a[1:], b[1:]):
        e = np.einsum("xy,bxu,byv->uv", e, np.conj(ta), tb)
    return complex(e.item())


def fast_dot_gradient(circ, thetas, lvec, vh_phi, block_range=None, front_layer=True) -> np.ndarray:
    """MPS twin of grad_of_dot_product (mps_dot_objective.py:41-242) under no
    truncation: identical control flow on the densified states."""
    return grad_of_dot_product(circ, thetas, mps_to_vector(lvec), mps_to_vector(vh_phi), block_range, front_layer)


def random_mps(n: int, chi: int, rng: np.random.Generator):
    """Synthetic normalised MPS in (gammas, lambdas) format with bond cap chi
    (format: mps_operations.py:33,87-123)."""
    dims = [1] + [min(chi, 2 ** min(q + 1, n - 1 - q)) for q in range(n - 1)] + [1]
    gam, lam = [], []
    for q in range(n):
        shp = (dims[q], dims[q + 1])
        g0 = rng.standard_normal(shp) + 1j * rng.standard_normal(shp)
        g1 = rng.standard_normal(shp) + 1j * rng.standard_normal(shp)
        gam.append((g0, g1))
        if q < n - 1:
            lam.append(np.sort(rng.random(dims[q + 1]) + 0.1)[::-1].copy())
    nrm = np.sqrt(abs(mps_dot((gam, lam), (gam, lam))))
    g0, g1 = gam[0]
    gam[0] = (g0 / nrm, g1 / nrm)
    return gam, lam


# ---------------------------------------------------------------------------
# Objective glue (objective_lhs_sur_max.py, objective_base.py, sk_core.py).
# ---------------------------------------------------------------------------


def flip_state_indices(n: int, max_flips: int, base_index: int = 0) -> np.ndarray:
    """Indices of the single non-zero of |0>, X_i|0>, X_i X_j|0>, ...
    (objective_base.py:42-97); ``base_index`` XORs a computational-basis
    state-preparation (e.g. the Neel pattern) on top."""
    idx = [0]
    for f in range(1, max_flips + 1):
        for sub in itertools.combinations(range(n), f):
            v = 0
            for q in sub:
                v ^= 1 << q
            idx.append(v)
    return np.asarray(idx, dtype=np.int64) ^ np.int64(base_index)


class SurMaxOracle:
    """Surrogate objective 1-(1-w)|h0|^2 - w|h_max|^2 and its gradient with
    the reference's hysteresis / weight-smoothing state machine
    (objective_lhs_sur_max.py:82-191, objective_base.py:715-734)."""

    gamma = 0.1

    def __init__(self, circ, target, max_flips=1, block_range=None, front_layer=False, base_index=0):
        self.a = as_ansatz(circ)
        self.target = np.asarray(target, C128)
        self.idx = flip_state_indices(self.a.n, max_flips, base_index)
        self.block_range = _check_range(self.a, block_range)
        self.front_layer = bool(front_layer)
        self.hs = np.zeros(self.idx.size, C128)
        self.hs2 = np.zeros(self.idx.size)
        self.max_no, self.weight, self.fobj, self.fidelity = 0, 1.0, 1.0, -1.0
        self.vh = None
        self.last = np.empty(0)

    def objective(self, thetas) -> float:
        self.last = np.array(thetas, float)
        self.vh = v_dagger_mul_vec(self.a, thetas, self.target)
        self.hs[:] = self.vh[self.idx]
        self.hs2[:] = np.abs(self.hs) ** 2
        best = self.hs2[self.max_no]
        for i in range(self.idx.size):
            if 1.1 * best < self.hs2[i]:
                best, self.max_no = self.hs2[i], i
        w = self.weight
        self.fobj = 1.0 - (1.0 - w) * self.hs2[0] - w * self.hs2[self.max_no]
        self.fidelity = self.hs2[0]
        return float(self.fobj)

    def _basis(self, i) -> np.ndarray:
        x = np.zeros(self.a.dim, C128)
        x[self.idx[i]] = 1
        return x

    def gradient(self, thetas) -> np.ndarray:
        tol = float(np.sqrt(np.finfo(np.float64).eps))
        if self.last.size == 0 or not np.allclose(thetas, self.last, atol=tol, rtol=tol):
            self.objective(thetas)
        front = self.front_layer or self.block_range == (0, self.a.num_blocks)
        g0 = grad_of_dot_product(self.a, thetas, self._basis(0), self.vh, self.block_range, front)
        if self.max_no == 0:
            full = (g0 * (-2 * np.conj(self.hs[0]))).real.copy()
        else:
            full = (g0 * (-2 * (1 - self.weight) * np.conj(self.hs[0]))).real.copy()
            gm = grad_of_dot_product(self.a, thetas, self._basis(self.max_no), self.vh, self.block_range, front)
            full += (gm * (-2 * self.weight * np.conj(self.hs[self.max_no]))).real
        self.weight += self.gamma * (float(np.sqrt(abs(self.fobj))) - self.weight)
        return full


def sketching_objective_and_gradient(circ, thetas, x_mat, y_mat) -> Tuple[float, np.ndarray]:
    """fobj = 1 - Re<X|V^H Y>/k, grad = -Re(g)/k (sk_core.py:167-194)."""
    k = x_mat.shape[1]
    vh_y = v_dagger_mul_mat(circ, thetas, y_mat)
    fobj = 1 - np.real(np.vdot(x_mat, vh_y)) / k
    g = grad_of_matrix_dot_product(circ, thetas, x_mat, vh_y)
    return float(fobj), -np.real(g) / k


# ---------------------------------------------------------------------------
# Block-layout generators needed to build benchmark inputs
# (circuit_structures.py:133-178,263-349), restated as index arithmetic.
# ---------------------------------------------------------------------------


def spin_blocks(n: int, depth: int) -> np.ndarray:
    pairs = [(i, i + 1) for s in (0, 1) for i in range(s, n - 1, 2)]
    cyc = itertools.cycle(pairs)
    return np.array([next(cyc) for _ in range(depth)], dtype=np.int64).T.reshape(2, depth)


def cyclic_spin_blocks(n: int, depth: int) -> np.ndarray:
    b = np.zeros((2, depth), dtype=np.int64)
    for i in range(depth):
        off = (i // (n // 2)) % 2 if n % 2 == 0 else 0
        b[0, i] = (2 * i + off) % n
        b[1, i] = (2 * i + off + 1) % n
    return b


def trotter_blocks(n: int, num_layers: int) -> np.ndarray:
    """Triplets (t,c),(c,t),(t,c) over the spin pattern
    (circuit_structures.py:133-178)."""
    base = spin_blocks(n, num_layers * (n - 1))
    out = np.repeat(base, 3, axis=1)
    out[:, 0::3] = base[::-1]
    out[:, 2::3] = base[::-1]
    return out


def rand_thetas(num: int, rng: np.random.Generator) -> np.ndarray:
    """pi*(2u-1) (utils.py:63-68)."""
    return np.pi * (2 * rng.random(num) - 1)


def rand_state(n: int, rng: np.random.Generator) -> np.ndarray:
    """uniform+i*uniform, normalised (utils.py:71-79)."""
    s = rng.random(1 << n) + 1j * rng.random(1 << n)
    return s / np.linalg.norm(s)

"""CPU-only: the C-ABI library loads and exports every symbol of include/aqc_hip.h; the
host-side planner produces valid stage plans; host-side validation raises like the reference."""
import os
import re

import numpy as np
import pytest

from oracle import aqc_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from aqc_research_amd import _lib

    header = open(os.path.join(ROOT, "include", "aqc_hip.h")).read()
    declared = set(re.findall(r"\b(aqc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()  # resolves every symbol or raises
    assert L.aqc_version().startswith(b"aqc_hip")


def _circ(n, ent, blocks, trotter=False, order2=False):
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz

    return TrotterAnsatz(n, blocks, order2) if trotter else ParametricCircuit(n, ent, blocks)


@pytest.mark.parametrize("n,depth", [(2, 1), (3, 4), (6, 11), (12, 40), (16, 40), (20, 64)])
def test_planner_valid_plans(n, depth):
    from aqc_research_amd.engine import HipContext

    rng = np.random.default_rng(n)
    layouts = [orc.spin_blocks(n, depth), np.stack([rng.permutation(n)[:2] for _ in range(depth)], axis=1)]
    for blocks in layouts:
        ctx = HipContext.of(_circ(n, "cx", blocks.astype(np.int64)))
        G = ctx.num_gate_groups
        assert G == n + depth
        for which in (0, 1, 2):
            for k in (2, 4, 7, 10, 12, 13):
                for low in (0, 2, 3):
                    plan = ctx.plan(which, 1, k, low)  # library validates (check_plan) or raises
                    ops = [g for _, o in plan for g in o]
                    assert sorted(ops) == list(range(G))
                    for bits, o in plan:
                        assert len(bits) == min(k, n) and bits == sorted(bits)
                        if low and k - 2 >= low and n > k:
                            assert bits[:low] == list(range(low))


@pytest.mark.parametrize("n,depth,trot", [(9, 20, 0), (13, 40, 0), (16, 40, 0), (20, 40, 0), (12, 0, 3), (14, 0, 2), (20, 0, 2)])
def test_mirrored_vdag_plan_walks_the_sweep_backwards(n, depth, trot):
    """V^H of the matrix-core workspaces = the sweep's plan (sub-stages included) walked backwards (mirror_plan): the library
    validates it as a plan of the INVERSE program (check_plan: every group once, per-qubit order of the inverse, local and
    register bits), stage j of it has the local bits and the groups of the sweep's stage m - 1 - j, in reverse order -- which is
    what makes the state before its last stage the sweep's z after its first one."""
    from aqc_research_amd.engine import HipContext

    blocks = orc.trotter_blocks(n, trot) if trot else orc.spin_blocks(n, depth)
    ctx = HipContext.of(_circ(n, "cx", blocks.astype(np.int64), bool(trot), bool(trot)))
    for k in (8, 10, 12):
        for low in (2, 3):
            sweep = ctx.plan(1, 1, k, low)
            mirror = ctx.plan(3, 1, k, low)      # raises if check_plan rejects it
            assert len(mirror) == len(sweep)
            for j, (bits, ops) in enumerate(mirror):
                sbits, sops = sweep[len(sweep) - 1 - j]
                assert bits == sbits and sorted(ops) == sorted(sops)
            assert sorted(g for _, o in mirror for g in o) == list(range(ctx.num_gate_groups))


def test_planner_matrix_and_trotter():
    from aqc_research_amd.engine import HipContext

    ctx = HipContext.of(_circ(10, "cx", orc.spin_blocks(10, 40)))
    plan = ctx.plan(1, 1024, 12, 2)  # config 5: all ten qubits + 2 column bits in one LDS tile
    assert len(plan) == 1 and plan[0][0] == [0, 1] + list(range(10, 20))
    t = HipContext.of(_circ(12, "cx", orc.trotter_blocks(12, 2), True, True))
    assert t.num_gate_groups == 12 + 66 + 18 and t.num_thetas == 36 + 4 * 66
    assert len(t.plan(1, 1, 12, 3)) == 1  # 12 qubits: the whole sweep is one launch


def test_validation_errors():
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz

    with pytest.raises(ValueError):
        ParametricCircuit(3, "cx", np.array([[0, 1], [0, 2]]))  # ctrl == targ
    with pytest.raises(ValueError):
        ParametricCircuit(3, "cx", np.array([[0, 3], [1, 2]]))  # out of range
    with pytest.raises(ValueError):
        ParametricCircuit(3, "xx", np.array([[0], [1]]))
    with pytest.raises(ValueError):
        TrotterAnsatz(4, orc.spin_blocks(4, 9), second_order=False)  # not triplets
    tb = orc.trotter_blocks(5, 1)
    circ = TrotterAnsatz(5, tb, second_order=True)
    assert circ.num_layers == 1 and circ.bpl == 12 and circ.half_layer_num_blocks == 6
    with pytest.raises(ValueError):
        circ.insert_unit_blocks(3, tb)  # not aligned at a layer boundary
    th, idx = circ.insert_unit_blocks(12, tb, np.ones(circ.num_thetas))
    assert circ.num_blocks == 24 and th.size == circ.num_thetas and np.all(th[idx] == 0) and idx[0] == 15 + 48
    # C ABI rejects what the Python layer would (defence in depth)
    from aqc_research_amd import _lib
    import ctypes

    L = _lib.lib()
    h = ctypes.c_void_p()
    bad = np.array([0, 0], dtype=np.int32)
    assert L.aqc_create(3, 0, bad.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 1, 0, 0, ctypes.byref(h)) != 0
    assert b"unit-blocks" in L.aqc_last_error()


def test_fails_loudly_without_gpu():
    """On a box without an AMD GPU the product path must raise, never fall back."""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "from aqc_research_amd import ParametricCircuit\n"
        "import aqc_research_amd.core_operations as cop\n"
        "c = ParametricCircuit(2, 'cx', np.array([[0],[1]]))\n"
        "v = np.zeros(4, complex); v[0] = 1\n"
        "try:\n"
        "    cop.v_mul_vec(c, np.zeros(c.num_thetas), v, np.zeros(4, complex), None)\n"
        "    print('COMPUTED')\n"
        "except RuntimeError as e:\n"
        "    print('RAISED', e)\n" % ROOT
    )
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300).stdout
    assert "RAISED" in out and "no CPU fallback" in out, out


def test_mps_lanes_fail_loudly_without_gpu():
    """The lockstep lanes of the MPS engine have no CPU path either: creating a batch (and a single-lane state) raises."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from aqc_research_amd import mps_engine as me\n"
        "for make in (lambda: me.LockstepLanes(6, 4), lambda: me.DeviceMPS.basis_state(6, 3)):\n"
        "    try:\n"
        "        make()\n"
        "        print('CREATED')\n"
        "    except RuntimeError as e:\n"
        "        print('RAISED', e)\n" % ROOT
    )
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300).stdout
    assert out.count("RAISED") == 2 and out.count("no CPU fallback") == 2 and "CREATED" not in out, out


def test_projected_route_is_planned_on_the_host():
    """The projected route of the sparse-lhs sweep (csrc/aqc_ws_project.cpp) is decided and planned without a GPU: the qubits the
    stages after the first touch, those they share with the first stage, the virtual register and its (checked) plan.  Headline
    shape: 8 touched + 4 shared = 12 virtual qubits instead of 16, the second stage's 4 sub-stages on one 2^12 tile per lane;
    20 qubits: 14; circuits whose later stages touch every bit of the first stage, and single-stage plans, have no such route."""
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz
    from aqc_research_amd.circuit_structures import create_ansatz_structure, make_trotter_like_circuit
    from aqc_research_amd.engine import HipContext

    def route(n, blocks):
        return HipContext(ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", blocks))).plan_projected()

    head = route(16, 40)
    assert head["virtual_qubits"] == 12 and head["touched_qubits"] == 8 and head["shared_with_first_stage"] == 4
    assert head["stages"] == 1 and head["substages"] == 4 and head["substages_on_the_register"] == 7 and head["summed_bits"] == 8
    big = route(20, 40)
    assert big["virtual_qubits"] == 14 and big["virtual_qubits"] + 2 <= 20 and sum(big["substages_per_stage"]) == big["substages"]
    small = route(16, 20)   # 6 virtual qubits, padded to the smallest matrix-core tile
    assert small["virtual_qubits"] == 6 and small["padded_qubits"] == 8 and small["tile_bits"] == 8
    assert route(16, 80) == {} and route(12, 40) == {}
    tro = HipContext(TrotterAnsatz(20, make_trotter_like_circuit(20, 2), second_order=True)).plan_projected()
    assert tro["virtual_qubits"] == 16 and tro["touched_qubits"] == 12
    assert HipContext(TrotterAnsatz(20, make_trotter_like_circuit(20, 3), second_order=True)).plan_projected() == {}

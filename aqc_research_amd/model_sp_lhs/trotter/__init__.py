from .trotter import (  # noqa: F401
    BasisPrepCircuit,
    half_zero_circuit,
    identity_circuit,
    neel_init_state,
    init_ansatz_to_trotter,
    make_hamiltonian,
    neel_state_index,
    slice2q,
    trotter_alphas,
    trotter_ansatz,
    trotter_global_phase,
    trotter_state,
)

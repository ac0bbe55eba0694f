from .trotter import (  # noqa: F401
    init_ansatz_to_trotter,
    make_hamiltonian,
    neel_state_index,
    slice2q,
    trotter_alphas,
    trotter_ansatz,
    trotter_global_phase,
    trotter_state,
)

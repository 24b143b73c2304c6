"""Shared helpers of the BiCGSTAB / FCG / CGS tests: argument orders of the
step kernels (the same for the oracle's ref_* and the C ABI's gkomi_*), dense
-> CSR, and the kernel-case runner."""
import numpy as np

# op -> (vector args in call order, scalar args in call order); every vector is
# followed by its stride in the call; "stop" goes last
KERNEL_ARGS = {
    ("bicgstab", "step_1"): (["r", "p", "v"], ["rho", "prev_rho", "alpha", "omega"]),
    ("bicgstab", "step_2"): (["r", "s", "v"], ["rho", "alpha", "beta"]),
    ("bicgstab", "step_3"): (["x", "r", "s", "t", "y", "z"], ["alpha", "beta", "gamma", "omega"]),
    ("bicgstab", "finalize"): (["x", "y"], ["alpha"]),
    ("fcg", "step_1"): (["p", "z"], ["rho_t", "prev_rho"]),
    ("fcg", "step_2"): (["x", "r", "t", "p", "q"], ["beta", "rho"]),
    ("cgs", "step_1"): (["r", "u", "p", "q"], ["beta", "rho", "prev_rho"]),
    ("cgs", "step_2"): (["u", "v_hat", "q", "t"], ["alpha", "rho", "gamma"]),
    ("cgs", "step_3"): (["t", "u_hat", "r", "x"], ["alpha"]),
    ("bicg", "step_1"): (["p", "z", "p2", "z2"], ["rho", "prev_rho"]),
    ("bicg", "step_2"): (["x", "r", "r2", "p", "q", "q2"], ["beta", "rho"]),
}


def dense_to_csr(A):
    A = np.asarray(A, np.float64)
    n = A.shape[0]
    rp = np.arange(0, n * n + 1, n, dtype=np.int32)
    ci = np.tile(np.arange(n, dtype=np.int32), n)
    return n, rp, ci, A.reshape(-1).copy()

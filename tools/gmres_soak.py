#!/usr/bin/env python3
"""Soak of the blocked Arnoldi sweep: N solves of GMRES(30) on the 108^3 system (and a 64^3 one), checking that
every solve takes the same iterations, returns the same bits and never falls back (gkomi_gmres_meeting_fallbacks)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
import gkomi.solvers as solvers
from gkomi.formats import Csr
gk = gkomi.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
raw = ctypes.CDLL(os.path.join(ROOT, "repo-8852-ginkgo_amd", "lib", "libgkomi.so"))
raw.gkomi_gmres_meeting_fallbacks.restype = ctypes.c_longlong
for g in (108, 64):
    n, rp, ci, v = matgen.at_like(g)
    A = Csr(gk, n, n, dev(rp), dev(ci), dev(v))
    b = dev(np.cos(0.3 * np.arange(n)).reshape(n, 1))
    ref = None
    t0 = time.perf_counter()
    for i in range(reps):
        r = solvers.solve_op(gk, "gmres", A, b, krylov_dim=30, max_iters=3000, reduction=1e-10)
        key = (r["iterations"], r["x"].cpu().numpy().tobytes())
        if ref is None:
            ref = key
        assert key == ref, f"solve {i} differs: {r['iterations']} vs {ref[0]} iterations"
    torch.cuda.synchronize()
    print(f"{g}^3: {reps} solves, {ref[0]} iterations each, {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per solve incl. the host copy, "
          f"fallbacks so far {raw.gkomi_gmres_meeting_fallbacks()}", flush=True)

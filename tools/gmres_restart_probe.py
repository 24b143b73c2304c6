#!/usr/bin/env python3
"""GMRES(krylov_dim) per iteration with and without the single-launch Arnoldi step (GKOMI_GMRES_PERSISTENT is
read once per process: one child process per setting).  usage: gmres_restart_probe.py [child <kd>]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import gkomi, matgen, gkomi.solvers as solvers
    gk = gkomi.lib()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for name, (n, rp, ci, v) in (("5pt 1000^2", matgen.poisson_2d_5pt(1000)), ("AT-like 108^3", matgen.at_like(108))):
        rpd, cid, vd = d(rp), d(ci), d(v)
        s = torch.sin(torch.arange(n, dtype=torch.float64, device="cuda")); s /= torch.linalg.vector_norm(s)
        b = torch.zeros(n, dtype=torch.float64, device="cuda")
        gk.csr_spmv_f64_i32(torch.cuda.current_stream().cuda_stream, n, n, 1, len(v), rpd, cid, vd, s, 1, b, 1, None, None, 0, -1)
        for kd in (10, 20, 30, 50, 100):
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = solvers.gmres_solve(gk, n, rpd, cid, vd, b, krylov_dim=kd, max_iters=600, reduction=1e-10)
                torch.cuda.synchronize(); el = time.perf_counter() - t0
            print(f"{name} GMRES({kd:3d}) persistent={os.environ.get('GKOMI_GMRES_PERSISTENT', '1')}: {r['iterations']:4d} its {el * 1e3:8.2f} ms "
                  f"{el / r['iterations'] * 1e6:7.1f} us/it", flush=True)
    sys.exit(0)
for pers in ("1", "0"):
    env = dict(os.environ, GKOMI_GMRES_PERSISTENT=pers)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)

#!/usr/bin/env python3
"""Timing of the preconditioned-solver configs (BASELINE configs 3 and 4 with
the offline stand-ins of SURVEY 8(d)): CG + block-Jacobi on a randomly
permuted 2-D Poisson matrix ("T2-like") and GMRES(30) + ParILU on a 3-D
convection-diffusion stencil ("AT-like"), plus the triangular solves alone."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, gkomi.solvers as solvers, matgen, ilu_util
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s = torch.cuda.current_stream().cuda_stream
g3 = int(sys.argv[1]) if len(sys.argv) > 1 else 108
g2 = int(sys.argv[2]) if len(sys.argv) > 2 else 1108


def ev_time(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


# ---- AT-like: 3-D 7-pt convection-diffusion, GMRES(30) + ParILU ----
if os.environ.get("JACOBI_ONLY"):
    g3 = 8
n, rp, ci, v = matgen.poisson_3d_7pt(g3)
v = v.copy(); rows = np.repeat(np.arange(n), np.diff(rp))
v[ci == rows - 1] -= 0.5; v[ci == rows] += 0.5
rpd, cid, vd = d(rp), d(ci), d(v)
t0 = time.perf_counter()
f = ilu_util.gpu_par_ilu(gk, torch, n, rpd.clone(), cid, vd, iterations=5)   # benchmark default: 5 sweeps
torch.cuda.synchronize(); t_gen = time.perf_counter() - t0
L, U = f["L"], f["U"]
lnnz, unnz = int(L[2].numel()), int(U[2].numel())
b = d(np.cos(0.3 * np.arange(n)).reshape(n, 1))
y = torch.zeros((n, 1), dtype=torch.float64, device="cuda"); z = torch.zeros_like(y)
nb = gk.trs_workspace_bytes(); tws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
tl = ev_time(lambda: gk.lower_trs_solve_f64_i32(s, n, 1, L[0], L[1], L[2], 0, b, 1, y, 1, tws, nb))
tu = ev_time(lambda: gk.upper_trs_solve_f64_i32(s, n, 1, U[0], U[1], U[2], 0, y, 1, z, 1, tws, nb))
flag = ctypes.c_int(0); gk.trs_check_overrun(s, tws, ctypes.addressof(flag))
print(f"AT-like {g3}^3: n={n} nnz={len(v)}  ParILU(5 sweeps) generate {t_gen*1e3:.1f} ms, L nnz {lnnz}, U nnz {unnz}")
print(f"  lower trs {tl:9.1f} us ({(12*lnnz+20*n)/tl/1e3:6.1f} GB/s)   upper trs {tu:9.1f} us ({(12*unnz+20*n)/tu/1e3:6.1f} GB/s)  overrun={flag.value}")
pre_plain = solvers.ilu_from_factors(gk, n, L, U, analyse=False)
t0 = time.perf_counter(); pre = solvers.ilu_from_factors(gk, n, L, U, bricks=False); torch.cuda.synchronize()
print(f"  LowerTrs/UpperTrs generate (level analysis of both factors): {1e3*(time.perf_counter()-t0):.1f} ms, "
      f"levels {pre.l_plan.nlevels if pre.l_plan else '-'} / {pre.u_plan.nlevels if pre.u_plan else '-'}")
if pre.l_plan is not None:
    tl2 = ev_time(lambda: pre.l_plan.solve(b, y)); tu2 = ev_time(lambda: pre.u_plan.solve(y, z))
    print(f"  analysed: lower trs {tl2:9.1f} us   upper trs {tu2:9.1f} us   overrun={int(pre.l_plan.overrun() or pre.u_plan.overrun())}")
t0 = time.perf_counter(); pre_bk = solvers.ilu_from_factors(gk, n, L, U); torch.cuda.synchronize()
print(f"  LowerTrs/UpperTrs generate (brick plan where the factor has one): {1e3*(time.perf_counter()-t0):.1f} ms, "
      f"bricks {pre_bk.l_bricks.nbricks if pre_bk.l_bricks else '-'} / {pre_bk.u_bricks.nbricks if pre_bk.u_bricks else '-'}")
if pre_bk.l_bricks is not None and pre_bk.u_bricks is not None:
    tl3 = ev_time(lambda: pre_bk.l_bricks.solve(b, y)); tu3 = ev_time(lambda: pre_bk.u_bricks.solve(y, z))
    print(f"  brick plan: lower trs {tl3:9.1f} us   upper trs {tu3:9.1f} us   overrun={int(pre_bk.l_bricks.overrun() or pre_bk.u_bricks.overrun())}")
for prec, name in ((None, "none"), (pre_plain, "ParILU (analysis-free trs)"), (pre, "ParILU (level plan)"), (pre_bk, "ParILU (brick plan)")):
    solvers.gmres_solve(gk, n, rpd, cid, vd, b, krylov_dim=30, max_iters=3000, reduction=1e-10, precond=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = solvers.gmres_solve(gk, n, rpd, cid, vd, b, krylov_dim=30, max_iters=3000, reduction=1e-10, precond=prec)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"  GMRES(30) precond={name:26s}: {r['iterations']:5d} iters {el*1e3:9.2f} ms  {el/max(r['iterations'],1)*1e6:8.1f} us/it converged={r['converged']} rel_res={r['rel_residual']:.2e}")

# ---- T2-like: permuted 2-D Poisson, CG + block-Jacobi(32) ----
n, rp, ci, v = matgen.poisson_2d_5pt(g2)
rng = np.random.default_rng(42)
perm = rng.permutation(n); inv = np.argsort(perm)
rows = np.repeat(np.arange(n), np.diff(rp))
pr, pc = inv[rows], inv[ci]
order = np.lexsort((pc, pr))
rp2, ci2, v2 = matgen.coo_to_csr(n, pr[order].astype(np.int32), pc[order].astype(np.int32), v[order])
rpd, cid, vd = d(rp2), d(ci2), d(v2)
b = torch.ones((n, 1), dtype=torch.float64, device="cuda")
x = d(np.sin(0.01 * np.arange(n)).reshape(n, 1)); yy = torch.zeros_like(x)
nnz = len(v2)
tsp = ev_time(lambda: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, yy, 1, None, None, 0, 5), 50)
print(f"T2-like {g2}^2 permuted: n={n} nnz={nnz}  csr spmv {tsp:.1f} us ({(12*nnz+20*n)/tsp/1e3:.0f} GB/s algorithmic)")
t0 = time.perf_counter(); pre = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=32); torch.cuda.synchronize()
print(f"  block-Jacobi(32) generate {1e3*(time.perf_counter()-t0):.1f} ms, {pre.num_blocks} blocks")
t0 = time.perf_counter(); pre_ad = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=32, storage_optimization=solvers.AUTODETECT); torch.cuda.synchronize()
hist = np.bincount(pre_ad.block_precisions.cpu().numpy(), minlength=256)
print(f"  block-Jacobi(32, adaptive) generate {1e3*(time.perf_counter()-t0):.1f} ms, precisions " + ", ".join(f"0x{k:02x}: {c}" for k, c in enumerate(hist) if c))
yb = torch.zeros_like(b)
forced = [(f"0x{st:02x}", solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=32, storage_optimization=st))
          for st in (0x01, 0x10, 0x02, 0x11, 0x20)]
for name, pcd in [("fp64", pre), ("adaptive", pre_ad)] + forced:
    t = ev_time(lambda: gk.jacobi_apply_cb(pcd.ctx_ptr, s, b, yb), 50)
    print(f"  block-Jacobi(32) apply, {name:8s} storage: {t:7.1f} us")
if os.environ.get("JACOBI_ONLY"):
    sys.exit(0)
for prec, name in ((None, "none"), (pre, "Jacobi32"), (pre_ad, "Jac32-ad")):
    solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=20000, reduction=1e-10, mode=1, check_every=32, precond=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=20000, reduction=1e-10, mode=1, check_every=32, precond=prec)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"  CG precond={name:8s}: {r['iterations']:5d} iters {el*1e3:9.2f} ms {el/max(r['iterations'],1)*1e6:8.1f} us/it converged={r['converged']}")

"""<float, int32> CSR SpMV and Cg on the P2 / P3-size stencils: time per apply next to the double kernel of record."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, gkomi, matgen
from gkomi import formats
gk = gkomi.lib()
s = lambda: torch.cuda.current_stream().cuda_stream
def timed(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, gen in (("5-pt 1000^2", lambda: matgen.poisson_2d_5pt(1000)), ("7-pt 108^3", lambda: matgen.poisson_3d_7pt(108)), ("27-pt 100^3", lambda: matgen.stencil_3d_27pt(100))):
    n, rp, ci, v = gen()
    M = formats.Csr.from_host(gk, n, n, rp, ci, v)
    b64 = torch.from_numpy(np.cos(0.001 * np.arange(n))).cuda().reshape(n, 1)
    y64 = torch.zeros_like(b64)
    t64 = timed(lambda: M.apply(b64, y64))
    v32, b32, y32 = M.vals.float(), b64.float(), torch.zeros(n, 1, dtype=torch.float32, device="cuda")
    t32 = timed(lambda: gk.csr_spmv_f32_i32(s(), n, n, 1, M.nnz, M.row_ptrs, M.col_idxs, v32, b32, 1, y32, 1, None, None))
    by64, by32 = 12 * M.nnz + 4 * (n + 1) + 16 * n, 8 * M.nnz + 4 * (n + 1) + 8 * n
    err = float((y32.double() - y64).abs().max() / y64.abs().max())
    print(f"{name}: double {t64:6.1f} us ({by64 / t64 / 1e6:.2f} TB/s)   float {t32:6.1f} us ({by32 / t32 / 1e6:.2f} TB/s)   max rel diff {err:.1e}", flush=True)

#!/usr/bin/env python3
"""Launch timeline of the nonzero-split CSR SpMV kernel on the 1M-row 5-pt
matrix, cold (rotating copies) and warm: per-workgroup phase stamps from
tools/spmv_timeline.hip.  Prints when workgroups start, phase durations and
residency over time.  Diagnostic only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
tl = ctypes.CDLL(os.path.join(ROOT, "tools", "libspmv_timeline.so"))
tl.timeline_launch.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p]
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, rp, ci, v = matgen.poisson_2d_5pt(grid)
nnz = int(rp[-1]); tile = 1536; ntiles = nnz // tile + 1
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
ncopies = 8
copies = [(d(rp), d(ci), d(v), d(x), torch.empty((n, 1), dtype=torch.float64, device="cuda")) for _ in range(ncopies)]
s = torch.cuda.current_stream().cuda_stream
srows = []
for c in copies:
    t = torch.empty(int(gk.csr_srow_entries(nnz, tile)), dtype=torch.int32, device="cuda")
    gk.csr_make_srow_i32(s, n, nnz, c[0], tile, t, t.numel()); srows.append(t)
stamps = torch.zeros(8 * (ntiles + 8), dtype=torch.int64, device="cuda")


def launch(j, nt, swz, st):
    c = copies[j]
    rc = tl.timeline_launch(s, nt, swz, n, nnz, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(),
                            c[4].data_ptr(), srows[j].data_ptr(), 4, st.data_ptr() if st is not None else None)
    assert rc == 0


def report(name, nt, swz, cold):
    for i in range(40):                       # settle clocks and cache state
        launch(i % ncopies if cold else 0, nt, swz, None)
    stamps.zero_()
    launch(1 if cold else 0, nt, swz, stamps)
    torch.cuda.synchronize()
    a = stamps.cpu().numpy().reshape(-1, 8)[:ntiles]
    t = (a[:, :4] - a[:, 0].min()) * 0.01   # us
    start, lds, bar, end = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    print(f"== {name}: {ntiles} workgroups; first start 0, last start {start.max():.2f} us, last end {end.max():.2f} us")
    q = lambda z: "  ".join(f"{np.percentile(z, p):6.2f}" for p in (5, 25, 50, 75, 95))
    print(f"   start time            p5/25/50/75/95: {q(start)}")
    print(f"   entry -> products     p5/25/50/75/95: {q(lds - start)}   (streaming loads + gathers)")
    print(f"   products -> barrier   p5/25/50/75/95: {q(bar - lds)}   (waiting for the slowest wave)")
    print(f"   barrier -> stored     p5/25/50/75/95: {q(end - bar)}   (row sums + store)")
    print(f"   lifetime              p5/25/50/75/95: {q(end - start)}")
    edges = np.arange(0, end.max() + 0.5, 0.5)
    res = [(int(((start <= e) & (end > e)).sum()), int(((start <= e) & (lds > e)).sum())) for e in edges]
    print("   t us : resident (of which still loading)  " + "  ".join(f"{e:.1f}:{r}({l})" for e, (r, l) in zip(edges, res)))
    first = start < np.percentile(start, 50)
    print(f"   first half of the starts: lifetime median {np.median((end - start)[first]):.2f} us, "
          f"second half {np.median((end - start)[~first]):.2f} us")
    xcc = a[:, 5] & 0xf
    print("   workgroups per XCC: " + " ".join(str(int((xcc == k).sum())) for k in range(8)))


for nt in (0, 1):
    for swz in (1, 0):
        for cold in (True, False):
            report(f"nt={nt} swizzle={swz} {'cold' if cold else 'warm'}", nt, swz, cold)

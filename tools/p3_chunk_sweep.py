#!/usr/bin/env python3
"""256^3 7-pt SpMV (nonzero-split kernel, nontemporal streams): XCD chunking of the tile -> workgroup map.
no swizzle = consecutive tiles on consecutive XCDs (x lines pulled by up to three XCDs: 1.19x traffic);
chunk c = XCD k takes c consecutive tiles of every 8c (c = one eighth of all tiles: 1.0x traffic, 6 % slower in
round 2).  Median of 5 x 20 launches per setting."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, rp, ci, v = matgen.poisson_3d_7pt(g)
nnz = int(rp[-1])
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
a = [dev(rp), dev(ci), dev(v)]
del rp, ci, v
x = dev(np.sin(0.01 * np.arange(n)).reshape(n, 1))
y = torch.empty((n, 1), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
SPLIT, NT = 4, 2 << 8
literal = [int(t) for t in os.environ.get("CHUNKS", "").split(",") if t]
for tile in (3072, 2048):
    srow = torch.empty(int(gk.csr_srow_entries(nnz, tile)), dtype=torch.int32, device="cuda")
    gk.csr_make_srow_i32(s, n, nnz, a[0], tile, srow, srow.numel())
    ref = None
    for name, word in [("no swizzle", SPLIT | NT | (1 << 16))] + [(f"chunk {1 << (c - 1)}", SPLIT | NT | (c << 17)) for c in (5, 7, 9, 10, 11, 12, 13)] + [(f"literal {c}", ("lit", c)) for c in literal] + [("one eighth each", SPLIT | NT)]:
        if isinstance(word, tuple):
            os.environ["GKOMI_CSR_XCD_CHUNK"] = str(word[1])
            word = SPLIT | NT | (127 << 17)
        run = lambda: gk.csr_spmv_srow_f64_i32(s, n, n, 1, nnz, a[0], a[1], a[2], x, 1, y, 1, None, None, word, 7, srow, tile)
        for _ in range(3):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        got = y.clone()
        if ref is None:
            ref = got
        assert torch.equal(got, ref)
        t = statistics.median(ts)
        print(f"tile {tile} {name:16s}: {t:7.1f} us  {bytes_ / t / 1e6:6.2f} TB/s = {bytes_ / t / 8e6:.3f} of 8 TB/s", flush=True)

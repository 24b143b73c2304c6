"""Power-law class, column-partitioned copy: what if the few long virtual rows (hubs) left the virtual matrix and were
applied by the load-balanced kernel on their own, so that the short rest can take the row-cut stream / split kernel?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import colpart_probe as cp
from gkomi import formats
import benchmark_spmv as bs
gk = cp.gk
M = bs.random_matrix(gk, {"random": "powerlaw", "rows": 1000000, "nnz_per_row": 8}, 7)
n = M.nrows
b = torch.from_numpy(np.cos(0.001 * np.arange(n))).cuda().reshape(n, 1)
for nb in (4,):
    V = cp.partition(M, nb)
    vr = nb * n
    lens = (V.row_ptrs[1:] - V.row_ptrs[:-1]).long()
    part = torch.zeros(vr, 1, dtype=torch.float64, device="cuda")
    print(f"nb {nb}: whole virtual matrix (automatic = load-balanced): {cp.timed(lambda: V.apply(b, part)):.1f} us", flush=True)
    for H in (64, 256, 1024):
        hub = lens > H
        rows_of = torch.repeat_interleave(torch.arange(vr, device="cuda"), lens)
        keep = ~hub[rows_of]
        ls = torch.where(hub, torch.zeros_like(lens), lens)
        rp_s = torch.zeros(vr + 1, dtype=torch.int32, device="cuda"); rp_s[1:] = torch.cumsum(ls, 0).int()
        S = formats.Csr(gk, vr, n, rp_s, V.col_idxs[keep].contiguous(), V.vals[keep].contiguous())
        hub_rows = torch.nonzero(hub).flatten()
        lh = lens[hub_rows]
        rp_h = torch.zeros(len(hub_rows) + 1, dtype=torch.int32, device="cuda"); rp_h[1:] = torch.cumsum(lh, 0).int()
        Hm = formats.Csr(gk, len(hub_rows), n, rp_h, V.col_idxs[~keep].contiguous(), V.vals[~keep].contiguous(), strategy=3)
        hp = torch.zeros(len(hub_rows), 1, dtype=torch.float64, device="cuda")
        ts = {name: cp.timed(lambda W=S.to(name): W.apply(b, part)) for name in ("csr", "csrm")}
        th = cp.timed(lambda: Hm.apply(b, hp))
        print(f"   hubs = virtual rows longer than {H}: {len(hub_rows)} rows, {int(lh.sum())} of {M.nnz} nonzeros; short part: split {ts['csr']:.1f} us, "
              f"stream {ts['csrm']:.1f} us; hub part (load-balanced, compact): {th:.1f} us; + reduce ~8 us -> ~{min(ts.values()) + th + 8:.1f} us", flush=True)

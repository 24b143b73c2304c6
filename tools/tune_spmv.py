#!/usr/bin/env python3
"""Interleaved A/B timing of CSR SpMV kernel variants in ONE process
(cdna_hip_programming.md §5.4 rule 24).  Prints median/min us per launch, cold
(rotating copies) and warm, and the algorithmic GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gkomi
import matgen

gk = gkomi.lib()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, rp, ci, v = matgen.poisson_2d_5pt(grid)
nnz = int(rp[-1])
bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
ncopies = max(2, int(700e6 // bytes_))
def arena_copy():
    """all five arrays of one copy carved out of ONE allocation (2 MiB-aligned pieces):
    does the cold number depend on how the allocator scatters them?"""
    al = 1 << 21
    sizes = [rp.nbytes, ci.nbytes, v.nbytes, x.nbytes, 8 * n]
    offs, off = [], 0
    for sz in sizes:
        offs.append(off)
        off += (sz + al - 1) // al * al
    buf = torch.empty(off + al, dtype=torch.uint8, device="cuda")
    base = (-buf.data_ptr()) % al
    def view(o, sz, dtype, shape=None):
        t = buf[base + o: base + o + sz].view(dtype)
        return t.reshape(shape) if shape else t
    out = (view(offs[0], sizes[0], torch.int32), view(offs[1], sizes[1], torch.int32), view(offs[2], sizes[2], torch.float64),
           view(offs[3], sizes[3], torch.float64, (n, 1)), view(offs[4], sizes[4], torch.float64, (n, 1)))
    out[0].copy_(d(rp)); out[1].copy_(d(ci)); out[2].copy_(d(v)); out[3].copy_(d(x))
    return out + (buf,)


if os.environ.get("TUNE_ARENA"):
    copies = [arena_copy() for _ in range(ncopies)]
else:
    copies = [(d(rp), d(ci), d(v), d(x), torch.empty((n, 1), dtype=torch.float64, device="cuda")) for _ in range(ncopies)]
s = torch.cuda.current_stream().cuda_stream
STREAM, VECTOR = 1, 2
variants = {"stream_v0(256,1,2048)": STREAM, "v1(256,2,4096)": STREAM | (1 << 8), "v2(512,1,4096)": STREAM | (2 << 8),
            "v3(256,4,8192)": STREAM | (3 << 8), "v4(128,1,1024)": STREAM | (4 << 8), "v5(256,1,1536)": STREAM | (5 << 8),
            "v6(512,1,3072)": STREAM | (6 << 8), "v7(1024,1,6144)": STREAM | (7 << 8),
            "v0_noswz": STREAM | (1 << 16), "v5_noswz": STREAM | (5 << 8) | (1 << 16),
            "vector4": VECTOR | (4 << 8),
            "v8(64,1,384)": STREAM | (8 << 8), "v9(192,1,1152)": STREAM | (9 << 8), "v10(320,1,1920)": STREAM | (10 << 8),
            "v11(128,1,768)": STREAM | (11 << 8), "v12(256,1,1024)": STREAM | (12 << 8), "v13(384,1,2304)": STREAM | (13 << 8),
            "v9_noswz": STREAM | (9 << 8) | (1 << 16), "v11_noswz": STREAM | (11 << 8) | (1 << 16),
            "v14(256,1,1536,nt)": STREAM | (14 << 8), "v14_noswz": STREAM | (14 << 8) | (1 << 16)}
SPLIT = 4
# nonzero-split kernel over srow: (tile, variant bits: 2 = nontemporal, 256 = no XCD chunking)
for tile in (1024, 1536, 2048):
    for name, bits in (("", 0), ("_nt", 2), ("_noswz", 256), ("_nt_noswz", 258)):
        variants[f"split{tile}{name}"] = (SPLIT | ((bits & 0xff) << 8) | ((bits >> 8) << 16), tile)
if os.environ.get("TUNE_ONLY"):
    variants = {k: v for k, v in variants.items() if any(t in k for t in os.environ["TUNE_ONLY"].split(","))}
extra = [a for a in sys.argv[2:]]
for e in extra:
    variants[f"custom_{e}"] = int(e, 0)


srows = {}


def srow_of(copy_index, tile):
    """every copy carries its own srow (like its own row_ptrs): cold means cold"""
    key = (copy_index, tile)
    if key not in srows:
        cnt = int(gk.csr_srow_entries(nnz, tile))
        t = torch.empty(cnt, dtype=torch.int32, device="cuda")
        gk.csr_make_srow_i32(s, n, nnz, copies[copy_index][0], tile, t, cnt)
        srows[key] = t
    return srows[key]


def run(strategy, cold, reps=200):
    tile = None
    if isinstance(strategy, tuple):
        strategy, tile = strategy
        for i in range(ncopies):
            srow_of(i, tile)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        j = i % ncopies if cold else 0
        c = copies[j]
        if tile is None:
            gk.csr_spmv_f64_i32(s, n, n, 1, nnz, c[0], c[1], c[2], c[3], 1, c[4], 1, None, None, strategy, 5)
        else:
            gk.csr_spmv_srow_f64_i32(s, n, n, 1, nnz, c[0], c[1], c[2], c[3], 1, c[4], 1, None, None, strategy, 5,
                                     srows[(j, tile)], tile)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {k: {"cold": [], "warm": []} for k in variants}
for rnd in range(7):
    for k, st in variants.items():
        for mode in ("cold", "warm"):
            t = run(st, mode == "cold")
            if rnd > 0:
                res[k][mode].append(t)
print(f"grid {grid}: n={n} nnz={nnz} algorithmic bytes={bytes_} copies={ncopies}")
print(f"{'variant':26s} {'cold med us':>11s} {'min':>7s} {'GB/s':>8s} | {'warm med us':>11s} {'min':>7s} {'GB/s':>8s}")
for k in variants:
    c, w = np.array(res[k]["cold"]), np.array(res[k]["warm"])
    print(f"{k:26s} {np.median(c):11.2f} {c.min():7.2f} {bytes_/np.median(c)/1e3:8.0f} | "
          f"{np.median(w):11.2f} {w.min():7.2f} {bytes_/np.median(w)/1e3:8.0f}")

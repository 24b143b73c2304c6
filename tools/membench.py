#!/usr/bin/env python3
"""Runs tools/membench.hip: practical bandwidth ceiling for the SpMV byte mix."""
import ctypes, os, sys
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmembench.so"))
lib.membench_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = grid * grid
nnz = 5 * n - 4 * grid
bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
ncopies = max(2, int(700e6 // bytes_))
mk = lambda: (torch.ones(nnz, dtype=torch.float64, device="cuda"), torch.ones(nnz, dtype=torch.int32, device="cuda"),
              torch.ones(n, dtype=torch.float64, device="cuda"), torch.ones(n + 4, dtype=torch.int32, device="cuda"),
              torch.empty(n, dtype=torch.float64, device="cuda"))
copies = [mk() for _ in range(ncopies)]
s = torch.cuda.current_stream().cuda_stream
def run(blocks, cold, reps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        c = copies[i % ncopies] if cold else copies[0]
        lib.membench_launch(s, blocks, nnz, n, c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), c[4].data_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"grid {grid} bytes {bytes_}")
for blocks in (512, 1024, 2048, 4096, 8192):
    for rnd in range(2):
        c, w = run(blocks, True), run(blocks, False)
    print(f"blocks {blocks:5d}: cold {c:7.2f} us {bytes_/c/1e3:6.0f} GB/s | warm {w:7.2f} us {bytes_/w/1e3:6.0f} GB/s")

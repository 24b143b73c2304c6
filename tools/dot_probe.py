#!/usr/bin/env python3
"""What the dot epilogue of the nonzero-split SpMV kernel costs on the 256^3 7-pt matrix, part by part:
tools/dot_probe.hip compiled with GKOMI_DOT_PROBE = 0..7 (bit 0 no status check, bit 1 no load of the other
factor, bit 2 no reduction across the workgroup), each timed with HIP events next to the plain kernel on the same
box.  Diagnostic only; build with tools/dot_probe.sh."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, rp, ci, v = matgen.poisson_3d_7pt(g)
nnz = int(rp[-1]); tile = 3072
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rp_d, ci_d, v_d = d(rp), d(ci), d(v)
x = d(np.sin(0.01 * np.arange(n)).reshape(n, 1)); y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
srow = torch.empty(int(gk.csr_srow_entries(nnz, tile)), dtype=torch.int32, device="cuda")
gk.csr_make_srow_i32(s, n, nnz, rp_d, tile, srow, srow.numel())
partial = torch.zeros(nnz // tile + 64, dtype=torch.float64, device="cuda")
status = torch.zeros(64, dtype=torch.uint8, device="cuda")
libs = {}
for k in range(8):
    path = os.path.join(ROOT, "tools", "bin", f"libdot_probe{k}.so")
    if os.path.exists(path):
        libs[k] = ctypes.CDLL(path)
        libs[k].probe_launch.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p] * 2


def run(k, dot, reps=60):
    def one():
        rc = libs[k].probe_launch(s, dot, n, nnz, rp_d.data_ptr(), ci_d.data_ptr(), v_d.data_ptr(), x.data_ptr(),
                                  y.data_ptr(), srow.data_ptr(), 6, partial.data_ptr(), status.data_ptr())
        assert rc == 0
    for _ in range(10):
        one()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); one(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


names = {0: "full dot epilogue", 1: "no status check", 2: "no load of the other factor", 4: "no reduction",
         3: "no status, no load", 5: "no status, no reduction", 6: "no load, no reduction", 7: "none of the three"}
print(f"{g}^3 7-pt, {n} rows, {nnz} nonzeros, tile {tile}; median of 60 launches, us")
for rnd in range(2):
    base = run(0, 0)
    print(f"round {rnd}: plain kernel {base:.1f}")
    for k in sorted(libs):
        t = run(k, 1)
        print(f"   probe {k} ({names[k]}): {t:.1f}  (+{t - base:.1f})")

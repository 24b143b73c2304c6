"""Differential fuzz of every SpMV entry point against the oracle on small odd shapes (tools/fuzz_spmv.py: empty
matrices, nnz 0 / 1 / 2 / 3, single rows and columns, runs of empty rows, rows longer than a tile, nnz on the tile
boundaries, unsorted rows, several right-hand sides, advanced applies) -- two fixed seeds."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 11])
def test_fuzz_against_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_spmv.py"), "150", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]

import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO_ROOT, os.path.join(REPO_ROOT, "repo-8852-ginkgo_amd"),
          os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gk():
    """The HIP library through its C ABI.  Fails loudly when it is missing."""
    import gkomi
    return gkomi.lib()


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    return torch.device("cuda:0")

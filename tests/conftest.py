import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    so = os.path.join(ge.PKG_DIR, "csrc", "libbdpt_amd.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(so) and os.path.exists(orc)):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def ob(pkg):
    import oracle_binding
    oracle_binding.load_oracle(pkg.abi)
    return oracle_binding


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible — the HIP path has no CPU fallback")
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()

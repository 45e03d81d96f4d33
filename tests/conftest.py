import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load_acx():
    mod = importlib.import_module("arithmetic-circuits_amd")
    sys.modules.setdefault("acx", mod)
    return mod


@pytest.fixture(scope="session")
def acx():
    return _load_acx()


def _gpu_context(acx, field):
    """GPU context.  On a box with no GPU at all the GPU tests SKIP (so that a plain `pytest tests` is green
    there); wherever a GPU is visible -- the driver's GPU box -- or with ACX_REQUIRE_GPU=1, a failure to create the
    context is an error: there is no CPU path to fall into."""
    import torch
    if not torch.cuda.is_available() and os.environ.get("ACX_REQUIRE_GPU") != "1":
        pytest.skip("no GPU visible (set ACX_REQUIRE_GPU=1 to make this an error)")
    return acx.Context(field, 0)


@pytest.fixture(scope="session")
def ctx_bn254(acx):
    c = _gpu_context(acx, "bn254")
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_bls(acx):
    c = _gpu_context(acx, "bls12_381")
    yield c
    c.close()


@pytest.fixture(scope="session")
def c_oracle_bn254():
    from oracle.c_oracle import COracle
    return COracle("bn254")


@pytest.fixture(scope="session")
def c_oracle_bls():
    from oracle.c_oracle import COracle
    return COracle("bls12_381")

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


_GPU_REQUIRED = False


def pytest_collection_modifyitems(config, items):
    """`pytest -m gpu` (the GPU job: the marker expression selects gpu tests and does not negate them) REQUIRES a GPU:
    a box that has lost its device must fail, not skip its way to green.  ACX_REQUIRE_GPU=1 forces the same."""
    global _GPU_REQUIRED
    expr = (config.option.markexpr or "").replace(" ", "")
    _GPU_REQUIRED = os.environ.get("ACX_REQUIRE_GPU") == "1" or ("gpu" in expr and "notgpu" not in expr)


def gpu_required() -> bool:
    return _GPU_REQUIRED


def _gpu_context(acx, field):
    """GPU context.  A plain `pytest tests` on a box with no GPU SKIPS the GPU tests; under `-m gpu` or with
    ACX_REQUIRE_GPU=1, and wherever a GPU is visible, a failure to create the context is an error: there is no CPU
    path to fall into."""
    import torch
    if not torch.cuda.is_available() and not _GPU_REQUIRED:
        pytest.skip("no GPU visible (run with -m gpu or ACX_REQUIRE_GPU=1 to make this an error)")
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

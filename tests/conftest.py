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


@pytest.fixture(scope="session")
def ctx_bn254(acx):
    """GPU context; raises (never skips silently into a CPU path) if the device is missing."""
    c = acx.Context("bn254", 0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_bls(acx):
    c = acx.Context("bls12_381", 0)
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

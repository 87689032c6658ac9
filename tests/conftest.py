import os
import sys

# The oracle's C loops carry `omp parallel for` and numpy's BLAS keeps a thread pool: on the GPU box (256 hardware threads) every
# one of the chain tests' ~1e5 small calls would wake a 256-thread team (measured: a chain test 100 s there against 14 s on 8
# cores).  Capped before anything loads a threading runtime; an explicit setting in the environment wins.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from thunder_amd import capi
    capi.load()  # fails loudly when the HIP extension is missing
    return torch.device("cuda:0")


@pytest.fixture
def knob_env(monkeypatch):
    """sets a THX_* switch of the library for the duration of a test: the library reads its environment once, at load,
    so the harness asks it to re-read (thx_knobs_reload) after every change and again on teardown"""
    from thunder_amd import capi

    def set_(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, value)
        capi.call("thx_knobs_reload")
    yield set_
    monkeypatch.undo()
    capi.call("thx_knobs_reload")

import os
import sys

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

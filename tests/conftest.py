import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _kernel_env_follows_os_environ():
    """libmarius_hip.so reads its MARIUS_* switches once, at load (no getenv in per-call code).  Tests that monkeypatch one call hip.reload_env()
    after setting it; this fixture (set up before monkeypatch, torn down after it has restored the environment) puts the library back."""
    yield
    mod = sys.modules.get("marius_amd.hip")
    if mod is not None:
        mod.reload_env()

"""pytest configuration: registers the `gpu` marker, puts the repo root on sys.path, and
exposes the product package (directory ``stheno.jl_amd/``) as module ``stheno_jl_amd``."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

entry.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with `-m gpu`; when run without a device they must fail loudly,
    # never silently pass -- so no auto-skip here.
    pass


@pytest.fixture(scope="session")
def pkg():
    import stheno_jl_amd
    return stheno_jl_amd


@pytest.fixture(scope="session")
def rng_seed():
    return 123456

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """the library under test is the one built from the sources present: rebuild it (hipcc cross-compiles without a GPU) when the digest inside it
    says otherwise, instead of testing a stale git-ignored binary that travelled with a snapshot"""
    from seal_amd import _build
    if _build.stale():
        _build.build(verbose=True)
    from oracle import seal_oracle
    seal_oracle.build_oracle_lib()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

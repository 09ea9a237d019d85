import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """A fresh checkout has no libtheora_hip.so (build artefacts are not in the history): build it
    once per session, exactly as __graft_entry__.build() does.  Nothing is rebuilt if it exists."""
    from theora_amd import build
    if not os.path.exists(build.OUT):
        build.build()
    yield


@pytest.fixture(scope="session")
def hip():
    """The product library; on the GPU box it must load and a device must be visible."""
    import torch
    import theora_amd
    from theora_amd import _lib
    _lib.load()   # raises if libtheora_hip.so is missing: no silent fallback
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    return theora_amd

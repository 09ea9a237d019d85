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


@pytest.fixture(autouse=True)
def _options_back_at_their_defaults():
    """A test that changes a run-time option of the library puts it back: the table is process-wide, and a test that leaves
    fe_assign or tl_levels changed makes every later test of the session run a setting nobody asked for (round 4 had one)."""
    from theora_amd import _lib
    from tests import util
    try:
        L = _lib.load()
    except Exception:
        yield
        return
    before = util.options_snapshot(L)
    yield
    after = util.options_snapshot(L)
    changed = {k.decode(): (before[k], after[k]) for k in before if after.get(k) != before[k]}
    for k, (b, _) in changed.items():
        L.thip_set_option(k.encode(), b)     # (the next test starts clean whatever this one did)
    assert not changed, "options left changed (before, after): %r" % changed

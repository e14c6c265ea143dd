import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build what is missing (host lib, oracle; the HIP lib cross-compiles without a GPU)."""
    from cutrace_amd import build
    build.build_host()
    build.build_hip()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libctr_oracle.so")):
        build.build_oracle()
    yield


@pytest.fixture(scope="session")
def ca():
    import cutrace_amd
    return cutrace_amd


def load_scene(ca, name, w=None, h=None):
    s = ca.HostScene.load(f"scene/{name}.json")
    assert s.ok, f"scene/{name}.json failed to load"
    if w is not None:
        s.set_size(w, h)
    return s

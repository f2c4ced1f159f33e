import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def fr():
    """The product package; importing it loads libfractalrenderer_amd.so (built on demand)."""
    lib = os.path.join(ROOT, "fractalrenderer_amd", "libfractalrenderer_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    import fractalrenderer_amd
    return fractalrenderer_amd


@pytest.fixture(scope="session")
def renderer(fr):
    """One fr_ctx on cuda:0 -- only gpu-marked tests may request it."""
    r = fr.Renderer(0)
    yield r
    r.close()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    return {"frames": np.load(os.path.join(g, "frames.npz")),
            "palettes": np.load(os.path.join(g, "palettes.npz")),
            "franim": os.path.join(g, "reference_sample.franim")}


@pytest.fixture(scope="session")
def spv_golden():
    """Vectors produced by executing the reference's compiled shaders (tests/golden/make_spv_golden.py)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "spv_frames.npz"))

import glob
import os
import sys

import numpy as np
import pytest

try:  # load torch's bundled HIP runtime BEFORE libdmi_hip.so pulls in the system one: a process that ends up with
    # the two in the other order finds no GPU when torch initialises (the distributed GPU test needs torch)
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for everything but the distributed tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(params=golden_names())
def golden(request):
    d = load_golden(request.param)
    d["name"] = request.param
    return d


@pytest.fixture(autouse=True)
def _brick_classes_on_tiny_grids(monkeypatch):
    """See tests/helpers.py: contexts created by tests keep their brick classes whatever the grid size."""
    from cudadepthmapintegration_amd import capi
    from helpers import variant_for_tests
    for cls in (capi.FusionContext, capi.MultiContext):
        orig = cls.__init__

        def init(self, *args, _orig=orig, kernel_variant=0, **kw):
            _orig(self, *args, kernel_variant=variant_for_tests(int(kernel_variant)), **kw)

        monkeypatch.setattr(cls, "__init__", init)
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getattr__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G()


@pytest.fixture(autouse=True, scope="session")
def _extra_engine_flags():
    """Validation aid: IRSDE_TEST_EXTRA_FLAGS=<int> ORs engine flags into EVERY engine the test session creates, so that the whole GPU
    parity suite (goldens, tolerances unchanged) can be replayed in an opt-in mode, e.g. 32768 = IRSDE_FLAG_SPLIT_F16X2:
        IRSDE_TEST_EXTRA_FLAGS=32768 pytest tests -m gpu
    Engines whose own flags select a 16-bit mode are left alone (the split flags do not combine with them)."""
    extra = int(os.environ.get("IRSDE_TEST_EXTRA_FLAGS", "0"))
    if not extra:
        yield
        return
    from image_restoration_sde_amd import _lib, unet
    orig = unet._Engine.__init__

    def patched(self, module, device_index, flags=0):
        if not flags & (_lib.FLAG_BF16 | _lib.FLAG_BF16_ACT | _lib.FLAG_FP16):
            flags |= extra
        orig(self, module, device_index, flags)
    unet._Engine.__init__ = patched
    yield
    unet._Engine.__init__ = orig

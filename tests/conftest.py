import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tools/run_sanitized_tests.sh: the same tests over the sanitizer build of the library (host side under ASan + UBSan).  The
    # package itself only ever loads lib/libvstab.so; another build has to be asked for by path, here, explicitly.
    alt = os.environ.get("VSTAB_TEST_LIB")
    if alt:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import devlib
        sys.modules["video-annotator_amd"] = devlib.load(alt)


@pytest.fixture(scope="session")
def vs():
    """The product binding (ctypes over libvstab.so).  Import fails loudly if the .so is missing."""
    return importlib.import_module("video-annotator_amd")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch.device("cuda:0")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build the product library (hipcc cross-compiles without a GPU) and the
    # CPU oracle once per session; both are git-ignored artefacts.
    import importlib.util
    spec = importlib.util.spec_from_file_location("_svoxt_build", os.path.join(ROOT, "svox_t_amd", "build.py"))
    _build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(_build)     # by path: importing the package loads the library
    _build.build()
    from oracle import oracle as _oracle
    _oracle.build()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")

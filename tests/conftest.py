import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# tests/ablation/: parity tests of experiment kernels that exist only in -DSIMT_ABLATION builds of the library (csrc/experiments/): not part of
# the product, not collected unless asked for (they would be permanent skips on the shipped library)
collect_ignore_glob = [] if os.environ.get("SIMT_ABLATION_TESTS") == "1" else ["ablation/*"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ablation: needs the -DSIMT_ABLATION build of the library (SIMT_ABLATION_TESTS=1; see tests/ablation/)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")

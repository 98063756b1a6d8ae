import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "akaze-rust_amd", "python"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref():
    """The CPU oracle (test infrastructure)."""
    import akaze_ref
    akaze_ref.build()
    return akaze_ref


@pytest.fixture(scope="session")
def amd():
    """The product binding; building the library is __graft_entry__.build()'s job."""
    import akaze_amd
    if not os.path.exists(akaze_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    akaze_amd.lib()
    return akaze_amd


@pytest.fixture(scope="session")
def ctx(amd):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()

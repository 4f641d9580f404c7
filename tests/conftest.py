import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def tune():
    """Pin vdn_gemm's kernel selection for one test (the knobs travel per launch in vdn_gemm_desc.tuning) and restore it after:
    `tune(force_bm=192)`. The library reads its VDN_GEMM_* environment defaults once per process, never per launch."""
    from vdn import _abi
    saved = _abi.OVERRIDE

    def _set(**kw):
        _abi.set_tuning(**kw)

    yield _set
    _abi.restore_tuning(saved)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the Gaussian taps reproduce upstream's mixed numpy / torch arithmetic on purpose (mv_utils.py:204-220); numpy 2 warns about it
    config.addinivalue_line('filterwarnings', 'ignore:__array_wrap__:DeprecationWarning')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('this test is marked gpu and needs a GPU; run the CPU suite with -m "not gpu"')
    return torch.device('cuda:0')


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """Every test session needs the in-tree C-ABI library (symbol checks on CPU, compute on GPU)."""
    from vilgod_amd import build
    build.build(verbose=False)
    yield

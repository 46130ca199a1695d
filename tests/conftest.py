import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def v3d(sub=None):
    """The package directory starts with a digit -> importlib."""
    return importlib.import_module('3dvnet_amd' + ('.' + sub if sub else ''))


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no HIP device is visible')
    return torch.device('cuda:0')

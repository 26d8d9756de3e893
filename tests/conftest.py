import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture
def experiment_build(monkeypatch):
    """Engines created inside the test load libcraftingworld_exp.so (same sources, -DCW_EXPERIMENT): the launch-shape knobs
    (CW_TUNE_FUSED_RENDER, CW_TUNE_OVERLAP, CW_TUNE_FUSED_STEP, CW_TUNE_RENDER_LINEAR, CW_TUNE_RENDER_QALL, ...) are compiled out of
    the product library."""
    monkeypatch.setenv('CW_EXPERIMENT_BUILD', '1')

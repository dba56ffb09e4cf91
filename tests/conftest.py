import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'na-fwebsod_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box)')
    config.addinivalue_line('markers', 'ab: checks a kernel form that LOST its A/B and exists only in '
                                       'the A/B build (make AB=1 -> libnaws_hip_ab.so); collected '
                                       'only when NAWS_LIB points at that library')


def pytest_collection_modifyitems(config, items):
    """Tests of A/B-only kernel forms are not part of the product suite: deselected (not skipped)
    unless the run is pointed at the A/B library, so `-m gpu` reports no skips for them."""
    if 'libnaws_hip_ab' in os.environ.get('NAWS_LIB', ''):
        return
    drop = [it for it in items if it.get_closest_marker('ab') is not None]
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = [it for it in items if it.get_closest_marker('ab') is None]


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture
def cfgmod():
    from detectron.core import config as c
    c.reset_cfg()
    yield c
    c.reset_cfg()

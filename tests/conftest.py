import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from ifcb_classifier_amd import _lib
    c = _lib.Context(0)
    c.reserve(512 << 20)
    return c


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """after every GPU test: drop the engines it built and hand the cached blocks back to the driver.  One pytest process runs the
    whole GPU suite (engines of batch 256 / 768 / 1024 among them); torch's caching allocator never returns a block by itself, so
    without this the reserved memory only grows and the HIP runtime's own allocations (graph launches, kernel arguments) compete
    with hundreds of GB of idle cache.  Prints the high-water mark when IFCBK_TEST_MEM=1."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    import gc
    import torch
    if not torch.cuda.is_available():
        return
    gc.collect()
    if os.environ.get('IFCBK_TEST_MEM'):
        print('\n[mem] %s: reserved %.1f GB, allocated %.1f GB' % (request.node.name, torch.cuda.memory_reserved() / 1e9,
                                                                   torch.cuda.memory_allocated() / 1e9))
    torch.cuda.empty_cache()

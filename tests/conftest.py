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


# Measured figures that must be visible in a GREEN log (`pytest -q` shows the output of failing tests only): a test appends lines to
# MEASURED (tests/local_parity.py: measured(...)) and they are printed after the run, under the pass / fail summary line's section.
MEASURED = []


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if MEASURED:
        terminalreporter.section('measured (printed by passing tests)')
        for ln in MEASURED:
            terminalreporter.write_line(ln)

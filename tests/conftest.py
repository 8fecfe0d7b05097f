import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: takes the better part of an hour (the sanitizer leg); runs only with DISCO_RUN_SLOW=1')


GOLDEN = os.path.join(REPO, 'tests', 'golden')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hp():
    from tacotron_amd.hparams import HParams
    return HParams()


@pytest.fixture(scope='session')
def weights(hp):
    """Random-init weights with non-trivial BN statistics and biases (exercise every fold)."""
    from tacotron_amd.weights import init_weights
    return init_weights(hp, seed=1234, randomize_bn=True, randomize_bias=True)


@pytest.fixture(scope='session')
def engine(hp, weights):
    from tacotron_amd.engine import TacoEngine
    eng = TacoEngine(weights, hp, device=0)
    yield eng
    eng.close()

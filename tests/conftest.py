import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def load_generator(kind):
    """oracle GeneratorRef from the committed weight fixtures."""
    from oracle.gen_ref import CNNWeights, GeneratorRef
    d = golden(f'weights_{kind}.npz')
    nets = [CNNWeights.from_npz_dict(d, 'net0_')]
    if kind == 'gz':
        nets.append(CNNWeights.from_npz_dict(d, 'net1_'))
    return GeneratorRef(kind, nets, d['x_std'], d['y_std'])

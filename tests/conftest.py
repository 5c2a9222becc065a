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


def load_generator(kind, regression=False):
    """oracle GeneratorRef from the committed weight fixtures.  regression: 'gan' / 'vae' with GZ's net_mean as the
    regression net — the combination tests/golden/make_golden_regression.py ran through the reference."""
    from oracle.gen_ref import CNNWeights, GeneratorRef
    if kind.endswith('+reg'):
        kind, regression = kind[:-4], True
    d = golden(f'weights_{kind}.npz')
    nets = [CNNWeights.from_npz_dict(d, 'net0_')]
    if kind == 'gz':
        nets.append(CNNWeights.from_npz_dict(d, 'net1_'))
    elif regression:
        nets.append(CNNWeights.from_npz_dict(golden('weights_gz.npz'), 'net0_'))
    return GeneratorRef(kind, nets, d['x_std'], d['y_std'])

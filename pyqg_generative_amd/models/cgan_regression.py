"""CGAN generator parameterization, inference surface of
pyqg_generative/models/cgan_regression.py (:22-66 constructor arguments, :133-137 generate,
:154-171 generate_latent_noise / predict_snapshot / predict_mean_snapshot)."""
import numpy as np
import torch

from .parameterization import Parameterization
from ..tools.cnn_tools import apply_function


class _LatentCNN(Parameterization):
    """Shared by the CGAN generator and the CVAE decoder: S = y_std * Net([q/x_std, z])."""
    n_latent = 2

    def generate_latent_noise(self, ny, nx):
        return np.random.randn(1, self.n_latent, ny, nx).astype('float32')

    def predict_mean_snapshot(self, m, M=100):
        q = np.asarray(m.q, dtype='float64')
        if q.ndim != 3:
            raise ValueError('predict_mean_snapshot expects a single member (2,N,N)')
        X = self.x_scale.normalize(q.astype('float32'))                  # (1,2,N,N)
        z = np.random.randn(M, self.n_latent, q.shape[-2], q.shape[-1]).astype('float32')
        Y = apply_function(self._gen, np.tile(X, (M, 1, 1, 1)), z).mean(0, keepdims=True)
        return self.y_scale.denormalize(Y).squeeze().astype('float64')


class CGANRegression(_LatentCNN):
    kind = 'gan'

    def __init__(self, regression='None', nx=64, generator='Andrew', folder='model', div=False,
                 hidden_channels=[128, 64, 32, 32, 32, 32, 32], device=0):
        if regression != 'None' or generator != 'Andrew' or div or \
                list(hidden_channels) != [128, 64, 32, 32, 32, 32, 32]:
            raise NotImplementedError('only the shipped configuration (regression="None", '
                                      'generator="Andrew", div=False) has a device path')
        self.regression, self.generator, self.nx, self.div = regression, generator, nx, div
        self.hidden_channels = hidden_channels
        self._load(folder, device)          # needs G.pt, x_scale.json, y_scale.json (D.pt is training-only)

"""Guillaumin-Zanna mean/variance parameterization, inference surface of
pyqg_generative/models/mean_var_model.py (:14-17 VarCNN softplus, :24-39 constructor,
:102-115 generate_latent_noise / predict_snapshot / predict_mean_snapshot)."""
import numpy as np

from .parameterization import Parameterization
from ..tools.cnn_tools import apply_function


class MeanVarModel(Parameterization):
    kind = 'gz'
    NET_NAMES = ('net_mean', 'net_var')

    def __init__(self, folder='model', hidden_channels=[128, 64, 32, 32, 32, 32, 32], device=0):
        if list(hidden_channels) != [128, 64, 32, 32, 32, 32, 32]:
            raise NotImplementedError('only the shipped channel configuration has a device path')
        self.hidden_channels = hidden_channels
        self._load(folder, device)          # needs net_mean.pt and net_var.pt

    def generate_latent_noise(self, ny, nx):
        return np.random.randn(2, ny, nx)

    def predict_mean_snapshot(self, m, M=100):
        X = self.x_scale.normalize(np.asarray(m.q, 'float64').astype('float32'))
        return self.y_scale.denormalize(apply_function(self.net_mean, X)).squeeze().astype('float64')

    def predict(self, ds, M=1000, seed=None):
        """mean_var_model.py:117-135: mean net, softplus variance net, one Gaussian sample."""
        from ..tools.simulate import dataset_backend
        xr = dataset_backend()
        qv = np.asarray(ds['q'].values)
        X = self.x_scale.normalize(qv.reshape((-1,) + qv.shape[2:]).astype('float32'))
        mean = self.y_scale.denormalize(apply_function(self.net_mean, X)).reshape(qv.shape)
        var = self.y_scale.denormalize_var(np.logaddexp(0, apply_function(self.net_var, X))).reshape(qv.shape)
        rng = np.random if seed is None else np.random.RandomState(seed)
        Y = mean + np.sqrt(var) * rng.randn(*var.shape)
        dims = ['run', 'time', 'lev', 'y', 'x']
        return xr.Dataset({'q_forcing_advection': (dims, Y), 'q_forcing_advection_mean': (dims, mean),
                           'q_forcing_advection_var': (dims, var)})

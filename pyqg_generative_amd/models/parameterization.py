"""Plugin base class (reference: pyqg_generative/models/parameterization.py:13-34).

``__call__(m)`` keeps the reference's stand-alone semantics (host sampler + one generator
call + per-layer de-mean, returning float64 (nz,ny,nx)).  When the object is attached to a
``QGModel`` of this package the model does NOT call it per step: it hands the device
generator to qgx_step, which runs sampler, generator, de-mean and the spectral step
back-to-back on the GPU (``device_generator``).
"""
import os
import numpy as np
import torch

from ..qgmodel import QParameterization
from ..engine import Generator
from .. import weights as _weights
from ..tools.cnn_tools import ChannelwiseScaler, DeviceNet


class Parameterization(QParameterization):
    kind = None          # 'gan' | 'vae' | 'gz'

    def _load(self, folder, device=0):
        self.folder = folder
        nets, xs, ys = _weights.load_folder(folder, self.kind, regression=getattr(self, 'regression', 'None') != 'None')
        self.x_scale = ChannelwiseScaler(xs)
        self.y_scale = ChannelwiseScaler(ys)
        self._gen = Generator(self.kind, nets, xs, ys, device=device)
        self._bind_nets()

    NET_NAMES = ()       # the reference's attribute names of the nets, in the generator's net order

    def _bind_nets(self):
        """the reference's model objects expose their torch modules (G / decoder / net_mean, net_var); notebooks hand
        them to apply_function(net, *X, fun=...) — here handles on the device-resident nets (tools/cnn_tools.py)"""
        for inet, name in enumerate(self.NET_NAMES):
            setattr(self, name, DeviceNet(self._gen, inet))

    @classmethod
    def from_arrays(cls, nets, x_std, y_std, device=0, **kw):
        """Build from in-memory weights (fixtures, synthetic) instead of a model folder."""
        self = cls.__new__(cls)
        self.folder = None
        # a second net beside a generator / decoder is the regression net (cgan_regression.py:59-60)
        self.regression = kw.get('regression', 'full_loss' if cls.kind != 'gz' and len(nets) == 2 else 'None')
        if (self.regression != 'None') != (cls.kind != 'gz' and len(nets) == 2):
            raise ValueError("regression != 'None' takes two nets (generator / decoder, net_mean); 'None' one")
        if self.regression != 'None':
            self.NET_NAMES = tuple(cls.NET_NAMES[:1]) + ('net_mean',)
        self.n_latent = 2
        self.x_scale = ChannelwiseScaler(x_std)
        self.y_scale = ChannelwiseScaler(y_std)
        self._gen = Generator(cls.kind, nets, x_std, y_std, device=device)
        self._bind_nets()
        return self

    def device_generator(self):
        return self._gen

    # ---- hooks of the reference API -----------------------------------------------------
    def generate_latent_noise(self, ny, nx):
        raise NotImplementedError

    def _forward(self, q, noise, demean):
        """q (2,N,N) or (B,2,N,N) float64, noise matching -> S float64, same leading shape."""
        q = np.asarray(q, dtype='float64')
        single = q.ndim == 3
        qd = torch.as_tensor(np.ascontiguousarray(q.reshape((-1, 2) + q.shape[-2:]))).cuda()
        z = np.asarray(noise).reshape(qd.shape)
        z = torch.as_tensor(np.ascontiguousarray(z), dtype=self._gen.noise_dtype).cuda()
        S = self._gen.forward(qd, z, demean=demean).cpu().numpy()
        return S[0] if single else S

    def predict_snapshot(self, m, noise):
        return self._forward(m.q, noise, demean=False)

    def predict_mean_snapshot(self, m, M=100):
        raise NotImplementedError

    def __call__(self, m):
        if m.sampling_type == 'deterministic':
            S = self.predict_mean_snapshot(m)
            m_forcing = S - S.mean(axis=(-2, -1), keepdims=True)
        elif m.noise_sampler.update(lambda: self.generate_latent_noise(m.ny, m.nx)):
            m_forcing = self._forward(m.q, m.noise_sampler.noise, demean=True)
        else:
            return self._last
        self._last = m_forcing
        try:
            m.PV_forcing = m_forcing
        except AttributeError:        # read-only device view on this package's QGModel
            pass
        return m_forcing

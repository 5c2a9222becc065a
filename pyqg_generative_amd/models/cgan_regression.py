"""CGAN generator parameterization, inference surface of
pyqg_generative/models/cgan_regression.py (:22-66 constructor arguments, :133-137 generate,
:154-171 generate_latent_noise / predict_snapshot / predict_mean_snapshot)."""
import numpy as np
import torch

from .parameterization import Parameterization
from ..tools.cnn_tools import apply_function


REGRESSIONS = ('None', 'full_loss', 'residual_loss')


class _LatentCNN(Parameterization):
    """Shared by the CGAN generator and the CVAE decoder: S = y_std * Net([q/x_std, z]), and with regression != 'None'
    S = y_std * (Net([q/x_std, z]) + net_mean(q/x_std)) (cgan_regression.py:157-162, cvae_regression.py:131-136; the two
    regression modes differ in training only)."""
    n_latent = 2

    def _set_regression(self, regression):
        if regression not in REGRESSIONS:
            raise ValueError(f'regression must be one of {REGRESSIONS}')
        self.regression = regression
        if regression != 'None':
            self.NET_NAMES = tuple(type(self).NET_NAMES[:1]) + ('net_mean',)

    def _mean_correction(self, X):
        """net_mean(X) for normalised PV X (B,2,N,N) float32 numpy, or 0 without a regression net"""
        if self.regression == 'None':
            return 0.0
        return apply_function(self.net_mean, X)

    def generate_latent_noise(self, ny, nx):
        return np.random.randn(1, self.n_latent, ny, nx).astype('float32')

    def generate(self, x, z=None):
        """device tensors: normalised PV x (B,2,Ny,Nx) and latent noise z (B,n_latent,Ny,Nx; drawn here if omitted) ->
        normalised forcing (cgan_regression.py:133-137, cvae_regression.py:114-118); the `fun` the reference passes to
        apply_function"""
        if z is None:
            z = torch.randn((x.shape[0], self.n_latent, x.shape[2], x.shape[3]), device=x.device)
        return getattr(self, self.NET_NAMES[0])(torch.cat([x, z.to(x.dtype)], dim=1))

    def predict_mean_snapshot(self, m, M=100, seed=None):
        """Deterministic sampling: the mean of M forcing realisations for ONE PV snapshot
        (cgan_regression.py:164-171, cvae_regression.py:138-145).  seed=None draws the latent noise from
        numpy's global stream as the reference does; an integer seed draws realisation j on the device from
        the Philox stream (seed, counter j), which the oracle reproduces (tests)."""
        q = np.asarray(m.q, dtype='float64')
        if q.ndim != 3:
            raise ValueError('predict_mean_snapshot expects a single member (2,N,N)')
        N = q.shape[-1]
        X = self.x_scale.normalize(q.astype('float32'))                  # (1,2,N,N)
        if seed is None:
            z = np.random.randn(M, self.n_latent, q.shape[-2], N).astype('float32')
            Y = apply_function(getattr(self, self.NET_NAMES[0]), np.tile(X, (M, 1, 1, 1)), z, fun=self.generate).mean(0, keepdims=True)
            Y = Y + self._mean_correction(X)
        else:
            from .._lib import lib, check
            from ..engine import _ptr, _stream
            x = torch.empty((M, 4, N, N), dtype=torch.float32, device='cuda')
            x[:, :2] = torch.as_tensor(np.ascontiguousarray(X)).cuda()
            z = torch.empty((M, 2, N, N), dtype=torch.float32, device='cuda')
            check(lib.qgx_noise_normal(_ptr(z), 0, M, 2 * N * N, int(seed), 0, 0, 0.0, 1.0, _stream()))
            x[:, 2:] = z
            Y = self._gen.cnn_forward(x).to(torch.float64).mean(0, keepdim=True).cpu().numpy().astype('float32')
            Y = Y + self._mean_correction(X)
        return self.y_scale.denormalize(Y).squeeze().astype('float64')

    def predict(self, ds, M=1000, seed=0):
        """Offline Monte-Carlo prediction for a dataset with q (run, time, lev, y, x)
        (cgan_regression.py:173-189, cvae_regression.py:147-163): one sample, the mean and the variance
        of M realisations per snapshot."""
        from ..tools.simulate import dataset_backend
        xr = dataset_backend()
        qv = np.asarray(ds['q'].values)
        Y, mean, var = self.generate_mean_var(qv.reshape((-1,) + qv.shape[2:]), M=M, seed=seed)
        dims = ['run', 'time', 'lev', 'y', 'x']
        return xr.Dataset({'q_forcing_advection': (dims, Y.reshape(qv.shape)),
                           'q_forcing_advection_mean': (dims, mean.reshape(qv.shape)),
                           'q_forcing_advection_var': (dims, var.reshape(qv.shape))})

    def generate_mean_var(self, q, M=1000, seed=0, batch_size=64):
        """Offline Monte-Carlo sampling (reference: generate_mean_var + predict,
        cgan_regression.py:139-146,173-189): for PV snapshots q (T,2,N,N) draw M realisations of the
        forcing with fresh on-device latent noise each; returns (one sample, mean, variance), float64,
        in physical units (y_scale applied; variance with y_scale^2, unbiased as torch.var)."""
        import ctypes as C
        from .._lib import lib, check
        from ..engine import _ptr, _stream
        q = np.asarray(q, dtype='float64').reshape((-1, 2) + np.shape(q)[-2:])
        T, _, N, _ = q.shape
        sample = np.empty((T, 2, N, N)); mean = np.empty_like(sample); var = np.empty_like(sample)
        ys = self.y_scale.std.reshape(1, 2, 1, 1).astype('float64')
        for s0 in range(0, T, batch_size):
            X = self.x_scale.normalize(q[s0:s0 + batch_size].astype('float32'))
            b = X.shape[0]
            x = torch.empty((b, 4, N, N), dtype=torch.float32, device='cuda')
            x[:, :2] = torch.as_tensor(np.ascontiguousarray(X)).cuda()
            z = torch.empty((b, 2, N, N), dtype=torch.float32, device='cuda')
            ssum = torch.zeros((b, 2, N, N), dtype=torch.float64, device='cuda')
            ssq = torch.zeros_like(ssum)

            def draws():                    # M launches back to back; the range guard is read once after them
                ssum.zero_()
                ssq.zero_()
                first = None
                for m in range(M):
                    check(lib.qgx_noise_normal(_ptr(z), 0, b, 2 * N * N, int(seed), s0, m, 0.0, 1.0, _stream()))
                    x[:, 2:] = z
                    y = self._gen.cnn_forward(x)
                    if first is None:
                        first = y.clone()
                    check(lib.qgx_moments_accumulate(_ptr(y), _ptr(ssum), _ptr(ssq), y.numel(), _stream()))
                return first
            first = self._gen.guarded_loop(draws)
            sm, sq = ssum.cpu().numpy(), ssq.cpu().numpy()
            mu = sm / M
            corr = self._mean_correction(X)         # Y += mean_correction; mean += mean_correction (cgan_regression.py:176-179)
            sample[s0:s0 + b] = (first.cpu().numpy() + np.asarray(corr, dtype='float32')).astype('float64') * ys
            mean[s0:s0 + b] = (mu + corr) * ys
            var[s0:s0 + b] = np.maximum(sq - M * mu * mu, 0.0) / max(M - 1, 1) * ys ** 2
        return sample, mean, var


class CGANRegression(_LatentCNN):
    kind = 'gan'
    NET_NAMES = ('G',)

    def __init__(self, regression='None', nx=64, generator='Andrew', folder='model', div=False,
                 hidden_channels=[128, 64, 32, 32, 32, 32, 32], device=0):
        if generator != 'Andrew' or div or list(hidden_channels) != [128, 64, 32, 32, 32, 32, 32]:
            raise NotImplementedError('only generator="Andrew", div=False with the default hidden channels has a device path')
        self._set_regression(regression)
        self.generator, self.nx, self.div = generator, nx, div
        self.hidden_channels = hidden_channels
        # needs G.pt, x_scale.json, y_scale.json (D.pt is training-only), and net_mean.pt with regression != 'None'
        self._load(folder, device)

"""CPU restatement (torch-CPU float32 functional ops) of the generator half:
the 8-layer circular-padded CNN, channel-wise scalers, the three stochastic
parameterizations and the per-step plugin call.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Pinned by tests/golden/*.npz
(vectors produced by importing the reference's own code,
tests/golden/make_golden.py).

Reference followed (paths relative to /root/reference/pyqg_generative):
  tools/cnn_tools.py:79-98     make_block: Conv2d('same', circular) -> ReLU -> BatchNorm2d
  tools/cnn_tools.py:125-176   AndrewCNN: channels [n_in,128,64,32,32,32,32,32,n_out],
                               kernels [5,5,3,3,3,3,3,3]; last block is a bare conv
  tools/cnn_tools.py:524-530   ChannelwiseScaler.normalize / denormalize (std only)
  tools/cnn_tools.py:702-735   apply_function: eval mode (running BN statistics), no grad
  models/cgan_regression.py:133-137,154-162   generate / generate_latent_noise / predict_snapshot
  models/cvae_regression.py:114-118,128-136   same dataflow with the decoder
  models/mean_var_model.py:14-17,102-109      VarCNN softplus; mean + z*sqrt(var)
  models/parameterization.py:23-34            sampler dispatch + per-layer de-mean
"""
import numpy as np
import torch
import torch.nn.functional as F

KERNELS = [5, 5, 3, 3, 3, 3, 3, 3]
HIDDEN = [128, 64, 32, 32, 32, 32, 32]
BN_EPS = 1e-5


class CNNWeights:
    """Flat container of one AndrewCNN's parameters (numpy float32).

    conv_w[i]: (cout, cin, k, k); conv_b[i]: (cout,); for i < 7 BatchNorm
    gamma/beta/running_mean/running_var of size cout.
    """

    def __init__(self, conv_w, conv_b, bn_g, bn_b, bn_m, bn_v):
        self.conv_w, self.conv_b = conv_w, conv_b
        self.bn_g, self.bn_b, self.bn_m, self.bn_v = bn_g, bn_b, bn_m, bn_v

    @property
    def n_in(self):
        return self.conv_w[0].shape[1]

    @property
    def n_out(self):
        return self.conv_w[-1].shape[0]

    @classmethod
    def from_state_dict(cls, sd):
        """From a torch state_dict of AndrewCNN (keys conv.{0,3,6,...}.weight ...)."""
        sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v)
              for k, v in sd.items()}
        cw, cb, g, b, m, v = [], [], [], [], [], []
        for i in range(8):
            base = 3 * i
            cw.append(sd[f'conv.{base}.weight'].astype('float32'))
            cb.append(sd[f'conv.{base}.bias'].astype('float32'))
            if i < 7:
                g.append(sd[f'conv.{base + 2}.weight'].astype('float32'))
                b.append(sd[f'conv.{base + 2}.bias'].astype('float32'))
                m.append(sd[f'conv.{base + 2}.running_mean'].astype('float32'))
                v.append(sd[f'conv.{base + 2}.running_var'].astype('float32'))
        return cls(cw, cb, g, b, m, v)

    def to_npz_dict(self, prefix=''):
        d = {}
        for i in range(8):
            d[f'{prefix}w{i}'] = self.conv_w[i]
            d[f'{prefix}b{i}'] = self.conv_b[i]
            if i < 7:
                d[f'{prefix}g{i}'] = self.bn_g[i]
                d[f'{prefix}be{i}'] = self.bn_b[i]
                d[f'{prefix}m{i}'] = self.bn_m[i]
                d[f'{prefix}v{i}'] = self.bn_v[i]
        return d

    @classmethod
    def from_npz_dict(cls, d, prefix=''):
        cw = [np.asarray(d[f'{prefix}w{i}'], 'float32') for i in range(8)]
        cb = [np.asarray(d[f'{prefix}b{i}'], 'float32') for i in range(8)]
        g = [np.asarray(d[f'{prefix}g{i}'], 'float32') for i in range(7)]
        b = [np.asarray(d[f'{prefix}be{i}'], 'float32') for i in range(7)]
        m = [np.asarray(d[f'{prefix}m{i}'], 'float32') for i in range(7)]
        v = [np.asarray(d[f'{prefix}v{i}'], 'float32') for i in range(7)]
        return cls(cw, cb, g, b, m, v)

    @classmethod
    def synthetic(cls, n_in, n_out, seed=0):
        """Seeded random weights of the AndrewCNN architecture (He-style scale so
        activations stay O(1)); used for throughput runs without fixtures."""
        rs = np.random.RandomState(seed)
        chans = [n_in] + HIDDEN + [n_out]
        cw, cb, g, b, m, v = [], [], [], [], [], []
        for i in range(8):
            cin, cout, k = chans[i], chans[i + 1], KERNELS[i]
            cw.append((rs.randn(cout, cin, k, k) * np.sqrt(2.0 / (cin * k * k))).astype('float32'))
            cb.append((0.1 * rs.randn(cout)).astype('float32'))
            if i < 7:
                g.append((1 + 0.1 * rs.randn(cout)).astype('float32'))
                b.append((0.1 * rs.randn(cout)).astype('float32'))
                m.append((0.5 + 0.1 * rs.randn(cout)).astype('float32'))
                v.append((0.5 + 0.2 * rs.rand(cout)).astype('float32'))
        return cls(cw, cb, g, b, m, v)


def cnn_forward(w: CNNWeights, x: np.ndarray, return_layers=False, dtype='float32'):
    """AndrewCNN.forward in eval mode (cnn_tools.py:164-176 with div=False,
    final_activation='None').  x: (B, n_in, N, N) float32 -> (B, n_out, N, N) float32.
    dtype='float64' evaluates the same float32 parameters in double precision (the ground truth the
    rounding error of every float32 / split-f16 evaluation order is measured against)."""
    td = torch.float64 if dtype == 'float64' else torch.float32
    T = lambda a: torch.as_tensor(np.asarray(a)).to(td)
    t = torch.as_tensor(np.ascontiguousarray(x, dtype='float32')).to(td)
    layers = []
    with torch.no_grad():
        for i in range(8):
            k = KERNELS[i]
            p = k // 2
            tp = F.pad(t, (p, p, p, p), mode='circular')
            t = F.conv2d(tp, T(w.conv_w[i]), T(w.conv_b[i]))
            if i < 7:
                t = F.relu(t)
                t = F.batch_norm(t, T(w.bn_m[i]), T(w.bn_v[i]), T(w.bn_g[i]), T(w.bn_b[i]),
                                 training=False, eps=BN_EPS)
            if return_layers:
                layers.append(t.numpy().copy())
    out = t.numpy()
    return (out, layers) if return_layers else out


class ScalerRef:
    """ChannelwiseScaler restricted to normalize/denormalize (cnn_tools.py:524-530)."""

    def __init__(self, std):
        self.std = np.asarray(std, dtype='float32').reshape(1, -1, 1, 1)

    def normalize(self, X):
        return X / self.std

    def denormalize(self, X):
        return X * self.std


def demean(x):
    """parameterization.py:25"""
    return x - x.mean(axis=(1, 2), keepdims=True)


class GeneratorRef:
    """One of 'gan' | 'vae' | 'gz'.  nets: [G] / [decoder] / [net_mean, net_var]; 'gan' / 'vae' with a second net:
    regression != 'None', nets = [G | decoder, net_mean] (cgan_regression.py:59-60, cvae_regression.py:49-50)."""

    def __init__(self, kind, nets, x_std, y_std):
        assert kind in ('gan', 'vae', 'gz')
        self.kind = kind
        self.nets = nets
        self.x_scale = ScalerRef(x_std)
        self.y_scale = ScalerRef(y_std)
        self.n_latent = 2

    def generate_latent_noise(self, ny, nx, rng=None):
        rng = rng if rng is not None else np.random
        if self.kind == 'gz':
            return rng.randn(2, ny, nx)                                  # mean_var_model.py:102-103
        return rng.randn(1, self.n_latent, ny, nx).astype('float32')      # cgan_regression.py:154-155

    def predict_snapshot(self, q, noise):
        """q: (2,N,N) float64 -> S (2,N,N) float64 (NOT yet de-meaned)."""
        X = self.x_scale.normalize(q.astype('float32'))                   # (1,2,N,N) f32
        if self.kind == 'gz':                                             # mean_var_model.py:105-109
            mean = cnn_forward(self.nets[0], X)
            var = F.softplus(torch.as_tensor(cnn_forward(self.nets[1], X))).numpy()
            return self.y_scale.denormalize(mean + noise * var ** 0.5).squeeze().astype('float64')
        Y = cnn_forward(self.nets[0], np.concatenate([X, noise.astype('float32')], axis=1))
        if len(self.nets) == 2:                                           # regression != 'None': cgan_regression.py:160-161
            Y += cnn_forward(self.nets[1], X)
        return self.y_scale.denormalize(Y).squeeze().astype('float64')    # cgan_regression.py:157-162

    def predict_mean_snapshot(self, q, M=100, rng=None, z=None):
        """z: the M latent fields (M, 2, N, N) instead of drawing them from rng"""
        X = self.x_scale.normalize(q.astype('float32'))
        if self.kind == 'gz':                                             # mean_var_model.py:111-115
            return self.y_scale.denormalize(cnn_forward(self.nets[0], X)).squeeze().astype('float64')
        rng = rng if rng is not None else np.random
        XX = np.tile(X, (M, 1, 1, 1))                                      # cgan_regression.py:164-171
        if z is None:
            z = rng.randn(M, self.n_latent, X.shape[2], X.shape[3]).astype('float32')
        Y = cnn_forward(self.nets[0], np.concatenate([XX, z], axis=1)).mean(0, keepdims=True)
        if len(self.nets) == 2:                                           # cgan_regression.py:169-170
            Y += cnn_forward(self.nets[1], X)
        return self.y_scale.denormalize(Y).squeeze().astype('float64')


class ParameterizationRef:
    """Per-step plugin call (models/parameterization.py:23-34) bound to a sampler.

    Called by qg_ref.QGModelRef as ``param(m)``; ``m`` must carry
    ``sampling_type`` and ``noise_sampler`` (tools/stochastic_pyqg.py:74-88).
    """

    def __init__(self, gen: GeneratorRef, rng=None, weight=1.0):
        self.gen = gen
        self.rng = rng
        self.weight = weight

    def __call__(self, m):
        latent_noise = lambda: self.gen.generate_latent_noise(m.ny, m.nx, self.rng)
        if m.sampling_type == 'deterministic':
            m.PV_forcing = demean(self.gen.predict_mean_snapshot(m.q, rng=self.rng))
        else:
            if m.noise_sampler.update(latent_noise):
                noise = m.noise_sampler.noise
                m.PV_forcing = demean(self.gen.predict_snapshot(m.q, noise))
        return self.weight * m.PV_forcing

"""Host-side helpers of the generator path (reference: pyqg_generative/tools/cnn_tools.py):
ChannelwiseScaler (:502-553) and apply_function (:702-735), the latter executing on the
MI355X through qgx_cnn_forward instead of PyTorch modules."""
import ast
import json
import numpy as np
import torch


class ChannelwiseScaler:
    """Per-channel std scaling; only the inference methods of the reference class."""

    def __init__(self, std=None, mean=None):
        if std is not None:
            self.std = np.asarray(std, dtype='float32').reshape(1, -1, 1, 1)
            self.mean = np.zeros_like(self.std) if mean is None else np.asarray(mean, 'float32').reshape(1, -1, 1, 1)

    def normalize(self, X):
        return X / self.std

    def denormalize(self, X):
        return X * self.std

    def normalize_var(self, X):
        return X / (self.std ** 2)

    def denormalize_var(self, X):
        return X * (self.std ** 2)

    def direct(self, X):
        return (X - self.mean) / self.std

    def inverse(self, X):
        return X * self.std + self.mean

    def read(self, name, folder='model'):
        with open(f'{folder}/{name}') as f:
            d = json.load(f)
        self.std = np.array(ast.literal_eval(d['std'])).astype('float32')
        self.mean = np.array(ast.literal_eval(d['mean'])).astype('float32')
        return self

    def write(self, name, folder='model'):
        with open(f'{folder}/{name}', 'w') as f:
            json.dump(dict(mean=str(self.mean.tolist()), std=str(self.std.tolist())), f)


class DeviceNet:
    """What the reference's model objects hold as ``G`` / ``decoder`` / ``net_mean`` / ``net_var`` (torch modules built by
    AndrewCNN, cnn_tools.py:125-176): here a handle on net ``inet`` of a device-resident generator.  Callable on a CUDA
    float32 tensor (B, C_in, Ny, Nx) -> (B, 2, Ny, Nx) in eval mode (running BatchNorm statistics); ``to`` / ``eval`` /
    ``train`` exist because the reference's apply_function calls them (cnn_tools.py:710-712,725) — the weights already
    live on the GPU and there is no training mode."""

    def __init__(self, generator, inet=0):
        self.generator, self.inet = generator, int(inet)

    def forward(self, x):
        x = x.to(device='cuda', dtype=torch.float32).contiguous()
        return self.generator.cnn_forward(x, inet=self.inet)

    __call__ = forward

    def to(self, device=None):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self


def minibatch(*arrays, batch_size=64, shuffle=True):
    """Batches of torch tensors (batch_size, C, Ny, Nx) cut from equally long numpy arrays (reference:
    cnn_tools.py:607-622); shuffle draws the order from numpy's global stream as the reference does."""
    n = len(arrays[0])
    if any(len(a) != n for a in arrays):
        raise AssertionError('arrays of different length')
    order = np.arange(n)
    if shuffle:
        np.random.shuffle(order)
    for s in range(0, n, batch_size):
        idx = order[s:s + batch_size]
        yield tuple(torch.as_tensor(np.ascontiguousarray(a[idx])) for a in arrays)


def apply_function(net, *X, fun=None, batch_size=64, inet=0, **kw):
    """The reference's call form (cnn_tools.py:702-735): ``apply_function(net, *X, fun=None, batch_size=64, **kw)``.
    X: numpy arrays (Nbatch, C_i, Ny, Nx); they are cut into batches, moved to the GPU and handed to ``fun(*x, **kw)`` —
    a function of device tensors that is pointwise in the batch dimension, default ``net.forward`` — whose output tensor
    (or tuple of tensors) is brought back and stacked along the batch axis.  Returns one array, or a list of arrays if
    ``fun`` returns several.  ``net`` is a DeviceNet (what the model objects of this package hold as G / decoder /
    net_mean / net_var) or, for convenience, a device Generator (then net ``inet`` of it).  With ``fun=None`` and several
    arrays in X they are concatenated along the channel axis first (the generator nets take [q, z] as one tensor)."""
    if not isinstance(net, DeviceNet):
        net = DeviceNet(net, inet)
    net.to('cuda')
    net.eval()
    if fun is None:
        fun = lambda *x: net.forward(x[0] if len(x) == 1 else torch.cat(x, dim=1))
    arrays = [np.asarray(a, dtype='float32') for a in X]

    def batches():
        preds = []
        for x in minibatch(*arrays, batch_size=batch_size, shuffle=False):
            with torch.no_grad():
                y = fun(*[t.cuda() for t in x], **kw)
            y = y if isinstance(y, tuple) else (y,)
            preds.append([t.cpu().numpy() for t in y])
        return preds
    preds = net.generator.guarded_loop(batches)       # one f16x3 range check for the whole loop, not one per launch
    net.train()
    outs = [np.vstack(col) for col in zip(*preds)]
    return outs[0] if len(outs) == 1 else outs

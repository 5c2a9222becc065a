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


def apply_function(generator, *X, inet=0, batch_size=64):
    """Batched eval-mode forward of net `inet` of a device generator on host arrays.
    X: float32 arrays (Nbatch, C_i, Ny, Nx), concatenated along channels."""
    x = np.concatenate([np.asarray(a, dtype='float32') for a in X], axis=1)
    out = []
    for s in range(0, len(x), batch_size):
        xb = torch.as_tensor(np.ascontiguousarray(x[s:s + batch_size])).cuda()
        out.append(generator.cnn_forward(xb, inet=inet).cpu().numpy())
    return np.vstack(out)

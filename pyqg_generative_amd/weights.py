"""Loading generator weights into the layout the C ABI takes (qgx_cnn_weights).

Sources: the reference's on-disk model folders (``*.pt`` state dicts +
``x_scale.json``/``y_scale.json``, written by pyqg_generative/tools/cnn_tools.py:543-553
and models/*.save_model) or this repo's flat ``.npz`` fixtures.
"""
import json
import os
import numpy as np


def net_from_state_dict(sd):
    """AndrewCNN state_dict (keys conv.{3i}.weight/bias, conv.{3i+2}.* BatchNorm) -> dict."""
    get = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], 'detach') else sd[k],
                               dtype=np.float32)
    net = dict(conv_w=[], conv_b=[], bn_g=[], bn_b=[], bn_m=[], bn_v=[])
    for i in range(8):
        net['conv_w'].append(get(f'conv.{3 * i}.weight'))
        net['conv_b'].append(get(f'conv.{3 * i}.bias'))
        if i < 7:
            net['bn_g'].append(get(f'conv.{3 * i + 2}.weight'))
            net['bn_b'].append(get(f'conv.{3 * i + 2}.bias'))
            net['bn_m'].append(get(f'conv.{3 * i + 2}.running_mean'))
            net['bn_v'].append(get(f'conv.{3 * i + 2}.running_var'))
    return net


def net_from_npz(d, prefix):
    return dict(conv_w=[np.asarray(d[f'{prefix}w{i}'], np.float32) for i in range(8)],
                conv_b=[np.asarray(d[f'{prefix}b{i}'], np.float32) for i in range(8)],
                bn_g=[np.asarray(d[f'{prefix}g{i}'], np.float32) for i in range(7)],
                bn_b=[np.asarray(d[f'{prefix}be{i}'], np.float32) for i in range(7)],
                bn_m=[np.asarray(d[f'{prefix}m{i}'], np.float32) for i in range(7)],
                bn_v=[np.asarray(d[f'{prefix}v{i}'], np.float32) for i in range(7)])


def load_npz(path, kind, regression_npz=None):
    """-> (nets, x_std, y_std) from a flat fixture written by tests/golden/make_golden.py.  regression_npz: 'gan' / 'vae'
    with a regression net (regression != 'None'), read from net0_ of that second fixture."""
    d = np.load(path, allow_pickle=False)
    nets = [net_from_npz(d, 'net0_')]
    if kind == 'gz':
        nets.append(net_from_npz(d, 'net1_'))
    elif regression_npz is not None:
        nets.append(net_from_npz(np.load(regression_npz, allow_pickle=False), 'net0_'))
    return nets, np.asarray(d['x_std'], np.float32), np.asarray(d['y_std'], np.float32)


def read_scaler_std(path):
    """ChannelwiseScaler.read (cnn_tools.py:547-553): JSON of stringified nested lists."""
    import ast
    with open(path) as f:
        d = json.load(f)
    return np.array(ast.literal_eval(d['std'])).astype('float32').reshape(-1)


def load_folder(folder, kind, regression=False):
    """Reference model folder -> (nets, x_std, y_std).  kind: 'gan' | 'vae' | 'gz'; regression ('gan' / 'vae' trained with
    regression != 'None'): the folder also holds net_mean.pt (cgan_regression.py:98-101, cvae_regression.py:75-76)."""
    import torch
    files = {'gan': ['G.pt'], 'vae': ['decoder.pt'], 'gz': ['net_mean.pt', 'net_var.pt']}[kind]
    if regression and kind != 'gz':
        files = files + ['net_mean.pt']
    nets = []
    for f in files:
        sd = torch.load(os.path.join(folder, f), map_location='cpu', weights_only=True)
        nets.append(net_from_state_dict(sd))
    return nets, read_scaler_std(os.path.join(folder, 'x_scale.json')), \
        read_scaler_std(os.path.join(folder, 'y_scale.json'))


def synthetic(kind, seed=0, regression=False):
    """Seeded random weights of the architecture (throughput runs without fixtures)."""
    rs = np.random.RandomState(seed)
    hidden = [128, 64, 32, 32, 32, 32, 32]
    ks = [5, 5, 3, 3, 3, 3, 3, 3]

    def one(n_in):
        ch = [n_in] + hidden + [2]
        net = dict(conv_w=[], conv_b=[], bn_g=[], bn_b=[], bn_m=[], bn_v=[])
        for i in range(8):
            cin, cout, k = ch[i], ch[i + 1], ks[i]
            net['conv_w'].append((rs.randn(cout, cin, k, k) * np.sqrt(2.0 / (cin * k * k))).astype('float32'))
            net['conv_b'].append((0.1 * rs.randn(cout)).astype('float32'))
            if i < 7:
                net['bn_g'].append((1 + 0.1 * rs.randn(cout)).astype('float32'))
                net['bn_b'].append((0.1 * rs.randn(cout)).astype('float32'))
                net['bn_m'].append((0.5 + 0.1 * rs.randn(cout)).astype('float32'))
                net['bn_v'].append((0.5 + 0.2 * rs.rand(cout)).astype('float32'))
        return net
    nets = [one(2), one(2)] if kind == 'gz' else ([one(4), one(2)] if regression else [one(4)])
    x_std = np.array([7.784383342368528e-06, 1.0471941322975908e-06], np.float32)
    y_std = np.array([7.60611105349307e-12, 1.656513061486578e-13], np.float32)
    return nets, x_std, y_std

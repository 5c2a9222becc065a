#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's
own Python from /root/reference (build container only — the reference never
travels to the GPU box; only the .npz data files written here do).

The reference imports ``xarray``, ``pyqg`` and ``gcm_filters`` at module top
(cnn_tools.py:6,10; operators.py:1-4; stochastic_pyqg.py:1); none is installed
and none can be fetched offline.  Inert placeholder modules (empty classes, no
arithmetic) are registered so that the import succeeds; every function
exercised below is one whose arithmetic is entirely the reference's own
numpy/torch code.  Functions that need pyqg's grid arithmetic
(gauss_filter, model_filter, advect, PV_subgrid_forcing) are NOT captured.

Run:  python tests/golden/make_golden.py      (writes *.npz next to this file)
"""
import os
import sys
import types
import json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
COLAB = os.path.join(REF, 'Google-Colab')


def install_inert_stubs():
    xr = types.ModuleType('xarray')

    class _Inert:
        def __init__(self, *a, **k):
            pass
    xr.DataArray = type('DataArray', (_Inert,), {})
    xr.Dataset = type('Dataset', (_Inert,), {})
    sys.modules['xarray'] = xr

    pq = types.ModuleType('pyqg')
    pq.QGModel = type('QGModel', (_Inert,), {})
    pq.Model = type('Model', (_Inert,), {})
    pq.QParameterization = type('QParameterization', (_Inert,), {})
    pq.Parameterization = type('Parameterization', (_Inert,), {})
    pq.__path__ = []                       # simulate.py star-imports pyqg.parameterizations
    sys.modules['pyqg'] = pq
    pqp = types.ModuleType('pyqg.parameterizations')
    pq.parameterizations = pqp
    sys.modules['pyqg.parameterizations'] = pqp
    sys.modules['gcm_filters'] = types.ModuleType('gcm_filters')


def main():
    install_inert_stubs()
    sys.path.insert(0, REF)
    import torch
    torch.set_num_threads(4)
    from pyqg_generative.tools import cnn_tools as ct
    from pyqg_generative.tools import stochastic_pyqg as sp
    from pyqg_generative.tools import operators as op
    from pyqg_generative.tools import spectral_tools as st
    from pyqg_generative.tools import simulate as sim
    from pyqg_generative.tools import parameters as par
    from pyqg_generative.models.cgan_regression import CGANRegression
    from pyqg_generative.models.cvae_regression import CVAERegression
    from pyqg_generative.models.mean_var_model import MeanVarModel

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.gen_ref import CNNWeights

    # ------------------------------------------------------------ models
    tmp = '/tmp/qgx_golden_nomodel'
    gan = CGANRegression(folder=tmp)       # D.pt is a missing blob: load G only
    gan.G.load_state_dict(torch.load(f'{COLAB}/GAN/G.pt', map_location='cpu', weights_only=True))
    gan.x_scale = ct.ChannelwiseScaler().read('x_scale.json', f'{COLAB}/GAN')
    gan.y_scale = ct.ChannelwiseScaler().read('y_scale.json', f'{COLAB}/GAN')
    vae = CVAERegression(folder=f'{COLAB}/VAE')
    gz = MeanVarModel(folder=f'{COLAB}/GZ')
    models = {'gan': gan, 'vae': vae, 'gz': gz}

    # G3: weights in the build's own flat layout
    np.savez_compressed(os.path.join(HERE, 'weights_gan.npz'),
                        x_std=gan.x_scale.std.reshape(-1), y_std=gan.y_scale.std.reshape(-1),
                        **CNNWeights.from_state_dict(gan.G.state_dict()).to_npz_dict('net0_'))
    np.savez_compressed(os.path.join(HERE, 'weights_vae.npz'),
                        x_std=vae.x_scale.std.reshape(-1), y_std=vae.y_scale.std.reshape(-1),
                        **CNNWeights.from_state_dict(vae.decoder.state_dict()).to_npz_dict('net0_'))
    np.savez_compressed(os.path.join(HERE, 'weights_gz.npz'),
                        x_std=gz.x_scale.std.reshape(-1), y_std=gz.y_scale.std.reshape(-1),
                        **CNNWeights.from_state_dict(gz.net_mean.state_dict()).to_npz_dict('net0_'),
                        **CNNWeights.from_state_dict(gz.net_var.state_dict()).to_npz_dict('net1_'))

    # G1: Parameterization.__call__ on seeded q, z for each model and resolution
    class _M:  # the attributes parameterization.py:23-34 and predict_snapshot touch
        pass
    g1 = {}
    for name, model in models.items():
        for N in (48, 64, 96):
            rs = np.random.RandomState(1000 + N)
            x_std = model.x_scale.std.reshape(2, 1, 1).astype('float64')
            q = (rs.randn(2, N, N) * x_std).astype('float32').astype('float64')
            if name == 'gz':
                z = rs.randn(2, N, N)
            else:
                z = rs.randn(1, 2, N, N).astype('float32')
            m = _M()
            m.q, m.nx, m.ny = q, N, N
            m.sampling_type = 'AR1'
            m.noise_sampler = sp.AR1_sampler(1)
            model.generate_latent_noise = (lambda zz: (lambda ny, nx: zz))(z)
            S = model(m)
            assert S.shape == (2, N, N) and S.dtype == np.float64
            g1[f'{name}_{N}_q'] = q.astype('float32')
            g1[f'{name}_{N}_z'] = z
            g1[f'{name}_{N}_S'] = S
            # raw (before de-mean) output too
            g1[f'{name}_{N}_Sraw'] = model.predict_snapshot(m, z)
    np.savez_compressed(os.path.join(HERE, 'generator.npz'), **g1)

    # G2: per-layer activations of AndrewCNN (eval mode) on a small input
    g2 = {}
    rs = np.random.RandomState(7)
    x = rs.randn(2, 4, 16, 16).astype('float32')
    gan.G.eval()
    t = torch.as_tensor(x)
    with torch.no_grad():
        for idx, layer in enumerate(gan.G.conv):
            t = layer(t)
            nxt = gan.G.conv[idx + 1] if idx + 1 < len(gan.G.conv) else None
            # record the output of every conv block (after BN for blocks 0..6)
            if nxt is None or isinstance(nxt, torch.nn.Conv2d):
                g2[f'act{len(g2)}'] = t.numpy().copy()
    g2['x'] = x
    np.savez_compressed(os.path.join(HERE, 'layers.npz'), **g2)

    # G4: sampler sequences with a deterministic generator
    g4 = {}
    for kind, cls, ns in (('ar1', sp.AR1_sampler, (1, 5, -1)), ('const', sp.constant_sampler, (1, 3))):
        for n in ns:
            rs = np.random.RandomState(11)
            s = cls(n)
            seq, flags = [], []
            for _ in range(8):
                flags.append(bool(s.update(lambda: rs.randn(6))))
                seq.append(np.array(s.noise, copy=True))
            g4[f'{kind}_{n}_seq'] = np.stack(seq)
            g4[f'{kind}_{n}_flags'] = np.array(flags)
    np.savez_compressed(os.path.join(HERE, 'samplers.npz'), **g4)

    # G5: numpy-only coarse-graining / re-gridding operators
    g5 = {}
    rs = np.random.RandomState(3)
    X = rs.randn(2, 64, 64)
    g5['X'] = X
    g5['cut_off_32'] = op.cut_off(X, 32)
    g5['cut_off_48'] = op.cut_off(X, 48)
    g5['clean_2h'] = op.clean_2h(X)
    g5['interp_64_96'] = op.fft_interpolate(X, 64, 96)
    g5['interp_64_32'] = op.fft_interpolate(X, 64, 32)
    g5['interp_64_96_keep2h'] = op.fft_interpolate(X, 64, 96, truncate_2h=False)
    g5['op5_32'] = op.Operator5(X, 32)
    np.savez_compressed(os.path.join(HERE, 'operators.npz'), **g5)

    # G6: isotropic spectrum with a duck-typed grid
    g6 = {}
    for N in (48, 64):
        L = 1e6
        g = _M()
        g.dk = g.dl = 2 * np.pi / L
        g.ll = g.dl * np.append(np.arange(0., N / 2), np.arange(-N / 2, 0.))
        g.kk = g.dk * np.arange(0., N // 2 + 1)
        kx, ly = np.meshgrid(g.kk, g.ll)
        g.wv = np.sqrt(kx ** 2 + ly ** 2)
        rs = np.random.RandomState(N)
        dens = rs.rand(N, N // 2 + 1) ** 2
        g6[f'dens_{N}'] = dens
        for av in (True, False):
            for tr in (True, False):
                kr, ph = st.calc_ispec(g, dens, averaging=av, truncate=tr)
                g6[f'kr_{N}_{int(av)}{int(tr)}'] = kr
                g6[f'ph_{N}_{int(av)}{int(tr)}'] = ph
    np.savez_compressed(os.path.join(HERE, 'ispec.npz'), **g6)

    # G7: initial condition with a fixed seed and a duck-typed model
    g7 = {}
    for N in (48, 64, 96):
        captured = {}
        m = _M()
        m.nx = m.ny = N
        m.L = 1e6
        dk = 2 * np.pi / m.L
        ll = dk * np.append(np.arange(0., N / 2), np.arange(-N / 2, 0.))
        kk = dk * np.arange(0., N // 2 + 1)
        kx, ly = np.meshgrid(kk, ll)
        m.wv = np.sqrt(kx ** 2 + ly ** 2)
        m.x = np.zeros((N, N))
        m.set_q1q2 = lambda a, b: captured.update(q1=np.array(a), q2=np.array(b))
        m._invert = lambda: None
        np.random.seed(N)
        sim.set_initial_condition(m)
        g7[f'q1_{N}'] = captured['q1']
        g7[f'q2_{N}'] = captured['q2']
    np.savez_compressed(os.path.join(HERE, 'initial_condition.npz'), **g7)

    # constants
    consts = dict(YEAR=par.YEAR, DAY=par.DAY, ANDREW_1000_STEPS=par.ANDREW_1000_STEPS,
                  EDDY=dict(par.EDDY_PARAMS), JET=dict(par.JET_PARAMS),
                  dt={str(n): par.EDDY_PARAMS.nx(n)['dt'] for n in (32, 48, 64, 96, 128, 256, 512)})
    with open(os.path.join(HERE, 'parameters.json'), 'w') as f:
        json.dump(consts, f, indent=1, sort_keys=True)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()

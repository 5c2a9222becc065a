"""GPU: the generator's 16-bit matrix-core paths against the float32 path and a float64 ground truth.

The CNN's parameters and inputs are float32; `oracle.gen_ref.cnn_forward(dtype='float64')` evaluates
them in double precision, so |result - truth| is the rounding error of an evaluation order.
  * precision 0: exact f32 MFMA (v_mfma_f32_32x32x2_f32)
  * precision 3: f16 hi/lo split, three f16 MFMAs per product, f32 accumulate — must stay in the
    float32 error class: it has to pass the SAME 2e-5 golden-vector tolerance as precision 0 and
    its error against the float64 truth may not exceed 4x that of the reference's own float32
    evaluation (torch CPU) on the same input
  * precision 1: plain f16 operands (TF32-class, 2^-11 operand rounding): 1e-2 of the field maximum; a mode of the
    A/B library only (test_ab_library_variants_agree)
"""
import os
import sys
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from conftest import golden, GOLDEN
from oracle import gen_ref


def _gpu_generator(kind):
    import pyqg_generative_amd as qa
    from pyqg_generative_amd import weights
    nets, xs, ys = weights.load_npz(os.path.join(GOLDEN, f'weights_{kind}.npz'), kind)
    return qa.Generator(kind, nets, xs, ys)


def _oracle_nets(kind):
    d = golden(f'weights_{kind}.npz')
    nets = [gen_ref.CNNWeights.from_npz_dict(d, 'net0_')]
    if kind == 'gz':
        nets.append(gen_ref.CNNWeights.from_npz_dict(d, 'net1_'))
    return nets


def _maxrel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize('kind,N,B', [('gan', 64, 32), ('vae', 96, 24), ('gz', 48, 48), ('gan', 32, 32), ('gan', 128, 16),
                                      ('gz', 64, 16), ('vae', 64, 16), ('gan', 64, 2)])
def test_split_f16_is_float32_class(kind, N, B):
    gen = _gpu_generator(kind)
    nets = _oracle_nets(kind)
    rs = np.random.RandomState(N + B)
    n_in = nets[0].n_in
    x = rs.randn(B, n_in, N, N).astype('float32')
    x[:, :2] *= 1.5                                   # q / x_std is O(1) with a heavier tail than z
    xd = torch.as_tensor(x, device='cuda')
    for inet, w in enumerate(nets):
        truth = gen_ref.cnn_forward(w, x[:4], dtype='float64')
        ref32 = gen_ref.cnn_forward(w, x[:4])
        err_ref = _maxrel(ref32, truth)
        errs = {}
        gen.set_option('wino', 0)                     # the 25-tap 5x5 layer: the float32-class claim is made for THIS form
        for prec in (0, 3):
            gen.set_option('precision', prec)
            y = gen.cnn_forward(xd, inet).cpu().numpy()
            errs[prec] = _maxrel(y[:4], truth)
        gen.set_option('auto', 0)                     # what calibration chose (64 x 64: possibly the Winograd 5x5 layer)
        errs['default'] = _maxrel(gen.cnn_forward(xd, inet).cpu().numpy()[:4], truth)
        gen.set_option('precision', 0)
        print(f'\n{kind} net{inet} N={N}: max err / max|y| vs float64 truth: torch-f32 {err_ref:.2e}, '
              f'f32 MFMA {errs[0]:.2e}, f16x3 {errs[3]:.2e}, default path {errs["default"]:.2e} {gen.wino_info()}')
        assert errs[0] < 2e-5
        assert errs[3] < 2e-5
        assert errs[3] < 4 * max(err_ref, errs[0]) + 1e-7
        assert errs['default'] < 2e-5


@pytest.mark.parametrize('kind,B', [('gan', 1), ('vae', 3), ('gz', 5)])
def test_single_member_strips_are_float32_class(kind, B):
    """Tiny ensembles at 64 x 64 (the split-K path of a single member, BASELINE configs[1]): layers (3, 4), (5, 6) and (7, 8)
    each run as ONE fused launch on 2-row strips (option `tiny_pairs`, bits 2 / 1 / 0; default all) instead of split-K +
    combine + four layer launches + the VALU last layer.  Every combination stays in the float32 error class against the
    float64 evaluation of the reference's arithmetic (cnn_tools.py:79-98,125-176), and within 1e-5 of the unfused path."""
    gen = _gpu_generator(kind)
    nets = _oracle_nets(kind)
    rs = np.random.RandomState(40 + B)
    x = rs.randn(B, nets[0].n_in, 64, 64).astype('float32')
    x[:, :2] *= 1.5
    xd = torch.as_tensor(x, device='cuda')
    for inet, w in enumerate(nets):
        truth = gen_ref.cnn_forward(w, x, dtype='float64')
        err_ref = _maxrel(gen_ref.cnn_forward(w, x), truth)
        ys = {}
        for tp in (0, 1, 2, 4, 7):
            gen.set_option('tiny_pairs', tp)
            ys[tp] = gen.cnn_forward(xd, inet).cpu().numpy()
            assert gen.layer2_kernel(B, 64, inet) == 2          # the split-K path is the one in use
            err = _maxrel(ys[tp], truth)
            assert err < 4 * err_ref + 1e-7 and err < 2e-5, (tp, err, err_ref)
            assert _maxrel(ys[tp], ys[0]) < 1e-5
        assert not np.array_equal(ys[7], ys[0])                  # (the strips are another order of summation: they did run)
        gen.set_option('tiny_pairs', 7)


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
def test_split_f16_golden_vectors(kind):
    """the reference's own outputs (Parameterization.__call__, tests/golden/generator.npz) at the
    float32 tolerance, with the ensemble padded by random members so that the 16-bit path engages"""
    d = golden('generator.npz')
    gen = _gpu_generator(kind)
    gen.set_option('precision', 3)
    for N in (48, 64, 96):
        q = d[f'{kind}_{N}_q'].astype('float64')
        z = d[f'{kind}_{N}_z']
        S_ref = d[f'{kind}_{N}_S']
        B = 40
        rs = np.random.RandomState(N)
        qb = np.concatenate([q[None], rs.randn(B - 1, 2, N, N) * np.abs(q).max() / 3], 0)
        if kind == 'gz':
            zb = np.concatenate([z.reshape(1, 2, N, N), rs.randn(B - 1, 2, N, N)], 0).astype('float64')
        else:
            zb = np.concatenate([z.reshape(1, 2, N, N), rs.randn(B - 1, 2, N, N)], 0).astype('float32')
        S = gen.forward(torch.as_tensor(qb, device='cuda'), torch.as_tensor(zb, device='cuda'), demean=True)
        S = S.cpu().numpy()[0]
        err = (np.abs(S - S_ref) / np.abs(S_ref).max(axis=(1, 2), keepdims=True)).max()
        print(f'\n{kind} N={N}: f16x3 vs golden {err:.2e}')
        assert err < 2e-5


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
@pytest.mark.parametrize('fold', [1, 0])
def test_winograd_5x5_layer_is_float32_class(kind, fold):
    """64 x 64: the 5x5 layer as a 1-D Toom-Cook / Winograd convolution F(4, 5) along x (k_convw: 0.4 x the MFMAs of the
    25-tap form, input transform in float32, f16x3 split after it) against the float64 truth of the same float32
    parameters: inside 2e-5 of max|y| (the tolerance of the golden vectors; the 25-tap form beside it stays within 4 x the
    error of the reference's own float32 evaluation — the Toom-Cook matrices cost the Winograd form a factor 3-8 on these
    nets, which is why calibration has to admit it per generator); and against the reference-generated golden vectors"""
    gen = _gpu_generator(kind)
    info = gen.wino_info()
    print(kind, info)
    assert info['N'] == 64 and info['calibration_error'] > 0 and info['enabled'] == (info['calibration_error'] <= 1e-5)
    nets = _oracle_nets(kind)
    N, B = 64, 8
    rs = np.random.RandomState(11)
    n_in = nets[0].n_in
    x = rs.randn(B, n_in, N, N).astype('float32')
    x[:, :2] *= 1.5
    xd = torch.as_tensor(x, device='cuda')
    gen.set_option('fold', fold)
    for inet, w in enumerate(nets):
        truth = gen_ref.cnn_forward(w, x[:4], dtype='float64')
        err_ref = _maxrel(gen_ref.cnn_forward(w, x[:4]), truth)
        gen.set_option('precision', 0)
        err_f32 = _maxrel(gen.cnn_forward(xd, inet).cpu().numpy()[:4], truth)
        gen.set_option('precision', 3)
        errs = {}
        for name, opts in (('25-tap', dict(wino=0)), ('winograd', dict(wino=1, wino_min_tiles=1))):
            for k, v in opts.items():
                gen.set_option(k, v)
            gen.set_option('part_max_tiles', 0)
            errs[name] = _maxrel(gen.cnn_forward(xd, inet).cpu().numpy()[:4], truth)
        print(f'\n{kind} net{inet} fold={fold}: vs float64 truth: torch-f32 {err_ref:.2e}, f32 MFMA {err_f32:.2e}, '
              f'f16x3 25-tap {errs["25-tap"]:.2e}, f16x3 winograd {errs["winograd"]:.2e}')
        assert errs['25-tap'] < 4 * max(err_ref, err_f32) + 1e-7        # float32 class
        assert errs['winograd'] < 2e-5                                   # inside the golden-vector tolerance (measured: 3-9e-6)
        assert gen.range_ok() is None
    if fold:
        d = golden('generator.npz')
        q = d[f'{kind}_64_q'].astype('float64')
        z = d[f'{kind}_64_z']
        zt = torch.as_tensor(z.reshape(1, 2, N, N).astype('float64' if kind == 'gz' else 'float32'), device='cuda')
        S = gen.forward(torch.as_tensor(q[None], device='cuda'), zt, demean=True).cpu().numpy()[0]     # one member: 8 tiles
        S_ref = d[f'{kind}_64_S']
        assert (np.abs(S - S_ref) / np.abs(S_ref).max(axis=(1, 2), keepdims=True)).max() < 2e-5


@pytest.mark.parametrize('kind', ['gan', 'vae', 'gz'])
@pytest.mark.parametrize('N,B', [(32, 32), (48, 8), (96, 4), (128, 2)])
def test_winograd_5x5_layer_at_the_other_grid_sizes(kind, N, B):
    """the Winograd layer's other tile shapes (16 x 32 / 8 x 32 tiles at 32 x 32, 16 x 16 at 48 x 48, 12 / 16 x 32 at 96 x 96, 8 / 4 x
    64 at 128 x 128) against the float64 truth of the same float32 parameters, and the admission made AT that size: calibration
    measured each size on inputs of that size (`wino_info(N)`), the default path obeys it, and forced on the form stays inside
    the golden-vector tolerance for the shipped nets"""
    gen = _gpu_generator(kind)
    info = gen.wino_info(N)
    print(kind, info)
    assert info['N'] == N and info['calibration_error'] > 0 and info['enabled'] == (info['calibration_error'] <= 1e-5)
    nets = _oracle_nets(kind)
    rs = np.random.RandomState(5 + N)
    x = rs.randn(B, nets[0].n_in, N, N).astype('float32')
    x[:, :2] *= 1.5
    xd = torch.as_tensor(x, device='cuda')
    nb = min(B, 2)
    for inet, w in enumerate(nets):
        truth = gen_ref.cnn_forward(w, x[:nb], dtype='float64')
        err_ref = _maxrel(gen_ref.cnn_forward(w, x[:nb]), truth)
        errs = {}
        for name, opts in (('25-tap', dict(wino=0)), ('winograd', dict(wino=1, wino_min_tiles=1)), ('default', dict(wino=2, wino_min_tiles=1))):
            for k, v in opts.items():
                gen.set_option(k, v)
            gen.set_option('part_max_tiles', 0)
            errs[name] = _maxrel(gen.cnn_forward(xd, inet).cpu().numpy()[:nb], truth)
            kk = gen.layer2_kernel(B, N, inet)
            assert (kk >= 3) == (name == 'winograd' or (name == 'default' and info['enabled'])), (name, kk)
        print(f'\n{kind} net{inet} {N}x{N}: vs float64 truth: torch-f32 {err_ref:.2e}, f16x3 25-tap {errs["25-tap"]:.2e}, '
              f'winograd {errs["winograd"]:.2e}, default {errs["default"]:.2e}')
        assert errs['25-tap'] < 4 * err_ref + 1e-7
        assert errs['winograd'] < 2e-5 and errs['default'] < 2e-5
        assert gen.range_ok() is None


def test_winograd_layer_with_the_transform_under_the_mfmas_is_bit_identical():
    """k_convw2 (conv_wino2.hpp: the eight positions as two teams of waves in ping-pong, the input transform under the MFMAs)
    against k_convw (option wino2 = 0): the same arithmetic in the same order — bit-identical outputs at the sizes it is
    built for, full tiles and a ragged tile count"""
    gen = _gpu_generator('gan')
    gen.set_option('wino', 1)
    gen.set_option('wino_min_tiles', 1)
    for N, B, opts in ((64, 128, {}), (64, 37, dict(wino_rows64=8)), (96, 32, dict(wino_rows96=12)), (96, 11, dict(wino_rows96=12))):
        for k, v in opts.items():
            gen.set_option(k, v)
        rs = np.random.RandomState(N + B)
        xd = torch.as_tensor(rs.randn(B, 4, N, N).astype('float32'), device='cuda')
        out = {}
        for w2 in (0, 1):
            gen.set_option('wino2', w2)
            assert gen.layer2_kernel(B, N) == (4 if w2 else 3), (N, B, w2, gen.layer2_kernel(B, N))
            out[w2] = gen.cnn_forward(xd).clone()
        assert torch.equal(out[0], out[1]), (N, B)
        gen.set_option('wino_rows64', 0)
        gen.set_option('wino_rows96', 0)
    assert gen.range_ok() is None


VARIANTS = [                                     # selectable variants of the product library
    dict(part_max_tiles=100000),                 # split-K on the wide layers (single-member path)
    dict(pair=0, fuse=0),                        # k_convh2 without the line-pair fetch
    dict(fuse=7),                                # every 3x3 pair fused (k_convh_pair)
    dict(fuse=2), dict(fuse=0),
    dict(h2_w8=0), dict(h2_w8=1),                # 4-wave x 2 workgroups per CU instead of one 8-wave workgroup
    dict(h2_x96=0), dict(h2_w8_min96=1),         # 96 x 96: 4-wave full-row 3x3 kernels / x-tiled 8-wave 5x5 kernel at any size
    dict(first_h=0),                             # exact-f32 first layer writing the 16-bit layout
    dict(fold=0),                                # layer 1's BatchNorm applied in its epilogue instead of folded into layer 2
    dict(wino=0),                                # 5x5 layer as the 25-tap implicit GEMM instead of the 1-D Winograd form
    dict(wino2=0),                               # the Winograd layer as k_convw instead of k_convw2 (transform under the MFMAs)
    dict(wino_min_tiles=1),                      # ... the Winograd form at every ensemble size (64 x 64)
    dict(wino_min_tiles=1, fold=0),
    dict(wino_rows64=4), dict(wino_rows64=8),    # tile shapes the launchers otherwise choose by tile-count quantisation
    dict(wino_rows96=12), dict(wino_rows96=16), dict(h2_rows96=12), dict(h2_rows96=16),
    dict(fuse96=0), dict(fuse96=3),              # 96 x 96: no pair fused / both pairs (default: layers 7 + 8)
    dict(small_tiles=0),                         # 64 x 64, at most 4 members: the regular 4-row tiles
    dict(member_chunk=16),                       # member sub-batches
    dict(ascale_log2=3), dict(ascale_log2=-2),   # another activation pre-scale inside the window
]
AB_VARIANTS = [                                  # kernels of the A/B library only (libqgx_ab.so, `make ab`)
    dict(h2=0, half_nw=8, res=0, fuse=0),        # generic run-time-N kernel k_convh, 8 waves
    dict(h2=0, half_nw=4, res=0, fuse=0),        # ... 4 waves, swizzled 64-byte patch pixels
    dict(h2=0, half_nw=8, res=1, fuse=0),        # resident-weight 3x3 kernel k_convh_res
    dict(h3=1),                                  # 5x5 layer on 16x16x32 MFMAs (k_convh3)
    dict(h4=1), dict(h4=2),                      # 5x5 layer with full-line patch chunks (k_convh4)
]
DEFAULTS = dict(fuse=3, pair=1, first_h=1, member_chunk=0, part_max_tiles=0, fold=1, h2_w8=3, ascale_log2=0, h2_x96=1, h2_w8_min96=1024,
                wino=2, wino2=1, wino_min_tiles=48, wino_rows64=0, wino_rows96=0, h2_rows96=0, fuse96=2, small_tiles=1)
AB_DEFAULTS = dict(DEFAULTS, h2=3, half_nw=8, res=1, h3=0, h4=0)


def _variant_errors(gen, x, variants, defaults):
    gen.check_range = False
    gen.set_option('precision', 0)
    ref = gen.cnn_forward(x).cpu().numpy()
    gen.set_option('precision', 3)
    errs = []
    for v in variants:
        for k, d in defaults.items():
            gen.set_option(k, v.get(k, d))
        errs.append(_maxrel(gen.cnn_forward(x).cpu().numpy(), ref))
    return ref, errs


@pytest.mark.parametrize('N,B', [(64, 32), (96, 8), (96, 32), (48, 16), (64, 3)])
def test_optional_kernel_variants_agree(N, B):
    """every selectable f16x3 kernel variant of the product library against the exact-f32 path: float32
    tolerance (2e-5 of the maximum); options that do not apply to a grid size fall back to the default kernels"""
    from pyqg_generative_amd import _lib
    gen = _gpu_generator('gan')
    rs = np.random.RandomState(7 * N + B)
    x = torch.as_tensor(rs.randn(B, 4, N, N).astype('float32'), device='cuda')
    _, errs = _variant_errors(gen, x, VARIANTS, DEFAULTS)
    for v, err in zip(VARIANTS, errs):
        assert err < 2e-5, (v, err)
    assert gen.range_ok() is None
    if b'+ab' not in _lib.lib.qgx_version():
        # the product library refuses options that select A/B-only kernels instead of silently ignoring them
        with pytest.raises(_lib.QgxError):
            gen.set_option('h3', 1)
        with pytest.raises(_lib.QgxError):
            gen.set_option('precision', 1)


@pytest.mark.parametrize('N,B,opts', [(64, 16, dict(wino_rows64=[4, 8])), (64, 48, dict(wino_rows64=[4, 8])),
                                      (96, 32, dict(wino_rows96=[12, 16], h2_rows96=[12, 16])), (96, 12, dict(wino_rows96=[12, 16], h2_rows96=[12, 16])),
                                      (32, 96, dict(wino_rows64=[4, 8])), (128, 2, dict(wino_rows64=[4, 8])), (128, 12, dict(wino_rows64=[4, 8])),
                                      (64, 1, dict(small_tiles=[0, 1])), (64, 4, dict(small_tiles=[0, 1]))])
def test_tile_shapes_do_not_change_the_result(N, B, opts):
    """the launchers choose tile shapes per launch by tile-count quantisation (256 persistent workgroups take ceil(tiles / 256)
    rounds): 8- or 4-row tiles for the Winograd layer at 64 x 64, 12- or 16-row tiles at 96 x 96 (the 3x3 layers likewise, on 6 or
    8 waves), half-height tiles for ensembles of at most 4 members.  A tile's shape does not enter any summation order, so the
    network's output is BIT-identical whichever shape runs — a shard and the whole ensemble may pick different ones."""
    gen = _gpu_generator('gan')
    gen.check_range = False
    rs = np.random.RandomState(N + 31 * B)
    x = torch.as_tensor(rs.randn(B, 4, N, N).astype('float32'), device='cuda')
    if N in (32, 128):
        # shapes the launchers pick by themselves only at these ensemble sizes (k_convw<32,32,16> beyond 64 members, k_convw<128,64,4> for
        # 4 members or fewer, or 12): with the Winograd form forced on, and against the 25-tap form at the golden-vector tolerance
        gen.set_option('wino', 1)
        gen.set_option('wino_min_tiles', 1)
    ref = gen.cnn_forward(x).cpu().numpy()                                   # the automatic choice
    for name, values in opts.items():
        for v in values:
            gen.set_option(name, v)
            assert np.array_equal(gen.cnn_forward(x).cpu().numpy(), ref), (name, v)
        gen.set_option(name, 0 if name != 'small_tiles' else 1)
    if N in (32, 128):
        assert gen.layer2_kernel(B, N) == 3
        gen.set_option('wino', 0)
        y25 = gen.cnn_forward(x).cpu().numpy()
        assert np.abs(y25 - ref).max() < 2e-5 * np.abs(y25).max()
    gen.close()


def test_ab_library_variants_agree():
    """the measured-slower kernels kept for A/B timing (k_convh, k_convh_res, k_convh3, k_convh4, plain-f16 mode)
    still compute the same function: run in a child process on libqgx_ab.so (QGX_LIB)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ab = os.path.join(root, 'pyqg_generative_amd', 'libqgx_ab.so')
    if not os.path.exists(ab):
        pytest.skip('libqgx_ab.so is not built (make -C pyqg_generative_amd/csrc ab)')
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'ab-child'], env=dict(os.environ, QGX_LIB=ab),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['version'].endswith('+ab')
    from pyqg_generative_amd import _lib
    # both libraries were built from the same sources (fingerprint in qgx_version, tests/test_abi_cpu.py)
    assert out['version'].split(' +ab')[0] == _lib.lib.qgx_version().decode().split(' +ab')[0]
    for key, errs in out['f16x3'].items():
        assert max(errs) < 2e-5, (key, errs)
    for key, errs in out['f16'].items():
        assert max(errs) < 1e-2, (key, errs)          # plain f16 operands: TF32-class tolerance


def _ab_child():
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyqg_generative_amd import _lib
    out = {'version': _lib.lib.qgx_version().decode(), 'f16x3': {}, 'f16': {}}
    for N, B in ((64, 32), (96, 8), (48, 16)):
        gen = _gpu_generator('gan')
        rs = np.random.RandomState(7 * N + B)
        x = torch.as_tensor(rs.randn(B, 4, N, N).astype('float32'), device='cuda')
        ref, errs = _variant_errors(gen, x, VARIANTS + AB_VARIANTS, AB_DEFAULTS)
        out['f16x3'][f'{N}x{B}'] = errs
        gen.set_option('precision', 1)
        e16 = []
        for nw in (8, 4):
            for k, d in AB_DEFAULTS.items():
                gen.set_option(k, d)
            gen.set_option('half_nw', nw)
            e16.append(_maxrel(gen.cnn_forward(x).cpu().numpy(), ref))
        out['f16'][f'{N}x{B}'] = e16
    print(json.dumps(out))


@pytest.mark.parametrize('kind', ['gan', 'gz'])
def test_full_size_properties(kind):
    """BASELINE's full single-GPU size (128 members, 64 x 64) through size-independent properties of a
    circular convolution stack, on the default kernels (f16x3, fused pairs):
      * members are independent: copies of a member at different positions of the ensemble give
        bit-identical outputs (every tile / workgroup / chunk schedule computes a pixel the same way)
      * translation equivariance: circularly shifting the input shifts the output, bit for bit
        (a pixel's summation order does not depend on where its tile lies) — for shifts in y and for shifts in x by
        whole quads of 4 columns: the 5x5 layer's 1-D Winograd form works on quads of output columns, so other shifts in
        x change the rounding (float32 tolerance there); with the 25-tap form (option wino = 0) every shift is exact"""
    gen = _gpu_generator(kind)
    N, B = 64, 128
    rs = np.random.RandomState(3)
    n_in = 2 if kind == 'gz' else 4
    base = rs.randn(4, n_in, N, N).astype('float32')
    x = torch.as_tensor(np.tile(base, (B // 4, 1, 1, 1)), device='cuda')
    y = gen.cnn_forward(x)
    for r in range(4):
        assert torch.equal(y[r::4], y[r:r + 1].expand(B // 4, -1, -1, -1)), r
    ref = gen_ref.cnn_forward(_oracle_nets(kind)[0], base)
    assert _maxrel(y[:4].cpu().numpy(), ref) < 2e-5
    for dy, dx in ((1, 0), (0, 4), (5, 8), (37, 60), (0, 3), (5, 7)):
        ys = gen.cnn_forward(torch.roll(x, shifts=(dy, dx), dims=(2, 3)))
        if dx % 4 == 0:
            assert torch.equal(ys, torch.roll(y, shifts=(dy, dx), dims=(2, 3))), (dy, dx)
        else:
            assert _maxrel(ys.cpu().numpy(), torch.roll(y, shifts=(dy, dx), dims=(2, 3)).cpu().numpy()) < 2e-5, (dy, dx)
    gen.set_option('wino', 0)
    y = gen.cnn_forward(x)
    for dy, dx in ((0, 3), (37, 61)):
        ys = gen.cnn_forward(torch.roll(x, shifts=(dy, dx), dims=(2, 3)))
        assert torch.equal(ys, torch.roll(y, shifts=(dy, dx), dims=(2, 3))), (dy, dx)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'ab-child':
    _ab_child()

"""Developer tool (A/B library): layer-1 / layer-2 activations of the planar (wino_pl=1) path against the pixel-major one."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from pyqg_generative_amd import _lib
B, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 64
gen, _ = bench.load_generator('gan', 0)
gen.check_range = False
lib = _lib.lib
lib.qgx_debug_read_act.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
rs = np.random.RandomState(1)
x = torch.as_tensor(rs.randn(B, 4, N, N).astype('float32'), device='cuda')
gen.set_option('wino', 1); gen.set_option('wino_min_tiles', 1); gen.set_option('part_max_tiles', 0)
def act(pl, stop, which, ch):
    gen.set_option('wino_pl', pl); gen.set_option('stop_layer', stop)
    gen.cnn_forward(x)
    buf = torch.zeros(B * N * N * ch * 2, dtype=torch.float16, device='cuda')
    assert lib.qgx_debug_read_act(gen._h, which, buf.data_ptr(), buf.numel() * 2, None) == 0
    torch.cuda.synchronize()
    return buf.cpu().numpy().astype(np.float64)
a0 = act(0, 1, 0, 128).reshape(B, N, N, 16, 2, 8)          # [b][y][x][octet][hi|lo][8]
v0 = (a0[..., 0, :] + a0[..., 1, :]).reshape(B, N, N, 128)
a1 = act(1, 1, 0, 128).reshape(B, N, 128, 2, N)            # [b][y][c][hi|lo][x]
v1 = (a1[:, :, :, 0, :] + a1[:, :, :, 1, :]).transpose(0, 1, 3, 2)
d = np.abs(v0 - v1)
print('layer 1: max|pixel-major| %.4g  max diff %.4g' % (np.abs(v0).max(), d.max()))
if d.max() > 1e-6:
    bad = np.argwhere(d > 1e-6)
    print('  mismatches', len(bad), 'of', d.size, 'first', bad[:8].tolist())
    for k, nm in enumerate('byxc'):
        print('   distinct', nm, np.unique(bad[:, k])[:40])
    # does the planar buffer hold the right values somewhere else?
    b, y, x_, c = bad[0]
    tgt = v0[b, y, x_, c]
    where = np.argwhere(np.abs(v1[b] - tgt) < 1e-9 * max(1, abs(tgt)))
    print('   value', tgt, 'found in planar image at (y, x, c):', where[:6].tolist())
b0 = act(0, 2, 1, 64).reshape(B, N, N, 8, 2, 8)
b1 = act(1, 2, 1, 64).reshape(B, N, N, 8, 2, 8)
w0 = (b0[..., 0, :] + b0[..., 1, :]).reshape(B, N, N, 64); w1 = (b1[..., 0, :] + b1[..., 1, :]).reshape(B, N, N, 64)
d2 = np.abs(w0 - w1)
print('layer 2: max|pixel-major path| %.4g  max diff %.4g' % (np.abs(w0).max(), d2.max()))
if d2.max() > 1e-3 * np.abs(w0).max():
    bad = np.argwhere(d2 > 1e-3 * np.abs(w0).max())
    print('  mismatches', len(bad), 'of', d2.size)
    for k, nm in enumerate('byxc'):
        print('   distinct', nm, np.unique(bad[:, k])[:64])
gen.set_option('stop_layer', 0)

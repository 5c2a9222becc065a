#!/usr/bin/env python
"""Developer tool: in-kernel timeline of one 16-bit conv layer (k_convh / k_convh_res) from s_memtime stamps.
    python bench_tools/conv_stamps.py MEMBERS LAYER
Needs the diagnostic library: make -C pyqg_generative_amd/csrc stamps; QGX_LIB=.../libqgx_stamps.so"""
import ctypes as C
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('QGX_LIB', os.path.join(ROOT, 'pyqg_generative_amd', 'libqgx_stamps.so'))
import pyqg_generative_amd as qa
from pyqg_generative_amd import weights, _lib

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 64
LAYER = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nets, xs, ys = weights.load_npz(os.path.join(ROOT, 'tests', 'golden', 'weights_gan.npz'), 'gan')
gen = qa.Generator('gan', nets, xs, ys)
x = torch.randn((B, 4, N, N), dtype=torch.float32, device='cuda')
for _ in range(3):
    gen.cnn_forward(x)
stamps = torch.zeros((512, 64), dtype=torch.int64, device='cuda')
lib = _lib.lib
lib.qgx_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.qgx_debug_set_stamps(gen._h, stamps.data_ptr(), LAYER)
gen.cnn_forward(x)
torch.cuda.synchronize()
s = stamps.cpu().numpy()
# s_memtime counts shader-clock cycles
s = s[s[:, 0] > 0]
for wg in (0, 1, len(s) // 2, len(s) - 1):
    t = s[wg][s[wg] > 0]
    print(f'wg {wg}: {len(t)} stamps over {t[-1] - t[0]} cycles; deltas: ' + ' '.join(str(v) for v in np.diff(t)))
if os.environ.get('QGX_STAMPS_REALTIME'):
    # library built with -DQGX_STAMPS_REALTIME: 100 MHz counter shared by the whole chip
    last = np.array([r[r > 0][-1] for r in s])
    t0 = s[:, 0].min()
    st = (s[:, 0] - t0) * 0.01
    en = (last - t0) * 0.01
    print(f'start after the first workgroup (us): median {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}')
    print(f'end after the first start (us): min {en.min():.1f} median {np.median(en):.1f} max {en.max():.1f}; lifetime median {np.median(en - st):.1f}')
    print('mean end by workgroup index mod 8 (XCD):', ' '.join(f'{en[x::8].mean():.1f}' for x in range(8)))
    print('mean end by index quarter:', ' '.join(f'{q.mean():.1f}' for q in np.array_split(en, 4)))
    print('spread within XCD 0 (min/median/max):', f'{en[0::8].min():.1f} {np.median(en[0::8]):.1f} {en[0::8].max():.1f}')

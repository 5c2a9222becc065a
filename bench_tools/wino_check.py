"""Developer tool: generator layer 2 as a 1-D Winograd convolution (k_convw, option 'wino') against the 25-tap f16x3 kernel and
the exact-f32 path on the same inputs — largest difference of the net's output, and the layer's kernel time (HIP events).
    python bench_tools/wino_check.py [members]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = 64
for kind in (('gan', 'vae', 'gz') if len(sys.argv) < 3 else sys.argv[2:]):
    gen, _ = bench.load_generator(kind, 0)
    gen.check_range = False
    rs = np.random.RandomState(1)
    n_in = 2 if kind == 'gz' else 4
    x = rs.randn(B, n_in, N, N).astype('float32')
    x[:, :2] *= 1.5
    xd = torch.as_tensor(x, device='cuda')
    gen.set_option('precision', 0)
    ref = gen.cnn_forward(xd).double()
    gen.set_option('precision', 3)
    out = {}
    ab = b'+ab' in __import__('pyqg_generative_amd')._lib.lib.qgx_version()
    for wino in ((0, 1, 2) if ab else (0, 1)):
        gen.set_option('wino', 1 if wino else 0)
        if ab:
            gen.set_option('wino_pl', 1 if wino == 2 else 0)     # 2: channel-planar layer 1 + MFMA input transform (A/B only)
        gen.set_option('wino_min_tiles', 1)
        y = gen.cnn_forward(xd)
        torch.cuda.synchronize()
        gen.set_option('prof_every', 1)
        gen.profile(1)
        for _ in range(20):
            y = gen.cnn_forward(xd)
        ms, n = gen.profile_read()
        gen.profile(-1)
        out[wino] = y.double()
        err = float((out[wino] - ref).abs().max() / ref.abs().max())
        print(f'{kind} B={B} wino={wino}: max |y - y_f32| / max|y| = {err:.2e}; layer 2 {1e3 * ms / max(n, 1):.1f} us; flags {gen.range_read()}', flush=True)
    print(f'   wino vs 25-tap: {float((out[1] - out[0]).abs().max() / ref.abs().max()):.2e}')
    if ab:
        gen.set_option('wino_pl', 0)
        for exp, what in ((2, 'no MFMAs'), (4, 'weights loaded once'), (5, 'no raw-patch copy'), (1, 'no input transform'), (6, 'no output stores'), (7, 'no MFMAs, no input transform'), (8, 'no MFMAs, no output stores')):
            gen.set_option('wino_exp', exp)
            gen.cnn_forward(xd)
            gen.profile(1)
            for _ in range(20):
                gen.cnn_forward(xd)
            ms, n = gen.profile_read()
            gen.profile(-1)
            print(f'   experiment {what}: layer 2 {1e3 * ms / max(n, 1):.1f} us')
        gen.set_option('wino_exp', 0)

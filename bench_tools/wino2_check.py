"""Developer tool: generator layer 2 as k_convw2 (option 'wino2', conv_wino2.hpp: input transform under the MFMAs) against
k_convw on the same inputs through the product library — bit identity of the net's output, and the layer's kernel time (HIP events).
    python bench_tools/wino2_check.py [members] [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for kind in ('gan', 'vae', 'gz'):
    gen, _ = bench.load_generator(kind, 0)
    gen.check_range = False
    rs = np.random.RandomState(1)
    n_in = 2 if kind == 'gz' else 4
    x = rs.randn(B, n_in, N, N).astype('float32')
    x[:, :2] *= 1.5
    xd = torch.as_tensor(x, device='cuda')
    gen.set_option('wino_min_tiles', 1)
    out, t = {}, {}
    for w2 in (0, 1, 0, 1):
        gen.set_option('wino2', w2)
        y = gen.cnn_forward(xd)
        torch.cuda.synchronize()
        gen.set_option('prof_every', 1)
        gen.profile(1)
        for _ in range(30):
            y = gen.cnn_forward(xd)
        ms, n = gen.profile_read()
        gen.profile(-1)
        out[w2] = y.clone()
        t[w2] = 1e3 * ms / max(n, 1)
    same = bool(torch.equal(out[0], out[1]))
    print(f'{kind} N={N} B={B}: layer 2 k_convw {t[0]:.1f} us, k_convw2 {t[1]:.1f} us ({t[0] / t[1]:.2f}x); outputs bit-identical: {same}', flush=True)

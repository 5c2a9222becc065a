#!/usr/bin/env python
"""Developer tool: HIP-event time of one generator layer (0-based index) under option sets, in one process.
    python bench_tools/layer_time.py <layer> "opt=val,..." ...      (env NX, MEMBERS, KIND)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

N, B = int(os.environ.get('NX', 64)), int(os.environ.get('MEMBERS', 128))
gen, _ = bench.load_generator(os.environ.get('KIND', 'gan'), 0)
gen.check_range = False
layer = int(sys.argv[1])
x = torch.randn((B, gen.n_in, N, N), dtype=torch.float32, device='cuda')
specs = ['base'] + sys.argv[2:] + ['base']
for rnd in range(2):
    for spec in specs:
        opts = dict(kv.split('=') for kv in spec.split(',')) if spec != 'base' else {}
        for k, v in opts.items():
            gen.set_option(k, int(v))
        for _ in range(3):
            gen.cnn_forward(x)
        gen.profile(layer)
        for _ in range(20):
            gen.cnn_forward(x)
        ms, n = gen.profile_read()
        gen.profile(-1)
        if rnd:
            print(f'{spec:30s} layer {layer}: {1e3 * ms / max(n, 1):7.1f} us over {n} launches')
        for k in opts:
            gen.set_option(k, {'h2_x96': 1, 'h2_w8_min96': 1024, 'fuse': 3, 'h2_w8': 3, 'pair': 1}.get(k, 0))

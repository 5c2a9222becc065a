#!/usr/bin/env python
"""Developer tool: interleaved in-process A/B timing of the generator's conv-kernel variants
(cdna_hip_programming.md rule 24: N variants x M rounds in ONE process on ONE device).
Per variant and per layer: median / min kernel time from HIP events (qgx_generator_profile)."""
import argparse
import itertools
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the A/B library (make -C pyqg_generative_amd/csrc ab) carries every kernel variant; the product library only
# the fastest path per layer and size
_AB = os.path.join(ROOT, 'pyqg_generative_amd', 'libqgx_ab.so')
if 'QGX_LIB' not in os.environ and os.path.exists(_AB):
    os.environ['QGX_LIB'] = _AB
import pyqg_generative_amd as qa
from pyqg_generative_amd import weights

OPTS = {'chunk': 32, 'last_valu': 1, 'first_split': 2, 'v3': -1, 'precision': 0, 'ascale_log2': 0, 'first_h': 1, 'half_nw': 8, 'member_chunk': 0, 'res': 1, 'h2': 3, 'pair': 1, 'fuse': 3, 'half_min_tiles': 1, 'h3': 0, 'last_rows': 0, 'part_max_tiles': 96, 'fold': 1, 'h2_grid': 0, 'h4': 0, 'prio_alt': 1, 'h2_w8': 3, 'h2_tw32': 0, 'pair_lp': 1, 'wino_pl': 0, 'wino': 1, 'wino_min_tiles': 48, 'fuse96': 2, 'wino_rows96': 0, 'h2_rows96': 0, 'small_tiles': 1, 'wino_rows64': 0}      # option -> default
MAC = [12800, 204800, 18432, 9216, 9216, 9216, 9216, 576]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=128)
    ap.add_argument('--nx', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--layers', default='1,3,2,0,7')
    ap.add_argument('--variants', default='v3=0,chunk=16;v3=0,chunk=32;v3=1;v3=2')
    args = ap.parse_args()
    B, N = args.members, args.nx
    nets, xs, ys = weights.load_npz(os.path.join(ROOT, 'tests', 'golden', 'weights_gan.npz'), 'gan')
    gen = qa.Generator('gan', nets, xs, ys)
    x = torch.randn((B, 4, N, N), dtype=torch.float32, device='cuda')
    variants = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in v.split(',')) for v in args.variants.split(';')]
    layers = [int(l) for l in args.layers.split(',')]
    res = {(vi, l): [] for vi in range(len(variants)) for l in layers}
    # correctness of every variant against variant 0 (exact f32 MFMA: identical sums up to ordering)
    outs = []
    for v in variants:
        for k in OPTS:
            gen.set_option(k, v.get(k, OPTS[k]))
        outs.append(gen.cnn_forward(x).clone())
    torch.cuda.synchronize()
    for v, o in zip(variants, outs):
        print(f'  check {str(v):50s} max|diff| vs first = {(o - outs[0]).abs().max().item():.3e}  (max|y| = {outs[0].abs().max().item():.3e})')
    for _ in range(3):
        gen.cnn_forward(x)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for l in layers:
            gen.profile(l)
            for vi in np.random.permutation(len(variants)):
                v = variants[vi]
                for k in OPTS:
                    gen.set_option(k, v.get(k, OPTS[k]))
                gen.cnn_forward(x)
                torch.cuda.synchronize()
                ms, n = gen.profile_read()
                res[(vi, l)].append(ms / max(n, 1))
    gen.profile(-1)
    # whole forward, interleaved
    tot = {vi: [] for vi in range(len(variants))}
    for r in range(args.rounds):
        for vi in np.random.permutation(len(variants)):
            v = variants[vi]
            for k in OPTS:
                gen.set_option(k, v.get(k, OPTS[k]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gen.cnn_forward(x)
            e1.record()
            torch.cuda.synchronize()
            tot[vi].append(e0.elapsed_time(e1) / 5 * 1e3)
    for vi, v in enumerate(variants):
        print(f'  whole forward {str(v):55s} {np.median(tot[vi]):9.1f} us')
    print(f'B={B} N={N}; times in us (median / min over {args.rounds} rounds); TF = algorithmic TFLOP/s at the median')
    for l in layers:
        flop = 2.0 * MAC[l] * N * N * B
        for vi, v in enumerate(variants):
            t = np.array(res[(vi, l)]) * 1e3
            print(f'  layer {l}  {str(v):55s} {np.median(t):9.1f} / {t.min():9.1f}   {flop / (np.median(t) * 1e-6) / 1e12:6.1f} TF')


if __name__ == '__main__':
    main()

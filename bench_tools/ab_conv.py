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
import pyqg_generative_amd as qa
from pyqg_generative_amd import weights

MAC = [12800, 204800, 18432, 9216, 9216, 9216, 9216, 576]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--members', type=int, default=128)
    ap.add_argument('--nx', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--layers', default='1,3,2,0,7')
    ap.add_argument('--variants', default='chunk=16,stage_batched=0;chunk=16,stage_batched=1;'
                                          'chunk=32,stage_batched=0;chunk=32,stage_batched=1;persistent=1,chunk=32')
    args = ap.parse_args()
    B, N = args.members, args.nx
    nets, xs, ys = weights.load_npz(os.path.join(ROOT, 'tests', 'golden', 'weights_gan.npz'), 'gan')
    gen = qa.Generator('gan', nets, xs, ys)
    x = torch.randn((B, 4, N, N), dtype=torch.float32, device='cuda')
    variants = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in v.split(',')) for v in args.variants.split(';')]
    layers = [int(l) for l in args.layers.split(',')]
    res = {(vi, l): [] for vi in range(len(variants)) for l in layers}
    for _ in range(3):
        gen.cnn_forward(x)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for l in layers:
            gen.profile(l)
            for vi, v in enumerate(variants):
                for k in ('chunk', 'stage_batched', 'persistent', 'last_valu'):
                    gen.set_option(k, v.get(k, {'chunk': 16, 'last_valu': 1}.get(k, 0)))
                gen.cnn_forward(x)
                torch.cuda.synchronize()
                ms, n = gen.profile_read()
                res[(vi, l)].append(ms / max(n, 1))
    gen.profile(-1)
    print(f'B={B} N={N}; times in us (median / min over {args.rounds} rounds); TF = algorithmic TFLOP/s at the median')
    for l in layers:
        flop = 2.0 * MAC[l] * N * N * B
        for vi, v in enumerate(variants):
            t = np.array(res[(vi, l)]) * 1e3
            print(f'  layer {l}  {str(v):55s} {np.median(t):9.1f} / {t.min():9.1f}   {flop / (np.median(t) * 1e-6) / 1e12:6.1f} TF')


if __name__ == '__main__':
    main()

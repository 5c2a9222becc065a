#!/usr/bin/env python
"""Developer tool: does splitting one GPU's shard into sub-ensembles advanced on separate HIP streams raise the
throughput (kernels of one stream filling the ramp / tail / latency-bound phases of the other's)?
    python bench_tools/two_stream.py [members] [parts...]           (64 x 64 eddy + CGAN)
    QGX_TS=96 python bench_tools/two_stream.py 32 1 2                (developer switch of this tool: 96 x 96 jet + CVAE)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyqg_generative_amd as qa
import bench

N = int(os.environ.get('QGX_TS', '64'))
KIND = 'vae' if N == 96 else 'gan'
total = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for parts in [int(a) for a in sys.argv[2:]] or [1, 2, 4]:
    B = total // parts
    gens = [bench.load_generator(KIND, 0)[0] for _ in range(parts)]
    engs = [qa.EnsembleEngine(nx=N, n_members=B, dt=14400. if N == 64 else 7200.) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for p, e in enumerate(engs):
        e.set_q(bench.eddy_like_q(np.arange(p * B, (p + 1) * B), N))

    def run(K):
        # interleave short chunks so that the host keeps every stream fed
        for c in range(K // 10):
            for p in range(parts):
                with torch.cuda.stream(streams[p]):
                    engs[p].step(10, generator=gens[p], sampling='constant', nsteps_decor=1, seed=1, member_offset=p * B)
        torch.cuda.synchronize()
    run(20)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(200); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f'{parts} stream(s) x {B} members: {total * 200 / t:9.0f} steps/s  ({1e3 * t / 200:.3f} ms per {total}-member step)')
    for e in engs: e.close()
    for g in gens: g.close()

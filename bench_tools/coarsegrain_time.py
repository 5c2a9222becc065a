"""Developer tool: time of the subgrid-forcing diagnostic of one snapshot (64 members 256 x 256 -> 64 x 64, Operator2 +
Operator5, 3/2-rule), as bench.py --leg config4 runs it; under rocprofv3 --kernel-trace --stats for the kernel split."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import pyqg_generative_amd as qa
from pyqg_generative_amd.tools.operators import Dev

B, N, nc = 64, 256, 64
q = torch.as_tensor(bench.eddy_like_q(list(range(B)), N)).cuda()
pp = dict(qa.engine.PYQG_DEFAULTS)


def coarsegrain():
    qh, adv_hat = Dev.hires_tendency_hat(q, pp, '3/2-rule')
    for op in (Dev.Operator2, Dev.Operator5):
        Dev.subgrid_forcing_from_hat(qh, adv_hat, nc, op, pp, '3/2-rule')


coarsegrain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    coarsegrain()
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) * 100:.2f} ms per snapshot')
Dev.close()

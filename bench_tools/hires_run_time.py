"""Developer tool: time per step of a 256 x 256 unparameterized ensemble at the reference's cadence (time-averaged
diagnostics every day = 24 steps, a snapshot every 1000 steps), as generate_subgrid_forcing steps its members."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import math
import torch
import bench
import pyqg_generative_amd as qa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = qa.EnsembleEngine(nx=256, n_members=B, device=0, dt=3600.)
e.set_q(bench.eddy_like_q(list(range(B)), 256))
loop = bench.OnlineLoop(e, 3600., {})
loop.run(100)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop.run(2000)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2000
print(f'B={B}: {1e6 * dt:.1f} us/step at the reference cadence ({loop.nsnap} snapshots, {e.diag_count} diagnostic increments), '
      f'{B / dt:.0f} member-steps/s')
e.close()

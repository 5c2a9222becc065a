"""Developer tool: largest difference between the XCD-resident run kernel and the three-launch step after N
unparameterized 256 x 256 steps from the same state (where, and how many elements).   python bench_tools/team_debug.py [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
import pyqg_generative_amd as qa
import pyqg_generative_amd._lib as L
B = 8
q0 = bench.eddy_like_q(list(range(B)), 256)
res = []
for team in (True, False):
    e = qa.EnsembleEngine(nx=256, n_members=B, device=0, dt=3600.)
    if not team: e.set_option('team', 0)
    e.set_q(q0)
    e.step(int(sys.argv[1]) if len(sys.argv) > 1 else 2, refresh_diag=False)
    res.append({f: e.get(f).cpu().numpy() for f in (L.F_QH, L.F_DQHDT)})
    e.close()
for f in res[0]:
    a, b = res[0][f], res[1][f]
    d = np.abs(a - b)
    print(f, 'max rel', d.max() / np.abs(b).max(), 'argmax', np.unravel_index(d.argmax(), d.shape), 'nbad', (d > 1e-10 * np.abs(b).max()).sum(), 'of', d.size)
    bad = d > 1e-10 * np.abs(b).max()
    if bad.any():
        idx = np.argwhere(bad)
        print(' members', np.unique(idx[:, 0]), 'layers', np.unique(idx[:, 1]), 'rows', np.unique(idx[:, 2])[:20], 'cols', np.unique(idx[:, 3])[:20])

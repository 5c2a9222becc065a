#!/usr/bin/env python
"""Generates tests/golden/oracle_forcing_dataset_stats.npz: the reference's forcing-dataset protocol
(scripts/run_forcing_datasets.py:17-25 -> tools/simulate.py:62-106: 256 x 256 eddy run of 10 years, a snapshot every
1000 steps coarse-grained to 64 x 64 with Operator1, subgrid forcing without dealiasing) run with the CPU ORACLE
(oracle/qg_ref.py + oracle/operators_ref.py), a few members, ~20 CPU-minutes each.  Stored: per-member sums that give the
two statistics the reference publishes as checksums of its dataset `eddy/64/sharp` (Google-Colab/dataset.ipynb cell 16):
std of the coarse-grained PV and of the subgrid forcing.  tests/test_oracle_qg.py compares them — this is the direct pin of
the oracle's spectral half against real pyqg output; the HIP path is held to the same numbers in
tests/test_gpu_statistics.py.

    python tests/golden/make_oracle_forcing_stats.py [n_members] [n_procs]
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
YEAR = 360 * 86400.


def member(b):
    from oracle import qg_ref, operators_ref
    params = dict(nx=256, dt=3600., tmax=10 * YEAR, tavestart=5 * YEAR)
    m = qg_ref.QGModelRef(twrite=10 ** 9, **params)
    qg_ref.set_initial_condition(m, np.random.RandomState(7000 + b))
    acc = np.zeros(6)            # n, sum q, sum q^2, n, sum f, sum f^2  (float32-stored values, as the dataset holds them)
    t0 = time.time()
    nsnap = 0
    for _ in m.run_with_snapshots(tsnapint=3600000.):
        forcing, mf, _ = operators_ref.PV_subgrid_forcing(m.q, 64, operators_ref.Operator1, {}, 'none')
        q = mf.q.astype('float32').astype('float64')
        f = forcing.astype('float32').astype('float64')
        acc += [q.size, q.sum(), (q ** 2).sum(), f.size, f.sum(), (f ** 2).sum()]
        nsnap += 1
        if b == 0 and nsnap % 10 == 0:
            print(f'member 0: snapshot {nsnap}/86, KE {m._calc_ke():.3e}, {time.time() - t0:.0f} s', flush=True)
    return np.append(acc, nsnap)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else min(n, os.cpu_count() or 1)
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(procs) as pool:
        res = np.array(pool.map(member, range(n)))
    path = os.path.join(ROOT, 'tests', 'golden', 'oracle_forcing_dataset_stats.npz')
    np.savez(path, sums=res, columns=np.array(['n_q', 'sum_q', 'sum_q2', 'n_f', 'sum_f', 'sum_f2', 'snapshots']))
    tot = res.sum(0)
    print('wrote', path)
    print('std q', np.sqrt(tot[2] / tot[0] - (tot[1] / tot[0]) ** 2), 'std forcing', np.sqrt(tot[5] / tot[3] - (tot[4] / tot[3]) ** 2))


if __name__ == '__main__':
    main()

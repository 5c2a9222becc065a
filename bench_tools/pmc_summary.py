#!/usr/bin/env python
"""Developer tool: per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate
runs as MI355X_MICROARCH.md prescribes).  Counters are KiB per dispatch; gfx950 correction: wide
(16 B/lane) reads are tallied at 1/2 -> traffic = (2*FETCH + WRITE) * 1024 B.

    python bench_tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> [<out.json> <kernel substring> [<config json> [<command note>]]]

With several kernels matching the substring (e.g. "k_l_" = every kernel of the large-grid spectral step) the
JSON holds the SUM of their per-launch traffic times their launches per step (launches / steps given as
"steps" in the config json).
"""
import json
import sys
import pandas as pd

f, w, out = sys.argv[1:4]
F = pd.read_csv(f)
W = pd.read_csv(w)
F = F[F.Counter_Name == 'FETCH_SIZE'].groupby('Kernel_Name').Counter_Value.agg(['mean', 'count'])
W = W[W.Counter_Name == 'WRITE_SIZE'].groupby('Kernel_Name').Counter_Value.agg(['mean', 'count'])
d = F.join(W, lsuffix='_f', rsuffix='_w', how='outer').fillna(0.0)
d['traffic'] = (2 * d.mean_f + d.mean_w) * 1024
d = d.sort_values('traffic', ascending=False)
with open(out, 'w') as fh:
    note = sys.argv[7] if len(sys.argv) > 7 else 'bench.py --steps 5 --warmup 2 (B=128, N=64, GAN)'
    fh.write(f'# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {note}\n')
    fh.write('# counters in KiB per launch; gfx950 correction: wide (16 B/lane) reads are tallied at 1/2 -> traffic = (2*FETCH + WRITE)*1024 B\n')
    fh.write('kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,traffic_bytes_per_launch\n')
    for k, r in d.iterrows():
        fh.write(f'"{k}",{int(r.count_f)},{r.mean_f:.1f},{r.mean_w:.1f},{r.traffic:.0f}\n')
if len(sys.argv) > 5:
    sel = d[d.index.str.contains(sys.argv[5], regex=False)]
    print(sel)
    cfg = json.loads(sys.argv[6]) if len(sys.argv) > 6 else {'nx': 64, 'members_per_gpu': 128, 'kind': 'gan'}
    steps = cfg.pop('steps', None)
    if steps:       # a multi-kernel step: bytes per STEP = sum over kernels of traffic x launches / steps
        tot = float((sel.traffic * sel.count_f).sum() / steps)
    else:
        tot = float(sel.traffic.iloc[0])
    json.dump({'kernel': sys.argv[5], 'config': cfg, 'traffic_bytes_per_launch': tot, 'source': out},
              open(sys.argv[4], 'w'), indent=1)

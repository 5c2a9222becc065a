#!/usr/bin/env python
"""Developer tool: busy time and idle gaps between consecutive kernels of the steady-state online step, from a
rocprofv3 --kernel-trace database (rocpd):  python bench_tools/gap_report.py <results.db> [first-kernel substring]"""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else 'k_prep_noise'
rows = c.execute('select name,start,end from kernels order by start').fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
span, busy, gaps = [], [], []
for a, b in zip(idx[len(idx) // 3:-1], idx[len(idx) // 3 + 1:]):
    seq = rows[a:b]
    gaps.append(sum(max(0, seq[i + 1][1] - seq[i][2]) for i in range(len(seq) - 1)) + max(0, rows[b][1] - seq[-1][2]))
    busy.append(sum(r[2] - r[1] for r in seq))
    span.append(rows[b][1] - rows[a][1])
print(f'steps {len(span)}: span {np.median(span) / 1e3:.1f} us, kernels busy {np.median(busy) / 1e3:.1f} us, idle {np.median(gaps) / 1e3:.1f} us')
a, b = idx[-3], idx[-2]
for i in range(a, b):
    r, n = rows[i], rows[i + 1]
    print(f'  {r[0][:70]:70s} {(r[2] - r[1]) / 1e3:7.1f} us  gap after {(n[1] - r[2]) / 1e3:6.1f}')

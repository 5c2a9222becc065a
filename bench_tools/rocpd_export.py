#!/usr/bin/env python
"""Developer tool: rocprofv3 (ROCm 7.2 default output = rocpd SQLite) -> the CSV summaries kept under profiles/.

    python bench_tools/rocpd_export.py stats <results.db> <out.csv> "<command note>"
    python bench_tools/rocpd_export.py pmc <fetch.db> <write.db> <out.csv> "<command note>" [<out.json> <kernel substring> '<config json>']
    python bench_tools/rocpd_export.py mfma <counters.db> <out.csv> "<command note>"

stats: per-kernel calls / total / average / min / max duration (the --kernel-trace --stats table).
pmc:   per-kernel HBM traffic from the two separate PMC passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch);
       gfx950 correction of MI355X_MICROARCH.md: wide (16 B/lane) reads are tallied at 1/2 ->
       traffic = (2*FETCH + WRITE) * 1024 B.  With "steps" in the config json the JSON holds the bytes per
       STEP summed over every kernel matching the substring (a multi-kernel step).
mfma:  per-kernel matrix-core utilisation from ONE pass with GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
       SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16: clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration,
       mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8).
"""
import json
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]) for r in rows]


def pmc_mean(db, counter):
    c = sqlite3.connect(db)
    return {r[0]: (r[1], r[2]) for r in c.execute(
        "select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name",
        (counter,))}


def main():
    mode = sys.argv[1]
    if mode == 'stats':
        db, out, note = sys.argv[2:5]
        with open(out, 'w') as fh:
            fh.write(f'# rocprofv3 --kernel-trace --stats -- {note}\n')
            fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
            for r in kernel_stats(db):
                fh.write(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{r[4]:.2f},{r[5]},{r[6]}\n')
        return
    if mode == 'mfma':
        db, out, note = sys.argv[2:5]
        names = ['GRBM_GUI_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES']
        cnt = {n: pmc_mean(db, n) for n in names}
        dur = {r[0]: (r[1], r[3]) for r in kernel_stats(db)}
        kernels = sorted(cnt['GRBM_GUI_ACTIVE'], key=lambda k: -cnt['GRBM_GUI_ACTIVE'][k][0] * cnt['GRBM_GUI_ACTIVE'][k][1])
        with open(out, 'w') as fh:
            fh.write(f'# rocprofv3 --kernel-trace --pmc {" ".join(names)} -- {note}\n')
            fh.write('# clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)\n')
            fh.write('Kernel_Name,' + ','.join(names) + ',launches,avg_us,clock_GHz,mfma_busy_frac\n')
            for k in kernels:
                v = [cnt[n].get(k, (0.0, 0))[0] for n in names]
                n_l, avg_ns = dur.get(k, (0, 0.0))
                gui = v[0]
                clock = gui / 8 / avg_ns if avg_ns else 0.0
                busy = v[3] / 1024 / (gui / 8) if gui else 0.0
                fh.write(f'"{k}",' + ','.join(f'{x:.4g}' for x in v) + f',{n_l},{avg_ns / 1e3:.2f},{clock:.3f},{busy:.4f}\n')
        return
    fdb, wdb, out, note = sys.argv[2:6]
    F, W = pmc_mean(fdb, 'FETCH_SIZE'), pmc_mean(wdb, 'WRITE_SIZE')
    rows = []
    for k in set(F) | set(W):
        f, n = F.get(k, (0.0, 0))
        w, _ = W.get(k, (0.0, 0))
        rows.append((k, n, f, w, (2 * f + w) * 1024))
    rows.sort(key=lambda r: -r[4] * r[1])
    with open(out, 'w') as fh:
        fh.write(f'# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {note}\n')
        fh.write('# counters in KiB per launch; gfx950 correction: wide (16 B/lane) reads are tallied at 1/2 -> traffic = (2*FETCH + WRITE)*1024 B\n')
        fh.write('kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,traffic_bytes_per_launch\n')
        for r in rows:
            fh.write(f'"{r[0]}",{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]:.0f}\n')
    if len(sys.argv) > 8:
        jout, sub, cfg = sys.argv[6], sys.argv[7], json.loads(sys.argv[8])
        sel = [r for r in rows if any(x in r[0] for x in sub.split('|'))]
        steps = cfg.pop('steps', None)
        val = sum(r[4] * r[1] for r in sel) / steps if steps else sel[0][4]
        for r in sel:
            print(r)
        json.dump({'kernel': sub, 'config': cfg, 'traffic_bytes_per_launch': float(val), 'source': out,
                   'per': 'step (all matching kernels)' if steps else 'launch'}, open(jout, 'w'), indent=1)


if __name__ == '__main__':
    main()

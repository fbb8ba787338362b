"""Coarse timeline of the last train step of a rocpd kernel trace: per stream, runs of same-named kernels with start
offset, count and busy time -- enough to read the critical path off the trace.
usage: python scripts/rocpd_phases.py <results.db> [min_run_us]"""
import re, sqlite3, sys
db = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, stream_id, grid_x/workgroup_x, grid_y from kernels order by start"))
adam = [i for i, r in enumerate(rows) if 'clip_adam' in r[0]]
rows = rows[adam[-2] + 1: adam[-1] + 1]
t0 = rows[0][1]
short = lambda n: re.sub(r'\(.*', '', n.replace('void ptv::', '').replace('ptv::', '').replace('ptv::BF16, ', ''))[:58]
streams = sorted({r[3] for r in rows})
for sid in streams:
    rs = [r for r in rows if r[3] == sid]
    print('--- stream %s: %d kernels, busy %.2f ms' % (sid, len(rs), sum(r[2] - r[1] for r in rs) / 1e6))
    run = None
    out = []
    for n, s, e, _, gx, gy in rs:
        key = (short(n), gx, gy)
        if run and run[0] == key and s - run[3] < 200e3:
            run[2] += e - s; run[3] = e; run[4] += 1
        else:
            if run: out.append(run)
            run = [key, s, e - s, e, 1]
    if run: out.append(run)
    for key, s, busy, e, cnt in out:
        if (e - s) / 1e3 >= min_us:
            print('  t=%7.2f ms  span %7.1f us  busy %7.1f us  x%-3d %s grid=%dx%d' % ((s - t0) / 1e6, (e - s) / 1e3, busy / 1e3, cnt, key[0], key[1], key[2]))

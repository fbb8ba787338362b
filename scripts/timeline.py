"""Timeline of ONE train step from a rocprofv3 rocpd database (kernel trace): per kernel start (ms since the step's first kernel),
duration, stream / queue, grid; kernels shorter than --min-us are folded into runs.  The step = the dispatches between the last two
`clip_adam` kernels.   python scripts/timeline.py <results.db> [--min-us 15] [--step -1]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument('db')
ap.add_argument('--min-us', type=float, default=15.0)
ap.add_argument('--step', type=int, default=-1)
ap.add_argument('--marker', default='clip_adam')
a = ap.parse_args()
con = sqlite3.connect(a.db)
cur = con.cursor()
cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
qcol = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
sel = 'name, start, end, grid_x, grid_y, grid_z' + (', ' + qcol if qcol else ', 0')
rows = list(cur.execute('select %s from kernels order by start' % sel))
marks = [i for i, r in enumerate(rows) if a.marker in r[0]]
assert len(marks) >= 2, 'need two optimiser kernels in the trace'
hi = marks[a.step]
lo = marks[a.step - 1] + 1
step = rows[lo:hi + 1]
t0 = step[0][1]
print('# step: %d kernels, %.3f ms first start -> last end; columns: start ms | dur us | stream | grid | kernel' %
      (len(step), (max(r[2] for r in step) - t0) / 1e6))
streams = {}
for r in step:
    streams.setdefault(r[6], len(streams))
fold = {}


def flush(q):
    f = fold.pop(q, None)
    if f:
        print('%8.3f %8.1f  s%d  (%d short kernels, busy %.1f us: %s)' % ((f[0] - t0) / 1e6, (f[1] - f[0]) / 1e3, streams[q], f[2], f[3] / 1e3,
                                                                      ', '.join(sorted(f[4]))[:150]))


for n, s, e, gx, gy, gz, q in step:
    d = (e - s) / 1e3
    short = n.split('(')[0].replace('ptv::', '').replace('void ', '')[:90]
    if d < a.min_us:
        f = fold.get(q)
        if f is None:
            fold[q] = [s, e, 1, e - s, {short[:40]}]
        else:
            f[1] = e; f[2] += 1; f[3] += e - s; f[4].add(short[:40])
        continue
    flush(q)
    print('%8.3f %8.1f  s%d  %dx%dx%d  %s' % ((s - t0) / 1e6, d, streams[q], gx, gy, gz, short))
for q in list(fold):
    flush(q)

"""Concurrency timeline of the last train step in a rocprofv3 rocpd kernel trace: how long the GPU runs 0 / 1 / 2+
kernels at once, and which kernels run ALONE (the serial, latency-bound part of the step).
usage: python scripts/rocpd_timeline.py <results.db> <steps_in_trace>"""
import collections
import re
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
# last step = dispatches after the last-but-one optimizer kernel
adam = [i for i, r in enumerate(rows) if 'clip_adam' in r[0]]
lo, hi = adam[-2] + 1, adam[-1] + 1
rows = rows[lo:hi]
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set()
hist = collections.defaultdict(float)
alone = collections.defaultdict(float)
prev = t0
for t, d, i in ev:
    dt = t - prev
    if dt > 0:
        hist[min(len(active), 4)] += dt
        if len(active) == 1:
            alone[rows[next(iter(active))][0]] += dt
    prev = t
    if d > 0: active.add(i)
    else: active.discard(i)
wall = (t1 - t0) / 1e6
print('last step: %.2f ms wall, %d kernels, sum of kernel time %.2f ms' % (wall, len(rows), sum(e - s for _, s, e in rows) / 1e6))
for k in sorted(hist):
    print('  %s kernels running: %.2f ms (%.0f%%)' % (('%d' % k) if k < 4 else '4+', hist[k] / 1e6, 100 * hist[k] / 1e6 / wall))
print('kernels running alone:')
short = lambda n: re.sub(r'\(.*', '', n.replace('void ptv::', '').replace('ptv::', ''))[:90]
for n, t in sorted(alone.items(), key=lambda x: -x[1])[:14]:
    print('  %.2f ms  %s' % (t / 1e6, short(n)))

"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a markdown table.
usage: python scripts/rocpd_summary.py <results.db> <out.md> <steps_in_trace> "<title>" "<command>" """
import collections
import sqlite3
import sys

db, out, steps, title, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
con = sqlite3.connect(db)
cur = con.cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z, vgpr_count, accum_vgpr_count, lds_size "
                        "from kernels order by start"))
tot = collections.defaultdict(float)
cnt = collections.Counter()
meta = {}
for n, s, e, gx, gy, gz, v, a, l in rows:
    tot[n] += (e - s)
    cnt[n] += 1
    meta[n] = (v, a, l)
T = sum(tot.values())
with open(out, 'w') as f:
    f.write('# %s\n\nCommand: `%s`\n\n' % (title, cmd))
    f.write('%d train steps in the trace.  Total kernel time %.1f ms over %d dispatches = **%.2f ms/step**, '
            '%d launches/step.\n\n' % (steps, T / 1e6, len(rows), T / 1e6 / steps, len(rows) // steps))
    fam = lambda pred: (sum(t for n, t in tot.items() if pred(n)) / 1e6 / steps, sum(c for n, c in cnt.items() if pred(n)) / steps)
    wg, wgn = fam(lambda n: 'wgrad_' in n)
    cs, csn = fam(lambda n: 'colsum_kernel' in n)
    f.write('Weight-gradient family (`wgrad_kernel*` + `wgrad_batch_kernel*` + `wgrad_reduce*`): **%.3f ms/step**, %.1f launches/step; '
            '`colsum_kernel*`: %.3f ms/step, %.1f launches/step.\n\n' % (wg, wgn, cs, csn))
    f.write('| kernel | calls | ms/step | avg us | % | vgpr | agpr | lds |\n|---|---|---|---|---|---|---|---|\n')
    for n, t in sorted(tot.items(), key=lambda x: -x[1])[:32]:
        v, a, l = meta[n]
        f.write('| `%s` | %d | %.3f | %.1f | %.1f | %s | %s | %s |\n'
                % (n[:100], cnt[n], t / 1e6 / steps, t / cnt[n] / 1e3, 100 * t / T, v, a, l))
print(open(out).read()[:2500])

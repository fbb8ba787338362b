"""per-launch-shape breakdown of the plain GEMM kernels in a rocprofv3 rocpd trace
usage: python scripts/rocpd_gemms.py <results.db> <steps_in_trace> [name-substring]"""
import sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else 'gemm_plain'
con = sqlite3.connect(db)
rows = list(con.execute("select name, grid_x/256, grid_y, grid_z, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels "
                        "where name like '%%%s%%' group by name, grid_x, grid_y, grid_z order by 7 desc" % pat))
tot = 0
for r in rows:
    tot += r[6] / steps
for r in rows[:28]:
    n = r[0].split('<')[1].split('>')[0] if '<' in r[0] else r[0][:40]
    print('%-52s tiles n,m,z=%4d,%5d,%4d  calls/step=%5.1f avg=%7.1fus total=%.2fms' % (n, r[1], r[2], r[3], r[4] / steps, r[5], r[6] / steps))
print('total %.2f ms/step' % tot)

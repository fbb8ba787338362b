"""Group rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv rows by (kernel, grid) -> mean HBM bytes per launch.
Units and gfx950 corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count
kilobytes... see the guide; the raw means are kept in the json next to the corrected bytes."""
import collections, csv, glob, json, re, sys
d = sys.argv[1]
out = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('%s/%s_counter_collection.csv' % (d, c))
    if not f:
        continue
    per = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != c:
            continue
        per[r['Dispatch_Id']] += float(r['Counter_Value'])
        meta[r['Dispatch_Id']] = (r['Kernel_Name'], r.get('Grid_Size', r.get('Grid_Size_X', '')))
    grp = collections.defaultdict(list)
    for k, v in per.items():
        grp[meta[k]].append(v)
    for k, v in grp.items():
        out[k][c] = (sum(v) / len(v), len(v))
res = []
for (name, grid), v in out.items():
    f, nf = v.get('FETCH_SIZE', (0, 0)); w, nw = v.get('WRITE_SIZE', (0, 0))
    res.append({'kernel': name, 'grid': grid, 'launches': nf, 'fetch_raw': f, 'write_raw': w})
res.sort(key=lambda r: -(r['fetch_raw'] + r['write_raw']) * r['launches'])
json.dump(res, open('%s/pmc_by_kernel.json' % d, 'w'), indent=1)
short = lambda n: re.sub(r'\(.*', '', n.replace('void ptv::', '').replace('ptv::', ''))[:70]
for r in res[:40]:
    print('%-72s grid=%-9s n=%3d fetch=%.1f write=%.1f' % (short(r['kernel']), r['grid'], r['launches'], r['fetch_raw'], r['write_raw']))

"""rocprofv3 --pmc csv files of scripts/gpu_pmc.sh -> per-kernel means per launch.  HBM bytes follow MI355X_MICROARCH.md (HBM
section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B,
so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is used as reported.  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 * 1024): busy cycles are summed over the chip's 1024 SIMDs (= 16 cycles x SQ_INSTS_MFMA for the 16x16x32
bf16 MFMA, which checks against the instruction count), GRBM_GUI_ACTIVE is summed over the 8 XCDs (one eighth of it is the
dispatch's duration in cycles: 2.63 M cycles for the 1.15 ms forward launch); the raw means stay in the json."""
import collections
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))      # (kernel, counter) -> dispatch -> value
for f in glob.glob('%s/*_counter_collection.csv' % d):
    for r in csv.DictReader(open(f)):
        per[(r['Kernel_Name'], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
kern = collections.defaultdict(dict)
for (k, c), v in per.items():
    kern[k][c] = (sum(v.values()) / len(v), len(v))
short = lambda n: re.sub(r'\(.*', '', n.replace('void ptv::', '').replace('ptv::', '').replace('nr::', '').replace('nb::', ''))
res = []
for k, v in kern.items():
    g = lambda c: v.get(c, (0.0, 0))[0]
    rd, wr = 2 * g('FETCH_SIZE') * 1024, g('WRITE_SIZE') * 1024
    act = g('GRBM_GUI_ACTIVE')
    res.append({'kernel': short(k), 'launches': v.get('FETCH_SIZE', (0, 0))[1], 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                'hbm_bytes_per_launch': rd + wr, 'mfma_busy_frac': (g('SQ_VALU_MFMA_BUSY_CYCLES') / (act / 8.0 * 1024.0)) if act else None,
                'raw': {c: x[0] for c, x in v.items()}})
res.sort(key=lambda r: -r['hbm_bytes_per_launch'] * max(1, r['launches']))
json.dump(res, open('%s/pmc_by_kernel.json' % d, 'w'), indent=1)
pick = {}
for r in res:
    for name in ('notes_bwd_kernel', 'notes_fwd_kernel', 'row_gru_bwd_kernel<128>', 'row_gru_fwd_kernel<128>', 'wgrad_dma_kernel', 'pgru_bwd_sk_kernel<2'):
        if r['kernel'].startswith(name[:-1] if name.endswith('>') else name):          # 'row_gru_bwd_kernel<512' matches the <512, false> instantiation; 'nr::notes_fwd_kernel' below
            pick[name] = {k: r[k] for k in ('launches', 'read_bytes_per_launch', 'write_bytes_per_launch', 'hbm_bytes_per_launch', 'mfma_busy_frac')}
# ---- per-step aggregates (bench.py's roofline: the weight-gradient family, the whole step's executed MFMA work and HBM traffic).
# steps in the trace = launches of the notes forward kernel (one per step)
nsteps = max([r['launches'] for r in res if r['kernel'].startswith('notes_fwd_kernel')] or [1])
fam = [r for r in res if r['kernel'].startswith('wgrad_')]
inst = lambda r: r['raw'].get('SQ_INSTS_MFMA', 0.0) * r['launches']
busy = sum(r['raw'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) * r['launches'] for r in fam)
act = sum(r['raw'].get('GRBM_GUI_ACTIVE', 0.0) * r['launches'] for r in fam)
pick['wgrad_family'] = {'launches_per_step': sum(r['launches'] for r in fam) / nsteps,
                        'hbm_bytes_per_step': sum(r['hbm_bytes_per_launch'] * r['launches'] for r in fam) / nsteps,
                        'executed_tflop_per_step': sum(inst(r) for r in fam) * 16384.0 / nsteps / 1e12,
                        'mfma_busy_frac': (busy / (act / 8.0 * 1024.0)) if act else None,
                        'kernels': sorted({r['kernel'] for r in fam})}
pick['_step'] = {'steps_in_trace': nsteps, 'executed_tflop_per_step': sum(inst(r) for r in res) * 16384.0 / nsteps / 1e12,
                 'hbm_bytes_per_step': sum(r['hbm_bytes_per_launch'] * r['launches'] for r in res) / nsteps,
                 'note': 'SQ_INSTS_MFMA x 16384 FLOP (every MFMA of the bf16 step is v_mfma_f32_16x16x32_bf16) summed over all kernels / steps'}
import subprocess
try:
    pick['_commit'] = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    pick['_commit'] = None                       # (the GPU box has no .git: scripts/gpu_pmc.sh passes PTV_COMMIT)
import os
pick['_commit'] = os.environ.get('PTV_COMMIT') or pick['_commit']
pick['_how'] = ('rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA, three passes) '
                '-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity; scripts/gpu_pmc.sh; corrections in '
                'scripts/pmc_summary.py')
json.dump(pick, open('%s/row_gru_pmc.json' % d, 'w'), indent=1)
json.dump(pick, open('%s/pmc_pick.json' % d, 'w'), indent=1)
for r in res[:30]:
    print('%-64s n=%4d rd=%8.1f MB wr=%8.1f MB mfma_busy=%s' % (r['kernel'][:64], r['launches'], r['read_bytes_per_launch'] / 1e6,
                                                              r['write_bytes_per_launch'] / 1e6,
                                                              ('%.3f' % r['mfma_busy_frac']) if r['mfma_busy_frac'] is not None else '-'))

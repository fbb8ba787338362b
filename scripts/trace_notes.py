"""per-wave event timeline of one workgroup of the wave-role notes GRU forward (ptv_debug_notes_trace): where a note step's time goes.
python scripts/trace_notes.py [dbg bits]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
R, T, H, E = 16384, 15, 512, 128
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
w_hh, w_tok, b_hh = rn(3 * H, H) / H ** 0.5, rn(3 * H, E) / H ** 0.5, rn(3 * H) * 0.1
gc16 = (rn(R, 3 * H) * 0.5).to(bf).view(R, 3 * H // 16, 16).permute(1, 0, 2).contiguous()
emb = rn(T, R, E) * 0.5
HN = torch.zeros(T + 1, R, H, device=dev); HN[0] = rn(R, H) * 0.5
HN16 = torch.zeros(T + 1, R, H, device=dev, dtype=bf)
G = torch.zeros(T, 4, R, H, device=dev, dtype=bf)
pk = (F_.pack_mfma_b(w_hh, pairs=False), F_.pack_mfma_b(w_tok, pairs=False))
buf = torch.zeros(8 * 2048, device=dev, dtype=torch.int64)


def run():
    call('ptv_notes_gru_roles_fwd', ptr(pk[0]), ptr(pk[1]), ptr(b_hh), ptr(gc16), ptr(emb), ptr(HN[0]), ptr(HN16), ptr(G), R, T | (dbg << 8), stream_ptr())


for _ in range(3):
    run()
call('ptv_debug_notes_trace', ptr(buf))
run()
torch.cuda.synchronize()
call('ptv_debug_notes_trace', None)
ev = buf.view(8, 2048).cpu().tolist()
names = {1: 'k-loop start', 2: 'k-loop end', 3: 'slot free', 4: 'dumped', 5: 'B1', 6: 'B2', 11: 'wait fill', 12: 'fill seen', 13: 'cells done',
         14: 'B1', 15: 'h16 written', 16: 'B2'}
t0 = min(e[0] >> 8 for e in ev if e[0])
CLK = 100.0   # s_memtime ticks at 100 MHz on gfx9 (constant clock)
for w in (0, 4):
    print('--- wave %d (%s)' % (w, 'product' if w < 4 else 'cell'))
    prev = None
    for x in ev[w]:
        if not x:
            break
        t, c = (x >> 8) - t0, x & 0xff
        print('  %9.2f us  +%7.2f  %s' % (t / CLK, 0 if prev is None else (t - prev) / CLK, names.get(c, c)))
        prev = t
        if t / CLK > 200:
            break

"""Time of the persistent note-loop kernel (csrc/freerun.hip) alone, with phases switched off, to see where a note step goes.
python scripts/bench_freerun.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev)
P = dict(m.decoder.named_parameters())
R, M = 32 * B, 15 * 32 * B
pk = FF_._free_packs(P, 1024)
w_ih_d, b_ih_d = P['dec_dur_gru.weight_ih_l0'], P['dec_dur_gru.bias_ih_l0']
tab0 = F_.gemm(P['dur_sos_token'].view(1, -1), w_ih_d, bias=b_ih_d, prec=0)
tab = F_.gemm(F_._onehot2x5(dev), w_ih_d, bias=b_ih_d, prec=0)
wl = F_._parr([pk['wg_h'], pk['wg_t'], pk['wp'], pk['wd_h'], pk['wd_p'], pk['wdur'], P['dec_notes_gru.bias_hh_l0'], P['pitch_out_linear.bias'],
               P['dur_hid_linear.bias'], P['dec_dur_gru.bias_hh_l0'], tab0, tab, P['dur_out_linear.weight'], P['dur_out_linear.bias'],
               pk['w_embT'], P['note_embedding.bias']])
bf = torch.bfloat16
GC = torch.randn(B, 1536, device=dev) * 0.3
HN = torch.randn(16, R, 512, device=dev) * 0.3
gates_n = torch.empty(15, 4, R, 512, device=dev, dtype=bf)
pitch = torch.empty(M, 136, device=dev)
HD = torch.empty(6, M, 64, device=dev)
gates_d = torch.empty(5, 4, M, 64, device=dev, dtype=bf)
dur = torch.empty(M, 10, device=dev)
idx = torch.empty(5, M, device=dev, dtype=torch.int32)
TOK = torch.randn(15, R, 128, device=dev) * 0.3
PRED = torch.zeros(16, R, 128, device=dev)
xhat = torch.zeros(B, 32, 16, 6, device=dev, dtype=torch.long)
plen = torch.zeros(R, device=dev, dtype=torch.int32)
dbg_out = torch.zeros(3 * ((B + 15) // 16), device=dev, dtype=torch.long)
S = int(os.environ.get('S', '0'))                  # cluster mode: S workgroups per panel (0 = off)
xch = torch.empty(((B + 15) // 16) * 2 * 16 * 512, device=dev, dtype=bf)
cnt = torch.zeros((B + 15) // 16 + 1, device=dev, dtype=torch.int32)
io = F_._parr([GC, None, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen, None, None, None, None, dbg_out, None, xch, cnt])


def run(flags):
    if S:
        cnt.zero_()                                # (the arrival counters run on from launch to launch of one forward pass: t = 0 here)
        call('ptv_free_note_loop', wl, io, 136, B, 0, 0, flags | (S << 18), stream_ptr())
    else:
        call('ptv_free_note_loop', wl, io, 136, B, 3, 0, flags, stream_ptr())


def timeit(flags, n=10):
    for _ in range(2):
        run(flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, dbg in (('all phases', 0), ('no gate products', 1), ('no GRU cell (P1)', 2), ('no pitch/dur_hid (P2-4) ', 4), ('no duration GRU (P5)', 8),
                  ('only P1', 4 | 8), ('only P5', 2 | 4), ('only P2-4', 2 | 8), ('nothing but token/embed', 2 | 4 | 8)):
    for train in (0, 1):
        t = timeit(train | (dbg << 8))
        print('B=%d %-28s train=%d  %8.1f us per launch  %6.1f us per note step' % (B, name, train, t, t / 15), flush=True)

if S:
    sys.exit(0)
# the same launch with the caches disturbed in between (what the real step loop does: the time-GRU step and its products stream
# ~20 MB of other weights between two note-loop launches) and walking t
big = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
for train in (0, 1):
  for thrash in (False, True):
    for tvar in (False, True):
        tot = 0.0
        for i in range(12):
            if thrash:
                big.zero_()
            tt = (i % 32) if tvar else 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call('ptv_free_note_loop', wl, io, 136, B, tt, 0, train, stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                tot += e0.elapsed_time(e1)
        print('train=%d thrash=%d vary_t=%d: %.1f us per launch' % (train, thrash, tvar, tot / 10 * 1e3), flush=True)

call('ptv_free_note_loop', wl, io, 136, B, 3, 0, 64 << 8, stream_ptr())
torch.cuda.synchronize()
d = dbg_out.cpu().view(-1, 3)
print('per workgroup: xcc', d[:, 0].tolist())
print('hw_id words', [hex(int(v) & 0xffffffff) for v in d[:, 1].tolist()])
print('memtime ticks (100 MHz?)', d[:, 2].tolist())

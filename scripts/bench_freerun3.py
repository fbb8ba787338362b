"""note-loop kernel variants (csrc/freerun.hip), per launch: 4 waves (flag 0x10000) vs producers / heads split over 8 waves, inference
and light-train mode.  python scripts/bench_freerun3.py [B]"""
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
io = F_._parr([GC, None, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen, None, None, None, None, dbg_out, None])




def timeit(flags, n=20):
    def run():
        call('ptv_free_note_loop', wl, io, 136, B, 3, 0, flags, stream_ptr())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, fl in (('4 waves, inference', 0x10000), ('8 waves (producers / heads), inference', 0x20000), ('4 waves, light train', 0x10002), ('8 waves, light train', 0x20002),
                 ('4 waves, full train', 0x10001), ('8 waves, full train', 0x20001)):
    us = timeit(fl)
    print('B=%d  %-42s %8.1f us per launch  (%.1f us per note step)' % (B, name, us, us / 15), flush=True)

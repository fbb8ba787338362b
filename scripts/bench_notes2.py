"""notes GRU forward at the B = 512 train-step shape: the wave-role kernel (csrc/notes_roles.hip) against the 4-wave row kernel
(csrc/notes_persist.hip) -- same inputs, results compared, both timed.  python scripts/bench_notes2.py [R] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 15
H, E = 512, 128
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
w_hh, w_tok, b_hh = rn(3 * H, H) / H ** 0.5, rn(3 * H, E) / H ** 0.5, rn(3 * H) * 0.1
gc_rm, emb = (rn(R, 3 * H) * 0.5).to(bf), rn(T, R, E) * 0.5
gc32 = gc_rm.view(R, 3 * H // 32, 32).permute(1, 0, 2).contiguous()
gc16 = gc_rm.view(R, 3 * H // 16, 16).permute(1, 0, 2).contiguous()
h0 = rn(R, H) * 0.5


def fresh():
    HN = torch.zeros(T + 1, R, H, device=dev); HN[0] = h0
    return HN, torch.zeros(T + 1, R, H, device=dev, dtype=bf), torch.zeros(T, 4, R, H, device=dev, dtype=bf)


old_p = (F_.pack_mfma_b(w_hh, pairs=True), F_.pack_mfma_b(w_tok, pairs=True))
new_p = (F_.pack_mfma_b(w_hh, pairs=False), F_.pack_mfma_b(w_tok, pairs=False))
HNo, HN16o, Go = fresh()
HNn, HN16n, Gn = fresh()


def f_old():
    call('ptv_notes_gru_persist_fwd', ptr(old_p[0]), ptr(old_p[1]), ptr(b_hh), ptr(gc32), ptr(emb), ptr(HNo), ptr(HN16o), ptr(Go), R, T, stream_ptr())


def f_new(flags=0):
    call('ptv_notes_gru_roles_fwd', ptr(new_p[0]), ptr(new_p[1]), ptr(b_hh), ptr(gc16), ptr(emb), ptr(h0), ptr(HN16n), ptr(Gn), R, T | flags, stream_ptr())


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


f_old(); f_new(); torch.cuda.synchronize()
# gate planes to [T][4][R][H]
go = Go.view(T, 4, H // 32, R, 32).permute(0, 1, 3, 2, 4).reshape(T, 4, R, H).float()
gn = Gn.view(T, 4, H // 16, R, 16).permute(0, 1, 3, 2, 4).reshape(T, 4, R, H).float()
print('max |HN16 new - old| %.3e' % (HN16n.float() - HN16o.float()).abs().max().item())
print('max |gates new - old| %.3e   mean %.3e' % ((gn - go).abs().max().item(), (gn - go).abs().mean().item()))
for t in (0, 1, T // 2, T):
    d = (HN16n[t].float() - HN16o[t].float()).abs()
    print('  step %2d: max |dHN16| %.3e  mean %.3e  differing %.4f %%' % (t, d.max().item(), d.mean().item(), 100.0 * (d > 0).float().mean().item()))
runs = [('4-wave row kernel', f_old), ('wave roles, ring 15', lambda: f_new(0)), ('wave roles, ring 12', lambda: f_new(12 << 16)),
        ('wave roles, ring 30', lambda: f_new(30 << 16)),
        ('wave roles, ring 15, no stagger', lambda: f_new(8 << 8)), ('4-wave row kernel', f_old), ('wave roles, ring 15', lambda: f_new(0))]
if os.environ.get('PRIO'):
    for bits in (16, 32, 16, 32, 0):
        runs.append(('prio dbg=%d (16: cells low, 32: products low)' % bits, lambda bits=bits: f_new(bits << 8)))
if os.environ.get('ABL'):
    for bits in (1, 2, 3, 4, 7):
        runs.append(('ABL dbg=%d (1 no st, 2 no ld, 4 no w, 16 no hp ld, 32 no hp ld+st, 48 same no w)' % bits, lambda bits=bits: f_new(bits << 8)))
for name, fn in runs:
    t = timeit(fn)
    print('R=%d T=%d  %-36s %8.1f us  (%.1f us per step)' % (R, T, name, t, t / T), flush=True)

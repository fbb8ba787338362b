"""notes GRU forward with wave roles (csrc/notes_roles.hip) at the B = 512 train-step shape: checked against the per-step kernels on the same
operands, timed at its ring depths, and -- ABL=1 -- with parts of its memory traffic switched off (timing only: which stream costs what).
python scripts/bench_notes2.py [R] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 15
H, E = 512, 128
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
w_hh, w_tok, b_hh = rn(3 * H, H) / H ** 0.5, rn(3 * H, E) / H ** 0.5, rn(3 * H) * 0.1
gc_rm, emb = (rn(R, 3 * H) * 0.5).to(bf), rn(T, R, E) * 0.5
gc16 = gc_rm.view(R, 3 * H // 16, 16).permute(1, 0, 2).contiguous()
h0 = rn(R, H) * 0.5
pk = (F_.pack_mfma_b(w_hh, pairs=False), F_.pack_mfma_b(w_tok, pairs=False))
HN16 = torch.zeros(T + 1, R, H, device=dev, dtype=bf)
G = torch.zeros(T, 4, R, H, device=dev, dtype=bf)
# reference: per-step kernels + separate token product
HN2 = torch.zeros(T + 1, R, H, device=dev); HN2[0] = h0
HN16_2, G2 = torch.zeros_like(HN16), torch.zeros_like(G)
w16, wtok16 = w_hh.to(bf).contiguous(), w_tok.to(bf)
FL = 1 | 2 | 4 | 8 | 16


def f_ref():
    GT = F_.gemm(emb.view(T * R, E), wtok16, prec=1, out_dtype=bf)
    call('ptv_gru_seq_fwd', 1, R, H, T, ptr(GT), R * 3 * H, 3 * H, ptr(gc_rm), 0, 3 * H, ptr(w16), ptr(b_hh), ptr(HN2), ptr(HN16_2), ptr(G2),
         None, 0, None, FL, stream_ptr())


def f_new(flags=0):
    call('ptv_notes_gru_persist_fwd', ptr(pk[0]), ptr(pk[1]), ptr(b_hh), ptr(gc16), ptr(emb), ptr(h0), ptr(HN16), ptr(G), R, T | flags, stream_ptr())


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


f_ref(); f_new(); torch.cuda.synchronize()
gn = G.view(T, 4, H // 16, R, 16).permute(0, 1, 3, 2, 4).reshape(T, 4, R, H).float()
print('max |gates - per-step| %.3e   mean %.3e' % ((gn - G2.float()).abs().max().item(), (gn - G2.float()).abs().mean().item()))
for t in (0, 1, T // 2, T):
    d = (HN16[t].float() - HN16_2[t].float()).abs()
    print('  step %2d: max |dHN16| %.3e  mean %.3e  differing %.4f %%' % (t, d.max().item(), d.mean().item(), 100.0 * (d > 0).float().mean().item()))
runs = [('per-step kernels + token product', f_ref), ('wave roles, ring 15', lambda: f_new(0)), ('wave roles, ring 12', lambda: f_new(12 << 16)),
        ('wave roles, ring 30', lambda: f_new(30 << 16)), ('wave roles, ring 15, no stagger', lambda: f_new(8 << 8)),
        ('wave roles, ring 15', lambda: f_new(0))]
if os.environ.get('PRIO'):
    for bits in (16, 32, 16, 32, 0):
        runs.append(('prio dbg=%d (16: cells low, 32: products low)' % bits, lambda bits=bits: f_new(bits << 8)))
if os.environ.get('ABL'):
    for bits in (1, 2, 3, 4, 7):
        runs.append(('ABL dbg=%d (1 no stores, 2 no operand loads, 4 no weights)' % bits, lambda bits=bits: f_new(bits << 8)))
for name, fn in runs:
    t = timeit(fn)
    print('R=%d T=%d  %-36s %8.1f us  (%.1f us per step)' % (R, T, name, t, t / T), flush=True)

"""notes GRU at the B = 512 train-step shape (R = 16384 rows, 15 steps): persistent kernels (csrc/notes_persist.hip) next to the
per-step kernels + token product they replace.  python scripts/bench_notes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
R, T, H, E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 15, 512, 128
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
w_hh, w_tok, b_hh = rn(3 * H, H) / H ** 0.5, rn(3 * H, E) / H ** 0.5, rn(3 * H) * 0.1
gc, emb = (rn(R, 3 * H) * 0.5).to(bf), rn(T, R, E) * 0.5
gc_rm = gc                                                     # row-major for the per-step kernels
gc = gc.view(R, 3 * H // 16, 16).permute(1, 0, 2).contiguous()   # column-blocked by 16 for the row kernel (ptv_gemm dtypes bit 4)
ext = (rn(T, R, H) * 0.1).to(bf)
ext_b = ext.view(T * R, H // 32, 32).permute(1, 0, 2).contiguous()      # column-blocked for the row BPTT kernel
wg_h, wg_t, wt = F_.pack_mfma_b(w_hh, pairs=False), F_.pack_mfma_b(w_tok, pairs=False), F_.pack_mfma_b(w_hh.t().contiguous(), pairs=True)
HN = torch.zeros(T + 1, R, H, device=dev); HN[0] = rn(R, H) * 0.5
HN16 = torch.zeros(T + 1, R, H, device=dev, dtype=bf)
gates = torch.zeros(T, 4, R, H, device=dev, dtype=bf)
dgi = torch.zeros(T, R, 3 * H, device=dev, dtype=bf); dgh = torch.zeros(T, R, H, device=dev, dtype=bf)
dgh_steps = torch.zeros_like(dgi)
dh0 = torch.zeros(R, H, device=dev)
scratch = torch.empty(lib().ptv_notes_gru_persist_scratch_elems(R), device=dev, dtype=bf)
w16, wt16, wtok16 = w_hh.to(bf).contiguous(), w_hh.t().contiguous().to(bf), w_tok.to(bf)
dhz = torch.empty(2, R, H, device=dev)
FL = 1 | 2 | 4 | 8 | 16


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


def p_fwd():
    call('ptv_notes_gru_persist_fwd', ptr(wg_h), ptr(wg_t), ptr(b_hh), ptr(gc), ptr(emb), ptr(HN), ptr(HN16), ptr(gates), R, T, stream_ptr())


def p_bwd():
    call('ptv_notes_gru_persist_bwd', ptr(wt), ptr(HN16), ptr(gates), ptr(ext_b), ptr(dgi), ptr(dgh), ptr(dh0), ptr(scratch), R, T, None, stream_ptr())


def s_fwd():
    GT = F_.gemm(emb.view(T * R, E), wtok16, prec=1, out_dtype=bf)
    call('ptv_gru_seq_fwd', 1, R, H, T, ptr(GT), R * 3 * H, 3 * H, ptr(gc_rm), 0, 3 * H, ptr(w16), ptr(b_hh), ptr(HN), ptr(HN16), ptr(gates),
         None, 0, None, FL, stream_ptr())


def s_bwd():
    call('ptv_gru_seq_bwd', 1, R, H, T, ptr(HN), ptr(gates), ptr(wt16), ptr(ext), ext.stride(0), ext.stride(1), None, 0, None, 0, 0, 0, None,
         ptr(dgi), ptr(dgh_steps), ptr(dhz), ptr(dh0), 0, FL | 64, stream_ptr())


def p_bwd4():
    call('ptv_notes_bwd_variant', 0)
    p_bwd()
    call('ptv_notes_bwd_variant', 1)


# the two BPTT kernels against each other (same operands): dgi / dgh / dh0
p_fwd(); p_bwd(); torch.cuda.synchronize()
r8 = (dgi.clone(), dgh.clone(), dh0.clone())
p_bwd4(); torch.cuda.synchronize()
for nm, x8, x4 in zip(('dgi', 'dgh', 'dh0'), r8, (dgi, dgh, dh0)):
    d = (x8.float() - x4.float()).abs()
    print('BPTT 8-wave vs 4-wave  %-4s max |d| %.3e  (max |x| %.3e)  differing %.3f %%' % (nm, d.max().item(), x4.float().abs().max().item(), 100.0 * (d > 0).float().mean().item()))

for name, fn in (('forward  per-step kernels + token product', s_fwd), ('forward  persistent', p_fwd),
                 ('backward per-step kernels', s_bwd), ('backward persistent (8 waves)', p_bwd), ('backward persistent (4 waves)', p_bwd4),
                 ('backward persistent (8 waves)', p_bwd)):
    t = timeit(fn)
    print('R=%d T=%d  %-42s %8.1f us  (%.1f us per step)' % (R, T, name, t, t / T), flush=True)

if os.environ.get('ABL'):
    def p_bwd_abl(bits):
        call('ptv_notes_gru_persist_bwd', ptr(wt), ptr(HN16), ptr(gates), ptr(ext_b), ptr(dgi), ptr(dgh), ptr(dh0), ptr(scratch), R, T | (bits << 8), None, stream_ptr())
    for bits, what in ((2, 'no products'), (4, 'no cell loads / stores'), (6, 'neither')):
        t = timeit(lambda: p_bwd_abl(bits))
        print('R=%d T=%d  BPTT 8 waves, ABLATION %-28s %8.1f us  (%.1f us per step)' % (R, T, what, t, t / T), flush=True)

for name, dbg in (('fwd persistent: default', 0), ('fwd persistent: no stagger of pass / k order', 8)):
    def f(dbg=dbg):
        call('ptv_row_gru_persist_fwd', 512, ptr(wg_h), ptr(wg_t), ptr(b_hh), None, ptr(gc), ptr(emb), R * 128, None, ptr(HN), ptr(HN16), ptr(gates),
             None, 0, R, T | (dbg << 8), 0, stream_ptr())
    t = timeit(f)
    print('R=%d T=%d  %-60s %8.1f us  (%.1f us per step)' % (R, T, name, t, t / T), flush=True)

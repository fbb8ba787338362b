"""Duration-GRU kernels alone at the B = 512 shape (M = 245,760 rows, H = 64): forward with / without gate planes, backward from saved
gates / recomputing them; about half of the rows carry no gradient (like the padded note slots of the synthetic batch).
    python scripts/bench_dur.py            (PTV_DUR_FWD_NB / PTV_DUR_BWD_NB override the grids)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
M, H = 480 * 512, 64
g = torch.Generator().manual_seed(12)
k = 1.0 / np.sqrt(H)
U = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev)
w_hh, b_hh, tab0, tab = U(3 * H, H), U(3 * H), U(3 * H), U(2, 3 * H)
w_out, b_out = U(2, H), U(2)
h0 = (torch.randn(M, H, generator=g) * 0.5).to(dev)
ddur = (torch.randn(M, 10, generator=g) * 0.1).to(dev)
ddur[M // 2:] = 0
HD16 = torch.zeros(6, M, H, device=dev, dtype=bf)
gates = torch.empty(5, 4, M, H, device=dev, dtype=bf)
dur = torch.empty(M, 10, device=dev)
idx = torch.empty(5, M, device=dev, dtype=torch.int32)
nblk = int(os.environ.get('PTV_DUR_BWD_NB', min(256, (M + 63) // 64)))
part = torch.zeros(nblk, lib().ptv_dur_gru_bwd_part_size(), device=dev)
dh0 = torch.empty(M, H, device=dev)


def fwd(save):
    call('ptv_dur_gru_fwd', H, M, ptr(h0), H, ptr(w_hh), ptr(b_hh), ptr(tab0), ptr(tab), ptr(w_out), ptr(b_out), None, M * H,
         ptr(HD16[1]), ptr(gates) if save else None, M * H, 4 * M * H, 1, ptr(dur), 10, ptr(idx), M, None, M, stream_ptr())


def bwd(saved):
    call('ptv_dur_gru_bwd', H, M, ptr(gates) if saved else None, M * H, 4 * M * H, ptr(HD16), M * H, 1, ptr(ddur), 10, ptr(w_hh), ptr(w_out),
         ptr(idx), M, ptr(dh0), ptr(part), nblk, ptr(b_hh), ptr(tab0), ptr(tab), stream_ptr())


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


call('ptv_cast_bf16', ptr(h0), ptr(HD16[0]), M * H, stream_ptr())
print('fwd  saving gates   %7.1f us' % timeit(lambda: fwd(True)))
print('fwd  no gates       %7.1f us' % timeit(lambda: fwd(False)))
print('bwd  saved gates    %7.1f us' % timeit(lambda: bwd(True)))
print('bwd  recompute      %7.1f us' % timeit(lambda: bwd(False)))

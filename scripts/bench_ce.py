"""Pitch cross-entropy kernels (csrc/loss.hip) at the B = 512 shape: 245760 rows x 130 classes in 136-float rows, ~53 % of the targets
ignore_index (the padded note slots).  python scripts/bench_ce.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
rows, C, ld = 245760, 130, 136
g = torch.Generator(device=dev).manual_seed(1)
logits = torch.randn(rows, ld, device=dev, generator=g)
tgt = torch.randint(0, C, (rows,), device=dev, generator=g, dtype=torch.int32)
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.53
tgt[torch.rand(rows, device=dev, generator=g) < frac] = 130
nll = torch.zeros(1, device=dev)
gs = torch.ones(1, device=dev)
dl = torch.empty(rows, ld, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


f = lambda: call('ptv_ce_fwd', ptr(logits), ld, ptr(tgt), rows, C, 130, ptr(nll), stream_ptr())
b = lambda: call('ptv_ce_bwd', ptr(logits), ld, ptr(tgt), rows, C, 130, ptr(gs), ptr(dl), ld, stream_ptr())
print('ignore fraction %.2f cap %s: fwd %.1f us  bwd %.1f us' % (frac, os.environ.get('PTV_CE_CAP', '16384'), timeit(f), timeit(b)), flush=True)
# check against torch
nll.zero_(); f(); torch.cuda.synchronize()
ref = torch.nn.functional.cross_entropy(logits[:, :C].double(), tgt.long(), ignore_index=130, reduction='sum')
print('nll %.6f torch %.6f' % (nll.item(), ref.item()))
lg = logits[:, :C].double().clone().requires_grad_(True)
torch.nn.functional.cross_entropy(lg, tgt.long(), ignore_index=130, reduction='sum').backward()
b(); torch.cuda.synchronize()
print('max |dlogits - torch| %.3e' % (dl[:, :C].double() - lg.grad).abs().max().item())

"""texture conv backward alone at B = 512: recomputed convolution vs the forward's arg-max map.  python scripts/bench_txt.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
B, C = 512, 10
W, ld = C * 29, 296
g = torch.Generator().manual_seed(1)
pr = ((torch.rand(B, 32, 128, generator=g) < 0.06).float() * 3).to(dev)
w, b = (torch.randn(C, 48, generator=g) * 0.15).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
feat, arg = torch.empty(B * 8, ld, device=dev), torch.empty(B * 8, W, device=dev, dtype=torch.int8)
dfeat = torch.randn(B * 8, ld, device=dev)
dw, db = torch.zeros(C, 48, device=dev), torch.zeros(C, device=dev)


def timeit(f, n=30):
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


print('fwd            %6.1f us' % timeit(lambda: call('ptv_txt_conv_relu_pool_fwd_rows', ptr(pr), ptr(w), ptr(b), ptr(feat), ld, B, C, None, stream_ptr())))
print('fwd + arg map  %6.1f us' % timeit(lambda: call('ptv_txt_conv_relu_pool_fwd_rows', ptr(pr), ptr(w), ptr(b), ptr(feat), ld, B, C, ptr(arg), stream_ptr())))
print('bwd recompute  %6.1f us' % timeit(lambda: call('ptv_txt_conv_relu_pool_bwd_rows', ptr(pr), ptr(w), ptr(b), ptr(dfeat), ld, ptr(dw), ptr(db), B, C, None, stream_ptr())))
print('bwd arg map    %6.1f us' % timeit(lambda: call('ptv_txt_conv_relu_pool_bwd_rows', ptr(pr), None, None, ptr(dfeat), ld, ptr(dw), ptr(db), B, C, ptr(arg), stream_ptr())))

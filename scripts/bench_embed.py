"""ptv_embed_fwd at B = 512 (262144 notes -> 134 MB of embeddings).  python scripts/bench_embed.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch  # noqa: E402

dev = torch.device('cuda:0')
B, E = 512, 128
x = torch.from_numpy(synth_batch(B, 1)[0]).to(dev)
w = torch.randn(E, 135, device=dev) * 0.1
b = torch.randn(E, device=dev) * 0.1
emb = torch.empty(16, 32, B, E, device=dev)
ln = torch.empty(32 * B, device=dev, dtype=torch.int32)


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


print('%.1f us' % (timeit(lambda: call('ptv_embed_fwd', ptr(x), ptr(w), ptr(b), ptr(emb), ptr(ln), B, E, stream_ptr()))))

"""one weight-gradient product alone (for PMC passes): python scripts/bench_wgrad_one.py M N K dtypes [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402

M, N, K, dt = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device('cuda:0')
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(K, M, device=dev, generator=g).to(bf if dt & 1 else torch.float32)
B = torch.randn(K, N, device=dev, generator=g).to(bf if dt & 2 else torch.float32)
C = torch.zeros(M, N, device=dev)
for _ in range(reps):
    call('ptv_wgrad', M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), 1.0, 1, dt, 0, None, None, 0, 0, stream_ptr())
torch.cuda.synchronize()
print('done')

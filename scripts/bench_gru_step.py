"""micro-benchmark of the fused GRU step kernels (bf16 storage) at one shape: us per fwd / BPTT step
usage: python scripts/bench_gru_step.py M H T [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
dev = torch.device('cuda:0')
M, H, T = [int(v) for v in sys.argv[1:4]]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
use_gi2 = len(sys.argv) > 5 and sys.argv[5] == 'gi2'
bf = torch.bfloat16
gi = (torch.randn(T, M, 3 * H, device=dev) * 0.5).to(bf)
gi2 = (torch.randn(M, 3 * H, device=dev) * 0.5).to(bf) if use_gi2 else None
w = (torch.randn(3 * H, H, device=dev) / H ** 0.5)
w16, wt16 = w.to(bf).contiguous(), w.t().contiguous().to(bf)
b = torch.randn(3 * H, device=dev) * 0.1
hall = torch.zeros(T + 1, M, H, device=dev); hall16 = torch.zeros(T + 1, M, H, device=dev, dtype=bf)
gates = torch.empty(T, 4, M, H, device=dev, dtype=bf)
dgi = torch.empty(T, M, 3 * H, device=dev, dtype=bf); dgh = torch.empty_like(dgi)
dhz = torch.empty(2, M, H, device=dev); dh0 = torch.empty(M, H, device=dev)
dh_ext = torch.randn(T, M, H, device=dev) * 0.1
FL = 1 | 2 | 8 | 16 | (4 if use_gi2 else 0)
def fwd():
    call('ptv_gru_seq_fwd', 1, M, H, T, ptr(gi), M * 3 * H, 3 * H, ptr(gi2), 0, 3 * H if use_gi2 else 0, ptr(w16), ptr(b), ptr(hall), ptr(hall16),
         ptr(gates), None, 0, None, FL, stream_ptr())
def bwd():
    call('ptv_gru_seq_bwd', 1, M, H, T, ptr(hall), ptr(gates), ptr(wt16), ptr(dh_ext), dh_ext.stride(0), dh_ext.stride(1),
         None, 0, None, 0, 0, 0, None, ptr(dgi), ptr(dgh), ptr(dhz), ptr(dh0), 0, FL, stream_ptr())
for name, fn in (('fwd', fwd), ('bwd', bwd)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * T)
    if name == 'fwd':
        by = M * H * (2 + 3 * 2 * (2 if use_gi2 else 1) + 4 + 4 + 2 + 4 * 2) + 3 * H * H * 2
    else:
        by = M * H * (3 * 2 + 4 * 2 + 4 + 4 + 4 + 6 * 2 + 4) + 3 * H * H * 2
    print(f'{name} M={M} H={H} T={T}: {us:.1f} us/step   {by / us / 1e6:.2f} TB/s algorithmic  {2 * M * 3 * H * H / us / 1e6:.1f} TFLOP/s')

"""GEMM micro-benchmark with operand dtypes: python scripts/bench_gemm2.py M N K ta tb a16 b16 c16 [acc]
prints us, TFLOP/s and the algorithmic HBM rate"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphonic_chord_texture_disentanglement_amd import functional as F_
dev = torch.device('cuda:0')
M, N, K, ta, tb, a16, b16, c16 = [int(v) for v in sys.argv[1:9]]
acc = len(sys.argv) > 9 and int(sys.argv[9])
splitk = int(sys.argv[10]) if len(sys.argv) > 10 else 0
bf = torch.bfloat16
a = torch.randn((K, M) if ta else (M, K), device=dev).to(bf if a16 else torch.float32)
b = torch.randn((K, N) if tb else (N, K), device=dev).to(bf if b16 else torch.float32)
out = torch.zeros(M, N, device=dev, dtype=bf if c16 else torch.float32)
def run():
    F_.gemm(a, b, out, ta=bool(ta), tb=bool(tb), prec=1, acc=bool(acc), splitk=splitk)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
by = a.numel() * a.element_size() + b.numel() * b.element_size() + out.numel() * out.element_size() * (2 if acc else 1)
print(f'M={M} N={N} K={K} ta={ta} tb={tb} a16={a16} b16={b16} c16={c16} acc={int(acc)} splitk={splitk}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s  {by/us/1e6:.2f} TB/s')

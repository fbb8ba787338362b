"""reference point: what the vendor GEMM library (through torch.matmul, bf16) does on the streaming shapes of the step"""
import torch
dev = torch.device('cuda:0'); bf = torch.bfloat16
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K, mode) in [(245760, 1536, 128, 'nt'), (262144, 384, 128, 'nt'), (245760, 130, 512, 'nt'), (245760, 512, 1536, 'nn'),
                        (245760, 512, 1536, 'nt'), (1536, 512, 245760, 'tn'), (3072, 1024, 16384, 'tn'), (245760, 512, 136, 'nn')]:
    if mode == 'nt':
        a = torch.randn(M, K, device=dev, dtype=bf); b = torch.randn(N, K, device=dev, dtype=bf); fn = lambda: a @ b.t()
    elif mode == 'nn':
        a = torch.randn(M, K, device=dev, dtype=bf); b = torch.randn(K, N, device=dev, dtype=bf); fn = lambda: a @ b
    else:
        a = torch.randn(K, M, device=dev, dtype=bf); b = torch.randn(K, N, device=dev, dtype=bf); fn = lambda: a.t() @ b
    us = t(fn)
    by = 2 * (M * K + N * K + M * N)
    print(f'{mode} M={M} N={N} K={K}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s  {by/us/1e6:.2f} TB/s')

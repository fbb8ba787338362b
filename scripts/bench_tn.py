import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphonic_chord_texture_disentanglement_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
shapes = [(1536, 512, 245760), (1536, 128, 245760), (130, 512, 245760), (3072, 1024, 16384), (1536, 1024, 16384), (192, 64, 1228800), (384, 128, 262144)]
for M, N, K in shapes:
    a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev); out = torch.zeros(M, N, device=dev)
    for sk in (0, 4, 8, 16, 32, 64):
        dt = timeit(lambda: ops.gemm(a, b, out, trans_a=True, trans_b=True, accumulate=True, prec='bf16', splitk=sk))
        print('TN M=%5d N=%5d K=%7d splitk=%3d  %8.1f us %7.1f TF' % (M, N, K, sk, dt*1e6, 2.0*M*N*K/dt/1e12), flush=True)
    del a, b, out

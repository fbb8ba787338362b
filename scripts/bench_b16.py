import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphonic_chord_texture_disentanglement_amd import functional as F_
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K, ta, tb, label) in [(245760, 512, 1536, 0, 1, 'dX NN'), (16384, 1536, 1024, 0, 0, 'gi NT'), (245760, 130, 512, 0, 0, 'pitch NT')]:
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((K, N) if tb else (N, K), device=dev)
    out = torch.empty(M, N, device=dev)
    for adt in (torch.float32, torch.bfloat16):
        for bdt in (torch.float32, torch.bfloat16):
            aa, bb = a.to(adt), b.to(bdt)
            dt = timeit(lambda: F_.gemm(aa, bb, out, ta=bool(ta), tb=bool(tb), prec=1))
            print('%-10s A=%s B=%s %8.1f us %7.1f TF' % (label, str(adt)[6:], str(bdt)[6:], dt * 1e6, 2.0 * M * N * K / dt / 1e12), flush=True)

"""Weight-gradient products of the B=512 train step (shapes and operand dtypes as scripts/gemm_shapes.py logs them): time and
error vs torch on the bf16-rounded operands.  PTV_WGRAD=0 python scripts/bench_wgrad.py  -> the generic TN GEMM;  default ->
csrc/wgrad.hip (ptv_gemm routes there)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
# (M, N, K, dtypes)  dtypes bit0: A bf16, bit1: B bf16
SHAPES = [(3072, 1024, 4096, 3), (1536, 512, 245760, 3), (3072, 256, 4096, 1), (1536, 128, 245760, 1), (130, 512, 245760, 2),
          (384, 128, 262144, 1), (384, 128, 262144, 3), (1536, 1024, 16384, 2), (3072, 1024, 16384, 3), (1536, 512, 4096, 1),
          (64, 512, 245760, 2), (3072, 256, 16384, 1), (128, 135, 262144, 0), (512, 1024, 16384, 2), (64, 130, 245760, 0),
          (3072, 36, 4096, 1), (1000, 290, 4096, 0), (256, 2048, 512, 0), (256, 1000, 4096, 0), (12, 512, 4096, 0)]
SWEEP = len(sys.argv) > 1 and sys.argv[1] == 'sweep'


def timeit(fn, n=10):
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


tot = 0.0
for M, N, K, dt in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N)
    lda = M + (8 - M % 8) % 8 if M % 8 else M
    A = torch.randn(K, lda, device=dev, generator=g)[:, :M]
    ldb = N + (8 - N % 8) % 8
    B = torch.randn(K, ldb, device=dev, generator=g)[:, :N]
    Ad = A.to(bf) if dt & 1 else A
    Bd = B.to(bf) if dt & 2 else B
    if dt & 1:
        Ad = Ad.contiguous() if M % 8 == 0 else torch.nn.functional.pad(Ad, (0, lda - M))[:, :M]
    if dt & 2:
        Bd = Bd.contiguous() if N % 8 == 0 else torch.nn.functional.pad(Bd, (0, ldb - N))[:, :N]
    C = torch.zeros(M, N, device=dev)

    def run():
        call('ptv_gemm', 1, 1, 1, M, N, K, ptr(Ad), Ad.stride(0), ptr(Bd), Bd.stride(0), ptr(C), C.stride(0), None, 1.0, 1, 0, 0, dt,
             stream_ptr())
    C.zero_()
    run()
    torch.cuda.synchronize()
    ref = torch.zeros(M, N, device=dev, dtype=torch.float64)
    ch = 32768
    for k0 in range(0, K, ch):
        ref += (A[k0:k0 + ch].to(bf).double().t() @ B[k0:k0 + ch].to(bf).double())
    err = (C.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    us = timeit(run)
    tot += us
    if SWEEP:
        res = []
        for sl in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
            if sl * 128 > K:
                break
            res.append((sl, timeit(lambda: call('ptv_wgrad', M, N, K, ptr(Ad), Ad.stride(0), ptr(Bd), Bd.stride(0), ptr(C), C.stride(0), 1.0, 1,
                                                  dt, sl, None, None, 0, 0, stream_ptr()), 5)))
        print('   slabs: ' + ' '.join('%d:%.0f' % r for r in res))
    byt = K * (M * (2 if dt & 1 else 4) + N * (2 if dt & 2 else 4))
    print('TN M=%5d N=%5d K=%7d dt=%d  %8.1f us %7.1f TF/s %6.2f TB/s  rel err %.2e' % (M, N, K, dt, us, 2.0 * M * N * K / us / 1e6, byt / us / 1e6, err),
          flush=True)
    del A, B, Ad, Bd, C, ref
print('sum %.0f us' % tot)

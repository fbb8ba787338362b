"""Micro-benchmark of the MFMA GEMM / GRU-step kernels at the shapes of the train step (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polyphonic_chord_texture_disentanglement_amd import ops

dev = torch.device('cuda:0')

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

shapes = [  # (M, N, K, transA, transB, label)
    (16384, 1536, 1024, 0, 0, 'notes gi_const NT'),
    (245760, 1536, 128, 0, 0, 'notes gi_tok NT'),
    (245760, 130, 512, 0, 0, 'pitch_out NT'),
    (16384, 3072, 512, 0, 0, 'time gi NT'),
    (245760, 512, 1536, 0, 1, 'dX notes NN'),
    (1536, 512, 245760, 1, 1, 'dW_hh notes TN'),
    (1536, 128, 245760, 1, 1, 'dW_ih tok TN'),
    (4096, 4096, 4096, 0, 0, 'square NT'),
]
for prec in ('bf16', 'fp32'):
    for M, N, K, ta, tb, label in shapes:
        a = torch.randn((K, M) if ta else (M, K), device=dev)
        b = torch.randn((K, N) if tb else (N, K), device=dev)
        out = torch.empty(M, N, device=dev)
        dt = timeit(lambda: ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), prec=prec))
        print('%-5s %-20s M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s' % (prec, label, M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12), flush=True)
        del a, b, out

# GRU step sequences
for prec in ('bf16', 'fp32'):
    for M, H, T, label in ((16384, 512, 15, 'notes gru'), (512, 1024, 32, 'time gru'), (16384, 128, 16, 'emb gru')):
        gi = torch.randn(T, M, 3 * H, device=dev)
        w = torch.randn(3 * H, H, device=dev) / H ** 0.5
        b = torch.randn(3 * H, device=dev)
        hall = torch.zeros(T + 1, M, H, device=dev)
        gates = torch.empty(T, 4, M, H, device=dev)
        dt = timeit(lambda: ops.gru_seq_fwd(gi, w, b, hall, gates, prec=prec), n=5)
        fl = 2.0 * M * H * 3 * H * T
        print('%-5s %-10s fwd M=%6d H=%4d T=%2d %8.1f us/step %7.1f TFLOP/s' % (prec, label, M, H, T, dt / T * 1e6, fl / dt / 1e12), flush=True)
        ext = torch.randn(T, M, H, device=dev)
        dt = timeit(lambda: ops.gru_seq_bwd(hall, gates, w, dh_ext=ext, prec=prec), n=5)
        print('%-5s %-10s bwd M=%6d H=%4d T=%2d %8.1f us/step %7.1f TFLOP/s' % (prec, label, M, H, T, dt / T * 1e6, fl / dt / 1e12), flush=True)
        del gi, hall, gates, ext

import os, sys
sys.path.insert(0, os.getcwd())
import torch
from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
dev = torch.device('cuda:0'); bf = torch.bfloat16
for (M, N, K, unit, adt, bdt) in [(200, 512, 7680, 512, bf, bf), (64, 130, 7680, 512, torch.float32, torch.float32), (200, 512, 15 * 16384, 16384, bf, bf),
                                  (1024, 512, 7680, 512, bf, bf), (512, 512, 7680, 512, bf, bf), (1536, 128, 7680, 512, bf, torch.float32)]:
    for top in (0, 3, 6, 13):
        g = torch.Generator(device=dev).manual_seed(1)
        A = torch.randn(K, M, device=dev, generator=g).to(adt); B = torch.randn(K, N + 6, device=dev, generator=g).to(bdt)[:, :N]
        lim = (top + 1) * unit
        A[lim:] = float('nan'); B[lim:] = float('nan')
        C = torch.zeros(M, N, device=dev); cs = torch.zeros(M, device=dev)
        kt = torch.tensor([top], device=dev, dtype=torch.int32)
        call('ptv_wgrad', M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), N, 1.0, 0, (1 if adt == bf else 0) | (2 if bdt == bf else 0), 0, ptr(cs), ptr(kt), unit, 0, stream_ptr())
        torch.cuda.synchronize()
        want = A[:lim].float().t() @ B[:lim].float()
        print(M, N, K, 'top', top, 'finite', bool(torch.isfinite(C).all()), bool(torch.isfinite(cs).all()), 'err %.3g' % ((C - want).abs().max() / want.abs().max()).item() if torch.isfinite(C).all() else '', flush=True)

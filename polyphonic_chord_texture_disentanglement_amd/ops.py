"""Thin tensor-level wrappers over the C ABI (one python function per entry point).

Tensors are plumbing only: every function passes raw device pointers, sizes and the current
HIP stream to libptvae_hip.so.  No function here computes anything with torch ops.
"""
import torch

from . import _lib
from ._lib import call, prec_code, ptr, stream_ptr


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'row-major 2-D view expected'
    return t.stride(0)


def gemm(a, b, out=None, *, trans_a=False, trans_b=False, bias=None, alpha=1.0, accumulate=False,
         act=0, prec='fp32', splitk=0):
    """out[M,N] = act(alpha * A.B^T + bias) (+ out).  a: [M,K] (or [K,M] if trans_a);
    b: [N,K] nn.Linear layout (or [K,N] if trans_b)."""
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    assert out.shape == (M, N)
    call('ptv_gemm', prec_code(prec), int(trans_a), int(trans_b), M, N, K, ptr(a), _ld(a), ptr(b),
         _ld(b), ptr(out), _ld(out), ptr(bias), float(alpha), int(accumulate), int(act), int(splitk),
         stream_ptr())
    return out


def gru_seq_fwd(gi, w_hh, b_hh, hall, gates=None, *, gi2=None, lengths=None, reverse=False,
                prec='fp32'):
    """gi: [T,M,3H] (any step/row strides), hall: [T+1,M,H] contiguous with slot 0 = h0."""
    T, M, H3 = gi.shape
    H = H3 // 3
    assert hall.shape == (T + 1, M, H) and hall.is_contiguous()
    assert gi.stride(2) == 1
    if gates is not None:
        assert gates.shape == (T, 4, M, H) and gates.is_contiguous()
    g2 = (ptr(gi2), gi2.stride(0), gi2.stride(1)) if gi2 is not None else (None, 0, 0)
    call('ptv_gru_seq_fwd', prec_code(prec), M, H, T, ptr(gi), gi.stride(0), gi.stride(1), *g2,
         ptr(w_hh), ptr(b_hh), ptr(hall), ptr(gates), ptr(lengths), int(reverse), stream_ptr())
    return hall


def gru_seq_bwd(hall, gates, w_hh, *, dh_ext=None, dh_last=None, reverse=False, prec='fp32',
                need_dh0=True):
    """Returns dgi [T,M,3H] (time order), dgh [T,M,3H] (processing order), dh0 [M,H]."""
    T1, M, H = hall.shape
    T = T1 - 1
    dev = hall.device
    dgi = torch.empty(T, M, 3 * H, device=dev, dtype=torch.float32)
    dgh = torch.empty(T, M, 3 * H, device=dev, dtype=torch.float32)
    dhz = torch.empty(2, M, H, device=dev, dtype=torch.float32)
    dh0 = torch.empty(M, H, device=dev, dtype=torch.float32) if need_dh0 else None
    ext = (ptr(dh_ext), dh_ext.stride(0), dh_ext.stride(1)) if dh_ext is not None else (None, 0, 0)
    last = (ptr(dh_last), dh_last.stride(0)) if dh_last is not None else (None, 0)
    call('ptv_gru_seq_bwd', prec_code(prec), M, H, T, ptr(hall), ptr(gates), ptr(w_hh), *ext, *last,
         ptr(dgi), ptr(dgh), ptr(dhz), ptr(dh0), int(reverse), stream_ptr())
    return dgi, dgh, dh0

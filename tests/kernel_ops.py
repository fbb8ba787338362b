"""Test helper: tensor-level aliases over functional.py's kernel wrappers with the keyword names the kernel parity tests use
(not part of the product package)."""
import torch

from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd._lib import prec_code


def gemm(a, b, out=None, *, trans_a=False, trans_b=False, bias=None, alpha=1.0, accumulate=False, act=0,
         prec='fp32', splitk=0):
    return F_.gemm(a, b, out, ta=trans_a, tb=trans_b, bias=bias, alpha=alpha, acc=accumulate, act=act,
                   prec=prec_code(prec), splitk=splitk)


def gru_seq_fwd(gi, w_hh, b_hh, hall, gates=None, *, gi2=None, lengths=None, reverse=False, prec='fp32'):
    """gi: [T,M,3H]; hall: [T+1,M,H] contiguous with slot 0 = h0; gates: [T,4,M,H] or None."""
    T, M, H3 = gi.shape
    assert gi.stride(2) == 1 and hall.shape == (T + 1, M, H3 // 3) and hall.is_contiguous()
    kw = {}
    if gi2 is not None:
        kw = dict(gi2=gi2, gi2_step=gi2.stride(0), gi2_ld=gi2.stride(1))
    F_.gru_fwd(prec_code(prec), gi, gi.stride(0), gi.stride(1), w_hh, b_hh, hall, gates, lengths=lengths,
               reverse=reverse, **kw)
    return hall


def gru_seq_bwd(hall, gates, w_hh, *, dh_ext=None, dh_last=None, reverse=False, prec='fp32', need_dh0=True):
    return F_.gru_bwd(prec_code(prec), hall, gates, w_hh, dh_ext=dh_ext, dh_last=dh_last, reverse=reverse,
                      need_dh0=need_dh0)

"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import ptvae_oracle as orc

pytestmark = pytest.mark.gpu

TOL = {'fp32': 2e-5, 'bf16': 3e-2}


def _dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('M,N,K', [(37, 130, 512), (512, 3072, 1024), (300, 64, 642), (1000, 290, 36),
                                   (2048, 1536, 128), (16, 5, 64), (130, 135, 7)])
def test_gemm_nt_bias(prec, M, N, K):
    from polyphonic_chord_texture_disentanglement_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = orc.linear(a, w, b)
    out = ops.gemm(a.to(_dev()), w.to(_dev()), bias=b.to(_dev()), prec=prec).cpu()
    err = (out - ref).abs().max().item()
    assert err < TOL[prec] * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_gemm_layouts_accumulate_exp_splitk(prec):
    from polyphonic_chord_texture_disentanglement_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    # NN: dX = dY . W     (asymmetric shapes catch transposes)
    dy = torch.randn(333, 96, generator=g)
    w = torch.randn(96, 200, generator=g) / 10
    out = ops.gemm(dy.to(dev), w.to(dev), trans_b=True, prec=prec).cpu()
    ref = dy @ w
    assert (out - ref).abs().max() < TOL[prec] * ref.abs().max()
    # TN: dW = dY^T . X with huge K -> split-K atomics, accumulate into existing grads
    K = 20000
    dy = torch.randn(K, 96, generator=g)
    x = torch.randn(K, 130, generator=g)
    base = torch.randn(96, 130, generator=g)
    out = base.clone().to(dev)
    ops.gemm(dy.to(dev), x.to(dev), out, trans_a=True, trans_b=True, accumulate=True, prec=prec)
    ref = base + dy.t() @ x
    assert (out.cpu() - ref).abs().max() < TOL[prec] * ref.abs().max()
    out2 = ops.gemm(dy.to(dev), x.to(dev), trans_a=True, trans_b=True, prec=prec)     # auto split, zero-init
    assert (out2.cpu() - dy.t() @ x).abs().max() < TOL[prec] * ref.abs().max()
    # exp epilogue + strided output rows
    a = torch.randn(50, 64, generator=g) / 8
    w = torch.randn(16, 64, generator=g) / 8
    b = torch.randn(16, generator=g) / 8
    big = torch.zeros(50, 40, device=dev)
    ops.gemm(a.to(dev), w.to(dev), big[:, 8:24], bias=b.to(dev), act=1, prec=prec)
    ref = torch.exp(orc.linear(a, w, b))
    assert (big[:, 8:24].cpu() - ref).abs().max() < TOL[prec] * ref.abs().max()
    assert big[:, :8].abs().max() == 0 and big[:, 24:].abs().max() == 0


def _gru_oracle_seq(x, h0, w_ih, w_hh, b_ih, b_hh, lengths, reverse):
    T = x.shape[0]
    h = h0
    hs = []
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        nh = orc.gru_cell(x[t], h, w_ih, w_hh, b_ih, b_hh)
        h = nh if lengths is None else torch.where((t < lengths).unsqueeze(1), nh, h)
        hs.append(h)
    return torch.stack(hs)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('M,H,I,T,masked,reverse', [(96, 64, 20, 5, False, False), (700, 128, 128, 16, True, False),
                                                    (700, 128, 128, 16, True, True), (512, 1024, 36, 4, False, True),
                                                    (2100, 512, 128, 3, False, False)])
def test_gru_seq_fwd_bwd(prec, M, H, I, T, masked, reverse):
    from polyphonic_chord_texture_disentanglement_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + H + T)
    k = 1.0 / np.sqrt(H)
    w_ih = ((torch.rand(3 * H, I, generator=g) * 2 - 1) * k).requires_grad_()
    w_hh = ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k).requires_grad_()
    b_ih = ((torch.rand(3 * H, generator=g) * 2 - 1) * k).requires_grad_()
    b_hh = ((torch.rand(3 * H, generator=g) * 2 - 1) * k).requires_grad_()
    x = torch.randn(T, M, I, generator=g).requires_grad_()
    h0 = (torch.randn(M, H, generator=g) * 0.5).requires_grad_()
    lengths = torch.randint(1, T + 1, (M,), generator=g) if masked else None
    hs = _gru_oracle_seq(x, h0, w_ih, w_hh, b_ih, b_hh, lengths, reverse)
    wgt = torch.randn(T, M, H, generator=g)
    wl = torch.randn(M, H, generator=g)
    ((hs * wgt).sum() + (hs[-1] * wl).sum()).backward()

    d = lambda t: t.detach().to(dev)
    gi = ops.gemm(d(x).view(T * M, I), d(w_ih), bias=d(b_ih), prec=prec).view(T, M, 3 * H)
    hall = torch.empty(T + 1, M, H, device=dev)
    hall[0] = d(h0)
    gates = torch.empty(T, 4, M, H, device=dev)
    ops.gru_seq_fwd(gi, d(w_hh), d(b_hh), hall, gates, lengths=None if lengths is None else lengths.int().to(dev),
                    reverse=reverse, prec=prec)
    tol = TOL[prec]
    assert (hall[1:].cpu() - hs.detach()).abs().max() < tol

    dgi, dgh, dh0 = ops.gru_seq_bwd(hall, gates, d(w_hh), dh_ext=wgt.to(dev), dh_last=wl.to(dev), reverse=reverse, prec=prec)
    gtol = tol * 20
    assert (dh0.cpu() - h0.grad).abs().max() < gtol * max(1.0, h0.grad.abs().max().item())
    dx = ops.gemm(dgi.view(T * M, 3 * H), d(w_ih), trans_b=True, prec=prec).view(T, M, I)
    assert (dx.cpu() - x.grad).abs().max() < gtol * max(1.0, x.grad.abs().max().item())
    dwhh = ops.gemm(dgh.view(T * M, 3 * H), hall[:T].reshape(T * M, H), trans_a=True, trans_b=True, prec=prec)
    assert (dwhh.cpu() - w_hh.grad).abs().max() < gtol * max(1.0, w_hh.grad.abs().max().item())
    dwih = ops.gemm(dgi.view(T * M, 3 * H), d(x).view(T * M, I), trans_a=True, trans_b=True, prec=prec)
    assert (dwih.cpu() - w_ih.grad).abs().max() < gtol * max(1.0, w_ih.grad.abs().max().item())
    assert (dgi.float().sum((0, 1)).cpu() - b_ih.grad).abs().max() < gtol * max(1.0, b_ih.grad.abs().max().item())
    assert (dgh.float().sum((0, 1)).cpu() - b_hh.grad).abs().max() < gtol * max(1.0, b_hh.grad.abs().max().item())

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
    import kernel_ops as ops
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
    import kernel_ops as ops
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
    import kernel_ops as ops
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


@pytest.mark.parametrize('M,H,T,use_gi2', [(1000, 128, 4, True), (4096, 512, 2, True), (4100, 512, 2, False), (512, 1024, 3, True)])
def test_gru_seq_bf16_storage_fast_path(M, H, T, use_gi2):
    """the bf16-storage fast path of the step kernels (bf16 gi / gi2 / gates / state and weight shadows: prefetching
    row-staged epilogues, 64x64 and 64x32 tiles, ragged M) against the fp32 oracle cell, forward and BPTT"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + H + T)
    k = 1.0 / np.sqrt(H)
    w_hh = ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k).to(bf).float().requires_grad_()      # bf16-representable weights
    b_hh = ((torch.rand(3 * H, generator=g) * 2 - 1) * k).requires_grad_()
    gi = (torch.randn(T, M, 3 * H, generator=g) * 0.5).to(bf).float().requires_grad_()
    gi2 = (torch.randn(M, 3 * H, generator=g) * 0.5).to(bf).float() if use_gi2 else None
    h0 = (torch.randn(M, H, generator=g) * 0.5).requires_grad_()
    dh_ext = torch.randn(T, M, H, generator=g) * 0.1
    # oracle: GRU cell with precomputed input-side pre-activations (b_ih folded into gi)
    h, hs = h0, []
    for t in range(T):
        x = gi[t] + (gi2 if use_gi2 else 0)
        gh = orc.linear(h, w_hh, b_hh)
        r = torch.sigmoid(x[:, :H] + gh[:, :H]); z = torch.sigmoid(x[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(x[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        hs.append(h)
    hs = torch.stack(hs)
    (hs * dh_ext).sum().backward()

    d = lambda t: t.detach().to(dev)
    gi_d = d(gi).to(bf); gi2_d = d(gi2).to(bf) if use_gi2 else None
    w16, wt16 = d(w_hh).to(bf).contiguous(), d(w_hh).t().contiguous().to(bf)
    hall = torch.zeros(T + 1, M, H, device=dev); hall[0] = d(h0)
    hall16 = torch.zeros(T + 1, M, H, device=dev, dtype=bf)
    gates = torch.empty(T, 4, M, H, device=dev, dtype=bf)
    FL = 1 | 2 | 8 | 16 | (4 if use_gi2 else 0)
    b_d = d(b_hh)
    call('ptv_gru_seq_fwd', 1, M, H, T, ptr(gi_d), M * 3 * H, 3 * H, ptr(gi2_d), 0, 3 * H if use_gi2 else 0, ptr(w16), ptr(b_d),
         ptr(hall), ptr(hall16), ptr(gates), None, 0, None, FL, stream_ptr())
    assert (hall[1:].cpu() - hs.detach()).abs().max() < 3e-2
    assert (hall16[1:].float() - hall[1:]).abs().max() < 1e-2                      # the shadow is the rounded state
    dgi = torch.empty(T, M, 3 * H, device=dev, dtype=bf); dgh = torch.empty_like(dgi)
    dhz = torch.empty(2, M, H, device=dev); dh0 = torch.empty(M, H, device=dev)
    de = dh_ext.to(dev)
    call('ptv_gru_seq_bwd', 1, M, H, T, ptr(hall), ptr(gates), ptr(wt16), ptr(de), de.stride(0), de.stride(1),
         None, 0, None, 0, 0, 0, None, ptr(dgi), ptr(dgh), ptr(dhz), ptr(dh0), 0, FL, stream_ptr())
    tol = 0.05
    assert (dh0.cpu() - h0.grad).abs().max() < tol * max(1.0, h0.grad.abs().max().item())
    assert (dgi.float().cpu() - gi.grad).abs().max() < tol * max(1.0, gi.grad.abs().max().item())
    db = dgh.float().sum((0, 1)).cpu()
    assert (db - b_hh.grad).abs().max() < tol * max(1.0, b_hh.grad.abs().max().item())


@pytest.mark.parametrize('NC,M,H,T,masked,use_gi2', [(1, 512, 1024, 6, False, True), (2, 512, 1024, 4, True, False),
                                                     (1, 512, 512, 8, False, True), (1, 1024, 1024, 3, False, False),
                                                     (1, 300, 512, 5, True, False), (2, 100, 1024, 3, False, True),
                                                     (4, 512, 1024, 3, False, False), (4, 300, 1024, 2, True, False),
                                                     (1, 128, 1024, 5, False, False), (2, 1024, 1024, 2, False, False)])
@pytest.mark.parametrize('splitk', [0, 2, 4])
def test_gru_persistent_kernels_vs_oracle_and_step_kernels(NC, M, H, T, masked, use_gi2, splitk, monkeypatch):
    """csrc/gru_persist.hip (one weight-stationary launch per sequence, state exchanged between workgroups per step)
    against the fp32 oracle cell and against the per-step kernels of csrc/gru.hip on the same bf16 operands: forward
    states + saved gates, BPTT dgi / dgh / dh0; NC chains per launch, reversed chains, masked rows, ragged M.
    splitk: the BPTT as split-K teams of 2 / 4 workgroups (pgru_bwd_sk_kernel; up to 512 rows per workgroup: four chains
    of 512 rows in one launch) or the round-2 kernel (0)."""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    monkeypatch.setattr(F_, 'PERSIST_SPLITK', splitk)
    if splitk == 0 and not lib().ptv_gru_persist_supported(NC, M, H):
        pytest.skip('shape does not fit one workgroup per CU on this device')
    if splitk and not lib().ptv_gru_persist_splitk_supported(NC, M, H, splitk):
        pytest.skip('shape does not fit the split-K teams on this device')
    import os
    if 'PTV_LP' in os.environ:                       # experiment switch: how consumers read the exchanged operand
        lib().ptv_gru_persist_load_policy(int(os.environ['PTV_LP']))
    g = torch.Generator().manual_seed(NC * 1000 + M + H + T)
    k = 1.0 / np.sqrt(H)
    d = lambda t: t.detach().to(dev)
    fw, bw, ref = [], [], []
    for ci in range(NC):
        reverse = bool(ci & 1)
        w_hh = ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k).to(bf).float().requires_grad_()
        b_hh = ((torch.rand(3 * H, generator=g) * 2 - 1) * k).requires_grad_()
        gi = (torch.randn(T, M, 3 * H, generator=g) * 0.5).to(bf).float().requires_grad_()
        gi2 = (torch.randn(M, 3 * H, generator=g) * 0.5).to(bf).float() if use_gi2 else None
        h0 = (torch.randn(M, H, generator=g) * 0.5).requires_grad_()
        lengths = torch.randint(1, T + 1, (M,), generator=g) if masked else None
        dh_ext = torch.randn(T, M, H, generator=g) * 0.1
        dh_last = torch.randn(M, H + 8, generator=g)[:, :H] * 0.1              # strided rows
        h, hs = h0, []
        for s_ in range(T):
            t = T - 1 - s_ if reverse else s_
            x = gi[t] + (gi2 if use_gi2 else 0)
            gh = orc.linear(h, w_hh, b_hh)
            r = torch.sigmoid(x[:, :H] + gh[:, :H]); z = torch.sigmoid(x[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(x[:, 2 * H:] + r * gh[:, 2 * H:])
            nh = (1 - z) * n + z * h
            h = nh if lengths is None else torch.where((t < lengths).unsqueeze(1), nh, h)
            hs.append(h)
        hs = torch.stack(hs)
        ((hs * dh_ext).sum() + (hs[-1] * dh_last).sum()).backward()
        ref.append((hs.detach(), h0.grad, gi.grad, b_hh.grad))
        hall = torch.zeros(T + 1, M, H, device=dev); hall[0] = d(h0)
        c = dict(gi=d(gi).to(bf), gi_step=M * 3 * H, gi_ld=3 * H, gi2=d(gi2).to(bf) if use_gi2 else None, gi2_step=0,
                 gi2_ld=3 * H if use_gi2 else 0, w16=d(w_hh).to(bf).contiguous(), b_hh=d(b_hh), hall=hall,
                 hall16=torch.zeros(T + 1, M, H, device=dev, dtype=bf), gates=torch.zeros(T, 4, M, H, device=dev, dtype=bf),
                 lengths=lengths.int().to(dev) if masked else None, reverse=reverse)
        fw.append(c)
        de = dh_ext.to(dev)
        dl = torch.zeros(M, H + 8, device=dev); dl[:, :H] = dh_last.to(dev)
        bw.append(dict(hall=hall, gates=c['gates'], wt16=d(w_hh).t().contiguous().to(bf), dh_ext=de, dh_last=dl[:, :H],
                       dgi=torch.zeros(T, M, 3 * H, device=dev, dtype=bf), dgh=torch.zeros(T, M, 3 * H, device=dev, dtype=bf),
                       dh0=torch.zeros(M, H, device=dev), reverse=reverse))
    if lib().ptv_gru_persist_supported(NC, M, H):
        F_.gru_persist_fwd(M, H, T, fw)
    else:                                            # the forward takes at most 256 rows per workgroup: more launches
        n = 2 if lib().ptv_gru_persist_supported(2, M, H) else 1
        for i in range(0, NC, n):
            F_.gru_persist_fwd(M, H, T, fw[i:i + n])
    F_.gru_persist_bwd(M, H, T, bw)
    F_.persist_check()
    for ci in range(NC):
        c, b = fw[ci], bw[ci]
        hs, dh0_ref, dgi_ref, dbhh_ref = ref[ci]
        assert (c['hall'][1:].cpu() - hs).abs().max() < 3e-2
        assert (c['hall16'][1:].float() - c['hall'][1:]).abs().max() < 1e-2
        tol = 0.05
        assert (b['dh0'].cpu() - dh0_ref).abs().max() < tol * max(1.0, dh0_ref.abs().max().item())
        assert (b['dgi'].float().cpu() - dgi_ref).abs().max() < tol * max(1.0, dgi_ref.abs().max().item())
        db = b['dgh'].float().sum((0, 1)).cpu()
        assert (db - dbhh_ref).abs().max() < tol * max(1.0, dbhh_ref.abs().max().item())
        # the per-step kernels on the same operands: same arithmetic up to summation order / bf16 rounding of the state
        hall2 = torch.zeros_like(c['hall']); hall2[0] = c['hall'][0]
        h16_2 = torch.zeros_like(c['hall16']); gates2 = torch.zeros_like(c['gates'])
        FL = 1 | 2 | 8 | 16 | (4 if use_gi2 else 0)
        call('ptv_gru_seq_fwd', 1, M, H, T, ptr(c['gi']), M * 3 * H, 3 * H, ptr(c['gi2']), 0, c['gi2_ld'], ptr(c['w16']), ptr(c['b_hh']),
             ptr(hall2), ptr(h16_2), ptr(gates2), ptr(c['lengths']), int(c['reverse']), None, FL, stream_ptr())
        assert (hall2 - c['hall']).abs().max() < 2e-2
        assert (gates2.float() - c['gates'].float()).abs().max() < 3e-2
        dgi2 = torch.zeros_like(b['dgi']); dgh2 = torch.zeros_like(b['dgh'])
        dhz = torch.empty(2, M, H, device=dev); dh02 = torch.empty(M, H, device=dev)
        de, dl = b['dh_ext'], b['dh_last']
        call('ptv_gru_seq_bwd', 1, M, H, T, ptr(c['hall']), ptr(c['gates']), ptr(b['wt16']), ptr(de), de.stride(0), de.stride(1),
             ptr(dl), dl.stride(0), None, 0, 0, 0, None, ptr(dgi2), ptr(dgh2), ptr(dhz), ptr(dh02), int(c['reverse']), FL, stream_ptr())
        sc = max(1.0, dgi_ref.abs().max().item())
        assert (dgi2.float() - b['dgi'].float()).abs().max() < 0.03 * sc
        assert (dgh2.float() - b['dgh'].float()).abs().max() < 0.03 * sc
        assert (dh02 - b['dh0']).abs().max() < 0.03 * max(1.0, dh0_ref.abs().max().item())


@pytest.mark.parametrize('M,H,T', [(16384, 512, 2), (512, 1024, 32)])
def test_gru_bwd_kernel_variants_of_the_b512_step(M, H, T):
    """the BPTT tile variants only B = 512 dispatches -- gru_bwd_step_kernel<BF16,128,128,...,FAST> (notes GRU, M = 16384,
    H = 512) and the 32-step M = 512, H = 1024 chain (per-step kernels; PERSIST off) -- against the fp32 oracle cell"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + H + T)
    k = 1.0 / np.sqrt(H)
    w_hh = ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k).to(bf).float().requires_grad_()
    b_hh = ((torch.rand(3 * H, generator=g) * 2 - 1) * k).requires_grad_()
    gi = (torch.randn(T, M, 3 * H, generator=g) * 0.5).to(bf).float().requires_grad_()
    h0 = (torch.randn(M, H, generator=g) * 0.5).requires_grad_()
    dh_ext = torch.randn(T, M, H, generator=g) * 0.1
    h, hs = h0, []
    for t in range(T):
        gh = orc.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[t][:, :H] + gh[:, :H]); z = torch.sigmoid(gi[t][:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[t][:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        hs.append(h)
    hs = torch.stack(hs)
    (hs * dh_ext).sum().backward()
    d = lambda t: t.detach().to(dev)
    w16, wt16 = d(w_hh).to(bf).contiguous(), d(w_hh).t().contiguous().to(bf)
    hall = torch.zeros(T + 1, M, H, device=dev); hall[0] = d(h0)
    hall16 = torch.zeros(T + 1, M, H, device=dev, dtype=bf)
    gates = torch.empty(T, 4, M, H, device=dev, dtype=bf)
    FL = 1 | 2 | 8 | 16
    gi_d, b_d = d(gi).to(bf), d(b_hh)
    call('ptv_gru_seq_fwd', 1, M, H, T, ptr(gi_d), M * 3 * H, 3 * H, None, 0, 0, ptr(w16), ptr(b_d),
         ptr(hall), ptr(hall16), ptr(gates), None, 0, None, FL, stream_ptr())
    assert (hall[1:].cpu() - hs.detach()).abs().max() < (3e-2 if T <= 4 else 8e-2)
    dgi = torch.empty(T, M, 3 * H, device=dev, dtype=bf); dgh = torch.empty_like(dgi)
    dhz = torch.empty(2, M, H, device=dev); dh0 = torch.empty(M, H, device=dev)
    de = dh_ext.to(dev)
    call('ptv_gru_seq_bwd', 1, M, H, T, ptr(hall), ptr(gates), ptr(wt16), ptr(de), de.stride(0), de.stride(1),
         None, 0, None, 0, 0, 0, None, ptr(dgi), ptr(dgh), ptr(dhz), ptr(dh0), 0, FL, stream_ptr())
    tol = 0.05 if T <= 4 else 0.1
    assert (dh0.cpu() - h0.grad).abs().max() < tol * max(1.0, h0.grad.abs().max().item())
    assert (dgi.float().cpu() - gi.grad).abs().max() < tol * max(1.0, gi.grad.abs().max().item())
    db = dgh.float().sum((0, 1)).cpu()
    assert (db - b_hh.grad).abs().max() < tol * max(1.0, b_hh.grad.abs().max().item())


def test_fused_duration_kernels_at_the_b512_grid_caps():
    """ptv_dur_gru_fwd / ptv_dur_gru_bwd at M = 245,760 rows (= 480 x 512): the forward's 1024-block cap and the backward's
    256-block grid-stride regime (functional.dur_bwd_fused), against the fp32 oracle cell with the kernel's argmax replayed"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    M, H = 480 * 512, 64
    g = torch.Generator().manual_seed(12)
    k = 1.0 / np.sqrt(H)
    U = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * k
    w_hh, b_hh, w_ih, b_ih = U(3 * H, H).to(bf).float(), U(3 * H), U(3 * H, 5), U(3 * H)
    w_out, b_out, sos = U(2, H), U(2), torch.rand(5, generator=g)
    h0 = torch.randn(M, H, generator=g) * 0.5
    ddur = torch.randn(M, 10, generator=g) * 0.1
    ddur[M // 2 + 100: M // 2 + 64 * 50] = 0                 # whole 64-row tiles without gradient (padded note slots): the backward skips them
    ddur[M - 64 * 3:] = 0
    d = lambda t: t.detach().to(dev).contiguous()
    tab0 = (orc.linear(sos.view(1, -1), w_ih, b_ih)).contiguous()
    oh = torch.zeros(2, 5); oh[0, 0] = 1; oh[1, 1] = 1
    tab = orc.linear(oh, w_ih, b_ih).contiguous()
    HD16 = torch.zeros(6, M, H, device=dev, dtype=bf)
    gates = torch.empty(5, 4, M, H, device=dev, dtype=bf)
    dur = torch.empty(M, 10, device=dev)
    idx = torch.empty(5, M, device=dev, dtype=torch.int32)
    h0d = d(h0)
    Wd = {n: d(t) for n, t in (('w_hh', w_hh), ('b_hh', b_hh), ('tab0', tab0), ('tab', tab), ('w_out', w_out), ('b_out', b_out),
                                ('ddur', ddur))}                      # keep the device copies alive across the launches
    call('ptv_dur_gru_fwd', H, M, ptr(h0d), H, ptr(Wd['w_hh']), ptr(Wd['b_hh']), ptr(Wd['tab0']), ptr(Wd['tab']), ptr(Wd['w_out']),
         ptr(Wd['b_out']), None, M * H, ptr(HD16[1]), ptr(gates), M * H, 4 * M * H, 1, ptr(dur), 10, ptr(idx), M, None, M, stream_ptr())
    call('ptv_cast_bf16', ptr(h0d), ptr(HD16[0]), M * H, stream_ptr())
    # oracle on a row sample (first / middle / last tiles), replaying the kernel's argmax decisions
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M)])
    idx_c = idx.cpu()[:, rows].long()
    hr = h0[rows].clone().requires_grad_()
    w_hh_r, b_hh_r, w_out_r = w_hh.clone().requires_grad_(), b_hh.clone().requires_grad_(), w_out.clone().requires_grad_()
    h, outs = hr, []
    for s in range(5):
        gi = tab0.expand(len(rows), -1) if s == 0 else tab[idx_c[s - 1]]
        gh = orc.linear(h, w_hh_r, b_hh_r)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H]); z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs.append(orc.linear(h, w_out_r, b_out))
    est = torch.cat(outs, 1)
    assert (dur.cpu()[rows] - est.detach()).abs().max() < 3e-2
    margin = (est.detach().view(-1, 5, 2)[:, :, 0] - est.detach().view(-1, 5, 2)[:, :, 1]).abs()
    flips = (idx_c.t() != est.detach().view(-1, 5, 2).argmax(-1)) & (margin > 5e-2)
    assert not flips.any()
    (est * ddur[rows]).sum().backward()
    nblk = min(256, (M + 63) // 64)
    psz = lib().ptv_dur_gru_bwd_part_size()
    part = torch.zeros(nblk, psz, device=dev)
    dh0 = torch.empty(M, H, device=dev)
    call('ptv_dur_gru_bwd', H, M, ptr(gates), M * H, 4 * M * H, ptr(HD16), M * H, 1, ptr(Wd['ddur']), 10, ptr(Wd['w_hh']), ptr(Wd['w_out']),
         ptr(idx), M, ptr(dh0), ptr(part), nblk, None, None, None, stream_ptr())
    assert (dh0.cpu()[rows] - hr.grad).abs().max() < 0.05 * max(1.0, hr.grad.abs().max().item())
    assert torch.isfinite(part).all() and torch.isfinite(dh0).all()
    assert (dh0[M - 64 * 3:] == 0).all() and (dh0[M // 2 + 128: M // 2 + 64 * 49] == 0).all()
    # RECOMPUTE mode (what the train step runs): the forward saves no gates, the backward rebuilds them from the bf16 states.  Same
    # states / logits / decisions bit for bit; gradients at least as close to the fp32 oracle as with the bf16-rounded saved gates
    HD16b = torch.zeros(6, M, H, device=dev, dtype=bf)
    dur_b = torch.empty(M, 10, device=dev)
    idx_b = torch.empty(5, M, device=dev, dtype=torch.int32)
    call('ptv_dur_gru_fwd', H, M, ptr(h0d), H, ptr(Wd['w_hh']), ptr(Wd['b_hh']), ptr(Wd['tab0']), ptr(Wd['tab']), ptr(Wd['w_out']),
         ptr(Wd['b_out']), None, M * H, ptr(HD16b[1]), None, M * H, 4 * M * H, 1, ptr(dur_b), 10, ptr(idx_b), M, None, M, stream_ptr())
    call('ptv_cast_bf16', ptr(h0d), ptr(HD16b[0]), M * H, stream_ptr())
    assert torch.equal(HD16b, HD16) and torch.equal(dur_b, dur) and torch.equal(idx_b, idx)
    part_b = torch.zeros(nblk, psz, device=dev)
    dh0_b = torch.empty(M, H, device=dev)
    assert lib().ptv_dur_gru_bwd(H, M, None, M * H, 4 * M * H, ptr(HD16b), M * H, 1, ptr(Wd['ddur']), 10, ptr(Wd['w_hh']), ptr(Wd['w_out']),
                                 ptr(idx_b), M, ptr(dh0_b), ptr(part_b), nblk, None, None, None, stream_ptr()) != 0     # tables are required
    call('ptv_dur_gru_bwd', H, M, None, M * H, 4 * M * H, ptr(HD16b), M * H, 1, ptr(Wd['ddur']), 10, ptr(Wd['w_hh']), ptr(Wd['w_out']),
         ptr(idx_b), M, ptr(dh0_b), ptr(part_b), nblk, ptr(Wd['b_hh']), ptr(Wd['tab0']), ptr(Wd['tab']), stream_ptr())
    gmax = max(1.0, hr.grad.abs().max().item())
    err_saved, err_rc = (dh0.cpu()[rows] - hr.grad).abs().max().item(), (dh0_b.cpu()[rows] - hr.grad).abs().max().item()
    assert err_rc < 0.05 * gmax and err_rc <= err_saved * 1.25 + 1e-4, (err_saved, err_rc)
    assert (dh0_b - dh0).abs().max() < 0.02 * gmax
    Sa, Sb = part.sum(0), part_b.sum(0)
    assert (Sa - Sb).abs().max() < 0.02 * max(1.0, Sa.abs().max().item())
    assert (dh0_b[M - 64 * 3:] == 0).all() and (dh0_b[M // 2 + 128: M // 2 + 64 * 49] == 0).all()


@pytest.mark.parametrize('B,C', [(3, 10), (70, 10), (512, 10), (5, 16)])
def test_texture_conv_front_end_vs_fp32_reference(B, C):
    """TextureEncoder's Conv2d(1,C,(4,12),stride (4,1)) + ReLU + MaxPool2d((1,4)) (ptvae.py:95-99,112-114) alone: the pooled map in the
    raw-view row layout (padded and plain), and the weight / bias gradients from (a) the recomputed convolution and (b) the forward's
    arg-max map (what the train step runs) -- against torch's own conv / pool autograd on the CPU, and (b) == (a) bit for bit"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    import torch.nn.functional as TF
    dev = _dev()
    g = torch.Generator().manual_seed(100 + B)
    pr = (torch.rand(B, 32, 128, generator=g) < 0.06).float() * torch.randint(1, 9, (B, 32, 128), generator=g).float()
    w = (torch.randn(C, 1, 4, 12, generator=g) * 0.15).requires_grad_()
    b = (torch.randn(C, generator=g) * 0.1).requires_grad_()
    ref = TF.max_pool2d(torch.relu(TF.conv2d(pr.unsqueeze(1), w, b, stride=(4, 1))), (1, 4))           # [B,C,8,29]
    dy = torch.randn(ref.shape, generator=g)
    (ref * dy).sum().backward()
    W = C * 29
    prd, wd, bd = pr.to(dev), w.detach().reshape(C, 48).to(dev).contiguous(), b.detach().to(dev)
    res = {}
    for ld in (W, (W + 7) // 8 * 8):
        feat = torch.full((B * 8, ld), float('nan'), device=dev)
        arg = torch.empty(B * 8, W, device=dev, dtype=torch.int8)
        call('ptv_txt_conv_relu_pool_fwd_rows', ptr(prd), ptr(wd), ptr(bd), ptr(feat), ld, B, C, ptr(arg), stream_ptr())
        # the reference's raw view (ptvae.py:114): [B,C,8,29] memory read as [B*8, C*29] rows
        np.testing.assert_allclose(feat[:, :W].cpu().numpy(), ref.detach().reshape(B * 8, W).numpy(), rtol=0, atol=2e-5)
        assert torch.isfinite(feat).all()                          # row padding zeroed
        dfeat = torch.zeros(B * 8, ld, device=dev)
        dfeat[:, :W] = dy.reshape(B * 8, W).to(dev)
        for mode in ('recompute', 'argmax'):
            dw, db = torch.zeros(C, 48, device=dev), torch.zeros(C, device=dev)
            call('ptv_txt_conv_relu_pool_bwd_rows', ptr(prd), ptr(wd) if mode == 'recompute' else None, ptr(bd) if mode == 'recompute' else None,
                 ptr(dfeat), ld, ptr(dw), ptr(db), B, C, ptr(arg) if mode == 'argmax' else None, stream_ptr())
            res[ld, mode] = (dw.cpu(), db.cpu())
            tol = 2e-6 + 2e-5 * float(w.grad.abs().max()) * max(1.0, B / 16)
            np.testing.assert_allclose(dw.cpu().numpy(), w.grad.reshape(C, 48).numpy(), rtol=0, atol=tol, err_msg='%s ld=%d' % (mode, ld))
            np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), rtol=0, atol=tol)
        assert torch.equal(res[ld, 'argmax'][0], res[ld, 'recompute'][0]) and torch.equal(res[ld, 'argmax'][1], res[ld, 'recompute'][1])


def test_integration_md_snippet_runs_verbatim():
    """the ctypes example a maintainer would copy out of INTEGRATION.md, executed as written against the shipped .so"""
    import os
    from test_abi_symbols import integration_snippet
    from polyphonic_chord_texture_disentanglement_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cwd = os.getcwd()
    os.chdir(root)                                         # the snippet opens the library by its repo-relative path
    try:
        ns = {}
        exec(integration_snippet(), ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(2)
    x, w, b = torch.randn(37, 290, generator=g), torch.randn(130, 290, generator=g) / 17, torch.randn(130, generator=g)
    y = ns['linear'](x.to(_dev()), w.to(_dev()), b.to(_dev()))
    torch.cuda.synchronize()
    ref = orc.linear(x, w, b)
    assert (y.cpu() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('w', [16, 32])
def test_gemm_column_blocked_output(w):
    """ptv_gemm dtypes bit 3 / bit 4: C written column-blocked by 32 / 16 ([N/w][M][w], the per-row operand layout of the row-partitioned
    recurrences) holds exactly the row-major result"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(w)
    M, N, K = 200, 96, 160
    a, b, bias = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    for dt in (bf, torch.float32):
        ref = F_.gemm(a, b, bias=bias, prec=1, out_dtype=dt)
        blk = F_.gemm(a, b, bias=bias, prec=1, out_dtype=dt, out_blocked=w)
        assert torch.equal(blk.view(N // w, M, w).permute(1, 0, 2).reshape(M, N), ref)


@pytest.mark.parametrize('R,T,zero_from', [(512, 15, None), (200, 4, None), (16384, 2, None), (512, 15, 7), (300, 6, 0)])
def test_notes_gru_persistent_kernels_vs_step_kernels_and_oracle(R, T, zero_from):
    """the row-partitioned notes GRU (a workgroup owns 64 rows for the whole sequence, token product fused) -- forward with wave roles
    (csrc/notes_roles.hip: fp32 state in registers, only the bf16 states leave the CU), BPTT with dgh through a K-blocked scratch tile
    (csrc/notes_persist.hip) -- against the per-step kernels + separate token product on the same bf16 operands, and against the
    fp32 oracle cell; whole / ragged last panel / the B = 512 row count / no gradient arriving at the late steps (skipped by the
    BPTT kernel, panel by panel)"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    H, E = 512, 128
    g = torch.Generator().manual_seed(R + T)
    k = 1.0 / np.sqrt(H)
    U = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * k
    w_hh, w_tok, b_hh = U(3 * H, H).to(bf).float(), U(3 * H, E).to(bf).float(), U(3 * H)
    gc = (torch.randn(R, 3 * H, generator=g) * 0.5).to(bf)
    emb = (torch.randn(T, R, E, generator=g) * 0.5).to(bf).float()
    h0 = torch.randn(R, H, generator=g) * 0.5
    ext = (torch.randn(T, R, H, generator=g) * 0.1).to(bf)
    if zero_from is not None:                                   # no gradient arrives at the late steps (padded note slots the loss ignores):
        ext[zero_from:] = 0                                     # the BPTT kernel passes over them; rows 64.. of step zero_from-1 too
        if zero_from > 0:
            ext[zero_from - 1, 64:] = 0
    d = lambda t: t.to(dev).contiguous()
    Wd = dict(w_hh=d(w_hh), w_tok=d(w_tok), b_hh=d(b_hh), gc=d(gc), emb=d(emb), ext=d(ext))
    gc_blocked = Wd['gc'].view(R, 3 * H // 16, 16).permute(1, 0, 2).contiguous()      # what ptv_gemm writes with dtypes bit 4
    wg_h, wg_t = F_.pack_mfma_b(Wd['w_hh'], pairs=False), F_.pack_mfma_b(Wd['w_tok'], pairs=False)
    wt = F_.pack_mfma_b(Wd['w_hh'].t().contiguous(), pairs=True)
    HN0 = d(h0)
    HN16 = torch.zeros(T + 1, R, H, device=dev, dtype=bf)
    gates = torch.zeros(T, 4, R, H, device=dev, dtype=bf)
    call('ptv_notes_gru_persist_fwd', ptr(wg_h), ptr(wg_t), ptr(Wd['b_hh']), ptr(gc_blocked), ptr(Wd['emb']), ptr(HN0), ptr(HN16), ptr(gates),
         R, T, stream_ptr())
    dgi = torch.zeros(T, R, 3 * H, device=dev, dtype=bf); dgh = torch.zeros(T, R, H, device=dev, dtype=bf)     # dgh: n third only
    dh0 = torch.zeros(R, H, device=dev)
    scratch = torch.empty(lib().ptv_notes_gru_persist_scratch_elems(R), device=dev, dtype=bf)
    ext_blocked = Wd['ext'].view(T * R, H // 32, 32).permute(1, 0, 2).contiguous()        # the [T*R][H] matrix as ptv_gemm writes it with dtypes bit 3
    call('ptv_notes_gru_persist_bwd', ptr(wt), ptr(HN16), ptr(gates), ptr(ext_blocked), ptr(dgi), ptr(dgh), ptr(dh0), ptr(scratch), R, T, None, stream_ptr())
    # (the 4-wave BPTT kernel of rounds 2-4 behind the same entry point: same results to the rounding of a re-associated fp32 sum)
    dgi4, dgh4, dh04 = torch.zeros_like(dgi), torch.zeros_like(dgh), torch.zeros_like(dh0)
    call('ptv_notes_bwd_variant', 0)
    try:
        call('ptv_notes_gru_persist_bwd', ptr(wt), ptr(HN16), ptr(gates), ptr(ext_blocked), ptr(dgi4), ptr(dgh4), ptr(dh04), ptr(scratch), R, T, None,
             stream_ptr())
        torch.cuda.synchronize()
    finally:
        call('ptv_notes_bwd_variant', 1)
    assert (dgi.float() - dgi4.float()).abs().max() < 1e-2 * max(1.0, dgi4.float().abs().max().item())
    assert (dgh.float() - dgh4.float()).abs().max() < 1e-2 * max(1.0, dgh4.float().abs().max().item())
    assert (dh0 - dh04).abs().max() < 2e-3 * max(1.0, dh04.abs().max().item())
    # ---- the per-step kernels on the same operands
    GT = F_.gemm(Wd['emb'].view(T * R, E), Wd['w_tok'].to(bf), prec=1, out_dtype=bf)
    HN2 = torch.zeros(T + 1, R, H, device=dev); HN2[0] = HN0
    HN16_2 = torch.zeros_like(HN16); gates2 = torch.zeros_like(gates)
    FL = 1 | 2 | 4 | 8 | 16
    w16, wt16 = Wd['w_hh'].to(bf).contiguous(), Wd['w_hh'].t().contiguous().to(bf)
    call('ptv_gru_seq_fwd', 1, R, H, T, ptr(GT), R * 3 * H, 3 * H, ptr(Wd['gc']), 0, 3 * H, ptr(w16), ptr(Wd['b_hh']),
         ptr(HN2), ptr(HN16_2), ptr(gates2), None, 0, None, FL, stream_ptr())
    assert torch.equal(HN16[0], HN0.to(bf))
    assert (HN16.float() - HN2).abs().max() < 3e-2
    # (the row kernels keep their gate planes unit-blocked, [T][4][H/16][R][16]: private to the forward / BPTT pair)
    gates_rm = gates.view(T, 4, H // 16, R, 16).permute(0, 1, 3, 2, 4).reshape(T, 4, R, H)
    assert (gates_rm.float() - gates2.float()).abs().max() < 4e-2
    dgi2 = torch.zeros_like(dgi); dgh2 = torch.zeros_like(dgi)
    dhz = torch.empty(2, R, H, device=dev); dh02 = torch.empty(R, H, device=dev)
    e = Wd['ext']
    gates_rm = gates_rm.contiguous()
    HNf = HN16.float()                                        # (the BPTT takes the previous state from the bf16 copy)
    call('ptv_gru_seq_bwd', 1, R, H, T, ptr(HNf), ptr(gates_rm), ptr(wt16), ptr(e), e.stride(0), e.stride(1), None, 0, None, 0, 0, 0, None,
         ptr(dgi2), ptr(dgh2), ptr(dhz), ptr(dh02), 0, FL | 64, stream_ptr())
    sc = max(1.0, dgi2.float().abs().max().item())
    assert (dgi.float() - dgi2.float()).abs().max() < 0.03 * sc
    assert (dgh.float() - dgh2[:, :, 2 * H:].float()).abs().max() < 0.03 * sc
    assert torch.equal(dgi2[:, :, :2 * H], dgh2[:, :, :2 * H])                 # the thirds the persistent kernel does not write twice
    assert (dh0 - dh02).abs().max() < 0.03 * max(1.0, dh02.abs().max().item())
    # ---- fp32 oracle cell on a row sample
    rows = torch.cat([torch.arange(0, min(R, 40)), torch.arange(R - min(R, 40), R)])
    hr = h0[rows].clone().requires_grad_()
    h, hs = hr, []
    for t in range(T):
        x = gc[rows].float() + orc.linear(emb[t][rows], w_tok)
        gh = orc.linear(h, w_hh, b_hh)
        r = torch.sigmoid(x[:, :H] + gh[:, :H]); z = torch.sigmoid(x[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(x[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        hs.append(h)
    hs = torch.stack(hs)
    (hs * ext[:, rows].float()).sum().backward()
    assert (HN16[1:, rows.to(dev)].float().cpu() - hs.detach()).abs().max() < 4e-2
    assert (dh0[rows.to(dev)].cpu() - hr.grad).abs().max() < 0.05 * max(1.0, hr.grad.abs().max().item())


@pytest.mark.parametrize('R,maxlen', [(16384, 16), (4100, 5), (70000, 38), (7, 3), (1024, 0)])
def test_rows_by_length_is_a_stable_descending_sort(R, maxlen):
    """ptv_rows_by_length (the row order of the note-summary bi-GRU's panels): a permutation, lengths non-increasing along it, ties in
    row order -- i.e. exactly torch's stable descending sort"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    g = torch.Generator().manual_seed(R)
    lengths = torch.randint(0, maxlen + 1, (R,), generator=g, dtype=torch.int32).to(dev)
    perm = torch.full((R,), -1, device=dev, dtype=torch.int32)
    call('ptv_rows_by_length', ptr(lengths), ptr(perm), R, maxlen, stream_ptr())
    torch.cuda.synchronize()
    want = torch.sort(lengths.cpu().long(), descending=True, stable=True)[1]
    assert torch.equal(perm.cpu().long(), want)


@pytest.mark.parametrize('sort_rows', [True, False])
@pytest.mark.parametrize('M,T,maxlen', [(4096, 16, 16), (4100, 5, 5), (4200, 16, 6), (4096, 16, 0), (8192, 16, -7), (4096, 16, -1)])
def test_row_gru_h128_bidirectional_with_lengths_vs_step_kernels_and_oracle(M, T, maxlen, sort_rows, monkeypatch):
    """dec_notes_emb_gru through the H = 128 instance of csrc/notes_persist.hip (lengths mask, reversed direction, final state into
    its half of the summary, gradient arriving at the final state only) against the per-step path on the same operands and the
    fp32 oracle's packed-sequence bi-GRU (oracle _bigru_final, ptvae.py:446-453); short sequences (panels whose late steps are skipped
    altogether) and the all-empty batch included"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from oracle.ptvae_oracle import Oracle
    dev = _dev()
    H = I = 128
    monkeypatch.setattr(F_, 'SORT_ROWS', sort_rows)             # panels of rows sorted by length (the default) / in row order
    g = torch.Generator().manual_seed(M + T)
    k = 1.0 / np.sqrt(H)
    U = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k)
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    shapes = [(3 * H, I), (3 * H, H), (3 * H,), (3 * H,)]
    cpu_w = {('g.' + n + s): U(*sh) for s in ('', '_reverse') for n, sh in zip(names, shapes)}
    order = ['g.' + n + s for s in ('', '_reverse') for n in names]
    x = torch.randn(T, M, I, generator=g) * 0.7
    # maxlen < T: whole 64-row panels have nothing to do at the late note positions (the kernels pass over those steps); one panel
    # keeps a full-length row, one row is empty
    # maxlen < 0: NO row longer than -maxlen (M a multiple of 32): the time indices beyond the longest row of the whole launch are dead
    # for every panel -- the kernels neither copy states nor write zero gradients there (round 6), so every buffer of the persistent run
    # is handed out NaN-filled: whatever read an unwritten slot would poison the results
    full_row = maxlen > 0
    maxlen = abs(maxlen)
    lengths = torch.randint(0, maxlen + 1, (M,), generator=g, dtype=torch.int32)
    if full_row:
        lengths[:3] = torch.tensor([0, T, 1], dtype=torch.int32)
    elif maxlen > 0:
        lengths[:3] = torch.tensor([0, maxlen, 1], dtype=torch.int32)
    dout = torch.randn(M, 2 * H, generator=g) * 0.3
    plain_empty = F_._empty

    def nan_empty(*shape, dev, dtype=torch.float32):
        t = plain_empty(*shape, dev=dev, dtype=dtype)
        return t.fill_(float('nan')) if t.dtype.is_floating_point else t

    def run(persist):
        monkeypatch.setattr(F_, 'NOTES_PERSIST', persist)
        monkeypatch.setattr(F_, '_empty', nan_empty if persist else plain_empty)
        w = [cpu_w[n].to(dev).requires_grad_() for n in order]
        xd = x.to(dev).requires_grad_()
        assert F_.row_gru_ok(1, H, I, M, torch.bfloat16) == persist
        out = F_.BiGruFinalFn.apply(xd, lengths.to(dev), 1, *w)
        out.backward(dout.to(dev))
        torch.cuda.synchronize()
        return out.detach().cpu(), xd.grad.cpu(), [p.grad.cpu() for p in w]

    o1, dx1, g1 = run(True)
    o0, dx0, g0 = run(False)
    assert (o1 - o0).abs().max() < 2e-2
    assert (dx1 - dx0).abs().max() < 0.03 * max(1.0, dx0.abs().max().item())
    for a, b in zip(g1, g0):
        assert (a - b).abs().max() < 0.03 * max(1.0, b.abs().max().item())
    # ---- fp32 oracle on a row sample (weight gradients need all rows: compare out and dx there)
    rows = torch.cat([torch.arange(0, 48), torch.arange(M - 48, M)])
    orc_ = Oracle.__new__(Oracle)                               # only the bi-GRU helper is used: no full parameter set needed
    orc_.p = cpu_w
    xr = x[:, rows].transpose(0, 1).contiguous().requires_grad_()
    ref = orc_._bigru_final('g', xr, lengths[rows])
    ref.backward(dout[rows])
    assert (o1[rows] - ref.detach()).abs().max() < 3e-2
    assert (dx1[:, rows] - xr.grad.transpose(0, 1)).abs().max() < 0.05 * max(1.0, xr.grad.abs().max().item())
    assert (o1[0] == 0).all()                                   # a row of length 0 never leaves the zero state


@pytest.mark.parametrize('M1,M2,N,K,bdt,pad', [(256, 128, 128, 4096, 1, 0), (1024, 512, 512, 2080, 1, 0), (128, 72, 130, 999, 0, 6)])
def test_wgrad_two_source_product_equals_the_two_products(M1, M2, N, K, bdt, pad):
    """ptv_wgrad_cat: C[M1 + M2, N] += [A1 | A2]^T B with the two column blocks in different arrays (the notes GRU's weight_hh gradient:
    dgi's r / z columns and dgh) -- the same slabs, the same tiles, the same order of additions as one ptv_wgrad per block, so the result
    and the fused bias gradient are bit-identical to the two calls; k_top included; and both agree with a float64 product"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M1 + M2 + K)
    A1 = torch.randn(K, 3 * M1 // 2, generator=g).to(bf).to(dev)[:, :M1]            # (a column block of a wider matrix: lda > M1)
    A2 = torch.randn(K, M2 + pad, generator=g).to(bf).to(dev)[:, :M2]
    B = torch.randn(K, N + pad, generator=g)
    Bd = (B.to(bf) if bdt else B).to(dev)[:, :N]
    C0, b0 = torch.randn(M1 + M2, N, generator=g).to(dev), torch.randn(M1 + M2, generator=g).to(dev)
    unit = 32
    ktop = torch.tensor([(K // unit) // 2], device=dev, dtype=torch.int32)
    for kt in (None, ktop):
        Az1, Az2 = A1.clone(), A2.clone()
        if kt is not None:
            Az1[(int(kt) + 1) * unit:] = 0; Az2[(int(kt) + 1) * unit:] = 0
        Ca, ba = C0.clone(), b0.clone()
        call('ptv_wgrad_cat', M1, ptr(Az1), Az1.stride(0), M2, ptr(Az2), Az2.stride(0), N, K, ptr(Bd), Bd.stride(0), ptr(Ca), Ca.stride(0), 1.0, 1,
             1 | (bdt << 1), 0, ptr(ba), ptr(kt), unit if kt is not None else 0, 0, stream_ptr())
        Cb, bb = C0.clone(), b0.clone()
        for A_, lo, hi in ((Az1, 0, M1), (Az2, M1, M1 + M2)):
            call('ptv_wgrad', hi - lo, N, K, ptr(A_), A_.stride(0), ptr(Bd), Bd.stride(0), ptr(Cb[lo:hi]), Cb.stride(0), 1.0, 1, 1 | (bdt << 1), 0,
                 ptr(bb[lo:hi]), ptr(kt), unit if kt is not None else 0, 0, stream_ptr())
        torch.cuda.synchronize()
        want = C0.cpu().double() + torch.cat([Az1, Az2], 1).cpu().double().t() @ Bd.cpu().to(bf).double()
        assert (Ca.cpu().double() - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item())
        if M1 % 128 == 0 and M2 % 128 == 0:                      # (same tile decomposition in both routes: not one bit apart)
            assert torch.equal(Ca, Cb) and torch.equal(ba, bb)
        else:
            assert (Ca - Cb).abs().max() < 2e-5 * max(1.0, want.abs().max().item()) and (ba - bb).abs().max() < 1e-3


def test_wgrad_batch_equals_the_single_calls_bit_for_bit():
    """ptv_wgrad_batch: several products behind ONE product launch and ONE reduction launch.  Every job keeps the slab plan and the
    reduction order of its own ptv_wgrad call, so outputs and fused bias sums are bit-identical to the calls made one by one: mixed operand
    dtypes, a guarded (unaligned) product, a single-slab product, one with a k_top limit, accumulate on and off; and vs float64."""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr, wgrad_batch
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    # (M, N, K, A bf16, B bf16, pad, bias, k_top, accumulate)
    specs = [(1024, 512, 8192, 1, 1, 0, True, False, 1), (512, 512, 8192, 1, 1, 0, True, True, 1), (1536, 1024, 2048, 0, 1, 0, True, False, 1),
             (130, 512, 2000, 0, 1, 6, False, False, 0), (64, 130, 999, 0, 0, 6, True, False, 1), (12, 512, 512, 0, 0, 0, True, False, 1),
             (3072, 256, 512, 0, 0, 0, True, False, 1), (384, 128, 16384, 1, 0, 0, True, True, 1), (256, 64, 96, 1, 1, 0, False, False, 1)]
    jobs, singles, wants = [], [], []
    unit = 32
    for M, N, K, abf, bbf, pad, bias, kt, acc in specs:
        A = torch.randn(K, M + pad, generator=g); B = torch.randn(K, N + pad, generator=g)
        top = (K // unit) // 2 - 1
        if kt:
            A[(top + 1) * unit:] = 0
        Ad = (A.to(bf) if abf else A).to(dev)[:, :M]; Bd = (B.to(bf) if bbf else B).to(dev)[:, :N]
        C0 = torch.randn(M, N, generator=g).to(dev); b0 = torch.randn(M, generator=g).to(dev)
        ktop = torch.tensor([top], device=dev, dtype=torch.int32) if kt else None
        Ca, ba, Cb, bb = C0.clone(), b0.clone(), C0.clone(), b0.clone()
        jobs.append(dict(M=M, N=N, K=K, A=Ad, B=Bd, C=Ca, alpha=0.5, accumulate=acc, colsum_a=ba if bias else None, k_top=ktop,
                         k_unit=unit if kt else 0))
        singles.append((M, N, K, Ad, Bd, Cb, acc, abf | (bbf << 1), bb if bias else None, ktop, unit if kt else 0))
        want = (C0.cpu().double() if acc else 0) + 0.5 * (A[:, :M].to(bf).double().t() @ B[:, :N].to(bf).double())
        wants.append((Ca, ba, Cb, bb, want, b0.cpu().double() + A[:, :M].to(bf).double().sum(0), bias))
    wgrad_batch(jobs)                                             # 9 jobs: two tables (8 + 1)
    for M, N, K, Ad, Bd, Cb, acc, dt, bb, ktop, ku in singles:
        call('ptv_wgrad', M, N, K, ptr(Ad), Ad.stride(0), ptr(Bd), Bd.stride(0), ptr(Cb), Cb.stride(0), 0.5, acc, dt, 0, ptr(bb), ptr(ktop), ku, 0,
             stream_ptr())
    torch.cuda.synchronize()
    for i, (Ca, ba, Cb, bb, want, want_b, bias) in enumerate(wants):
        assert torch.equal(Ca, Cb), (i, (Ca - Cb).abs().max())
        assert torch.equal(ba, bb), i
        assert (Ca.cpu().double() - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item()), i
        if bias:
            assert (ba.cpu().double() - want_b).abs().max() < 2e-5 * max(1.0, want_b.abs().max().item()), i


def test_wgrad_batch_k_segments_skip_the_dead_rows_of_every_unit_bit_for_bit():
    """ptv_wgrad_job.seg_n (round 6): K runs over 15 units of `unit` rows (a note step's decoder rows in length order) of which only a
    prefix is live.  With the dead rows of A zero, a clipped call equals -- bit for bit -- the call that multiplies everything (same slab
    plan: seg_n = unit everywhere), although its B holds NaN in every dead row (never read); also with a k_top limit on top, with a fused
    bias sum, for an unaligned product (guarded tail launch), accumulate on and off; and against float64.  ptv_rows_seg_counts: the
    prefixes of a sorted length vector by 128-row blocks."""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr, wgrad_batch
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(23)
    unit, T = 1024, 15
    K = unit * T
    # lengths in descending order -> seg_n through the library's own kernel
    lens = torch.sort(torch.randint(0, 12, (unit,), generator=g), descending=True).values.to(torch.int32)
    lens_d = lens.to(dev)
    seg = torch.empty(T, device=dev, dtype=torch.int32)
    call('ptv_rows_seg_counts', ptr(lens_d), unit, T, ptr(seg), stream_ptr())
    want_seg = torch.tensor([128 * int((lens[::128] > s_).sum()) for s_ in range(T)], dtype=torch.int32)
    assert torch.equal(seg.cpu(), want_seg), (seg.cpu(), want_seg)
    assert 0 < int(want_seg[5]) < unit and int(want_seg[-1]) == 0           # the case is not trivial
    full = torch.full((T,), unit, device=dev, dtype=torch.int32)
    live = torch.zeros(K, dtype=torch.bool)
    for s_ in range(T):
        live[s_ * unit: s_ * unit + int(want_seg[s_])] = True
    top = 8                                                        # a k_top limit below the last live unit (units 9.. are declared zero too)
    ktop = torch.tensor([top], device=dev, dtype=torch.int32)
    # (M, N, A bf16, B bf16, pad, bias, k_top, accumulate)
    specs = [(1024, 512, 1, 1, 0, True, False, 1), (512, 128, 1, 0, 0, False, True, 1), (64, 130, 0, 0, 6, False, False, 0), (200, 512, 1, 1, 0, True, True, 0)]
    jobs_c, jobs_f, chk = [], [], []
    for M, N, abf, bbf, pad, bias, kt, acc in specs:
        A = torch.randn(K, M + pad, generator=g); B = torch.randn(K, N + pad, generator=g)
        lv = live.clone()
        if kt:
            lv[(top + 1) * unit:] = False
        A[~lv] = 0
        Bn = B.clone(); Bn[~live] = float('nan')                   # what a forward that skipped the dead blocks leaves behind
        Ad = (A.to(bf) if abf else A).to(dev)[:, :M]
        Bd_nan = (Bn.to(bf) if bbf else Bn).to(dev)[:, :N]; Bd = (B.to(bf) if bbf else B).to(dev)[:, :N]
        C0 = torch.randn(M, N, generator=g).to(dev); b0 = torch.randn(M, generator=g).to(dev)
        Cc, bc, Cf, bfull = C0.clone(), b0.clone(), C0.clone(), b0.clone()
        common = dict(M=M, N=N, K=K, alpha=0.5, accumulate=acc, k_top=ktop if kt else None, k_unit=unit if kt else 0, seg_unit=unit, seg_period=T)
        jobs_c.append(dict(common, A=Ad, B=Bd_nan, C=Cc, colsum_a=bc if bias else None, seg_n=seg))
        jobs_f.append(dict(common, A=Ad, B=Bd, C=Cf, colsum_a=bfull if bias else None, seg_n=full))
        want = (C0.cpu().double() if acc else 0) + 0.5 * (A[:, :M].to(bf).double().t() @ B[:, :N].to(bf).double())
        chk.append((Cc, bc, Cf, bfull, want, b0.cpu().double() + A[:, :M].to(bf).double().sum(0), bias))
    wgrad_batch(jobs_c)
    wgrad_batch(jobs_f)
    torch.cuda.synchronize()
    for i, (Cc, bc, Cf, bfull, want, want_b, bias) in enumerate(chk):
        assert torch.isfinite(Cc).all(), i
        assert torch.equal(Cc, Cf), (i, (Cc - Cf).abs().max())
        assert (Cc.cpu().double() - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item()), i
        if bias:
            assert torch.equal(bc, bfull), i
            assert (bc.cpu().double() - want_b).abs().max() < 2e-5 * max(1.0, want_b.abs().max().item()), i


def test_row_segment_variants_of_step_sum_row_product_gather_and_scatter():
    """round 6: the other walkers of (note step, length-sorted row) take the live prefixes too (ptv_rows_seg_counts).  ptv_sum_steps_seg and
    ptv_gemm_mtop_seg on operands whose dead rows hold NaN (they are not read) equal the plain calls on operands whose dead rows are zero,
    bit for bit; ptv_gather_rows_seg leaves the dead rows of its output untouched, ptv_scatter_rows_seg writes zeros there without
    reading its (NaN) source."""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(29)
    R, T, W = 1024, 15, 96                                        # rows per step, steps, row width
    seg = torch.tensor([1024, 896, 640, 640, 384, 128, 128, 0, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32)
    live = torch.zeros(T, R, dtype=torch.bool)
    for s_ in range(T):
        live[s_, :int(seg[s_])] = True
    seg_d = seg.to(dev)
    top = torch.tensor([6], device=dev, dtype=torch.int32)
    # ---- step sum: out[r] = sum_s in[s][r]
    x = torch.randn(T, R, W, generator=g).to(bf)
    x0 = x.clone(); x0[~live] = 0
    xn = x.clone(); xn[~live] = float('nan')
    a_, b_ = torch.empty(R, W, device=dev), torch.empty(R, W, device=dev)
    x0d, xnd = x0.to(dev), xn.to(dev)
    call('ptv_sum_steps_top', ptr(a_), ptr(x0d), R * W, T, R * W, 0, 1, ptr(top), stream_ptr())
    call('ptv_sum_steps_seg', ptr(b_), ptr(xnd), R * W, T, R * W, 0, 1, ptr(top), ptr(seg_d), W, stream_ptr())
    torch.cuda.synchronize()
    assert torch.isfinite(b_).all() and torch.equal(a_, b_)
    assert (a_.cpu().double() - x0.double().sum(0)).abs().max() < 1e-4
    # ---- row product: C[(s, r)] = A[(s, r)] . B^T, dead row tiles store zeros without reading A
    K, N = 192, 128
    A = torch.randn(T * R, K, generator=g).to(bf); Wt = torch.randn(N, K, generator=g).to(bf)
    A0 = A.clone(); A0[~live.view(-1)] = 0
    An = A.clone(); An[~live.view(-1)] = float('nan')
    C0 = torch.full((T * R, N), 7.0, device=dev); C1 = torch.full((T * R, N), 7.0, device=dev)
    A0d, And, Wd = A0.to(dev), An.to(dev), Wt.to(dev)
    # (transB = 0: B in the nn.Linear weight layout [N][K])
    call('ptv_gemm_mtop', 1, 0, 0, T * R, N, K, ptr(A0d), K, ptr(Wd), K, ptr(C0), N, None, 1.0, 0, 0, 0, 3, ptr(top), R, stream_ptr())
    call('ptv_gemm_mtop_seg', 1, 0, 0, T * R, N, K, ptr(And), K, ptr(Wd), K, ptr(C1), N, None, 1.0, 0, 0, 0, 3, ptr(top), R, ptr(seg_d), R, T, stream_ptr())
    torch.cuda.synchronize()
    assert torch.isfinite(C1).all() and torch.equal(C0, C1)
    wantC = A0.double() @ Wt.double().t()
    assert (C1.cpu().double() - wantC).abs().max() < 2e-5 * max(1.0, wantC.abs().max().item())
    assert (C1.view(T, R, N)[~live.to(dev)] == 0).all()
    # ---- gather / scatter by a permutation
    perm = torch.randperm(R, generator=g).to(torch.int32).to(dev)
    src = torch.randn(T, R, W, generator=g).to(dev)
    dst = torch.full((T, R, W), -3.0, device=dev)
    call('ptv_gather_rows_seg', ptr(dst), ptr(src), ptr(perm), R, W, R * W, R * W, T, ptr(seg_d), stream_ptr())
    torch.cuda.synchronize()
    want = src[:, perm.long()]
    lv = live.to(dev)
    assert torch.equal(dst[lv], want[lv]) and (dst[~lv] == -3.0).all()
    srcn = want.clone(); srcn[~lv] = float('nan')                  # sorted order, dead rows poisoned
    back = torch.full((T, R, W), -3.0, device=dev)
    call('ptv_scatter_rows_seg', ptr(back), ptr(srcn), ptr(perm), R, W, R * W, R * W, T, ptr(seg_d), stream_ptr())
    torch.cuda.synchronize()
    expect = torch.zeros(T, R, W, device=dev)
    expect[:, perm.long()] = torch.where(lv.unsqueeze(-1), want, torch.zeros_like(want))
    assert torch.equal(back, expect)


@pytest.mark.parametrize('B', [24, 512])
def test_resummarize_with_resident_weights_is_bit_identical_to_the_streamed_kernel(B):
    """ptv_free_resummarize (round 6): a wave keeps its 48 weight fragments in registers for the 16 steps of a launch; train bit 1 selects the
    former path that streams them from L2 every step -- same products in the same k order: next tokens, saved states and gates bit-equal"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_
    from polyphonic_chord_texture_disentanglement_amd._lib import call, stream_ptr
    from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    m = DisentangleVAE.init_model(dev).to(dev)
    P = dict(m.decoder.named_parameters())
    pk = FF_._free_packs(P, 1024)
    wE = [P['dec_notes_emb_gru.' + n] for n in FF_.EMB_GRU]
    wr = F_._parr([pk['e_ih'], pk['e_hh'], pk['e_ih_r'], pk['e_hh_r'], wE[2], wE[3], wE[6], wE[7]])
    R = 32 * B
    g = torch.Generator(device=dev).manual_seed(9)
    PRED = torch.randn(16, R, 128, device=dev, generator=g) * 0.5
    plen = torch.randint(1, 16, (R,), device=dev, generator=g).to(torch.int32)
    outs = []
    for fl in (1, 3, 0, 2):
        XH = [torch.zeros(17, R, 128, device=dev) for _ in range(2)]
        XG = [torch.zeros(16, 4, R, 128, device=dev, dtype=torch.bfloat16) for _ in range(2)]
        tok = torch.zeros(B, 256, device=dev)
        io = F_._parr([PRED, plen, XH[0], XH[1], XG[0], XG[1], tok])
        call('ptv_free_resummarize', wr, io, B, 3, fl, stream_ptr())
        torch.cuda.synchronize()
        outs.append([tok] + XH + XG)
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    assert torch.equal(outs[2][0], outs[3][0]) and torch.equal(outs[0][0], outs[2][0])
    assert outs[0][0].abs().sum() > 0 and outs[0][1][1:, 3 * B:4 * B].abs().sum() > 0


@pytest.mark.parametrize('M,N,K,dt,pad', [(384, 128, 4096, 3, 0), (130, 512, 2000, 2, 6), (1536, 128, 1056, 1, 0), (64, 130, 999, 0, 6),
                                          (128, 135, 640, 0, 1), (3072, 36, 512, 1, 4), (12, 512, 4100, 0, 0), (256, 1000, 8192, 3, 0)])
def test_wgrad_kernel_vs_fp64_product_and_column_sums(M, N, K, dt, pad):
    """csrc/wgrad.hip (transposing-LDS-read weight-gradient kernel): C += A^T B and the fused bias gradient sum_k A[k, :] against
    a float64 product of the bf16-rounded operands -- fp32 / bf16 sources, widths that are not multiples of 8 or 128 (130, 135, 36,
    12), K that is not a multiple of the 32-row stage, row strides that are / are not 16-byte aligned (`pad`), every slab count"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(K, M + pad, generator=g)
    B = torch.randn(K, N + pad, generator=g)
    Ad = (A.to(bf) if dt & 1 else A).to(dev)[:, :M]
    Bd = (B.to(bf) if dt & 2 else B).to(dev)[:, :N]
    A64, B64 = A[:, :M].to(bf).double(), B[:, :N].to(bf).double()
    C0 = torch.randn(M, N, generator=g)
    b0 = torch.randn(M, generator=g)
    want = C0.double() + 0.5 * (A64.t() @ B64)
    want_b = b0.double() + A64.sum(0)
    for slabs, dma in ((0, 0), (1, 0), (3, 0), (8, 0), (0, 1), (3, 1)):          # dma: the LDS-DMA kernel (bf16 x bf16 operands only; else a no-op)
        C = C0.to(dev).clone()
        bias = b0.to(dev).clone()
        call('ptv_wgrad_dma', dma)
        try:
            call('ptv_wgrad', M, N, K, ptr(Ad), Ad.stride(0), ptr(Bd), Bd.stride(0), ptr(C), C.stride(0), 0.5, 1, dt, slabs, ptr(bias), None, 0, 0,
                 stream_ptr())
            torch.cuda.synchronize()
        finally:
            call('ptv_wgrad_dma', 0)
        sc = max(1.0, want.abs().max().item())
        assert (C.cpu().double() - want).abs().max() < 2e-5 * sc, (slabs, dma)
        assert (bias.cpu().double() - want_b).abs().max() < 2e-5 * max(1.0, want_b.abs().max().item()), (slabs, dma)
    # k_top: the caller knows the rows from (k_top + 1) * k_unit on are zero -- same result on an operand where they are
    if K >= 128:
        unit, top = 32, (K // 32) // 2 - 1
        Az = Ad.clone(); Az[(top + 1) * unit:] = 0
        ktop = torch.tensor([top], device=dev, dtype=torch.int32)
        Ca, Cb = C0.to(dev).clone(), C0.to(dev).clone()
        ba, bb = b0.to(dev).clone(), b0.to(dev).clone()
        call('ptv_wgrad', M, N, K, ptr(Az), Az.stride(0), ptr(Bd), Bd.stride(0), ptr(Ca), Ca.stride(0), 0.5, 1, dt, 0, ptr(ba), None, 0, 0, stream_ptr())
        call('ptv_wgrad', M, N, K, ptr(Az), Az.stride(0), ptr(Bd), Bd.stride(0), ptr(Cb), Cb.stride(0), 0.5, 1, dt, 0, ptr(bb), ptr(ktop), unit, 0,
             stream_ptr())
        torch.cuda.synchronize()
        assert (Ca - Cb).abs().max() < 2e-5 * max(1.0, Ca.abs().max().item()) and (ba - bb).abs().max() < 2e-5 * max(1.0, ba.abs().max().item())
        # the reversed order (gradients of a *_reverse GRU direction, indexed by processing step): a zero PREFIX
        U = K // unit
        Ar = Ad.clone(); Ar[:(U - top - 1) * unit] = 0
        Ca, Cb = C0.to(dev).clone(), C0.to(dev).clone()
        call('ptv_wgrad', M, N, K, ptr(Ar), Ar.stride(0), ptr(Bd), Bd.stride(0), ptr(Ca), Ca.stride(0), 0.5, 1, dt, 0, None, None, 0, 0, stream_ptr())
        call('ptv_wgrad', M, N, K, ptr(Ar), Ar.stride(0), ptr(Bd), Bd.stride(0), ptr(Cb), Cb.stride(0), 0.5, 1, dt, 0, None, ptr(ktop), unit, U,
             stream_ptr())
        torch.cuda.synchronize()
        assert (Ca - Cb).abs().max() < 2e-5 * max(1.0, Ca.abs().max().item())
    # accumulate = 0 overwrites C; through ptv_gemm (the route the autograd functions take)
    C = torch.full((M, N), 7.0, device=dev)
    call('ptv_wgrad', M, N, K, ptr(Ad), Ad.stride(0), ptr(Bd), Bd.stride(0), ptr(C), C.stride(0), 1.0, 0, dt, 0, None, None, 0, 0, stream_ptr())
    C2 = C0.to(dev).clone()
    call('ptv_gemm', 1, 1, 1, M, N, K, ptr(Ad), Ad.stride(0), ptr(Bd), Bd.stride(0), ptr(C2), C2.stride(0), None, 0.5, 1, 0, 0, dt, stream_ptr())
    torch.cuda.synchronize()
    full = A64.t() @ B64
    assert (C.cpu().double() - full).abs().max() < 2e-5 * max(1.0, full.abs().max().item())
    assert (C2.cpu().double() - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('B', [1, 5, 64])
def test_embed_fwd_and_multihot_kernels_vs_oracle_emb_x(B):
    """ptv_embed_fwd (column gather + 5 duration columns, ptvae.py:292-313,531-535) and ptv_multihot in isolation: the embedding
    and lengths of the synthetic grid against the oracle's dense one-hot Linear; the multi-hot rows bit-exact (0 / 1 / 2 values), in
    the step-major row order (note, step, sample) the backward's weight-gradient product reads"""
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    from oracle.ptvae_oracle import Oracle
    from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
    dev = _dev()
    x, _, _ = synth_batch(B, 4242 + B)
    xt = torch.from_numpy(x)
    g = torch.Generator().manual_seed(B)
    E = 128
    w, b = torch.randn(E, 135, generator=g) / 11, torch.randn(E, generator=g)
    orc_ = Oracle.__new__(Oracle)
    orc_.p = {'decoder.note_embedding.weight': w, 'decoder.note_embedding.bias': b}
    emb_ref, len_ref = orc_.emb_x(xt)                                          # [B,32,16,E], [B,32]
    xd = xt.to(dev)
    emb = torch.empty(16, 32, B, E, device=dev)
    lengths = torch.empty(32 * B, device=dev, dtype=torch.int32)
    wd, bd = w.to(dev), b.to(dev)                                              # kept alive across the launch
    call('ptv_embed_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(emb), ptr(lengths), B, E, stream_ptr())
    mh = torch.full((16 * 32 * B, 136), -1.0, device=dev)
    call('ptv_multihot', ptr(xd), ptr(mh), 136, B, stream_ptr())
    torch.cuda.synchronize()
    assert (emb.permute(2, 1, 0, 3).cpu() - emb_ref).abs().max() < 2e-5       # fp32, 7 addends in a different order
    assert torch.equal(lengths.view(32, B).t().cpu().long(), len_ref)
    onehot = torch.zeros(B, 32, 16, 131)
    onehot.scatter_(-1, xt[..., 0:1], 1.0)
    want = torch.cat([onehot[..., :130], xt[..., 1:].float()], -1).permute(2, 1, 0, 3).reshape(16 * 32 * B, 135)
    assert torch.equal(mh[:, :135].cpu(), want)
    assert (mh[:, 135] == -1.0).all()                                          # the padding column is the caller's


@pytest.mark.parametrize('acc', [False, True])
def test_gemm_row_limit_and_sum_steps_plane_limit_equal_the_dense_kernels(acc):
    """ptv_gemm_mtop / ptv_sum_steps_top: a producer declares the tail of an operand zero through a device int (the notes BPTT's
    top_step); the limited kernels give what the dense ones give on an operand whose tail IS zero -- bias rows included"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    T, R, K, N = 7, 192, 96, 40
    top = torch.tensor([3], device=dev, dtype=torch.int32)
    A = torch.randn(T, R, K, generator=g).to(dev)
    A[4:] = 0
    W = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    for prec, a_ in ((0, A), (1, A.to(torch.bfloat16))):
        c0 = torch.randn(T * R, N, generator=g).to(dev)
        dense = F_.gemm(a_.view(T * R, K), W, c0.clone() if acc else None, bias=None if acc else bias, acc=acc, prec=prec)
        lim = F_.gemm(a_.view(T * R, K), W, c0.clone() if acc else None, bias=None if acc else bias, acc=acc, prec=prec, m_top=top, m_unit=R)
        assert torch.equal(dense, lim)
        s0 = F_.sum_steps(a_)
        s1 = F_.sum_steps(a_, t_top=top)
        assert torch.equal(s0, s1)


@pytest.mark.parametrize('B', [40, 16, 250, 1000])
def test_note_loop_cluster_mode_is_bit_identical_to_one_workgroup_per_panel(B):
    """ptv_free_note_loop with S = 2 / 4 workgroups per 16-sample panel (each streams 1/S of the notes-GRU gate weights, the new
    bf16 state is all-gathered once per note step through agent-scope 8-byte stores / loads, the heads are computed redundantly):
    the arithmetic per unit is the same, so logits, decisions, tokens and states must equal the S = 1 launch BIT FOR BIT -- over
    two consecutive time steps (the arrival counters run on), inference and training mode; every panel counts 15 * S arrivals per
    launch and no member gave up waiting"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_
    from polyphonic_chord_texture_disentanglement_amd._lib import call, stream_ptr
    from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    m = DisentangleVAE.init_model(dev).to(dev)
    P = dict(m.decoder.named_parameters())
    R, M = 32 * B, 15 * 32 * B
    panels = (B + 15) // 16
    pk = FF_._free_packs(P, 1024)
    w_ih_d, b_ih_d = P['dec_dur_gru.weight_ih_l0'], P['dec_dur_gru.bias_ih_l0']
    tab0 = F_.gemm(P['dur_sos_token'].view(1, -1), w_ih_d, bias=b_ih_d, prec=0)
    tab = F_.gemm(F_._onehot2x5(dev), w_ih_d, bias=b_ih_d, prec=0)
    wl = F_._parr([pk['wg_h'], pk['wg_t'], pk['wp'], pk['wd_h'], pk['wd_p'], pk['wdur'], P['dec_notes_gru.bias_hh_l0'],
                   P['pitch_out_linear.bias'], P['dur_hid_linear.bias'], P['dec_dur_gru.bias_hh_l0'], tab0, tab,
                   P['dur_out_linear.weight'], P['dur_out_linear.bias'], pk['w_embT'], P['note_embedding.bias']])
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5)
    GC = torch.randn(2, B, 1536, device=dev, generator=g) * 0.6
    HN0 = torch.randn(R, 512, device=dev, generator=g) * 0.5
    TOK0 = torch.randn(R, 128, device=dev, generator=g) * 0.5
    emb = torch.randn(16, R, 128, device=dev, generator=g) * 0.5
    res = {}
    # S = -1: one workgroup per panel with the head weights STREAMED from L2 every note step (train bit 21: the kernel before round 6
    # kept nothing resident) -- same products in the same k order, so also bit-equal
    for S in (1, -1, 2, 4, 8):
        if panels * S > torch.cuda.get_device_properties(dev).multi_processor_count:       # (one member per CU at most: B = 1000 -> 63 x 4 = 252)
            continue
        for train in (0, 1):
            HN = torch.zeros(16, R, 512, device=dev); HN[0] = HN0
            gates_n = torch.zeros(15, 4, R, 512, device=dev, dtype=bf)
            pitch = torch.zeros(M, 136, device=dev)
            HD = torch.zeros(6, M, 64, device=dev)
            gates_d = torch.zeros(5, 4, M, 64, device=dev, dtype=bf)
            dur = torch.zeros(M, 10, device=dev)
            idx = torch.zeros(5, M, device=dev, dtype=torch.int32)
            TOK = torch.zeros(15, R, 128, device=dev); TOK[0] = TOK0
            PRED = torch.zeros(16, R, 128, device=dev)
            xhat = torch.zeros(B, 32, 16, 6, device=dev, dtype=torch.long)
            plen = torch.zeros(R, device=dev, dtype=torch.int32)
            xch = torch.zeros(panels * 2 * 16 * 512 * 2, device=dev, dtype=bf)
            cnt = torch.zeros(panels + 1, device=dev, dtype=torch.int32)
            for t in (0, 1):
                io = F_._parr([GC[t], emb, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen, None, None, None, None, None, None,
                               xch if S > 1 else None, cnt if S > 1 else None])
                call('ptv_free_note_loop', wl, io, 136, B, t, 0x15a5 if train else 0,
                     train | 0x10000 | (0x400000 if S == 8 else ((S if S > 1 else 0) << 18)) | (0x200000 if S < 0 else 0), stream_ptr())
            torch.cuda.synchronize()
            if S > 1:
                c = cnt.cpu()
                assert int(c[-1]) == 0 and (c[:-1] == 2 * 15 * S).all(), c
            res[S, train] = [v.clone() for v in (pitch, dur, idx, xhat, plen, PRED, TOK, HN, gates_n, HD, gates_d)]
    names = ('pitch', 'dur', 'idx', 'xhat', 'plen', 'PRED', 'TOK', 'HN', 'gates_n', 'HD', 'gates_d')
    for (S, train), got in res.items():
        if S == 1:
            continue
        ref = res[1, train]
        for n, a_, b_ in zip(names, ref, got):
            assert torch.equal(a_, b_), (S, train, n, float((a_.float() - b_.float()).abs().max()))
    assert (res[1, 1][3][:, :2, 1:, 0] != 0).any()             # the decisions are not trivially constant


def test_chain_priority_marker_changes_nothing_but_scheduling():
    """ptv_gemm_priority: plain and weight-gradient products enqueued with the marker set raise their wave priority (s_setprio) -- same
    arithmetic, bit-identical results"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd._lib import lib
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(2)
    a = torch.randn(700, 520, device=dev, generator=g)
    b = torch.randn(300, 520, device=dev, generator=g)
    dy = (torch.randn(4096, 256, device=dev, generator=g)).to(torch.bfloat16)
    x = (torch.randn(4096, 192, device=dev, generator=g)).to(torch.bfloat16)
    outs = []
    for p in (0, 1, 0):
        lib().ptv_gemm_priority(p)
        F_._SIDE_DEPTH[1] = p
        old, F_.CHAIN_PRIO = F_.CHAIN_PRIO, False            # (keep the wrapper from resetting the marker)
        try:
            c = F_.gemm(a, b, prec=1)
            gw = torch.zeros(256, 192, device=dev)
            F_.wgrad_bias(dy, x, gw, None, 1)                 # one K slab would be exact; several slabs add atomically: compare loosely
        finally:
            F_.CHAIN_PRIO = old
        torch.cuda.synchronize()
        outs.append((c.clone(), gw.clone()))
    lib().ptv_gemm_priority(0)
    F_._SIDE_DEPTH[1] = 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])
    ref = dy.float().t() @ x.float()
    for _, gw in outs:
        assert (gw - ref).abs().max() <= 2e-3 * ref.abs().max()


def test_binding_rejects_host_tensors_and_checks_arguments_in_the_test_suite():
    """_lib.ptr: a CPU tensor is always refused (there is no CPU fallback); dtype and current-device checks run under PTV_PTR_CHECKS,
    which tests/conftest.py switches on"""
    from polyphonic_chord_texture_disentanglement_amd import _lib
    assert _lib.PTR_CHECKS
    with pytest.raises(AssertionError):
        _lib.ptr(torch.zeros(4))
    with pytest.raises(AssertionError):
        _lib.ptr(torch.zeros(4, device='cuda:0', dtype=torch.float64))
    t = torch.zeros(4, device='cuda:0')
    assert _lib.ptr(t) == t.data_ptr() and _lib.ptr(None) is None
    assert int(_lib.stream_ptr() or 0) == int(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('M,top,unit', [(1000, None, 0), (4096, 1, 1024), (4096, -1, 1024), (300, 0, 96)])
def test_fused_heads_kernels_vs_fp32_products_of_the_same_bf16_operands(M, top, unit):
    """csrc/heads.hip: decode_note's two Linears (ptvae.py:343-352) for M note summaries in ONE pass, and their input gradients in one
    pass -- against plain fp32 products of the bf16-rounded operands (what the three ptv_gemm calls they replace compute): logits,
    duration state (+ its bf16 copy), dP in place, dNSUM row-major and column-blocked by 32, ragged M, the zero-row limit"""
    from polyphonic_chord_texture_disentanglement_amd import functional as F_
    from polyphonic_chord_texture_disentanglement_amd._lib import call, ptr, stream_ptr
    dev = _dev()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M)
    w_p = (torch.randn(130, 512, generator=g) * 0.05).to(dev)
    w_dh = (torch.randn(64, 642, generator=g) * 0.05).to(dev)
    b_p, b_dh = torch.randn(130, generator=g).to(dev) * 0.1, torch.randn(64, generator=g).to(dev) * 0.1
    hn = (torch.randn(M, 512, generator=g) * 0.5).to(dev).to(bf)
    hp = F_.heads_packs(w_p, w_dh)
    r = lambda t: t.to(bf).float()
    pitch = torch.full((M, 136), 7.0, device=dev)
    hd0 = torch.zeros(M, 64, device=dev)
    hd16 = torch.zeros(M, 64, device=dev, dtype=bf)
    call('ptv_heads_fwd', ptr(hn), ptr(hp['wp']), ptr(hp['wdh']), ptr(hp['wdp']), ptr(b_p), ptr(b_dh), ptr(pitch), 136, ptr(hd0), ptr(hd16), M,
         stream_ptr())
    p_ref = hn.float() @ r(w_p).t() + b_p
    h_ref = hn.float() @ r(w_dh[:, :512]).t() + r(p_ref) @ r(w_dh[:, 512:]).t() + b_dh
    assert (pitch[:, :130] - p_ref).abs().max() < 2e-4 * max(1.0, p_ref.abs().max().item())
    assert (pitch[:, 130:] == 7.0).all()                                  # the row padding is not touched
    assert (hd0 - h_ref).abs().max() < 3e-3 * max(1.0, h_ref.abs().max().item())      # (a logit that rounds the other way in bf16 moves it)
    assert torch.equal(hd16, hd0.to(bf))
    # ---- backward
    live = M if top is None else min(M, (top + 1) * unit)
    dP0 = torch.randn(M, 136, generator=g).to(dev) * 0.1
    dhd = torch.randn(M, 64, generator=g).to(dev) * 0.1
    dP0[live:] = 0
    dhd[live:] = 0
    top_t = torch.tensor([top], device=dev, dtype=torch.int32) if top is not None else None
    dp_ref = dP0[:, :130] + r(dhd) @ r(w_dh[:, 512:])
    dn_ref = r(dp_ref) @ r(w_p) + r(dhd) @ r(w_dh[:, :512])
    for blocked in (0, 1):
        dP = dP0.clone()
        dn = torch.full((M * 512,), 3.0, device=dev).to(bf)
        dy = torch.full((M, 200), 9.0, device=dev).to(bf)
        call('ptv_heads_bwd', ptr(dP), 136, ptr(dhd), ptr(hp['wdpT']), ptr(hp['wcat']), ptr(dn), blocked, ptr(dy), ptr(top_t), unit, M,
             stream_ptr())
        lv = ((live + 127) // 128) * 128                                  # (whole workgroups of dead rows write nothing)
        assert torch.equal(dy[:live, :130], dP[:live, :130].to(bf)) and (dy[:min(lv, M), 130:136] == 0).all()
        assert torch.equal(dy[:live, 136:], dhd[:live].to(bf))
        assert (dP[:, :130] - dp_ref).abs().max() < 2e-4 * max(1.0, dp_ref.abs().max().item())
        assert torch.equal(dP[:, 130:], dP0[:, 130:])
        got = dn.view(16, M, 32).permute(1, 0, 2).reshape(M, 512) if blocked else dn.view(M, 512)
        assert (got.float() - dn_ref).abs().max() < 1.5e-2 * max(1.0, dn_ref.abs().max().item())
        assert (got[live:] == 0).all()

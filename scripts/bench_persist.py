"""Per-step time of the persistent GRU launches (csrc/gru_persist.hip) next to the per-step kernels (csrc/gru.hip) at the
shapes the B=512 train step dispatches; BPTT variants: classic (every workgroup reads all of K) vs split-K teams of 2 / 4.
python scripts/bench_persist.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, lib, ptr, stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3       # us


def case(NC, M, H, T, step_kernels=True):
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    fw, bw = [], []
    for ci in range(NC):
        hall = torch.zeros(T + 1, M, H, device=dev)
        hall[0] = rn(M, H) * 0.5
        c = dict(gi=(rn(T, M, 3 * H) * 0.5).to(bf), gi_step=M * 3 * H, gi_ld=3 * H, gi2=(rn(M, 3 * H) * 0.5).to(bf), gi2_step=0,
                 gi2_ld=3 * H, w16=(rn(3 * H, H) / H ** 0.5).to(bf), b_hh=rn(3 * H) * 0.1, hall=hall,
                 hall16=torch.zeros(T + 1, M, H, device=dev, dtype=bf), gates=torch.rand(T, 4, M, H, device=dev).to(bf),
                 lengths=None, reverse=bool(ci & 1))
        fw.append(c)
        bw.append(dict(hall=hall, gates=c['gates'], wt16=c['w16'].t().contiguous(), dh_ext=rn(T, M, H) * 0.1, dh_last=None,
                       dgi=torch.zeros(T, M, 3 * H, device=dev, dtype=bf), dgh=torch.zeros(T, M, 3 * H, device=dev, dtype=bf),
                       dh0=torch.zeros(M, H, device=dev), reverse=bool(ci & 1)))
    FL = 1 | 2 | 4 | 8 | 16
    dhz = torch.empty(2, M, H, device=dev)

    def step_fwd():
        for c in fw:
            call('ptv_gru_seq_fwd', 1, M, H, T, ptr(c['gi']), M * 3 * H, 3 * H, ptr(c['gi2']), 0, 3 * H, ptr(c['w16']), ptr(c['b_hh']),
                 ptr(c['hall']), ptr(c['hall16']), ptr(c['gates']), None, int(c['reverse']), None, FL, stream_ptr())

    def step_bwd():
        for b in bw:
            de = b['dh_ext']
            call('ptv_gru_seq_bwd', 1, M, H, T, ptr(b['hall']), ptr(b['gates']), ptr(b['wt16']), ptr(de), de.stride(0), de.stride(1),
                 None, 0, None, 0, 0, 0, None, ptr(b['dgi']), ptr(b['dgh']), ptr(dhz), ptr(b['dh0']), int(b['reverse']), FL, stream_ptr())

    out = 'NC=%d M=%4d H=%4d T=%2d |' % (NC, M, H, T)
    if lib().ptv_gru_persist_supported(NC, M, H):
        if step_kernels:
            t_sf, t_sb = timeit(step_fwd), timeit(step_bwd)
            out += ' step kernels fwd %.1f bwd %.1f us/step/chain |' % (t_sf / T / NC, t_sb / T / NC)
        t_pf = timeit(lambda: F_.gru_persist_fwd(M, H, T, fw))
        out += ' persistent fwd %7.1f us (%.1f/step) |' % (t_pf, t_pf / T)
    else:
        out += ' forward / classic BPTT unsupported |'
    for S in (0, 2, 4):
        F_.PERSIST_SPLITK = S
        if S == 0 and not lib().ptv_gru_persist_supported(NC, M, H):
            continue
        if S and not lib().ptv_gru_persist_splitk_supported(NC, M, H, S):
            out += ' S=%d unsupported' % S
            continue
        t_pb = timeit(lambda: F_.gru_persist_bwd(M, H, T, bw))
        out += ' bwd S=%d %7.1f us (%.1f/step)' % (S, t_pb, t_pb / T)
    F_.persist_check()
    print(out, flush=True)


if __name__ == '__main__':
    case(1, 512, 1024, 32)      # dec_time_gru
    case(2, 512, 1024, 8)       # one encoder's bi-GRU
    case(4, 512, 1024, 8)       # both encoders' bi-GRUs in one launch (split-K only)
    case(1, 512, 512, 8)        # chord decoder
    case(1, 1024, 1024, 32)     # dec_time_gru at B = 1024 (configs[4])
    case(2, 1024, 1024, 8)
    case(1, 256, 1024, 32)
    case(1, 128, 1024, 32)

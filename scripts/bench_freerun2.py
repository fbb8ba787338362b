"""In-situ timing of the note-loop launches inside the real decode (events around each call), with optional removal of the
neighbouring launches, to find what makes the kernel slower in context than alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonic_chord_texture_disentanglement_amd import functional as F_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_  # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
m = DisentangleVAE.init_model(dev).to(dev).set_precision('bf16')
z = torch.randn(B, 512, device=dev)
orig_call = FF_.call
evs = []
skip = set()


def timed_call(name, *args):
    if name in skip:
        return
    if name == 'ptv_free_note_loop':
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *args)
        e1.record()
        evs.append((e0, e1))
    else:
        orig_call(name, *args)


FF_.call = timed_call
for label, sk, gemm_off in (('real decode', set(), False), ('no resummarize', {'ptv_free_resummarize'}, False),
                            ('no resummarize, no K1 products / step', {'ptv_free_resummarize'}, True)):
    skip = sk
    og, ogs = FF_.gemm, FF_.gru_step
    if gemm_off:
        FF_.gemm = lambda a, b, out=None, **kw: out if out is not None else torch.zeros(a.shape[0], b.shape[0], device=dev)
        FF_.gru_step = lambda *a, **k: None
    for it in range(3):
        del evs[:]
        with torch.no_grad():
            m.decoder(z, True, None, None, 0., 0.)
        torch.cuda.synchronize()
    FF_.gemm, FF_.gru_step = og, ogs
    ts = [a.elapsed_time(b) * 1e3 for a, b in evs]
    print('%-40s note_loop: n=%d mean %.1f us  min %.1f  max %.1f' % (label, len(ts), sum(ts) / len(ts), min(ts), max(ts)), flush=True)

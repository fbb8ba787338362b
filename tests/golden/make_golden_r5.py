"""Round-5 golden vectors at the batch sizes where the SIZE-SPECIFIC free-running kernels engage, produced by running the REFERENCE itself
(build container only; same recipe and helper classes as make_golden.py: `/root/reference` imported unmodified behind the two third-party
stubs).  Fixtures hold seeds and expected outputs only.

  full_infer_b2048.npz   BASELINE configs[3] at ITS batch: inference_decode(z_chd, z_rhy) with z ~ randn (torch.manual_seed(31), [2048, 256]
                         each -- the test re-draws them): est_x (int16), the decisions' top-2 margins where they are below 1e-3 (sparse:
                         index + value; every other decision has a margin >= 1e-3), 1024-element slices of the decoder's pitch / duration
                         logits (model.py:124-131, ptvae.py:370-428 with inference = True).  At 2048 samples = 128 panels the note loop runs
                         its producer / head split kernel (freerun.hip note_loop2_kernel), which no smaller fixture reaches.
  full_tf0_b1024.npz     one free-running TRAINING step at configs[4]'s per-GPU batch (B = 1024, tfr = 0: train.py's schedule from its third
                         batch on): eps, the 487 coins, the argmax trace (pitch int16, duration bits packed) with near-tie margins (sparse),
                         11 losses, logit slices, per-tensor gradient norm / sum and 64-element gradient slices.  64 panels: the note loop's
                         4-member cluster mode.

    python tests/golden/make_golden_r5.py [infer2048] [tf0_1024]      # needs /root/reference; ~2 min / ~25 min, <= 40 GB
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden_r2 import top2_margin  # noqa: E402
from make_golden_r4 import full_model, grad_slices  # noqa: E402

NEAR = 1e-3


def sparse_near(margin):
    """(flat indices, values) of the decisions whose margin is below NEAR"""
    flat = margin.reshape(-1)
    idx = np.nonzero(flat < NEAR)[0].astype(np.int64)
    return idx, flat[idx].astype(np.float32)


def slices(a, n=1024):
    flat = a.reshape(-1)
    idx = np.linspace(0, flat.size - 1, n).astype(np.int64)
    return idx, flat[idx].astype(np.float32)


def main():
    what = set(sys.argv[1:]) or {'infer2048', 'tf0_1024'}
    ref_model, ref_ptvae, ref_tp = mg.import_reference()
    mf = full_model(ref_model)

    if 'infer2048' in what:
        B = 2048
        torch.manual_seed(31)
        z_chd, z_rhy = torch.randn(B, 256), torch.randn(B, 256)
        with torch.no_grad():
            po, do = mf.decoder(torch.cat([z_chd, z_rhy], -1), True, None, None, 0., 0.)
        est_x = mf.inference_decode(z_chd, z_rhy)
        po, do = po.numpy(), do.numpy()
        pm, dm = top2_margin(po), np.abs(do[..., 0] - do[..., 1]).astype(np.float32)
        out = OrderedDict(B=np.int64(B), z_seed=np.int64(31), est_x=est_x.astype(np.int16))
        out['pitch_near.idx'], out['pitch_near.val'] = sparse_near(pm)
        out['dur_near.idx'], out['dur_near.val'] = sparse_near(dm)
        out['pitch_outs.idx'], out['pitch_outs.val'] = slices(po)
        out['dur_outs.idx'], out['dur_outs.val'] = slices(do)
        out['z_chd.sum'], out['z_rhy.sum'] = np.float64(z_chd.double().sum().item()), np.float64(z_rhy.double().sum().item())
        np.savez_compressed(os.path.join(HERE, 'full_infer_b2048.npz'), **out)
        print('full_infer_b2048: est_x', est_x.shape, 'near-tie pitch decisions', out['pitch_near.idx'].size, 'dur', out['dur_near.idx'].size,
              'pitch margin min %.2e' % pm.min())

    if 'tf0_1024' in what:
        B = 1024
        res = mg.run_case(mf, B, 1513, 17, (0., 0., 0.))
        po, do = res['pitch_outs'], res['dur_outs']
        pm, dm = top2_margin(po), np.abs(do[..., 0] - do[..., 1]).astype(np.float32)
        out = OrderedDict(B=np.int64(B), data_seed=np.int64(1513), rng_seed=np.int64(17), tfr=res['tfr'], beta=res['beta'], weights=res['weights'],
                          coins=res['coins'], losses=res['losses'],
                          eps_chd=res['eps_chd'].astype(np.float32), eps_rhy=res['eps_rhy'].astype(np.float32),
                          pitch_inds=po.argmax(-1).astype(np.int16), dur_bits=np.packbits(do.argmax(-1).astype(np.uint8).reshape(-1)),
                          root_inds=res['recon_root'].argmax(-1).astype(np.int8), bass_inds=res['recon_bass'].argmax(-1).astype(np.int8),
                          chroma_bits=np.packbits(res['recon_chroma'].argmax(-1).astype(np.uint8).reshape(-1)))
        out['pitch_near.idx'], out['pitch_near.val'] = sparse_near(pm)
        out['dur_near.idx'], out['dur_near.val'] = sparse_near(dm)
        out['pitch_outs.idx'], out['pitch_outs.val'] = slices(po)
        out['dur_outs.idx'], out['dur_outs.val'] = slices(do)
        for k in ('recon_root', 'recon_chroma', 'recon_bass'):
            out[k + '.idx'], out[k + '.val'] = slices(res[k], 512)
        for k, v in res.items():
            if k.startswith('grad.'):
                out['gnorm.' + k[5:]] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
                out['gsum.' + k[5:]] = np.float64(v.astype(np.float64).sum())
        grad_slices(res, out)
        np.savez_compressed(os.path.join(HERE, 'full_tf0_b1024.npz'), **out)
        print('full_tf0_b1024: losses', res['losses'][:4], 'near-tie pitch', out['pitch_near.idx'].size, 'dur', out['dur_near.idx'].size)


if __name__ == '__main__':
    main()

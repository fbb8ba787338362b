"""DisentangleVAE (alias PolyphonicVAE): chord/texture disentanglement VAE, MI355X train-step path.

Host-side mirror of the reference `model.py` (DisentangleVAE :11-96, inference family :117-184,
init_model :244-265): same constructor, `forward(mode, ...)` dispatch, `run` / `loss` /
`loss_function` / `kl_loss` / `chord_loss` signatures and return tuples, same `state_dict` keys.
All arithmetic runs in libptvae_hip.so kernels.
"""
import os

import torch

from . import functional as F_
from .amc_dl.torch_plus import PytorchModel
from .optim import refresh_weight_shadows
from .ptvae import HipNormal, PtvaeDecoder, RnnDecoder, RnnEncoder, TextureEncoder

LOSS_NAMES = ['loss', 'recon_loss', 'pl', 'dl', 'kl_loss', 'kl_chd', 'kl_rhy', 'chord_loss', 'root_loss',
              'chroma_loss', 'bass_loss']                       # train.py:54-55


# Stream slots of the two encoders (pool stream = slot mod 4; autograd replays a branch's backward on the stream of its forward).  Round 5,
# after loss() stopped computing the dead note steps and the backward passes moved behind the C ABI, the schedule was measured again
# (scripts/ab_combo.py, 5-6 interleaved rounds per setting, profiles/r05_ab_runs.txt): the chord encoder on pool stream 3 -- the stream of
# the decoder's deferred weight-gradient products, so its BPTT runs AFTER them in the tail instead of beside them -- with the bi-GRUs'
# reversed-direction products on pool stream 0 (functional.BIGRU_SLOT_BWD = 4): 6.97-7.15 ms per step against 7.49-7.68 (slots 1 / 7);
# both encoders on stream 3: 7.47-7.56 (their forwards serialise); the texture encoder on 1 instead of 2: the same.
CHD_ENC_SLOT = 3
RHY_ENC_SLOT = 2
CHD_DEC_SLOT = 4        # the chord decoder beside the PianoTree decoder
# The two encoders ARE the latency chain of the head of the step (the decoder waits for z; the embedding / note summaries beside them are
# needed later): their products keep the raised wave priority although they run inside sibling-stream calls, the note-summary GRUs drop it
ENC_CHAIN = True


class DisentangleVAE(PytorchModel):

    def __init__(self, name, device, chd_encoder, rhy_encoder, decoder, chd_decoder):
        super().__init__(name, device)
        self.chd_encoder = chd_encoder
        self.rhy_encoder = rhy_encoder
        self.decoder = decoder
        self.num_step = self.decoder.num_step
        self.chd_decoder = chd_decoder
        self.eps_source = None      # optional callable (name, shape, device) -> eps tensor (tests)
        self._philox = None         # (seed, global index of this process's first sample): see use_philox()
        self._draws = 0

    # ---- precision switch: 'fp32' (exact, parity) | 'bf16' (bf16 MFMA operands, fp32 accumulate)
    def set_precision(self, precision):
        assert precision in ('fp32', 'bf16')
        for m in (self.chd_encoder, self.rhy_encoder, self.decoder, self.chd_decoder):
            m.precision = precision
        return self

    # ---- reparameterisation noise.  Default = torch's device generator (the reference draws from torch's global generator,
    # train_utils.py:33-34).  use_philox(seed, sample_offset) makes eps a pure function of (seed, draw number, GLOBAL sample
    # index, column) -- the same batch sees the same noise whether it runs on one GPU or is sharded over N ranks
    # (sample_offset = rank * per-rank batch; SURVEY.md section 8 d/e).  Every rank must make the same sequence of draws.
    def use_philox(self, seed=7, sample_offset=0):
        self._philox = (int(seed), int(sample_offset))
        self._draws = 0
        return self

    def _rsample(self, name, dist):
        eps = None
        if self.eps_source is not None:
            eps = self.eps_source(name, dist.mean.shape, dist.mean.device)
        elif self._philox is not None:
            B, Z = dist.mean.shape
            eps = torch.empty(B, Z, device=dist.mean.device, dtype=torch.float32)
            F_.call('ptv_philox_normal', F_.ptr(eps), B, Z, self._philox[0], self._draws, self._philox[1], F_.stream_ptr())
            self._draws += 1
        return dist.rsample(eps=eps)

    # ---- model.py:22-40: two helpers the reference defines and never calls (both call sites are commented out, model.py:44-46,102).
    # Plain tensor glue, kept for the method surface; not part of the hot path
    def confuse_prmat(self, pr_mat):
        """model.py:22-29: every non-zero entry of the piano-roll is also written one semitone up or down (coin per entry, clamped to
        0..127), in place.  Draw order = the reference's: one torch.randint(0, 2, (nnz,)) from the default CPU generator."""
        nz = torch.nonzero(pr_mat.long())
        eps = ((2 * torch.randint(0, 2, (nz.size(0),))) - 1).long().to(pr_mat.device)
        tgt = torch.clamp(nz[:, 2] + eps, min=0, max=127)
        pr_mat[nz[:, 0], nz[:, 1], tgt] = pr_mat[nz[:, 0], nz[:, 1], nz[:, 2]]
        return pr_mat

    def get_chroma(self, pr_mat):
        """model.py:31-40: log(1 + per-beat, per-pitch-class sum of the [B,32,128] piano-roll) -> [B,8,12]"""
        bs = pr_mat.size(0)
        pr = torch.cat([pr_mat, torch.zeros(bs, 32, 4, device=pr_mat.device, dtype=pr_mat.dtype)], dim=-1)
        c = pr.view(bs, 32, -1, 12).sum(dim=-2).view(bs, 8, 4, 12).sum(dim=-2).float()
        return torch.log(c + 1)

    # ---- model.py:42-55
    def run(self, x, c, pr_mat, tfr1, tfr2, tfr3, confuse=True):
        F_.mark('run:start')
        refresh_weight_shadows()                         # bf16 operand copies of the flat parameter buffer (if any)
        F_.mark('run:shadows')
        # the two encoders are independent of each other and of the embedding: sibling HIP streams
        # (autograd replays each branch's backward on the stream its forward ran on).  They fork FIRST: a sibling stream waits for
        # what its parent has queued so far, and the embedding (queued on the parent next) is not their input
        from .ptvae import _require_cuda
        _require_cuda(x, 'DisentangleVAE.run')               # (fails loudly off-GPU before any stream is touched)
        s_chd, s_rhy = F_.Side(CHD_ENC_SLOT, chain=ENC_CHAIN), F_.Side(RHY_ENC_SLOT, chain=ENC_CHAIN)
        self.decoder.summaries_needed = tfr1 > 0             # with tfr1 = 0 no time step is fed a ground-truth note summary
        # (creating the embedding / note-summary nodes FIRST, so that autograd runs the encoders' BPTTs before the note-summary BPTT, measured
        # inside box noise, 8.11-9.0 ms per step: the plain order stays)
        dist_chd = s_chd(lambda: self.chd_encoder(c), c)
        dist_rhy = s_rhy(lambda: self.rhy_encoder(pr_mat), pr_mat)
        try:
            embedded_x, lengths = self.decoder.emb_x(x)
        finally:
            self.decoder.summaries_needed = True
        F_.mark('run:emb_x')
        s_chd.join()
        s_rhy.join()
        F_.mark('run:encoders')
        z_chd = self._rsample('chd', dist_chd)           # chd first, then rhy (train_utils.py:33-34)
        z_rhy = self._rsample('rhy', dist_rhy)
        dec_z = torch.cat([z_chd, z_rhy], dim=-1)
        # chord decoder (8 small steps) rides a sibling stream next to the PianoTree decoder.  Its coin
        # flips come AFTER the decoder's in the reference's draw order: draw them first
        # on the host in that order, then enqueue.
        dec_coins = self.decoder.draw_coins(tfr1, tfr2)
        chd_coins = self.chd_decoder.draw_coins(tfr3)
        s_cd = F_.Side(CHD_DEC_SLOT)
        recon_root, recon_chroma, recon_bass = s_cd(
            lambda: self.chd_decoder(z_chd, False, tfr3, c, coins=chd_coins), z_chd, c)
        pitch_outs, dur_outs = self.decoder(dec_z, False, embedded_x, lengths, tfr1, tfr2, coins=dec_coins)
        s_cd.join()
        return pitch_outs, dur_outs, dist_chd, dist_rhy, recon_root, recon_chroma, recon_bass

    # ---- model.py:57-68: one fused loss node (CE with ignore_index x2, KL x2, chord CE x3)
    def loss_function(self, x, c, recon_pitch, recon_dur, dist_chd, dist_rhy, recon_root, recon_chroma,
                      recon_bass, beta, weights, weighted_dur=False):
        out = F_.VaeLossFn.apply(recon_pitch, recon_dur, dist_chd.mean, dist_chd.scale, dist_rhy.mean,
                                 dist_rhy.scale, recon_root, recon_chroma, recon_bass, x.long(), c.float(),
                                 float(beta), float(weights[0]), float(weights[1]), bool(weighted_dur))
        return F_.SplitScalarsFn.apply(out)

    # ---- model.py:70-90 (stand-alone forms; loss_function computes them fused)
    def chord_loss(self, c, recon_root, recon_chroma, recon_bass):
        out = F_.ChordLossFn.apply(recon_root, recon_chroma, recon_bass, c.float())
        return out[0], out[1], out[2], out[3]

    def kl_loss(self, *dists):
        kl_chd = F_.KlFn.apply(dists[0].mean, dists[0].scale)
        kl_rhy = F_.KlFn.apply(dists[1].mean, dists[1].scale)
        return kl_chd + kl_rhy, kl_chd, kl_rhy

    # ---- model.py:92-96.  Same positional/keyword signature (x, c, pr_mat, tfr1=0., tfr2=0., tfr3=0.,
    # beta=0.1, weights=(1, 0.5)); additionally a 4th positional TENSOR (the reference's unused `dt_x`,
    # which makes its own trainer call fail -- SURVEY.md §0.2) is accepted and ignored.
    def loss(self, x, c, pr_mat, *args, **kwargs):
        args = list(args)
        while args and torch.is_tensor(args[0]):
            args.pop(0)
        names = ('tfr1', 'tfr2', 'tfr3', 'beta', 'weights')
        if len(args) > len(names):
            raise TypeError('loss() takes at most %d scalar arguments after pr_mat' % len(names))
        p = dict(tfr1=0., tfr2=0., tfr3=0., beta=0.1, weights=(1, 0.5))
        for n, v in zip(names, args):
            if n in kwargs:
                raise TypeError("loss() got multiple values for argument '%s'" % n)
            p[n] = v
        for k, v in kwargs.items():
            if k not in p:
                raise TypeError("loss() got an unexpected keyword argument '%s'" % k)
            p[k] = v
        # run()'s outputs go nowhere but into the loss, which ignores the padded note slots: the teacher-forced decoder may leave the note
        # steps after the batch's last target uncomputed (functional.arm_live_top; run() on its own always computes all of them)
        armed = F_.arm_live_top(x) if (torch.is_grad_enabled() and p['tfr1'] >= 1. and p['tfr2'] >= 1.) else None
        try:
            outputs = self.run(x, c, pr_mat, p['tfr1'], p['tfr2'], p['tfr3'])
        finally:
            if armed is not None:
                F_.disarm_live_top()
        return self.loss_function(x, c, *outputs, p['beta'], p['weights'])

    # ---- model.py:117-122
    def inference_encode(self, pr_mat, c):
        self.eval()
        refresh_weight_shadows()                         # no-op unless the parameters changed since the last cast
        with torch.no_grad():
            dist_chd = self.chd_encoder(c)
            dist_rhy = self.rhy_encoder(pr_mat)
        return dist_chd, dist_rhy

    # ---- model.py:124-131: free-running decode; est_x = the argmax grid the step loop produced on device
    # (identical to output_to_numpy's argmax of the returned logits, ptvae.py:537-544)
    def inference_decode(self, z_chd, z_rhy):
        self.eval()
        refresh_weight_shadows()
        with torch.no_grad():
            dec_z = torch.cat([z_chd, z_rhy], dim=-1)
            self.decoder(dec_z, True, None, None, 0., 0.)
            est_x = self.decoder.last_xhat[:, :, 1:, :].cpu().numpy()
        return est_x

    # ---- model.py:133-142
    def inference(self, pr_mat, c, sample):
        self.eval()
        refresh_weight_shadows()
        with torch.no_grad():
            dist_chd = self.chd_encoder(c)
            dist_rhy = self.rhy_encoder(pr_mat)
            z_chd = self._rsample('chd', dist_chd) if sample else dist_chd.mean
            z_rhy = self._rsample('rhy', dist_rhy) if sample else dist_rhy.mean
        return self.inference_decode(z_chd, z_rhy)

    # ---- model.py:144-148
    def swap(self, pr_mat1, pr_mat2, c1, c2, fix_rhy, fix_chd):
        pr_mat = pr_mat1 if fix_rhy else pr_mat2
        c = c1 if fix_chd else c2
        return self.inference(pr_mat, c, sample=False)

    # ---- model.py:150-172
    def posterior_sample(self, pr_mat, c, scale=None, sample_chd=True, sample_txt=True):
        if scale is None and sample_chd and sample_txt:
            return self.inference(pr_mat, c, sample=True)
        dist_chd, dist_rhy = self.inference_encode(pr_mat, c)
        if scale is not None:
            dist_chd = HipNormal(dist_chd.mean, dist_chd.scale * scale)
            dist_rhy = HipNormal(dist_rhy.mean, dist_rhy.scale * scale)
        with torch.no_grad():
            # the reference draws BOTH latents (get_zs_from_dists(..., True), model.py:165) and then overrides the unsampled one with
            # its mean: the generator / draw counter advances by two draws whatever the flags say, so the sampled latent sees the
            # same noise as in the reference under a seeded generator
            z_chd, z_rhy = self._rsample('chd', dist_chd), self._rsample('rhy', dist_rhy)
            if not sample_chd:
                z_chd = dist_chd.mean
            if not sample_txt:
                z_rhy = dist_rhy.mean
        return self.inference_decode(z_chd, z_rhy)

    # ---- model.py:174-184
    def prior_sample(self, x, c, sample_chd=False, sample_rhy=False, scale=1.):
        dist_chd, dist_rhy = self.inference_encode(x, c)
        mean = torch.zeros_like(dist_rhy.mean)
        loc = torch.ones_like(dist_rhy.mean) * scale
        if sample_chd:
            dist_chd = HipNormal(mean, loc)
        if sample_rhy:
            dist_rhy = HipNormal(mean, loc)
        with torch.no_grad():
            z_chd, z_rhy = self._rsample('chd', dist_chd), self._rsample('rhy', dist_rhy)
        return self.inference_decode(z_chd, z_rhy)

    # ---- model.py:186-188
    def gt_sample(self, x):
        return x[:, :, 1:].cpu().numpy()

    # ---- model.py:190-209: decode int_count points on the path between two items' latent codes
    def interp(self, pr_mat1, c1, pr_mat2, c2, interp_chd=False, interp_rhy=False, int_count=10):
        dist_chd1, dist_rhy1 = self.inference_encode(pr_mat1, c1)
        dist_chd2, dist_rhy2 = self.inference_encode(pr_mat2, c2)
        z_chd1, z_rhy1, z_chd2, z_rhy2 = dist_chd1.mean, dist_rhy1.mean, dist_chd2.mean, dist_rhy2.mean
        z_chds = self.interp_z(z_chd1, z_chd2, int_count) if interp_chd else z_chd1.unsqueeze(1).repeat(1, int_count, 1)
        z_rhys = self.interp_z(z_rhy1, z_rhy2, int_count) if interp_rhy else z_rhy1.unsqueeze(1).repeat(1, int_count, 1)
        bs = z_chds.size(0)
        estxs = self.inference_decode(z_chds.reshape(bs * int_count, -1).contiguous(),
                                      z_rhys.reshape(bs * int_count, -1).contiguous())
        return estxs.reshape((bs, int_count, 32, 15, -1))

    # ---- model.py:211-216: [B,D] x [B,D] -> [B,int_count,D]; the reference loops over numpy rows on the host, here the
    # whole batch is one launch on the device holding z (ptv_slerp_path)
    def interp_z(self, z1, z2, int_count=10):
        z1, z2 = z1.detach().float().contiguous(), z2.detach().float().contiguous()
        if not z1.is_cuda:
            z1, z2 = z1.to(self.device), z2.to(self.device)
        B, D = z1.shape
        out = torch.empty(B, int_count, D, device=z1.device, dtype=torch.float32)
        F_.call('ptv_slerp_path', F_.ptr(z1), F_.ptr(z2), F_.ptr(out), B, D, int_count, F_.stream_ptr())
        return out

    # ---- model.py:218-242: one pair of codes (any shape); spherical interpolation of the directions, geometric of the norms
    def interp_path(self, z1, z2, interpolation_count=10):
        t1 = torch.as_tensor(z1, dtype=torch.float32)
        shape = list(t1.shape)
        out = self.interp_z(t1.reshape(1, -1), torch.as_tensor(z2, dtype=torch.float32).reshape(1, -1), interpolation_count)
        return out.reshape([interpolation_count] + shape)

    # ---- model.py:244-265
    @staticmethod
    def init_model(device=None, chd_size=256, txt_size=256, num_channel=10):
        name = 'disvae'
        if device is None:
            device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        chd_encoder = RnnEncoder(36, 1024, chd_size)
        rhy_encoder = TextureEncoder(256, 1024, txt_size, num_channel)
        chd_decoder = RnnDecoder(z_dim=chd_size)
        pt_decoder = PtvaeDecoder(note_embedding=None, dec_dur_hid_size=64, z_size=chd_size + txt_size)
        return DisentangleVAE(name, device, chd_encoder, rhy_encoder, pt_decoder, chd_decoder)


PolyphonicVAE = DisentangleVAE      # the name BASELINE.json's north_star uses

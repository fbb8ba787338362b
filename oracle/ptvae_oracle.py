"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain-PyTorch (CPU, fp32, elementary ops only: matmul / sigmoid / tanh / exp / log)
restatement of the reference's polyphonic-VAE training step, written from the algorithm, with
every function citing the reference file:line it follows (`/root/reference/...`).  It follows
the reference AS WRITTEN (three nested python loops, 2912 GRU cells, python `random` coin
flips) so that the restructured HIP path is checked against the original semantics, not
against itself.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product package (`polyphonic_chord_texture_disentanglement_amd`) never does.

PARITY PINNING: the reference has no tests or golden vectors of its own, so this
oracle is pinned against outputs of the reference itself, produced in the build container by
`tests/golden/make_golden.py` (which imports `/root/reference` unmodified) and committed as
`tests/golden/*.npz`; `tests/test_oracle_vs_golden.py` checks the oracle against them.

Parameters are a dict keyed exactly like the reference `state_dict()`.
All dimensions are derived from the parameter shapes, so reduced configurations work.
"""
import random

import torch

PITCH_SOS, PITCH_EOS, PITCH_PAD, DUR_PAD = 128, 129, 130, 2
PITCH_RANGE = 130            # ptvae.py:236  max_pitch - min_pitch + 3
DUR_WIDTH = 5
NOTE_SIZE = PITCH_RANGE + DUR_WIDTH   # 135
MAX_SIMU_NOTE = 16
NUM_STEP = 32


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def sigmoid(x):
    return 1.0 / (1.0 + torch.exp(-x))


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU cell semantics: gate rows ordered [r; z; n]."""
    H = h.shape[-1]
    gi = linear(x, w_ih, b_ih)
    gh = linear(h, w_hh, b_hh)
    r = sigmoid(gi[..., :H] + gh[..., :H])
    z = sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h


def log_softmax(x):
    m = x.max(dim=-1, keepdim=True)[0]
    s = x - m
    return s - torch.log(torch.exp(s).sum(dim=-1, keepdim=True))


def cross_entropy(logits, target, ignore_index=None):
    """nn.CrossEntropyLoss(ignore_index): mean NLL over non-ignored rows."""
    lsm = log_softmax(logits)
    if ignore_index is None:
        valid = torch.ones_like(target, dtype=torch.bool)
    else:
        valid = target != ignore_index
    tgt = torch.where(valid, target, torch.zeros_like(target))
    nll = -lsm.gather(1, tgt.unsqueeze(1)).squeeze(1)
    nll = torch.where(valid, nll, torch.zeros_like(nll))
    return nll.sum() / valid.sum().to(nll.dtype)


def kl_with_normal(mu, std):
    """amc_dl/torch_plus/train_utils.py:45-49 -- MEAN over (B x z) of KL(N(mu,std)||N(0,1))."""
    return (-torch.log(std) + (std * std + mu * mu) / 2.0 - 0.5).mean()


class Oracle:
    def __init__(self, params):
        self.p = params
        p = params
        self.note_emb_size = p['decoder.note_embedding.weight'].shape[0]
        self.dec_emb_hid = p['decoder.dec_notes_emb_gru.weight_hh_l0'].shape[1]
        self.dec_time_hid = p['decoder.dec_time_gru.weight_hh_l0'].shape[1]
        self.dec_notes_hid = p['decoder.dec_notes_gru.weight_hh_l0'].shape[1]
        self.dec_dur_hid = p['decoder.dec_dur_gru.weight_hh_l0'].shape[1]
        self.trace = None          # filled by decoder(): argmax decisions
        self.force = None          # optional dict to force argmax decisions (replay mode)

    # ---- helpers -----------------------------------------------------------------------
    def _gru(self, prefix, x, h, reverse=False):
        s = '_reverse' if reverse else ''
        p = self.p
        return gru_cell(x, h, p[prefix + '.weight_ih_l0' + s], p[prefix + '.weight_hh_l0' + s],
                        p[prefix + '.bias_ih_l0' + s], p[prefix + '.bias_hh_l0' + s])

    def _bigru_final(self, prefix, x, lengths=None):
        """Bidirectional GRU over x [N,T,I]; returns [N, 2H] = [fwd final | bwd final].
        With lengths (packed-sequence semantics, ptvae.py:446-453): fwd state after the last
        valid element, bwd state after index 0 having started at index len-1."""
        N, T, _ = x.shape
        H = self.p[prefix + '.weight_hh_l0'].shape[1]
        hf = x.new_zeros(N, H)
        hb = x.new_zeros(N, H)
        for t in range(T):
            nf = self._gru(prefix, x[:, t], hf)
            if lengths is None:
                hf = nf
            else:
                hf = torch.where((t < lengths).unsqueeze(1), nf, hf)
        for t in range(T - 1, -1, -1):
            nb = self._gru(prefix, x[:, t], hb, reverse=True)
            if lengths is None:
                hb = nb
            else:
                hb = torch.where((t < lengths).unsqueeze(1), nb, hb)
        return torch.cat([hf, hb], dim=-1)

    # ---- ptvae.py:292-313, 531-535 -------------------------------------------------------
    def emb_x(self, x):
        lengths = MAX_SIMU_NOTE - (x[..., 0] == PITCH_PAD).sum(dim=-1)       # [B,32]
        onehot = torch.zeros(x.shape[:-1] + (PITCH_RANGE + 1,), dtype=torch.float32)
        onehot.scatter_(-1, x[..., 0:1], 1.0)
        multihot = torch.cat([onehot[..., :PITCH_RANGE], x[..., 1:].float()], dim=-1)
        emb = linear(multihot, self.p['decoder.note_embedding.weight'],
                     self.p['decoder.note_embedding.bias'])
        return emb, lengths

    # ---- ptvae.py:22-29 ------------------------------------------------------------------
    def _heads(self, prefix, h):
        mu = linear(h, self.p[prefix + '.linear_mu.weight'], self.p[prefix + '.linear_mu.bias'])
        std = torch.exp(linear(h, self.p[prefix + '.linear_var.weight'],
                               self.p[prefix + '.linear_var.bias']))
        return mu, std

    def chd_encode(self, c):
        return self._heads('chd_encoder', self._bigru_final('chd_encoder.gru', c))

    # ---- ptvae.py:110-122 ----------------------------------------------------------------
    def rhy_encode(self, pr_mat):
        B = pr_mat.shape[0]
        w = self.p['rhy_encoder.cnn.0.weight']            # [C,1,4,12]
        C = w.shape[0]
        # Conv2d(1,C,(4,12),stride (4,1)) + ReLU + MaxPool(1,4): explicit patch matmul
        pr = pr_mat.view(B, 8, 4, 128)
        patches = pr.unfold(3, 12, 1)                      # [B,8,4,117,12]
        patches = patches.permute(0, 1, 3, 2, 4).reshape(B, 8, 117, 48)
        conv = patches @ w.view(C, 48).t() + self.p['rhy_encoder.cnn.0.bias']   # [B,8,117,C]
        conv = torch.clamp(conv, min=0.0).permute(0, 3, 1, 2)                   # [B,C,8,117]
        pooled = conv[..., :116].reshape(B, C, 8, 29, 4).max(dim=-1)[0]         # [B,C,8,29]
        feat = pooled.reshape(B, 8, -1)                    # RAW view (ptvae.py:114): mixes C and beat
        feat = linear(feat, self.p['rhy_encoder.fc1.weight'], self.p['rhy_encoder.fc1.bias'])
        feat = linear(feat, self.p['rhy_encoder.fc2.weight'], self.p['rhy_encoder.fc2.bias'])
        return self._heads('rhy_encoder', self._bigru_final('rhy_encoder.gru', feat))

    # ---- ptvae.py:315-334 ----------------------------------------------------------------
    def _note_token(self, pitch_inds, dur_inds):
        B = pitch_inds.shape[0]
        tok = torch.zeros(B, NOTE_SIZE)
        tok[torch.arange(B), pitch_inds] = 1.0
        tok[:, PITCH_RANGE:] = dur_inds.float()
        return linear(tok, self.p['decoder.note_embedding.weight'],
                      self.p['decoder.note_embedding.bias'])

    # ---- ptvae.py:336-368 ----------------------------------------------------------------
    def decode_note(self, note_summary, key):
        p = self.p
        B = note_summary.shape[0]
        est_pitch = linear(note_summary, p['decoder.pitch_out_linear.weight'],
                           p['decoder.pitch_out_linear.bias'])
        dur_hid = linear(torch.cat([note_summary, est_pitch], dim=-1),
                         p['decoder.dur_hid_linear.weight'], p['decoder.dur_hid_linear.bias'])
        token = p['decoder.dur_sos_token'].unsqueeze(0).expand(B, -1)
        est_durs = []
        dur_inds = []
        for d in range(DUR_WIDTH):
            dur_hid = self._gru('decoder.dec_dur_gru', token, dur_hid)
            est_dur = linear(dur_hid, p['decoder.dur_out_linear.weight'],
                             p['decoder.dur_out_linear.bias'])
            est_durs.append(est_dur)
            ind = est_dur.max(1)[1]
            if self.force is not None:
                ind = self.force['dur_inds'][key][:, d]
            dur_inds.append(ind)
            token = torch.zeros(B, DUR_WIDTH)
            token[torch.arange(B), ind] = 1.0          # one-hot5 at index argmax in {0,1}
        return est_pitch, torch.stack(est_durs, dim=1), torch.stack(dur_inds, dim=1)

    # ---- ptvae.py:370-428 ----------------------------------------------------------------
    def decode_notes(self, notes_summary, notes, inference, tfr2, coin, t_idx):
        p = self.p
        B = notes_summary.shape[0]
        hid = linear(notes_summary, p['decoder.dec_time_to_notes_hid.weight'],
                     p['decoder.dec_time_to_notes_hid.bias'])
        if inference:
            sos = torch.zeros(NOTE_SIZE)
            sos[PITCH_SOS] = 1.0
            sos[PITCH_RANGE:] = 2.0
            token = linear(sos, p['decoder.note_embedding.weight'],
                           p['decoder.note_embedding.bias']).unsqueeze(0).expand(B, -1)
        else:
            token = notes[:, 0]
        predicted_notes = [token]
        lengths = torch.zeros(B)
        pitch_outs, dur_outs = [], []
        for n in range(1, MAX_SIMU_NOTE):
            hid = self._gru('decoder.dec_notes_gru', torch.cat([notes_summary, token], dim=-1), hid)
            est_pitch, est_durs, dur_inds = self.decode_note(hid, (t_idx, n - 1))
            pitch_outs.append(est_pitch)
            dur_outs.append(est_durs)
            pitch_inds = est_pitch.max(1)[1]
            if self.force is not None:
                pitch_inds = self.force['pitch_inds'][(t_idx, n - 1)]
            self.trace['pitch_inds'][(t_idx, n - 1)] = pitch_inds
            self.trace['dur_inds'][(t_idx, n - 1)] = dur_inds
            predicted = self._note_token(pitch_inds, dur_inds)
            predicted_notes.append(predicted)
            lengths = torch.where((pitch_inds == PITCH_EOS) & (lengths == 0),
                                  torch.full_like(lengths, float(n)), lengths)
            if n == MAX_SIMU_NOTE - 1:
                break
            teacher_force = coin() < tfr2
            if inference or not teacher_force:
                token = predicted
            else:
                token = notes[:, n]
        lengths = torch.where(lengths == 0, torch.full_like(lengths, float(MAX_SIMU_NOTE - 1)),
                              lengths)
        return (torch.stack(pitch_outs, dim=1), torch.stack(dur_outs, dim=1),
                torch.stack(predicted_notes, dim=1), lengths)

    # ---- ptvae.py:430-496 ----------------------------------------------------------------
    def decoder(self, z, inference, embedded, lengths, tfr1, tfr2, coin=random.random):
        p = self.p
        B = z.shape[0]
        self.trace = {'pitch_inds': {}, 'dur_inds': {}}
        z_hid = linear(z, p['decoder.z2dec_hid_linear.weight'], p['decoder.z2dec_hid_linear.bias'])
        z_in = linear(z, p['decoder.z2dec_in_linear.weight'], p['decoder.z2dec_in_linear.bias'])
        if not inference:
            flat = embedded.reshape(-1, MAX_SIMU_NOTE, self.note_emb_size)
            x_summ = self._bigru_final('decoder.dec_notes_emb_gru', flat, lengths.reshape(-1))
            x_summ = x_summ.view(B, NUM_STEP, 2 * self.dec_emb_hid)
        token = p['decoder.dec_init_input'].unsqueeze(0).expand(B, -1)
        pitch_outs, dur_outs = [], []
        for t in range(NUM_STEP):
            z_hid = self._gru('decoder.dec_time_gru', torch.cat([token, z_in], dim=-1), z_hid)
            notes = None if inference else embedded[:, t]
            po, do, pred_notes, pred_len = self.decode_notes(z_hid, notes, inference, tfr2, coin, t)
            pitch_outs.append(po)
            dur_outs.append(do)
            if t == NUM_STEP - 1:
                break
            teacher_force = coin() < tfr1
            if teacher_force and not inference:
                token = x_summ[:, t]
            else:
                token = self._bigru_final('decoder.dec_notes_emb_gru', pred_notes, pred_len)
        return torch.stack(pitch_outs, dim=1), torch.stack(dur_outs, dim=1)

    # ---- ptvae.py:51-87 ------------------------------------------------------------------
    def chd_decoder(self, z_chd, inference, tfr, c, coin=random.random):
        p = self.p
        B = z_chd.shape[0]
        hid = linear(z_chd, p['chd_decoder.z2dec_hid.weight'], p['chd_decoder.z2dec_hid.bias'])
        z_in = linear(z_chd, p['chd_decoder.z2dec_in.weight'], p['chd_decoder.z2dec_in.bias'])
        if inference:
            tfr = 0.0
        token = p['chd_decoder.init_input'].unsqueeze(0).expand(B, -1)
        roots, chromas, basses = [], [], []
        for t in range(NUM_STEP // 4):
            hid = self._gru('chd_decoder.gru', torch.cat([token, z_in], dim=-1), hid)
            r_root = linear(hid, p['chd_decoder.root_out.weight'], p['chd_decoder.root_out.bias'])
            r_chroma = linear(hid, p['chd_decoder.chroma_out.weight'],
                              p['chd_decoder.chroma_out.bias']).view(B, 12, 2)
            r_bass = linear(hid, p['chd_decoder.bass_out.weight'], p['chd_decoder.bass_out.bias'])
            roots.append(r_root)
            chromas.append(r_chroma)
            basses.append(r_bass)
            # QUIRK (ptvae.py:74-77): the reference indexes t_root[arange(B), 0, idx] with idx of
            # shape (B,1), which broadcasts to (B,B): EVERY row receives the union over the batch
            # of all rows' argmax one-hots (same for bass).  Reproduced as written.
            t_root = torch.zeros(B, 12)
            t_root[:, r_root.max(-1)[1]] = 1.0
            t_chroma = r_chroma.max(-1)[1].float()
            t_bass = torch.zeros(B, 12)
            t_bass[:, r_bass.max(-1)[1]] = 1.0
            token = torch.cat([t_root, t_chroma, t_bass], dim=-1)
            teacher_force = coin() < tfr        # ptvae.py:79-81: the break never fires -> 8 draws
            if teacher_force and not inference:
                token = c[:, t]
        return torch.stack(roots, 1), torch.stack(chromas, 1), torch.stack(basses, 1)

    # ---- model.py:42-55 ------------------------------------------------------------------
    def run(self, x, c, pr_mat, tfr1, tfr2, tfr3, eps_chd, eps_rhy, coin=random.random):
        embedded, lengths = self.emb_x(x)
        mu_c, std_c = self.chd_encode(c)
        mu_r, std_r = self.rhy_encode(pr_mat)
        z_chd = mu_c + std_c * eps_chd           # Normal.rsample, chd first (train_utils.py:33)
        z_rhy = mu_r + std_r * eps_rhy
        pitch_outs, dur_outs = self.decoder(torch.cat([z_chd, z_rhy], -1), False, embedded,
                                            lengths, tfr1, tfr2, coin)
        root, chroma, bass = self.chd_decoder(z_chd, False, tfr3, c, coin)
        return pitch_outs, dur_outs, (mu_c, std_c), (mu_r, std_r), root, chroma, bass

    # ---- model.py:57-90, ptvae.py:498-529 ------------------------------------------------
    def loss_function(self, x, c, pitch_outs, dur_outs, dist_chd, dist_rhy, root, chroma, bass,
                      beta, weights, weighted_dur=False):
        pl = cross_entropy(pitch_outs.reshape(-1, pitch_outs.shape[-1]),
                           x[:, :, 1:, 0].reshape(-1), PITCH_PAD)
        if not weighted_dur:
            dl = cross_entropy(dur_outs.reshape(-1, 2), x[:, :, 1:, 1:].reshape(-1), DUR_PAD)
        else:                                  # ptvae.py:512-527: one ignore-index mean per bit position, fixed weights
            rd = dur_outs.reshape(-1, DUR_WIDTH, 2)
            gd = x[:, :, 1:, 1:].reshape(-1, DUR_WIDTH)
            dl = sum(w * cross_entropy(rd[:, d, :], gd[:, d], DUR_PAD) for d, w in enumerate((1.0, 0.6, 0.4, 0.3, 0.3)))
        recon = weights[0] * pl + weights[1] * dl
        kl_chd = kl_with_normal(*dist_chd)
        kl_rhy = kl_with_normal(*dist_rhy)
        kl = kl_chd + kl_rhy
        root_l = cross_entropy(root.reshape(-1, 12), c[:, :, 0:12].max(-1)[1].reshape(-1))
        chroma_l = cross_entropy(chroma.reshape(-1, 2), c[:, :, 12:24].long().reshape(-1))
        bass_l = cross_entropy(bass.reshape(-1, 12), c[:, :, 24:].max(-1)[1].reshape(-1))
        chord = root_l + chroma_l + bass_l
        loss = recon + beta * kl + chord
        return loss, recon, pl, dl, kl, kl_chd, kl_rhy, chord, root_l, chroma_l, bass_l

    # ---- model.py:92-96 ------------------------------------------------------------------
    def loss(self, x, c, pr_mat, tfr1, tfr2, tfr3, beta, weights, eps_chd, eps_rhy,
             coin=random.random):
        out = self.run(x, c, pr_mat, tfr1, tfr2, tfr3, eps_chd, eps_rhy, coin)
        return self.loss_function(x, c, *out, beta, weights)

    # ---- model.py:124-131, ptvae.py:537-544 ----------------------------------------------
    def inference_decode(self, z_chd, z_rhy):
        with torch.no_grad():
            po, do = self.decoder(torch.cat([z_chd, z_rhy], -1), True, None, None, 0.0, 0.0)
            est_x = torch.cat([po.max(-1)[1].unsqueeze(-1), do.max(-1)[1]], dim=-1)
        return est_x, po, do


# ---- ptvae.py:125-215: PtvaeEncoder (params keyed like its state_dict) --------------------------
def ptvae_encoder(p, x, multihot=None, lengths=None, max_simu_note=MAX_SIMU_NOTE, num_step=32, pitch_range=PITCH_RANGE,
                  pitch_pad=PITCH_PAD):
    """x int64 [B,S,N,1+D] -> (mu, std, embedded [B,S,N,E], lengths [B,S]); any grid geometry the constructor accepts (:127-147).
    With `multihot` / `lengths` given, this is encoder() itself (:190-206) on the caller's multi-hot grid."""
    o = Oracle.__new__(Oracle)
    o.p = dict(p)
    S, N, P = num_step, max_simu_note, pitch_range
    if multihot is None:
        lengths = N - (x[..., 0] == pitch_pad).sum(dim=-1)                       # :167-172
        onehot = torch.zeros(x.shape[:-1] + (P + 1,), dtype=torch.float32)       # :174-188
        onehot.scatter_(-1, x[..., 0:1], 1.0)
        multihot = torch.cat([onehot[..., :P], x[..., 1:].float()], dim=-1)
    emb = linear(multihot, p['note_embedding.weight'], p['note_embedding.bias'])  # :191
    B = multihot.shape[0]
    E = emb.shape[-1]
    notes = o._bigru_final('enc_notes_gru', emb.reshape(B * S, N, E), lengths.reshape(-1))        # :193-198 packed by length
    h = o._bigru_final('enc_time_gru', notes.reshape(B, S, -1))                                     # :200-202
    mu = linear(h, p['linear_mu.weight'], p['linear_mu.bias'])
    std = torch.exp(linear(h, p['linear_std.weight'], p['linear_std.bias']))                        # :204-205
    return mu, std, emb, lengths


# ---- model.py:218-242: interp_path (float64 numpy in the reference) --------------------------------
def interp_path(z1, z2, n=10):
    import numpy as np
    z1 = np.asarray(z1, dtype=np.float64).reshape(-1)
    z2 = np.asarray(z2, dtype=np.float64).reshape(-1)
    n1, n2 = np.linalg.norm(z1), np.linalg.norm(z2)
    p0, p1 = z1 / n1, z2 / n2
    t = np.linspace(0.0, 1.0, n)
    omega = np.arccos(np.dot(p0, p1))
    so = np.sin(omega)
    dirs = np.sin((1.0 - t) * omega)[:, None] / so * p0[None] + np.sin(t * omega)[:, None] / so * p1[None]
    length = np.linspace(np.log(n1), np.log(n2), n)
    return dirs * np.exp(length[:, None])


# ---- optimiser / schedule restatements -------------------------------
def scheduled_sampling(i, high=0.7, low=0.05):
    """train_utils.py:17-21"""
    import numpy as np
    z = 1 / (1 + np.exp(10 * (i - 0.5)))
    return (high - low) * z + low


def kl_anealing(i, high=0.1, low=0.0):
    """train_utils.py:24-30"""
    import numpy as np
    hh, ll = 1 - low, 1 - high
    z = 1 / (1 + np.exp(10 * (i - 0.5)))
    return 1 - ((hh - ll) * z + ll)


def clip_and_adam_step(params, grads, m, v, step, lr, clip=1.0, b1=0.9, b2=0.999, eps=1e-8):
    """clip_grad_norm_(.,clip) (module.py:142) + torch.optim.Adam defaults (train.py:50).
    In-place on lists of tensors; `step` is the 1-based Adam step.  Returns pre-clip norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(clip / (total + 1e-6), max=1.0)
    for p_, g, m_, v_ in zip(params, grads, m, v):
        g = g * coef
        m_.mul_(b1).add_(g, alpha=1 - b1)
        v_.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v_.sqrt() / (bc2 ** 0.5)).add_(eps)
        p_.addcdiv_(m_, denom, value=-lr / bc1)
    return total

"""Network blocks of the polyphonic chord/texture VAE on MI355X HIP kernels.

Host-side mirror of the reference `ptvae.py`: same class names, constructor signatures, method
names, `state_dict` keys and default initialisation (same RNG draws, so `torch.manual_seed(s)`
gives the reference's weights).  The modules hold parameters only; every forward/backward is a
sequence of libptvae_hip.so kernels (see functional.py).  There is no CPU fallback: calling a
module with CPU tensors raises.

Reference: /root/reference/ptvae.py  (RnnEncoder :11-29, RnnDecoder :32-87, TextureEncoder :90-122,
PtvaeEncoder :125-215, PtvaeDecoder :218-575).
"""
import math
import os
import random

import torch
from torch import nn

from . import functional as F_
from . import functional_free as FF_
from ._lib import prec_code

# stream slot of the ground-truth note summaries (functional.Side; autograd replays their BPTT on the same stream): -1 = the caller's
SUMMARY_SLOT = 5


# ---------------------------------------------------------------------------------------------
# parameter containers (same names / shapes / init order as torch.nn.{Linear,GRU,Conv2d})
# ---------------------------------------------------------------------------------------------
class Linear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, prec=0):
        shp = x.shape
        y = F_.LinearFn.apply(x.reshape(-1, shp[-1]), self.weight, self.bias, prec)
        return y.view(shp[:-1] + (self.out_features,))


class GRU(nn.Module):
    """Parameters of a single-layer (bi)GRU, keys weight_ih_l0 ... bias_hh_l0[_reverse]."""

    def __init__(self, input_size, hidden_size, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        for sfx in ([''] + (['_reverse'] if bidirectional else [])):
            setattr(self, 'weight_ih_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size, input_size)))
            setattr(self, 'weight_hh_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size, hidden_size)))
            setattr(self, 'bias_ih_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size)))
            setattr(self, 'bias_hh_l0' + sfx, nn.Parameter(torch.empty(3 * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def weights(self):
        names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
        if self.bidirectional:
            names += [n + '_reverse' for n in names]
        return [getattr(self, n) for n in names]


class Conv2dParams(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch, *kernel_size))
        self.bias = nn.Parameter(torch.empty(out_ch))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = in_ch * kernel_size[0] * kernel_size[1]
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.bias, -bound, bound)


class HipNormal:
    """Duck-typed torch.distributions.Normal (`.mean`, `.scale`, `.loc`, `.stddev`, `.rsample()`,
    `.sample()`) whose rsample runs the reparameterisation kernel.  The model's run() returns
    these where the reference returns Normal (model.py:45-48)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale

    mean = property(lambda self: self.loc)
    stddev = property(lambda self: self.scale)
    variance = property(lambda self: self.scale * self.scale)

    def rsample(self, sample_shape=torch.Size(), eps=None):
        if eps is None:
            eps = torch.randn(self.loc.shape, device=self.loc.device, dtype=self.loc.dtype)
        return F_.ReparamFn.apply(self.loc, self.scale, eps.contiguous())

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)


def _require_cuda(t, who):
    if not t.is_cuda:
        raise RuntimeError('%s: input is on %s -- this build runs on MI355X HIP kernels only (no CPU '
                           'fallback); move the model and inputs to a cuda device' % (who, t.device))


class _PrecMixin:
    """`precision` = 'fp32' (exact fp32 MFMA, parity path) or 'bf16' (bf16 MFMA operands)."""
    precision = 'fp32'

    @property
    def _prec(self):
        return prec_code(self.precision)


# ---------------------------------------------------------------------------------------------
class RnnEncoder(nn.Module, _PrecMixin):
    """Chord encoder: bi-GRU over 8 chord steps -> Normal(mu, exp(linear_var)).  ptvae.py:11-29"""

    def __init__(self, input_dim, hidden_dim, z_dim):
        super().__init__()
        self.gru = GRU(input_dim, hidden_dim, bidirectional=True)
        self.linear_mu = Linear(hidden_dim * 2, z_dim)
        self.linear_var = Linear(hidden_dim * 2, z_dim)
        self.input_dim, self.hidden_dim, self.z_dim = input_dim, hidden_dim, z_dim

    def forward(self, x):
        _require_cuda(x, 'RnnEncoder')
        x_sm = F_.Transpose01Fn.apply(x.float())                       # [T,B,I]
        h = F_.BiGruFinalFn.apply(x_sm, None, self._prec, *self.gru.weights())
        mu, sd = F_.EncoderHeadsFn.apply(h, self.linear_mu.weight, self.linear_mu.bias, self.linear_var.weight,
                                         self.linear_var.bias, self._prec)
        return HipNormal(mu, sd)


class TextureEncoder(nn.Module, _PrecMixin):
    """Piano-roll texture encoder (ptvae.py:90-122): conv(4x12,stride 4x1)+ReLU+maxpool(1x4) ->
    raw view [B,8,C*29] -> fc1 -> fc2 -> bi-GRU(8 steps) -> Normal."""

    def __init__(self, emb_size, hidden_dim, z_dim, num_channel=10):
        super().__init__()
        self.cnn = nn.Sequential(Conv2dParams(1, num_channel, (4, 12)))     # key 'cnn.0.*' as in the reference
        self.fc1 = Linear(num_channel * 29, 1000)
        self.fc2 = Linear(1000, emb_size)
        self.gru = GRU(emb_size, hidden_dim, bidirectional=True)
        self.linear_mu = Linear(hidden_dim * 2, z_dim)
        self.linear_var = Linear(hidden_dim * 2, z_dim)
        self.emb_size, self.hidden_dim, self.z_dim = emb_size, hidden_dim, z_dim

    def forward(self, pr):
        _require_cuda(pr, 'TextureEncoder')
        bs = pr.size(0)
        conv = self.cnn[0]
        feat = F_.TextureFrontFn.apply(pr.float(), conv.weight, conv.bias)              # [B*8, C*29]: the reference's raw .view(bs, 8, -1) of [B,C,8,29]
        feat = F_.LinearFn.apply(feat, self.fc1.weight, self.fc1.bias, self._prec)
        feat = F_.LinearFn.apply(feat, self.fc2.weight, self.fc2.bias, self._prec)
        x_sm = F_.Transpose01Fn.apply(feat.view(bs, 8, -1))
        h = F_.BiGruFinalFn.apply(x_sm, None, self._prec, *self.gru.weights())
        mu, sd = F_.EncoderHeadsFn.apply(h, self.linear_mu.weight, self.linear_mu.bias, self.linear_var.weight,
                                         self.linear_var.bias, self._prec)
        return HipNormal(mu, sd)


class RnnDecoder(nn.Module, _PrecMixin):
    """Chord decoder (ptvae.py:32-87)."""

    def __init__(self, input_dim=36, z_input_dim=256, hidden_dim=512, z_dim=256, num_step=32):
        super().__init__()
        self.z2dec_hid = Linear(z_dim, hidden_dim)
        self.z2dec_in = Linear(z_dim, z_input_dim)
        self.gru = GRU(input_dim + z_input_dim, hidden_dim)
        self.init_input = nn.Parameter(torch.rand(36))
        self.input_dim, self.hidden_dim, self.z_dim = input_dim, hidden_dim, z_dim
        self.root_out = Linear(hidden_dim, 12)
        self.chroma_out = Linear(hidden_dim, 24)
        self.bass_out = Linear(hidden_dim, 12)
        self.num_step = num_step
        self.force_trace = None            # {'root' [B,8,12], 'chroma' [B,8,12,2], 'bass' [B,8,12]} logits: replay mode (tests)

    def _params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in F_.CHD_PARAM_NAMES]

    def draw_coins(self, tfr):
        """one teacher-forcing coin per chord step for the whole batch (ptvae.py:81; the `break` at :79
        never fires, so 8 draws)"""
        return [random.random() < tfr for _ in range(int(self.num_step / 4))]

    def forward(self, z_chd, inference, tfr, c=None, coins=None):
        _require_cuda(z_chd, 'RnnDecoder')
        T = int(self.num_step / 4)
        if inference:
            tfr = 0.
        if coins is None:
            coins = self.draw_coins(tfr)
        c_sm = F_.Transpose01Fn.apply(c.float()).detach() if (c is not None and not inference) else None
        if all(coins) and not inference:
            root, chroma, bass = F_.ChordDecoderTFFn.apply(z_chd, c_sm, self._prec, *self._params())
        else:
            force = None
            if self.force_trace is not None:
                ft = self.force_trace
                force = {'root': ft['root'].float().transpose(0, 1).contiguous(), 'bass': ft['bass'].float().transpose(0, 1).contiguous(),
                         'chroma': ft['chroma'].float().reshape(z_chd.size(0), T, 24).transpose(0, 1).contiguous()}
            root, chroma, bass = FF_.ChordDecoderStepFn.apply(z_chd, c_sm, list(coins), force, self._prec, *self._params())
        bs = z_chd.size(0)
        # reference shapes [B,8,12] / [B,8,12,2] / [B,8,12] as views of the step-major buffers
        return root.transpose(0, 1), chroma.view(T, bs, 12, 2).transpose(0, 1), bass.transpose(0, 1)


class PtvaeDecoder(nn.Module, _PrecMixin):
    """PianoTree decoder (ptvae.py:218-575): time GRU (32) -> notes GRU (15) -> pitch head +
    5-step duration GRU."""

    def __init__(self, device=None, note_embedding=None, max_simu_note=16, max_pitch=127, min_pitch=0,
                 pitch_sos=128, pitch_eos=129, pitch_pad=130, dur_pad=2, dur_width=5, num_step=32,
                 note_emb_size=128, z_size=512, dec_emb_hid_size=128, dec_time_hid_size=1024,
                 dec_notes_hid_size=512, dec_z_in_size=256, dec_dur_hid_size=16):
        super().__init__()
        self.max_pitch, self.min_pitch = max_pitch, min_pitch
        self.pitch_sos, self.pitch_eos, self.pitch_pad = pitch_sos, pitch_eos, pitch_pad
        self.pitch_range = max_pitch - min_pitch + 3
        self.dur_pad, self.dur_width = dur_pad, dur_width
        self.note_size = self.pitch_range + dur_width
        self.max_simu_note, self.num_step = max_simu_note, num_step
        self.device = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
        if (max_simu_note, num_step, dur_width, self.pitch_range, pitch_pad, dur_pad) != (16, 32, 5, 130, 130, 2):
            raise NotImplementedError('HIP kernels are specialised to the 32x16x(130+5) PianoTree grid')
        self.note_emb_size, self.z_size = note_emb_size, z_size
        self.dec_z_in_size, self.dec_emb_hid_size = dec_z_in_size, dec_emb_hid_size
        self.dec_time_hid_size, self.dec_notes_hid_size = dec_time_hid_size, dec_notes_hid_size
        self.dec_dur_hid_size = dec_dur_hid_size
        self.dec_init_input = nn.Parameter(torch.rand(2 * dec_emb_hid_size))
        self.dur_sos_token = nn.Parameter(torch.rand(dur_width))
        self.note_embedding = Linear(self.note_size, note_emb_size) if note_embedding is None else note_embedding
        self.z2dec_hid_linear = Linear(z_size, dec_time_hid_size)
        self.z2dec_in_linear = Linear(z_size, dec_z_in_size)
        self.dec_notes_emb_gru = GRU(note_emb_size, dec_emb_hid_size, bidirectional=True)
        self.dec_time_gru = GRU(dec_z_in_size + 2 * dec_emb_hid_size, dec_time_hid_size)
        self.dec_time_to_notes_hid = Linear(dec_time_hid_size, dec_notes_hid_size)
        self.dec_notes_gru = GRU(dec_time_hid_size + note_emb_size, dec_notes_hid_size)
        self.pitch_out_linear = Linear(dec_notes_hid_size, self.pitch_range)
        self.dec_dur_gru = GRU(dur_width, dec_dur_hid_size)
        self.dur_hid_linear = Linear(self.pitch_range + dec_notes_hid_size, dec_dur_hid_size)
        self.dur_out_linear = Linear(dec_dur_hid_size, 2)
        self.force_dur_idx = None          # [5, 480*B] int32: replay the oracle's duration argmaxes (tests)
        self.force_trace = None            # {'pitch': [15, 32B] int32, 'dur': [5, 480B] int32}: replay mode (tests)
        self.last_xhat = None              # predicted grid [B,32,16,6] int64 of the last step-loop decode
        self.use_graph = False             # replay inference decodes from a captured hipGraph
        self._graphs = {}
        self._train_graphs = {}
        self._summary = None
        self.summaries_needed = True       # emb_x() starts the ground-truth note summaries early unless told they are dead values
        self.last_dur_idx = None

    def _params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in F_.DEC_PARAM_NAMES]

    def _graph_decode(self, z, coins):
        """Free-running decode replayed from a captured hipGraph: the step loop is ~9,000 tiny launches whose
        order and arguments depend only on (B, precision) -- capture once, then one graph launch per call.
        The graph holds raw parameter pointers, so it is re-captured if the parameter storage moves."""
        ps = self._params_free()
        key = (z.shape[0], self._prec, z.device.index, tuple(p.data_ptr() for p in ps))
        ent = self._graphs.get(key)
        if ent is None:
            static_z = z.detach().clone()
            cur = torch.cuda.current_stream()
            s = torch.cuda.Stream(device=z.device)
            s.wait_stream(cur)
            with torch.cuda.stream(s):                         # warm-up outside capture (lazy inits, allocator)
                FF_.DecoderStepFn.apply(static_z, None, None, coins, True, None, self._prec, *ps)
            cur.wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = FF_.DecoderStepFn.apply(static_z, None, None, coins, True, None, self._prec, *ps)
            self._graphs.clear()
            ent = self._graphs[key] = (g, static_z, outs)
        g, static_z, outs = ent
        static_z.copy_(z)
        g.replay()
        return outs

    def _params_free(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in FF_.FREE_PARAM_NAMES]

    # ---- ptvae.py:531-535 (+ :292-313)
    def emb_x(self, x):
        """-> (embedded [B,32,16,E], lengths [B,32]).  Shapes are the reference's; the memory behind
        them is the decoder's step-major layout ([16,32,B,E] / [32,B]) exposed through permuted views,
        so handing them back to `decoder()` costs no transpose (134 MB each way at B=512)."""
        _require_cuda(x, 'PtvaeDecoder.emb_x')
        emb, lengths = F_.EmbedFn.apply(x.long(), self.note_embedding.weight, self.note_embedding.bias, self._prec)
        # The ground-truth note summaries (packed bi-GRU over the embedded notes, ptvae.py:446-453) depend
        # on the embedding only: start them now on a sibling stream so they overlap the encoders;
        # decoder() picks the result up.
        # (not when the caller knows that no time step will be teacher-forced -- DisentangleVAE.run with tfr1 = 0, the reference's
        # schedule from its third batch on: the summaries are then dead values with zero gradient, ptvae.py:476-478)
        self._summary = None
        if self.summaries_needed:
            if SUMMARY_SLOT < 0:                             # on the caller's stream (idle while the encoders run)
                self._summary = (emb, self._summarize(emb, lengths), None)
            else:
                side = F_.Side(SUMMARY_SLOT)
                xs = side(lambda: self._summarize(emb, lengths), emb, lengths)
                self._summary = (emb, xs, side)
        return emb.permute(2, 1, 0, 3), lengths.view(32, x.size(0)).t()

    def _summarize(self, emb, len32):
        n, t, b, e = emb.shape
        F_.emb_link_arm(emb)
        return F_.BiGruFinalFn.apply(emb.view(n, t * b, e), len32, self._prec, *self.dec_notes_emb_gru.weights())

    def draw_coins(self, tfr1, tfr2):
        """Teacher-forcing decisions in the reference's draw order (ptvae.py:420,476): per time step 14
        note-level coins (tfr2), then one time-level coin (tfr1) for t < 31 -> (notes [32][14], time [31])."""
        notes, time = [], []
        for t in range(self.num_step):
            notes.append([random.random() < tfr2 for _ in range(self.max_simu_note - 2)])
            if t < self.num_step - 1:
                time.append(random.random() < tfr1)
        return notes, time

    def decoder(self, z, inference, x, lengths, teacher_forcing_ratio1, teacher_forcing_ratio2, coins=None):
        _require_cuda(z, 'PtvaeDecoder')
        B = z.size(0)
        if inference:
            assert x is None and lengths is None
            assert teacher_forcing_ratio1 == 0 and teacher_forcing_ratio2 == 0
            coins = ([[False] * (self.max_simu_note - 2)] * self.num_step, [False] * (self.num_step - 1))
            if self.use_graph and not torch.is_grad_enabled() and self.force_trace is None:
                pitch, dur, xhat, idx = self._graph_decode(z, coins)
            else:
                pitch, dur, xhat, idx = FF_.DecoderStepFn.apply(z, None, None, coins, True, self.force_trace, self._prec,
                                                                *self._params_free())
            self.last_dur_idx, self.last_xhat = idx, xhat
            return pitch.permute(2, 1, 0, 3), dur.view(15, 32, B, 5, 2).permute(2, 1, 0, 3, 4)
        if coins is None:
            coins = self.draw_coins(teacher_forcing_ratio1, teacher_forcing_ratio2)
        all_tf = all(all(r) for r in coins[0]) and all(coins[1])
        emb = x.permute(2, 1, 0, 3)                                            # [16,32,B,E]
        if not emb.is_contiguous():                                            # caller built a plain [B,32,16,E]
            E = x.size(-1)
            emb = F_.Transpose01Fn.apply(x.transpose(1, 2).reshape(B, 16 * 32, E)).view(16, 32, B, E)
        len32 = lengths.t()
        len32 = (len32 if (len32.is_contiguous() and len32.dtype == torch.int32) else len32.contiguous().int()).reshape(-1)
        cached = getattr(self, '_summary', None)
        self._summary = None
        if cached is not None and cached[0].data_ptr() == emb.data_ptr() and cached[0].shape == emb.shape:
            xs = cached[1]
            if cached[2] is not None:
                cached[2].join()
        elif any(coins[1]):
            xs = self._summarize(emb, len32)
        else:
            xs = None                                                          # no time step reads a ground-truth summary
        if not all_tf:
            none_tf = not any(any(r) for r in coins[0]) and not any(coins[1])
            if self.use_graph and none_tf and self.force_trace is None and torch.is_grad_enabled():
                # free-running training step: forward replayed from a captured hipGraph
                pitch, dur, xhat, idx = FF_.graphed_decoder_step(self._train_graphs, z, emb, xs, self._prec, *self._params_free())
            else:
                pitch, dur, xhat, idx = FF_.DecoderStepFn.apply(z, emb, xs, coins, False, self.force_trace, self._prec,
                                                                *self._params_free())
            self.last_dur_idx, self.last_xhat = idx, xhat
            return pitch.permute(2, 1, 0, 3), dur.view(15, 32, B, 5, 2).permute(2, 1, 0, 3, 4)
        pitch, dur, idx = F_.DecoderTFFn.apply(z, emb, xs, self.force_dur_idx, self._prec, *self._params())
        self.last_dur_idx = idx
        # reference shapes [B,32,15,130] / [B,32,15,5,2] as permuted views of the step-major buffers
        return pitch.permute(2, 1, 0, 3), dur.view(15, 32, B, 5, 2).permute(2, 1, 0, 3, 4)

    def forward(self, z, inference, x, lengths, teacher_forcing_ratio1, teacher_forcing_ratio2, coins=None):
        return self.decoder(z, inference, x, lengths, teacher_forcing_ratio1, teacher_forcing_ratio2, coins=coins)

    # ---- the reference's helper METHODS (ptvae.py:292-428), callable by reference-side code.  Forward-only entry points onto the
    # kernels the fused path runs (`decoder()` never calls them: it runs DecoderTFFn / DecoderStepFn); results carry no autograd graph.
    def _P(self):
        return dict(zip(FF_.FREE_PARAM_NAMES, self._params_free()))

    def _geom(self):
        return (self.num_step, self.max_simu_note, self.pitch_range, self.dur_width, self.pitch_pad)

    def get_len_index_tensor(self, ind_x):
        """ptvae.py:292-297: lengths [B, num_step] (int64) = max_simu_note - number of <pad> pitches per step"""
        _require_cuda(ind_x, 'PtvaeDecoder.get_len_index_tensor')
        x = ind_x.long().contiguous()
        B = x.size(0)
        S, N, _P, D, pad = PtvaeDecoder._geom(self)
        lengths = torch.empty(S * B, device=x.device, dtype=torch.int32)
        F_.call('ptv_grid_lengths_geom', F_.ptr(x), F_.ptr(lengths), B, S, N, D, pad, F_.stream_ptr())
        return lengths.view(S, B).t().long()

    def index_tensor_to_multihot_tensor(self, ind_x):
        """ptvae.py:299-313: piano grid [B,32,16,6] -> multi-hot [B,32,16,135] (one-hot pitch of 130 | 5 duration bits); a permuted
        view of the step-major matrix the note_embedding weight gradient multiplies.  (Any grid geometry; the reference's own
        `view(-1, 32, ...)` fixes num_step = 32.)"""
        _require_cuda(ind_x, 'PtvaeDecoder.index_tensor_to_multihot_tensor')
        x = ind_x.long().contiguous()
        B = x.size(0)
        S, N, P, D, _pad = PtvaeDecoder._geom(self)
        out = torch.empty(N, S, B, self.note_size, device=x.device, dtype=torch.float32)
        F_.call('ptv_multihot_geom', F_.ptr(x), F_.ptr(out), self.note_size, B, S, N, P, D, 0, F_.stream_ptr())
        return out.permute(2, 1, 0, 3)

    def get_sos_token(self):
        """ptvae.py:315-320: the <sos> note token [135] = onehot(128) | 2 2 2 2 2"""
        dev = self.note_embedding.weight.device
        return self.index_tensor_to_multihot_tensor(FF_._sos_grid(dev))[0, 0, 0].clone()

    def dur_ind_to_dur_token(self, inds, batch_size):
        """ptvae.py:322-326: [B] indices -> one-hot rows [B, dur_width]"""
        dev = self.note_embedding.weight.device
        inds = torch.as_tensor(inds, device=dev).long().view(batch_size, 1)
        return torch.zeros(batch_size, self.dur_width, device=dev).scatter_(1, inds, 1.0)

    def pitch_dur_ind_to_note_token(self, pitch_inds, dur_inds, batch_size):
        """ptvae.py:328-334: note_embedding(onehot(pitch) | duration bits) -> [B, note_emb_size] (ptv_note_token)"""
        dev = self.note_embedding.weight.device
        with torch.no_grad():
            return FF_.note_token_rows(self._P(), torch.as_tensor(pitch_inds, device=dev).view(batch_size),
                                       torch.as_tensor(dur_inds, device=dev).view(batch_size, self.dur_width))

    def decode_note(self, note_summary, batch_size):
        """ptvae.py:336-368: note_summary [B,1,Hn] -> est_pitch [B,130], est_durs [B,5,2] (pitch head + the 5-step duration GRU
        with argmax feedback)"""
        _require_cuda(note_summary, 'PtvaeDecoder.decode_note')
        with torch.no_grad():
            p, d, _ = FF_.decode_note_rows(self._P(), note_summary.detach().reshape(batch_size, -1).float().contiguous(), self._prec)
        return p, d

    def decode_notes(self, notes_summary, batch_size, notes, inference, teacher_forcing_ratio=0.5):
        """ptvae.py:370-428: one time step's 15 note steps.  notes_summary [B,1,Ht], notes [B,16,E] or None ->
        pitch_outs [B,15,130], dur_outs [B,15,5,2], predicted_notes [B,16,E], lengths [B].  Coins in the reference's order
        (one random.random() after each of the first 14 note steps)."""
        _require_cuda(notes_summary, 'PtvaeDecoder.decode_notes')
        if inference:
            assert teacher_forcing_ratio == 0
            assert notes is None
        coins = [random.random() < teacher_forcing_ratio for _ in range(self.max_simu_note - 2)]
        with torch.no_grad():
            return FF_.decode_notes_rows(self._P(), notes_summary.detach().reshape(batch_size, -1).float().contiguous(),
                                         None if notes is None else notes.detach(), coins, bool(inference), self._prec)

    # ---- ptvae.py:537-544
    def output_to_numpy(self, recon_pitch, recon_dur):
        est_pitch = recon_pitch.max(-1)[1].unsqueeze(-1)
        est_dur = recon_dur.max(-1)[1]
        est_x = torch.cat([est_pitch, est_dur], dim=-1).cpu().numpy()
        return est_x, recon_pitch.detach().cpu().numpy(), recon_dur.detach().cpu().numpy()

    # ---- ptvae.py:498-529
    def recon_loss(self, x, recon_pitch, recon_dur, weights=(1, 0.5), weighted_dur=False):
        out = F_.recon_loss(x.long(), recon_pitch, recon_dur, float(weights[0]), float(weights[1]), bool(weighted_dur))
        return out[0], out[1], out[2]

    # ---- ptvae.py:546-575, MIDI-free: notes come back as (pitch, start, end) tuples (velocity is the constant 100 of the
    # reference's pretty_midi.Note calls); pretty_midi is not needed
    def pr_to_notes(self, pr, bpm=80, start=0., one_hot=False):
        """pr: [32,128] duration matrix (pr_mat layout).  The reference routes through an undefined `pr_to_pr_matrix`
        (ptvae.py:547); a duration matrix is what its loop body consumes."""
        import numpy as np
        pr_matrix = np.asarray(pr)
        alpha = 0.25 * 60 / bpm
        notes = []
        for t in range(32):
            for p in range(128):
                if pr_matrix[t, p] >= 1:
                    notes.append((int(p), alpha * t + start, alpha * (t + pr_matrix[t, p]) + start))
        return notes

    def grid_to_pr_and_notes(self, grid, bpm=60., start=0.):
        import numpy as np
        grid = np.asarray(grid)
        if grid.shape[1] == self.max_simu_note:
            grid = grid[:, 1:]
        pr = np.zeros((32, 128), dtype=int)
        alpha = 0.25 * 60 / bpm
        notes = []
        for t in range(32):
            for n in range(10):                          # the reference reads at most 10 notes per step (ptvae.py:563)
                note = grid[t, n]
                if note[0] == self.pitch_eos:
                    break
                pitch = int(note[0]) + self.min_pitch
                dur = int(''.join(str(int(v)) for v in note[1:]), 2) + 1
                pr[t, pitch] = min(dur, 32 - t)
                notes.append((pitch, start + t * alpha, start + (t + dur) * alpha))
        return pr, notes


class PtvaeEncoder(nn.Module, _PrecMixin):
    """PianoTree note -> time hierarchical encoder (ptvae.py:125-215): note embedding, bi-GRU over the <= 16 notes of each
    step (packed by length), bi-GRU over the 32 step summaries, Normal(linear_mu, exp(linear_std)).  The reference's train.py:32
    constructs it (unusable in that wiring); it is the encoder of the original PianoTree-VAE pairing.
    Runs on the same kernels as the hot path: `EmbedFn` (gather embedding + lengths), `BiGruFinalFn` with per-row lengths,
    `BiGruFinalFn` over time, `EncoderHeadsFn`.  No transposes: the step-major embedding [16][32*B] feeds the note GRU, whose
    [32*B, 2H] summaries ARE the step-major [32][B] input of the time GRU."""

    def __init__(self, device, max_simu_note=16, max_pitch=127, min_pitch=0, pitch_sos=128, pitch_eos=129,
                 pitch_pad=130, dur_pad=2, dur_width=5, num_step=32, note_emb_size=128, enc_notes_hid_size=256,
                 enc_time_hid_size=512, z_size=512):
        super().__init__()
        self.max_pitch, self.min_pitch = max_pitch, min_pitch
        self.pitch_sos, self.pitch_eos, self.pitch_pad = pitch_sos, pitch_eos, pitch_pad
        self.pitch_range = max_pitch - min_pitch + 3
        self.dur_pad, self.dur_width = dur_pad, dur_width
        self.note_size = self.pitch_range + dur_width
        self.max_simu_note, self.num_step = max_simu_note, num_step
        self.device = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
        self.note_emb_size, self.z_size = note_emb_size, z_size
        self.enc_notes_hid_size, self.enc_time_hid_size = enc_notes_hid_size, enc_time_hid_size
        self.note_embedding = Linear(self.note_size, note_emb_size)
        self.enc_notes_gru = GRU(note_emb_size, enc_notes_hid_size, bidirectional=True)
        self.enc_time_gru = GRU(2 * enc_notes_hid_size, enc_time_hid_size, bidirectional=True)
        self.linear_mu = Linear(2 * enc_time_hid_size, z_size)
        self.linear_std = Linear(2 * enc_time_hid_size, z_size)

    def _check_grid(self):
        """Every geometry the reference's constructor accepts runs (train.py:32 builds max_pitch = 31: pitch_range 34); the bounds are
        those of the kernels' fixed-size staging (dur_width <= 8 index columns per note, the [P+D][E] weight in 150 KB of LDS)"""
        if self.dur_width > 8 or (self.note_size * self.note_emb_size * 4) > 150 * 1024 or self.note_emb_size > 256:
            raise NotImplementedError('PtvaeEncoder on HIP: dur_width <= 8, note_emb_size <= 256, note_size * note_emb_size <= 38400')

    # ---- ptvae.py:160-206: the reference's method surface
    def get_len_index_tensor(self, ind_x):
        return PtvaeDecoder.get_len_index_tensor(self, ind_x)

    def index_tensor_to_multihot_tensor(self, ind_x):
        return PtvaeDecoder.index_tensor_to_multihot_tensor(self, ind_x)

    def encoder(self, x, lengths):
        """ptvae.py:190-206: MULTI-HOT x [B,32,16,135] + lengths [B,32] -> (Normal, embedded [B,32,16,E]); differentiable.  forward()
        embeds the index grid by a gather instead of multiplying the multi-hot matrix; this entry point takes what the reference's
        takes (the embedding is then a [B*512,135] product)"""
        _require_cuda(x, 'PtvaeEncoder.encoder')
        self._check_grid()
        B = x.size(0)
        E = self.note_emb_size
        S, N = self.num_step, self.max_simu_note
        emb_b = F_.LinearFn.apply(x.reshape(B * S * N, self.note_size).float(), self.note_embedding.weight, self.note_embedding.bias,
                                  self._prec).view(B, S, N, E)
        emb = F_.Transpose01Fn.apply(emb_b.transpose(1, 2).reshape(B, N * S, E)).view(N, S, B, E)
        len32 = lengths.t().contiguous().int().reshape(-1)
        notes = F_.BiGruFinalFn.apply(emb.view(N, S * B, E), len32, self._prec, *self.enc_notes_gru.weights())
        h = F_.BiGruFinalFn.apply(notes.view(S, B, -1), None, self._prec, *self.enc_time_gru.weights())
        mu, sd = F_.EncoderHeadsFn.apply(h, self.linear_mu.weight, self.linear_mu.bias, self.linear_std.weight, self.linear_std.bias,
                                         self._prec)
        return HipNormal(mu, sd), emb_b

    def forward(self, x, return_iterators=False):
        _require_cuda(x, 'PtvaeEncoder')
        self._check_grid()
        B = x.size(0)
        emb, lengths = F_.EmbedFn.apply(x.long(), self.note_embedding.weight, self.note_embedding.bias, self._prec,
                                        PtvaeDecoder._geom(self))                                                      # [N,S,B,E]
        n, t, b, e = emb.shape
        notes = F_.BiGruFinalFn.apply(emb.view(n, t * b, e), lengths, self._prec, *self.enc_notes_gru.weights())      # [S*B, 2Hn]
        h = F_.BiGruFinalFn.apply(notes.view(t, b, -1), None, self._prec, *self.enc_time_gru.weights())               # [B, 2Ht]
        mu, sd = F_.EncoderHeadsFn.apply(h, self.linear_mu.weight, self.linear_mu.bias, self.linear_std.weight,
                                         self.linear_std.bias, self._prec)
        embedded_x = emb.permute(2, 1, 0, 3)                                # reference shape [B,S,N,E], step-major memory
        if return_iterators:
            return mu, sd, embedded_x
        return HipNormal(mu, sd), embedded_x, lengths.view(t, b).t()

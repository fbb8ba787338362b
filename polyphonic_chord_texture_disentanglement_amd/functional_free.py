"""Scheduled-sampling / free-running PianoTree decoder and chord decoder (the step loop).

Reference: `PtvaeDecoder.decoder/decode_notes/decode_note` (ptvae.py:336-496) with arbitrary
teacher-forcing coins, `inference=True` (model.py:124-131), and `RnnDecoder.forward` (ptvae.py:51-87).

Forward is inherently sequential -- every next token is an argmax of the previous step -- so it walks
32 x 15 x (1 + 5) cell steps on [B]-row windows of the SAME step-major buffers the teacher-forced path
uses.  Backward is not: argmax is not differentiable, so given the recorded tokens every recurrent chain is
independent across time steps and the whole BPTT runs batched over all 32*B rows exactly like the
teacher-forced path; token gradients are routed to the ground-truth embedding or to the predicted-token
buffer (whose gradient reaches `note_embedding` and, through the re-summarising bi-GRU,
`dec_notes_emb_gru`) -- never into the logits that produced the argmax.
"""
import contextlib
import os

import weakref

import torch

from . import functional as F_
from ._lib import call, lib, ptr, stream_ptr
from .functional import (DEC_PARAM_NAMES, Side, _bgrad, _bigru_backward, _empty, _eye2, _gbuf, _onehot2x5, _zeros,
                         colsum, copy2d, gemm, gru_bwd, sum_steps)

FREE_PARAM_NAMES = DEC_PARAM_NAMES + ['note_embedding.weight', 'note_embedding.bias']
EMB_GRU = ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse',
           'bias_ih_l0_reverse', 'bias_hh_l0_reverse')


_SOS = {}



_ROUTE_MASKS = {}


def _route_mask(key, dev):
    """device int32 mask of a coin pattern (the backward's gradient routing): cached per pattern -- with tfr = 0 or 1 it is one
    constant, and a host-to-device copy per step is a synchronisation point (and impossible inside a graph capture)"""
    k = (key, str(dev))
    m = _ROUTE_MASKS.get(k)
    if m is None:
        if len(_ROUTE_MASKS) > 64:
            _ROUTE_MASKS.clear()
        if key[0] == 'tok':
            t_ = torch.ones(15, 32, dtype=torch.int32)
            for t in range(32):
                for n in range(14):
                    t_[n + 1, t] = 1 if key[1][t][n] else 0
        else:
            t_ = torch.tensor([1 if c else 0 for c in key[1]] + [1], dtype=torch.int32)
        m = _ROUTE_MASKS[k] = t_.to(dev)
    return m

def _sos_grid(dev):
    """a [1,32,16,6] grid whose every row is <sos> (ptvae.py:315-320); cached so graph capture sees no H2D copy"""
    k = str(dev)
    if k not in _SOS:
        _SOS[k] = torch.tensor([128, 2, 2, 2, 2, 2], dtype=torch.long).view(1, 1, 1, 6).expand(1, 32, 16, 6).contiguous().to(dev)
    return _SOS[k]


# ---------------------------------------------------------------------------------------------
# row-partitioned persistent kernels for the step loop (csrc/freerun.hip): one launch per time step walks all 15 note steps of
# a 16-sample panel, one more re-summarises the predicted notes.  bf16 precision, init_model() geometry.
# ---------------------------------------------------------------------------------------------
FREE_PERSIST = True
# the persistent path of DecoderStepFn.forward behind ONE C call (ptv_decoder_free_fwd, csrc/composite.hip: ~170 launches -- prologue, the
# 32-step loop, the batched recompute); PTV_FREE_COMPOSITE=0: the same launches sequenced from here (bit-identical)
FREE_COMPOSITE = os.environ.get('PTV_FREE_COMPOSITE', '1') != '0'
_DFF = {}
# training forward of the step loop: the panels store only decisions, logits and fed tokens; states and gates the backward needs are
# recomputed afterwards for all 480*B rows at once by the teacher-forced kernels (same tokens, same decisions forced)
FREE_REPLAY = True
# note-loop kernel: None = by panel count (csrc/freerun.hip), True / False = force the producers-heads split / the 4-wave kernel
NOTE_LOOP_SPLIT = None
# cluster mode of the 4-wave kernel: S workgroups per 16-sample panel, each streaming 1/S of the notes-GRU gate weights (bound by ONE
# CU's L2 port), the new state all-gathered once per note step.  None = by panel count (4 up to 32 panels, 2 up to 64), 0 = off
LAST_CLUSTER_COUNTERS = None
NOTE_LOOP_CLUSTER = None if os.environ.get('PTV_NOTE_CLUSTER') is None else int(os.environ['PTV_NOTE_CLUSTER'])


def note_loop_cluster(B):
    panels = (B + 15) // 16
    ncu = _num_cu()
    # round 6: from 96 panels on the 8-wave producer / head kernel used to take over; with the head weights resident and the state exchange
    # flag-in-data, two members per panel on the 4-wave kernel win up to one member per CU (B = 2048: 34.9 vs 41.7 us per note step)
    if NOTE_LOOP_SPLIT or (NOTE_LOOP_SPLIT is None and panels >= 96 and panels * 2 > ncu):
        return 0
    if NOTE_LOOP_CLUSTER is not None:
        return NOTE_LOOP_CLUSTER if NOTE_LOOP_CLUSTER in (2, 4, 8) and panels * NOTE_LOOP_CLUSTER <= ncu else 0
    if NOTE_CLUSTER8 and panels * 8 <= ncu:
        return 8                 # round 6: eight members per panel (B <= 512 on 256 CUs): each streams an eighth of the gate weights
    # round 4: four members per panel up to ONE MEMBER PER CU (64 panels = B 1024, the per-GPU batch of BASELINE configs[4]); it was capped at
    # half the chip.  Free-running training, same box, S = 2 -> 4: B = 640 21.4k -> 23.8k samples/s, 768 23.4k -> 26.5k, 1024 27.2k -> 30.6k
    cap4 = NOTE_CLUSTER4_MAX_WGS if NOTE_CLUSTER4_MAX_WGS > 0 else ncu
    return 4 if panels * 4 <= cap4 else (2 if panels * 2 <= ncu else 0)


NOTE_CLUSTER4_MAX_WGS = 0      # 0 = the CU count
NOTE_CLUSTER8 = os.environ.get('PTV_NOTE_CLUSTER8', '1') != '0'


def cluster_bits(c):
    """the cluster size in ptv_free_note_loop's `train` word: bits 18-20 = 2 / 4, bit 22 = eight members"""
    return 0x400000 if c == 8 else (c << 18)
_NCU = []


def _num_cu():
    if not _NCU:
        _NCU.append(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    return _NCU[0]
_PACKS = F_.PackCache()
_PACK_SRC = ('dec_notes_gru.weight_hh_l0', 'dec_notes_gru.weight_ih_l0', 'pitch_out_linear.weight', 'dur_hid_linear.weight',
             'dec_dur_gru.weight_hh_l0', 'note_embedding.weight', 'dec_notes_emb_gru.weight_ih_l0', 'dec_notes_emb_gru.weight_hh_l0',
             'dec_notes_emb_gru.weight_ih_l0_reverse', 'dec_notes_emb_gru.weight_hh_l0_reverse', 'dec_time_to_notes_hid.weight',
             'dec_time_to_notes_hid.bias', 'dec_notes_gru.bias_ih_l0')


def free_persist_ok(prec, E, He, Hn, Hd, NP):
    return FREE_PERSIST and prec == 1 and F_.BF16_STORAGE and (E, He, Hn, Hd, NP) == (128, 128, 512, 64, 130)


def _pack(w2d, K=None, pairs=False):
    """fp32 [N, >=K] (any row stride) -> MFMA B-fragment-major bf16 (ptv_pack_mfma_b)"""
    N = w2d.shape[0]
    K = w2d.shape[1] if K is None else K
    out = torch.empty(lib().ptv_pack_mfma_b_size(N, K), device=w2d.device, dtype=torch.bfloat16)
    call('ptv_pack_mfma_b', ptr(w2d), w2d.stride(0), N, K, ptr(out), int(pairs), stream_ptr())
    return out


def _free_packs(P, Ht):
    """fragment-major bf16 copies of the weights the persistent step-loop kernels stream, re-packed when a parameter changed
    (in-place version counters + the fused optimiser's step count, like the bf16 weight shadows of optim.py)"""
    from .optim import _SHADOW_OF
    src = [P[n] for n in _PACK_SRC]
    ent = _SHADOW_OF.get(src[0].data_ptr())
    opt = ent[0]() if ent is not None else None
    stamp = (sum(p._version for p in src), opt.step_count if opt is not None else -1, opt._dirty if opt is not None else -1)
    hit = _PACKS.get(src, stamp)
    if hit is not None:
        return hit
    w_ih_n, w_dh, w_emb = P['dec_notes_gru.weight_ih_l0'], P['dur_hid_linear.weight'], P['note_embedding.weight']
    w_embT = torch.empty(w_emb.shape[1], w_emb.shape[0], device=w_emb.device, dtype=torch.float32)
    call('ptv_transpose01', ptr(w_embT), ptr(w_emb), w_emb.shape[0], w_emb.shape[1], 1, stream_ptr())
    pk = dict(wg_h=_pack(P['dec_notes_gru.weight_hh_l0']), wg_t=_pack(w_ih_n[:, Ht:]), wp=_pack(P['pitch_out_linear.weight']),
              wd_h=_pack(w_dh[:, :512]), wd_p=_pack(w_dh[:, 512:]), wdur=_pack(P['dec_dur_gru.weight_hh_l0']), w_embT=w_embT,
              e_ih=_pack(P['dec_notes_emb_gru.weight_ih_l0']), e_hh=_pack(P['dec_notes_emb_gru.weight_hh_l0']),
              e_ih_r=_pack(P['dec_notes_emb_gru.weight_ih_l0_reverse']), e_hh_r=_pack(P['dec_notes_emb_gru.weight_hh_l0_reverse']),
              w_cat=torch.cat([P['dec_time_to_notes_hid.weight'], w_ih_n[:, :Ht]], 0).to(torch.bfloat16).contiguous(),
              b_cat=torch.cat([P['dec_time_to_notes_hid.bias'], P['dec_notes_gru.bias_ih_l0']], 0).contiguous())
    return _PACKS.put(src, stamp, pk)


def gru_step(prec, hprev, gi, gi_ld, w_hh, b_hh, hout, *, gi2=None, gates=None, plane=0, lengths=None, t=0, gi_idx=None, hout16=None,
             hprev16=None):
    """one GRU cell on a window of rows; with a bf16 copy of the previous state (hprev16) the weight may be its bf16 shadow"""
    M, H = hout.shape
    if hprev16 is None and w_hh.dtype == torch.bfloat16:
        raise ValueError('a bf16 weight shadow needs the bf16 copy of the previous state')
    call('ptv_gru_step_fwd', prec, M, H, ptr(hprev), hprev.stride(0), ptr(hprev16), ptr(hout16), ptr(gi), gi_ld, ptr(gi2),
         gi2.stride(0) if gi2 is not None else 0, ptr(w_hh), ptr(b_hh), ptr(hout), hout.stride(0), ptr(gates), plane,
         ptr(lengths), t, ptr(gi_idx), F_._gru_flags(gates, gi, gi2, w=w_hh), stream_ptr())


def _decoder_free_fwd_composite(P, z, xs, tok0_src, tok0_lds, pk, wl, wr, io_of, ior, dims, tens, coins, w_hh_t, w_ih_t16, prec, dev, M, R):
    """DecoderStepFn.forward's persistent path through ptv_decoder_free_fwd (one C call); True when it ran"""
    import ctypes
    if 't' not in _DFF:
        from ._lib import header_enum
        _DFF['t'], _DFF['d'] = header_enum('PtvDffTensor'), header_enum('PtvDffDim')
    T_, D_ = _DFF['t'], _DFF['d']
    B, He, Ht, Hn, Hd, E = (dims[k] for k in ('B', 'He', 'Ht', 'Hn', 'Hd', 'E'))
    replay, need_resum = bool(dims['replay']), ior is not None
    w_ih_n, w_hh_n = P['dec_notes_gru.weight_ih_l0'], P['dec_notes_gru.weight_hh_l0']
    t = dict(tens)
    t.update(Z=z, XS=xs, TOK0_SRC=tok0_src, W_ZHID=P['z2dec_hid_linear.weight'], B_ZHID=P['z2dec_hid_linear.bias'],
             W_ZIN=P['z2dec_in_linear.weight'], B_ZIN=P['z2dec_in_linear.bias'], W_IH_T=P['dec_time_gru.weight_ih_l0'],
             B_IH_T=P['dec_time_gru.bias_ih_l0'], INIT_INPUT=P['dec_init_input'], B_HH_T=P['dec_time_gru.bias_hh_l0'], W_IH_T_OP=w_ih_t16,
             W_HH_T_OP=w_hh_t, W_CAT=pk['w_cat'], B_CAT=pk['b_cat'], GI=_empty(B, 3 * Ht, dev=dev), H0GC=_empty(B, 4 * Hn, dev=dev))
    if replay:
        F_.zero_skip_sync()
        pkn = F_.notes_packs(w_ih_n, w_hh_n, Ht)
        t.update(W_IH_N=w_ih_n, B_IH_N=P['dec_notes_gru.bias_ih_l0'], B_HH_N=P['dec_notes_gru.bias_hh_l0'], W_DH=P['dur_hid_linear.weight'],
                 B_DH=P['dur_hid_linear.bias'], W_HH_D=P['dec_dur_gru.weight_hh_l0'], B_HH_D=P['dec_dur_gru.bias_hh_l0'],
                 W_OUT_D=P['dur_out_linear.weight'], B_OUT_D=P['dur_out_linear.bias'], PK_NOTES_H=pkn['wg_h'], PK_NOTES_T=pkn['wg_t'],
                 GC16=_empty(R, 3 * Hn, dev=dev, dtype=torch.bfloat16), DUR_SCR=_empty(M, 10, dev=dev),
                 IDX_SCR=torch.empty(5, M, device=dev, dtype=torch.int32))
        if need_resum:
            wE = [P['dec_notes_emb_gru.' + n] for n in EMB_GRU]
            for d_ in range(2):
                w_ih_e, w_hh_e, b_ih_e, b_hh_e = wE[4 * d_: 4 * d_ + 4]
                pke = F_.notes_packs(w_ih_e, w_hh_e, 0)
                t.update({'PK_E_H%d' % d_: pke['wg_h'], 'PK_E_T%d' % d_: pke['wg_t'], 'B_HH_E%d' % d_: b_hh_e, 'B_IH_E%d' % d_: b_ih_e})
    slots = [None] * T_['PTV_DFF_COUNT']
    for k, v in t.items():
        if v is not None:
            slots[T_['PTV_DFF_' + k]] = v.data_ptr()
    dvals = [0] * D_['PTV_DFF_D_COUNT']
    for k, v in (('B', B), ('ZS', dims['Zs']), ('ZI', dims['Zi']), ('HE', He), ('HT', Ht), ('HN', Hn), ('HD', Hd), ('E', E), ('NP', dims['NP']),
                 ('LDP', dims['ldp']), ('TRAIN', dims['train']), ('REPLAY', dims['replay']), ('INFERENCE', dims['inference']),
                 ('LOOP_FLAGS', dims['loop_flags']), ('CLUSTER', dims['cluster']), ('RESUM_TRAIN', dims['resum_train']), ('TOK0_LDS', tok0_lds),
                 ('W_IH_T_BF16', int(w_ih_t16.dtype == torch.bfloat16)), ('W_HH_T_BF16', int(w_hh_t.dtype == torch.bfloat16))):
        dvals[D_['PTV_DFF_D_' + k]] = int(v)
    coin_notes, coin_time = coins
    masks = (ctypes.c_uint * 32)()
    tc = (ctypes.c_ubyte * 31)()
    if not dims['inference']:
        for ts in range(32):
            m_ = 0
            for n in range(14):
                m_ |= int(bool(coin_notes[ts][n])) << n
            masks[ts] = m_
        for ts in range(31):
            tc[ts] = int(bool(coin_time[ts]))
    cur = F_.cur_stream()
    done_ev = None
    if dims['cluster']:                                     # the cluster-mode note loops take the persistent-launch turn (functional._PersistTurn)
        prev = F_._PERSIST_LAST.get(cur.device.index)
        done_ev = torch.cuda.Event()
        if prev is not None:
            slots[T_['PTV_DFF_WAIT_EVENT']] = prev.cuda_event
        done_ev.record(cur)                                 # creates the handle; the library records it again after the last note loop
        slots[T_['PTV_DFF_RECORD_EVENT']] = done_ev.cuda_event
    F_._chain_prio()
    rc = lib().ptv_decoder_free_fwd((ctypes.c_void_p * len(slots))(*slots), F_._larr(dvals), wl, io_of(t['H0GC']), wr, ior, masks, tc, stream_ptr())
    if rc == -3:
        return False
    F_.check(rc, 'ptv_decoder_free_fwd')
    if done_ev is not None:
        F_._PERSIST_LAST[cur.device.index] = done_ev
    _DFF['calls'] = _DFF.get('calls', 0) + 1
    _DFF['keep'] = (t, masks, tc)                           # (the scratch tensors stay referenced until the next call: their launches are queued, not run)
    return True


def _decoder_free_bwd_call(items, dTOK, dTOKS, demb, dPRED, dxs, dxsp, mask_tok, mask_time, dx_pred, xhat, mh, gw, gb, B, E, He):
    """the collected stages of DecoderStepFn.backward through ptv_decoder_free_bwd (one C call); True when it ran"""
    import ctypes
    if 'bt' not in _DFF:
        from ._lib import header_enum
        _DFF['bt'], _DFF['bd'] = header_enum('PtvDfbTensor'), header_enum('PtvDfbDim')
    T_, D_ = _DFF['bt'], _DFF['bd']
    tf = [p for k, p, _ in items if k == 'tf_bwd'][0]
    rows = [p for k, p, _ in items if k == 'rows_bwd']
    slots = [None] * T_['PTV_DFB_COUNT']
    for k, v in (('DTOK', dTOK), ('DTOKS', dTOKS), ('DEMB', demb), ('DPRED', dPRED), ('DXS', dxs), ('DXSP', dxsp), ('MASK_TOK', mask_tok),
                 ('MASK_TIME', mask_time), ('DX_PRED', dx_pred), ('XHAT', xhat), ('MH', mh), ('G_W_EMB', gw), ('G_B_EMB', gb)):
        if v is not None:
            slots[T_['PTV_DFB_' + k]] = v.data_ptr()
    dvals = [0] * D_['PTV_DFB_D_COUNT']
    dvals[D_['PTV_DFB_D_B']], dvals[D_['PTV_DFB_D_E']], dvals[D_['PTV_DFB_D_HE']] = B, E, He
    if dx_pred is not None and not dx_pred.is_contiguous():
        return False
    F_._chain_prio()
    rc = lib().ptv_decoder_free_bwd(tf[0], tf[1], rows[0][0] if rows else None, rows[0][1] if rows else None,
                                    (ctypes.c_void_p * len(slots))(*slots), F_._larr(dvals), stream_ptr())
    F_.check(rc, 'ptv_decoder_free_bwd')
    _DFF['bcalls'] = _DFF.get('bcalls', 0) + 1
    return True


class DecoderStepFn(torch.autograd.Function):
    """(z, emb [16,32,B,E] or None, xs [32B,2He] or None, coins, inference, force, prec, *params)
    -> pitch [15,32,B,130], dur [15*32*B,5,2], xhat int64 [B,32,16,6] (predicted grid), dur idx"""

    @staticmethod
    def forward(ctx, z, emb, xs, coins, inference, force, prec, *params):
        P = dict(zip(FREE_PARAM_NAMES, params))
        dev = z.device
        z = z.contiguous()
        B = z.shape[0]
        R = 32 * B
        E = P['note_embedding.weight'].shape[0]
        He = P['dec_notes_emb_gru.weight_hh_l0'].shape[1]
        Ht = P['dec_time_gru.weight_hh_l0'].shape[1]
        Hn = P['dec_notes_gru.weight_hh_l0'].shape[1]
        Hd = P['dec_dur_gru.weight_hh_l0'].shape[1]
        NP = P['pitch_out_linear.weight'].shape[0]
        M = 15 * R
        train = (not inference) and any(ctx.needs_input_grad)
        coin_notes, coin_time = coins
        w_emb, b_emb = P['note_embedding.weight'], P['note_embedding.bias']
        st = stream_ptr()

        NS = _empty(33, B, Ht, dev=dev)
        w_ih_t = P['dec_time_gru.weight_ih_l0']
        Zi = P['z2dec_in_linear.weight'].shape[0]
        TOKS = _empty(33, B, 2 * He, dev=dev)
        # the persistent path as ONE library call (ptv_decoder_free_fwd): then nothing is launched from here before that call
        comp = (FREE_COMPOSITE and free_persist_ok(prec, E, He, Hn, Hd, NP) and Ht % 8 == 0
                and (not train or F_._act_dtype(prec, Hn) == torch.bfloat16) and z.dtype == torch.float32)
        if comp:
            z_in, zg = _empty(B, Zi, dev=dev), _empty(B, 3 * Ht, dev=dev)
        else:
            gemm(z, P['z2dec_hid_linear.weight'], NS[0], bias=P['z2dec_hid_linear.bias'], prec=prec)
            z_in = gemm(z, P['z2dec_in_linear.weight'], bias=P['z2dec_in_linear.bias'], prec=prec)
            zg = gemm(z_in, w_ih_t[:, 2 * He:], bias=P['dec_time_gru.bias_ih_l0'], prec=prec)
            copy2d(TOKS[0], P['dec_init_input'].view(1, -1), lds=0)
        gates_t = _empty(32, 4, B, Ht, dev=dev, dtype=F_._act_dtype(prec, Ht)) if train else None

        HN = _empty(16, R, Hn, dev=dev)
        gates_n = _empty(15, 4, R, Hn, dev=dev, dtype=F_._act_dtype(prec, Hn)) if train else None
        TOK = _empty(15, R, E, dev=dev)
        PRED = None                                                # allocated below (zero-filled only where something may stay unwritten)
        xhat = torch.full((B, 32, 16, 6), 2, device=dev, dtype=torch.long)
        xhat[:, :, :, 0] = 130
        xhat[:, :, 0, 0] = 128
        plen = torch.zeros(R, device=dev, dtype=torch.int32)
        fast = free_persist_ok(prec, E, He, Hn, Hd, NP) and (not train or F_._act_dtype(prec, Hn) == torch.bfloat16)
        PRED = _empty(16, R, E, dev=dev) if fast else _zeros(16, R, E, dev=dev)      # the persistent note loop writes every slot
        pitch = _empty(M, F_._pad8(NP), dev=dev)[:, :NP] if fast else _empty(M, NP, dev=dev)
        HD = _empty(6, M, Hd, dev=dev)
        gates_d = _empty(5, 4, M, Hd, dev=dev, dtype=F_._act_dtype(prec, Hd)) if train else None
        idx = torch.empty(5, M, device=dev, dtype=torch.int32)
        dur = _empty(M, 5, 2, dev=dev)
        dur2 = dur.view(M, 10)
        need_resum = inference or not all(coin_time)

        if inference:
            sos = _sos_grid(dev)
            sos_emb = _empty(16, 32, 1, E, dev=dev)
            call('ptv_embed_fwd', ptr(sos), ptr(w_emb), ptr(b_emb), ptr(sos_emb), None, 1, E, st)
            sos_row = sos_emb.view(-1, E)[0:1]
        w_ih_n, w_hh_n, b_hh_n = P['dec_notes_gru.weight_ih_l0'], P['dec_notes_gru.weight_hh_l0'], P['dec_notes_gru.bias_hh_l0']
        w_dh = P['dur_hid_linear.weight']
        w_ih_d, b_ih_d = P['dec_dur_gru.weight_ih_l0'], P['dec_dur_gru.bias_ih_l0']
        w_hh_d, b_hh_d = P['dec_dur_gru.weight_hh_l0'], P['dec_dur_gru.bias_hh_l0']
        tab0 = gemm(P['dur_sos_token'].view(1, -1), w_ih_d, bias=b_ih_d, prec=0)
        tab = gemm(_onehot2x5(dev), w_ih_d, bias=b_ih_d, prec=0)
        emb3 = emb.view(16, R, E) if emb is not None else None
        wE = [P['dec_notes_emb_gru.' + n] for n in EMB_GRU]
        force_pitch = force.get('pitch') if force else None          # [15, R] int32
        force_dur = force.get('dur') if force else None              # [5, 15R] int32

        HN16 = HD16 = NS16 = None
        if fast and train:                                       # bf16 state copies: MFMA operands of the batched backward
            HN16 = _empty(16, R, Hn, dev=dev, dtype=torch.bfloat16)
            HD16 = _empty(6, M, Hd, dev=dev, dtype=torch.bfloat16)
        if fast and Ht % 8 == 0:                                 # ... and of the time GRU / the per-step products in the loop
            NS16 = _empty(33, B, Ht, dev=dev, dtype=torch.bfloat16)
            if not comp:
                call('ptv_cast_bf16', ptr(NS[0]), ptr(NS16[0]), B * Ht, st)
        # per time step the loop runs four M = B products whose cost is reading the weights: their bf16 shadows halve it
        w_hh_t = F_._W(P['dec_time_gru.weight_hh_l0'], prec) if NS16 is not None else P['dec_time_gru.weight_hh_l0']
        w_ih_t16, w_tn16, w_ih_n16 = (F_._W(w_ih_t, prec), F_._W(P['dec_time_to_notes_hid.weight'], prec), F_._W(w_ih_n, prec)) if fast \
            else (w_ih_t, P['dec_time_to_notes_hid.weight'], w_ih_n)
        replay = (fast and train and FREE_REPLAY and F_.notes_persist_ok(prec, Hn, E) and Hd == 64 and F_.FUSED_DUR and NS16 is not None
                  and force_dur is None and force_pitch is None)
        XH = XG = None
        if replay and F_.DUR_RECOMPUTE and HD16 is not None:
            gates_d = None                                         # the batched recompute's duration GRU saves no gates (csrc/dur_bwd.hip rebuilds them)
        if need_resum and replay:                                # the batched recompute writes every row of every slot but slot 0
            XH = [_empty(17, R, He, dev=dev) for _ in range(2)]
            for d in range(2):
                XH[d][0].zero_()
            XG = [_empty(16, 4, R, He, dev=dev, dtype=torch.bfloat16) for _ in range(2)]
        elif need_resum:
            XH = [_zeros(17, R, He, dev=dev) for _ in range(2)]
            XG = [torch.zeros(16, 4, R, He, device=dev, dtype=F_._act_dtype(prec, He)) for _ in range(2)] if train else [None, None]
        cluster, xch, xcnt = 0, None, None
        if fast:
            cluster = note_loop_cluster(B)
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing and torch.is_grad_enabled():
                cluster = 0          # a captured training forward replays next to other streams' persistent launches, unordered with them
            if cluster:
                global LAST_CLUSTER_COUNTERS
                xch = torch.zeros((B + 15) // 16 * 2 * 16 * Hn * 2, device=dev, dtype=torch.bfloat16)   # 8-byte words {2 units, step tag}: tag 0 = nothing sent yet
                xcnt = torch.zeros((B + 15) // 16 + 1, device=dev, dtype=torch.int32)    # arrival counters of t = 0..31 + error word
                LAST_CLUSTER_COUNTERS = xcnt             # (tests: every panel counts 32 * 15 * S arrivals, the error word stays 0)
                F_._CLUSTER_SYNC.append(xcnt)
                if len(F_._CLUSTER_SYNC) > 64:
                    del F_._CLUSTER_SYNC[:32]
            pk = _free_packs(P, Ht)
            wl = F_._parr([pk['wg_h'], pk['wg_t'], pk['wp'], pk['wd_h'], pk['wd_p'], pk['wdur'], b_hh_n, P['pitch_out_linear.bias'],
                           P['dur_hid_linear.bias'], b_hh_d, tab0, tab, P['dur_out_linear.weight'], P['dur_out_linear.bias'],
                           pk['w_embT'], b_emb])
            wr = F_._parr([pk['e_ih'], pk['e_hh'], pk['e_ih_r'], pk['e_hh_r'], wE[2], wE[3], wE[6], wE[7]])
        XH16 = [None, None]
        done = False
        if comp:
            assert fast and NS16 is not None
            if replay and need_resum:
                XH16 = [_empty(17, R, He, dev=dev, dtype=torch.bfloat16) for _ in range(2)]
            done = _decoder_free_fwd_composite(
                P, z, xs, sos_row if inference else emb3[0], 0 if inference else E, pk, wl, wr, io_of=lambda h0gc: F_._parr(
                    [None, emb3, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen, force_pitch, force_dur, HN16, HD16, None, h0gc,
                     xch, xcnt]),
                ior=F_._parr([PRED, plen, XH[0], XH[1], XG[0], XG[1], None]) if need_resum else None,
                dims=dict(B=B, Zs=z.shape[1], Zi=Zi, He=He, Ht=Ht, Hn=Hn, Hd=Hd, E=E, NP=NP, ldp=pitch.stride(0), train=int(train), replay=int(bool(replay)),
                          inference=int(bool(inference)), cluster=int(cluster and not capturing),
                          loop_flags=(2 if replay else int(train)) | (0 if NOTE_LOOP_SPLIT is None else (0x20000 if NOTE_LOOP_SPLIT else 0x10000))
                          | cluster_bits(cluster), resum_train=int(train and not replay)),
                tens=dict(NS=NS, NS16=NS16, Z_IN=z_in, ZG=zg, TOKS=TOKS, GATES_T=gates_t, TOK=TOK, PRED=PRED, PITCH=pitch, HN=HN, HN16=HN16,
                          GATES_N=gates_n, HD=HD, HD16=HD16, GATES_D=gates_d, IDX=idx, PLEN=plen, XH0=XH[0] if XH else None,
                          XH1=XH[1] if XH else None, XH16_0=XH16[0], XH16_1=XH16[1], XG0=XG[0] if XG else None, XG1=XG[1] if XG else None,
                          TAB0=tab0, TAB=tab),
                coins=(coin_notes, coin_time), w_hh_t=w_hh_t, w_ih_t16=w_ih_t16, prec=prec, dev=dev, M=M, R=R)
            assert done, 'ptv_decoder_free_fwd declined a configuration free_persist_ok() accepted'
        # the first note token of every time step is the <sos> embedding (ptvae.py:388-392): one copy for all 32 steps
        if not done:
            if inference:
                copy2d(TOK[0], sos_row, lds=0)
            else:
                copy2d(TOK[0], emb3[0])
            copy2d(PRED[0], TOK[0])
        for t in range(0 if not done else 32, 32):
            rows = slice(t * B, (t + 1) * B)
            gi = gemm(TOKS[t], w_ih_t16[:, :2 * He], prec=prec)
            gru_step(prec, NS[t], gi, 3 * Ht, w_hh_t, P['dec_time_gru.bias_hh_l0'], NS[t + 1], gi2=zg,
                     gates=gates_t[t] if train else None, plane=B * Ht, hout16=NS16[t + 1] if NS16 is not None else None,
                     hprev16=NS16[t] if NS16 is not None else None)
            ns = NS16[t + 1] if NS16 is not None else NS[t + 1]
            if fast:
                # initial notes-GRU state and the hoisted input part in ONE product against the stacked weights ([512 | 1536] rows):
                # 256 blocks, no split-K, no zero fill
                H0GC = gemm(ns, pk['w_cat'], bias=pk['b_cat'], prec=prec)
                GCt = None
            else:
                gemm(ns, w_tn16, HN[0][rows], bias=P['dec_time_to_notes_hid.bias'], prec=prec)
                GCt = gemm(ns, w_ih_n16[:, :Ht], bias=P['dec_notes_gru.bias_ih_l0'], prec=prec)
            if fast:
                # all 15 note steps of this time step in ONE launch (csrc/freerun.hip); then the next time-step token
                mask = 0
                if not inference:
                    for n in range(14):
                        mask |= int(bool(coin_notes[t][n])) << n
                io = F_._parr([GCt, emb3, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen, force_pitch, force_dur, HN16, HD16,
                               None, H0GC, xch, xcnt])
                # (cluster mode: the members of a panel spin on each other -- like every persistent launch it takes its turn, so that
                # it is never half-resident next to another spinning grid, e.g. the chord decoder's on its sibling stream)
                with (F_._PersistTurn() if cluster and not capturing else contextlib.nullcontext()):
                    call('ptv_free_note_loop', wl, io, pitch.stride(0), B, t, mask,
                         (2 if replay else int(train)) | (0 if NOTE_LOOP_SPLIT is None else (0x20000 if NOTE_LOOP_SPLIT else 0x10000))
                         | cluster_bits(cluster), st)
                if t == 31:
                    break
                if (not inference) and coin_time[t]:
                    copy2d(TOKS[t + 1], xs[rows])
                else:
                    io = F_._parr([PRED, plen, XH[0], XH[1], XG[0], XG[1], TOKS[t + 1]])
                    call('ptv_free_resummarize', wr, io, B, t, int(train and not replay), st)
                continue
            for n in range(15):
                gi_tok = gemm(TOK[n][rows], w_ih_n[:, Ht:], prec=prec)
                gru_step(prec, HN[n][rows], gi_tok, 3 * Hn, w_hh_n, b_hh_n, HN[n + 1][rows], gi2=GCt,
                         gates=gates_n[n][:, rows] if train else None, plane=R * Hn)
                h = HN[n + 1][rows]
                pr = slice(n * R + t * B, n * R + (t + 1) * B)
                gemm(h, P['pitch_out_linear.weight'], pitch[pr], bias=P['pitch_out_linear.bias'], prec=prec)
                gemm(h, w_dh[:, :Hn], HD[0][pr], bias=P['dur_hid_linear.bias'], prec=prec)
                gemm(pitch[pr], w_dh[:, Hn:], HD[0][pr], acc=True, prec=prec)
                if prec == 1 and Hd == 64 and F_.FUSED_DUR:
                    call('ptv_dur_gru_fwd', Hd, B, ptr(HD[0][pr]), Hd, ptr(w_hh_d), ptr(b_hh_d), ptr(tab0), ptr(tab),
                         ptr(P['dur_out_linear.weight']), ptr(P['dur_out_linear.bias']), ptr(HD[1][pr]), M * Hd, None,
                         ptr(gates_d[0][0][pr]) if train else None, M * Hd, 4 * M * Hd, F_._bf(gates_d), ptr(dur2[pr]), 10,
                         ptr(idx[0][pr]), M, ptr(force_dur[0][pr]) if force_dur is not None else None, M, st)
                else:
                    for d in range(5):
                        g_, g_ld, g_idx = (tab0, 0, None) if d == 0 else (tab, 3 * Hd, idx[d - 1][pr])
                        gru_step(prec, HD[d][pr], g_, g_ld, w_hh_d, b_hh_d, HD[d + 1][pr],
                                 gates=gates_d[d][:, pr] if train else None, plane=M * Hd, gi_idx=g_idx)
                        call('ptv_dur_out_token', ptr(HD[d + 1][pr]), Hd, ptr(P['dur_out_linear.weight']),
                             ptr(P['dur_out_linear.bias']), ptr(dur2[pr][:, 2 * d:]), 10, ptr(idx[d][pr]),
                             ptr(force_dur[d][pr]) if force_dur is not None else None, B, st)
                call('ptv_note_token', ptr(pitch[pr]), NP, ptr(idx[0][pr]), M, ptr(w_emb), ptr(b_emb), E,
                     ptr(PRED[n + 1][rows]), E, ptr(xhat[0, t, n + 1]), 32 * 16 * 6, ptr(plen[rows]), n + 1, int(n == 14),
                     ptr(force_pitch[n][rows]) if force_pitch is not None else None, B, st)
                if n < 14:
                    use_gt = (not inference) and coin_notes[t][n]
                    copy2d(TOK[n + 1][rows], emb3[n + 1][rows] if use_gt else PRED[n + 1][rows])
            if t == 31:
                break
            if (not inference) and coin_time[t]:
                copy2d(TOKS[t + 1], xs[rows])
            else:
                # token = final states of dec_notes_emb_gru over the predicted notes (ptvae.py:480-486)
                PT = _empty(16, B * E, dev=dev)
                copy2d(PT, PRED.view(16, R * E)[:, t * B * E:(t + 1) * B * E])
                for d in range(2):
                    gi_e = gemm(PT.view(16 * B, E), wE[4 * d], bias=wE[4 * d + 2], prec=prec).view(16, B, 3 * He)
                    for s_ in range(16):
                        tt = 15 - s_ if d else s_
                        gru_step(prec, XH[d][s_][rows], gi_e[tt], 3 * He, wE[4 * d + 1], wE[4 * d + 3], XH[d][s_ + 1][rows],
                                 gates=XG[d][s_][:, rows] if train else None, plane=R * He, lengths=plen[rows], t=tt)
                    copy2d(TOKS[t + 1][:, d * He:(d + 1) * He], XH[d][16][rows])

        if replay and not done:
            F_.zero_skip_sync()
            # ---- recompute what the backward reads, batched over all rows (the step loop above stored decisions and tokens only)
            GC16 = gemm(NS16[1:].view(R, Ht), w_ih_n[:, :Ht], bias=P['dec_notes_gru.bias_ih_l0'], prec=prec, out_dtype=torch.bfloat16,
                        out_blocked=16)      # (column-blocked by 16: the layout the row kernel reads)
            pkn = F_.notes_packs(w_ih_n, w_hh_n, Ht)
            call('ptv_notes_gru_persist_fwd', ptr(pkn['wg_h']), ptr(pkn['wg_t']), ptr(b_hh_n), ptr(GC16), ptr(TOK), ptr(HN), ptr(HN16),
                 ptr(gates_n), R, 15, st)
            NSUM_op = HN16[1:].view(M, Hn)
            gemm(NSUM_op, w_dh[:, :Hn], HD[0], bias=P['dur_hid_linear.bias'], prec=prec)
            gemm(pitch, w_dh[:, Hn:], HD[0], acc=True, prec=prec)
            dur_scr, idx_scr = _empty(M, 10, dev=dev), torch.empty(5, M, device=dev, dtype=torch.int32)
            call('ptv_dur_gru_fwd', Hd, M, ptr(HD[0]), Hd, ptr(w_hh_d), ptr(b_hh_d), ptr(tab0), ptr(tab), ptr(P['dur_out_linear.weight']),
                 ptr(P['dur_out_linear.bias']), None, M * Hd, ptr(HD16[1]), ptr(gates_d), M * Hd, 4 * M * Hd, 1, ptr(dur_scr), 10,
                 ptr(idx_scr), M, ptr(idx), M, st)
            call('ptv_cast_bf16', ptr(HD[0]), ptr(HD16[0]), M * Hd, st)
            if need_resum:
                for d in range(2):
                    w_ih_e, w_hh_e, b_ih_e, b_hh_e = wE[4 * d: 4 * d + 4]
                    pke = F_.notes_packs(w_ih_e, w_hh_e, 0)
                    XH16[d] = _empty(17, R, He, dev=dev, dtype=torch.bfloat16)
                    call('ptv_row_gru_persist_fwd', He, ptr(pke['wg_h']), ptr(pke['wg_t']), ptr(b_hh_e), ptr(b_ih_e), None, ptr(PRED), R * E,
                         ptr(plen), ptr(XH[d]), ptr(XH16[d]), ptr(XG[d]), None, 0, R, 16, d, st)
        if train:
            ctx.save_for_backward(z, emb, *params)
            ctx.st = dict(B=B, R=R, E=E, He=He, Ht=Ht, Hn=Hn, Hd=Hd, NP=NP, prec=prec, NS=NS, z_in=z_in, TOKS=TOKS,
                          gates_t=gates_t, HN=HN, gates_n=gates_n, gates_n_rowk=bool(replay), pitch=pitch, HD=HD, gates_d=gates_d, idx=idx, TOK=TOK,
                          PRED=PRED, xhat=xhat, XH=XH, XG=XG, XH16=XH16, plen=plen, skipped=F_.ZERO_SKIP, coins=coins, has_xs=xs is not None,
                          NS16=NS16, HN16=HN16, HD16=HD16, dur16_only=HD16 is not None, dur_tabs=(tab0, tab))
        ctx.mark_non_differentiable(xhat, idx)
        return pitch.view(15, 32, B, NP), dur, xhat, idx

    @staticmethod
    def backward(ctx, dpitch, ddur, _dx, _di):
        z, emb, *params = ctx.saved_tensors
        P = dict(zip(FREE_PARAM_NAMES, params))
        st = ctx.st
        if not getattr(ctx, 'keep_state', False):        # a captured forward (GraphedDecoderStepFn) reuses its buffers
            ctx.st = None
        B, R, E, He, Ht, Hn, Hd, NP, prec = (st[k] for k in ('B', 'R', 'E', 'He', 'Ht', 'Hn', 'Hd', 'NP', 'prec'))
        dev = z.device
        M = 15 * R
        PRED = st['PRED']
        coin_notes, coin_time = st['coins']
        sp = stream_ptr()
        # ---- the node behind ONE C entry point (ptv_decoder_free_bwd): its stages are COLLECTED (functional._DEFER) instead of launched -- the
        # decoder's composite backward, the two routings, the re-summarisation bi-GRU's composite backward, the note_embedding gradients --
        # and go out as one call; a stage that cannot go behind the C ABI flushes what was collected and the rest launches at once (same
        # order, same bits either way)
        collect = (FREE_COMPOSITE and prec == 1 and F_.DEC_BWD_COMPOSITE and F_.BIGRU_BWD_COMPOSITE and F_.WGRAD_FUSE_BIAS
                   and not torch.cuda.is_current_stream_capturing() and F_._DEFER is None)
        if collect:
            F_._DEFER = []
        try:
            # ---- duration GRU, heads, notes GRU, time GRU: the batched BPTT of the teacher-forced path on the recorded fed tokens
            dz, dTOK, dTOKS, G0, side = F_.decoder_bwd_core(P, st, z, st['TOK'].view(M, E), dpitch, ddur)
            G = {n: None for n in FREE_PARAM_NAMES}
            G.update(G0)

            # ---- route token gradients: ground-truth embedding (coin true / slot 0) vs predicted tokens
            demb = _zeros(16, R, E, dev=dev)
            dPRED = _zeros(16, R, E, dev=dev)
            mask_tok = _route_mask(('tok', tuple(tuple(bool(v) for v in row) for row in coin_notes)), dev)
            F_._defer_or_run('route', None, lambda: call('ptv_route_slices', ptr(dTOK), ptr(demb), ptr(dPRED), ptr(mask_tok), B * E, 15 * 32, 0, sp))

            # ---- time tokens: ground-truth summaries (coin true) vs re-summarised predictions
            dxs = _zeros(32, B, 2 * He, dev=dev)
            dxsp = _zeros(32, B, 2 * He, dev=dev)
            mask_time = _route_mask(('time', tuple(bool(v) for v in coin_time)), dev)
            F_._defer_or_run('route', None, lambda: call('ptv_route_slices', ptr(dTOKS[1:]), ptr(dxs), ptr(dxsp), ptr(mask_time), B * 2 * He, 32, 0, sp))
            dx_pred = None
            if st['XH'] is not None:
                wE = [P['dec_notes_emb_gru.' + n] for n in EMB_GRU]
                saved = [(st['XH'][d_], st['XG'][d_], st['XH16'][d_]) + ((st['plen'] if st['skipped'] else None,) if st['XH16'][d_] is not None else ()) for d_ in range(2)]
                ge, dx_pred = _bigru_backward(prec, PRED, wE, saved, dxsp.view(R, 2 * He), True)
                for n, gg in zip(EMB_GRU, ge):
                    G['dec_notes_emb_gru.' + n] = gg
                F_._defer_or_run('copy', None, lambda: copy2d(dPRED.view(16 * R, E), dx_pred.view(16 * R, E), acc=True))

            # ---- predicted tokens -> note_embedding (slot 0 is the ground-truth <sos> embedding); weight and bias gradient in ONE pass
            # over the gradient matrix (ptv_wgrad's colsum_a)
            mh = _empty(B * 512, 136, dev=dev)
            for n in ('note_embedding.weight', 'note_embedding.bias'):
                if G[n] is None:
                    G[n] = _gbuf(P[n])

            def tail():
                copy2d(demb[0], dPRED[0], acc=True)
                dPRED[0].zero_()
                call('ptv_multihot', ptr(st['xhat']), ptr(mh), 136, B, sp)
                F_.wgrad_bias(dPRED.view(16 * R, E), mh[:, :135], G['note_embedding.weight'], G['note_embedding.bias'], prec)
            F_._defer_or_run('tail', None, tail)
            items = F_._DEFER
            if items is None and collect:
                _DFF['bflush'] = 'a stage flushed'
            if items is not None:
                F_._DEFER = None
                kinds = [k for k, _, _ in items]
                want = ['tf_bwd', 'route', 'route'] + (['rows_bwd', 'copy'] if dx_pred is not None else []) + ['tail']
                if kinds == want and _decoder_free_bwd_call(items, dTOK, dTOKS, demb, dPRED, dxs, dxsp, mask_tok, mask_time, dx_pred, st['xhat'], mh,
                                                            G['note_embedding.weight'], G['note_embedding.bias'], B, E, He):
                    pass
                else:                                       # (not the whole pattern: the collected stages run as they are, in order)
                    _DFF['bflush'] = kinds
                    F_._DEFER = items
                    F_._defer_flush()
        finally:
            F_._DEFER = None

        side.join()
        for s2 in getattr(side, 'extra', []):
            s2.join()
        grads = tuple(G[n] for n in FREE_PARAM_NAMES)
        return (dz, demb.view(16, 32, B, E), dxs.view(R, 2 * He) if st['has_xs'] else None, None, None, None, None) + grads


# =============================================================================================
# RnnDecoder (chord decoder) with arbitrary coins: ptvae.py:51-87
# =============================================================================================
class _CapturedCtx:
    """stands in for the autograd ctx of DecoderStepFn while its forward is captured into a hipGraph"""

    def __init__(self, n_inputs):
        self.needs_input_grad = (True,) * n_inputs
        self.saved_tensors = ()
        self.st = None
        self.keep_state = True

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *a):
        pass


class _GraphEntry:
    """one captured free-running training forward + the buffers its backward reads.  `pending` is a weak reference to the
    autograd ctx of the replay whose backward has not run yet: a second replay before that would overwrite NS / HN / gates /
    PRED under it (two loss() calls and one backward, gradient accumulation over micro-batches, ...)"""

    def __init__(self, g, sz, se, cap, outs):
        self.g, self.sz, self.se, self.cap, self.outs = g, sz, se, cap, outs
        self.pending = None

    def busy(self):
        c = self.pending() if self.pending is not None else None
        return c is not None and not getattr(c, 'done', True)


GRAPH_CACHE_SLOTS = 3          # e.g. train and validation batch sizes alternating: no re-capture (~9,000 launches + warm-up) per switch


def _graph_key(z, emb, xs, prec, params):
    return (z.shape[0], prec, z.device.index, tuple(p.data_ptr() for p in params), tuple(emb.shape), xs is not None,
            F_.FUSED_DUR, F_.BF16_STORAGE, F_.NOTES_PERSIST, F_.PERSIST, FREE_PERSIST)


def graphed_decoder_step(cache, z, emb, xs, prec, *params):
    """GraphedDecoderStepFn when its captured buffers are free, the eager DecoderStepFn (same kernels, own buffers) while an
    earlier replay of the same graph still waits for its backward"""
    ent = cache.get(_graph_key(z, emb, xs, prec, params))
    if ent is not None and ent.busy():
        coins = ([[False] * 14] * 32, [False] * 31)
        return DecoderStepFn.apply(z, emb, xs, coins, False, None, prec, *params)
    return GraphedDecoderStepFn.apply(cache, z, emb, xs, prec, *params)


class GraphedDecoderStepFn(torch.autograd.Function):
    """Free-running TRAINING forward (every teacher-forcing coin false: what the reference's schedules reach after two
    steps) replayed from a captured hipGraph.  The launch order and arguments of the step loop depend only on (B, precision,
    parameter storage, the module-level kernel toggles): capture DecoderStepFn.forward once -- every buffer the backward needs
    is the graph's static storage -- then one graph launch per step; the backward is DecoderStepFn.backward on those buffers.
    The returned tensors are copies (they do not change under the caller at the next replay).  `cache` is a dict owned by the
    decoder module, a small LRU.  Call through graphed_decoder_step(), which falls back to the eager path while a replay's
    backward is pending."""

    @staticmethod
    def forward(ctx, cache, z, emb, xs, prec, *params):
        key = _graph_key(z, emb, xs, prec, params)
        ent = cache.get(key)
        coins = ([[False] * 14] * 32, [False] * 31)
        if ent is not None and ent.busy():
            raise RuntimeError('GraphedDecoderStepFn: the previous replay of this graph has not been back-propagated yet; '
                               'call graphed_decoder_step(), which runs such forwards eagerly')
        if ent is None:
            sz, se = z.detach().clone().contiguous(), emb.detach().clone()
            sx = xs.detach().clone() if xs is not None else None
            cur = torch.cuda.current_stream()
            s = torch.cuda.Stream(device=z.device)
            s.wait_stream(cur)
            with torch.cuda.stream(s):                       # warm-up outside capture (lazy inits, allocator)
                DecoderStepFn.forward(_CapturedCtx(7 + len(params)), sz, se, sx, coins, False, None, prec, *params)
            cur.wait_stream(s)
            cap = _CapturedCtx(7 + len(params))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = DecoderStepFn.forward(cap, sz, se, sx, coins, False, None, prec, *params)
            while len(cache) >= GRAPH_CACHE_SLOTS:           # least recently used first (dicts keep insertion order)
                victim = next((k for k, e in cache.items() if not e.busy()), None)
                if victim is None:
                    break
                del cache[victim]
            ent = cache[key] = _GraphEntry(g, sz, se, cap, outs)
        else:
            cache[key] = cache.pop(key)                      # most recently used last
        ent.sz.copy_(z)
        ent.se[0].copy_(emb[0])                               # only the <sos> slot of the ground truth is read when no coin is true
        ent.g.replay()
        ctx.cap = ent.cap
        ctx.done = not any(ctx.needs_input_grad)
        ent.pending = weakref.ref(ctx)
        outs = tuple(o.clone() for o in ent.outs)
        ctx.mark_non_differentiable(outs[2], outs[3])
        return outs

    @staticmethod
    def backward(ctx, dpitch, ddur, _dx, _di):
        r = DecoderStepFn.backward(ctx.cap, dpitch, ddur, None, None)
        ctx.done = True
        # DecoderStepFn inputs: (z, emb, xs, coins, inference, force, prec, *params) -> ours: (cache, z, emb, xs, prec, *params)
        return (None, r[0], r[1], r[2], None) + tuple(r[7:])


# ---------------------------------------------------------------------------------------------
# The reference's note-level METHODS (ptvae.py:315-428) as callable entry points: forward-only helpers for reference-side code that
# calls `decoder.decode_note(...)` / `decode_notes(...)` / the token builders directly (training goes through DecoderTFFn /
# DecoderStepFn, which run the same kernels fused).  Rows = whatever batch the caller hands over.
# ---------------------------------------------------------------------------------------------
def note_token_rows(P, pitch_inds, dur_inds):
    """pitch_dur_ind_to_note_token (ptvae.py:328-334): note_embedding(onehot(pitch) | 5 duration bits) -> [B, E]"""
    dev = pitch_inds.device
    B = pitch_inds.shape[0]
    w_emb, b_emb = P['note_embedding.weight'], P['note_embedding.bias']
    E = w_emb.shape[0]
    NP = w_emb.shape[1] - 5
    dummy = _zeros(B, NP, dev=dev)
    idx = dur_inds.reshape(B, 5).t().contiguous().int()
    pred = _empty(B, E, dev=dev)
    xhat = torch.empty(B, 6, device=dev, dtype=torch.long)
    plen = torch.zeros(B, device=dev, dtype=torch.int32)
    call('ptv_note_token', ptr(dummy), NP, ptr(idx), B, ptr(w_emb), ptr(b_emb), E, ptr(pred), E, ptr(xhat), 6, ptr(plen), 1, 0,
         ptr(pitch_inds.int().contiguous()), B, stream_ptr())
    return pred


def decode_note_rows(P, h, prec, force_dur=None):
    """decode_note (ptvae.py:336-368) for B rows: h [B, Hn] -> est_pitch [B, NP], est_durs [B, 5, 2], duration argmaxes [5, B]"""
    dev = h.device
    B, Hn = h.shape
    Hd = P['dec_dur_gru.weight_hh_l0'].shape[1]
    w_dh = P['dur_hid_linear.weight']
    w_ih_d, b_ih_d = P['dec_dur_gru.weight_ih_l0'], P['dec_dur_gru.bias_ih_l0']
    w_hh_d, b_hh_d = P['dec_dur_gru.weight_hh_l0'], P['dec_dur_gru.bias_hh_l0']
    pitch = gemm(h, P['pitch_out_linear.weight'], bias=P['pitch_out_linear.bias'], prec=prec)
    HD = _empty(6, B, Hd, dev=dev)
    gemm(h, w_dh[:, :Hn], HD[0], bias=P['dur_hid_linear.bias'], prec=prec)
    gemm(pitch, w_dh[:, Hn:], HD[0], acc=True, prec=prec)
    tab0 = gemm(P['dur_sos_token'].view(1, -1), w_ih_d, bias=b_ih_d, prec=0)
    tab = gemm(_onehot2x5(dev), w_ih_d, bias=b_ih_d, prec=0)
    idx = torch.empty(5, B, device=dev, dtype=torch.int32)
    dur = _empty(B, 5, 2, dev=dev)
    dur2 = dur.view(B, 10)
    for d in range(5):
        g_, g_ld, g_idx = (tab0, 0, None) if d == 0 else (tab, 3 * Hd, idx[d - 1])
        gru_step(prec, HD[d], g_, g_ld, w_hh_d, b_hh_d, HD[d + 1], plane=B * Hd, gi_idx=g_idx)
        call('ptv_dur_out_token', ptr(HD[d + 1]), Hd, ptr(P['dur_out_linear.weight']), ptr(P['dur_out_linear.bias']),
             ptr(dur2[:, 2 * d:]), 10, ptr(idx[d]), ptr(force_dur[d]) if force_dur is not None else None, B, stream_ptr())
    return pitch, dur, idx


def decode_notes_rows(P, ns, notes, coins, inference, prec):
    """decode_notes (ptvae.py:370-428) for B rows: ns [B, Ht] time-level summary, notes [B, 16, E] ground-truth embedded notes (or
    None when `inference`), coins [14] teacher-forcing decisions -> pitch_outs [B,15,NP], dur_outs [B,15,5,2],
    predicted_notes [B,16,E], lengths [B] float"""
    dev = ns.device
    B, Ht = ns.shape
    E = P['note_embedding.weight'].shape[0]
    Hn = P['dec_notes_gru.weight_hh_l0'].shape[1]
    NP = P['pitch_out_linear.weight'].shape[0]
    w_emb, b_emb = P['note_embedding.weight'], P['note_embedding.bias']
    w_ih_n, w_hh_n, b_hh_n = P['dec_notes_gru.weight_ih_l0'], P['dec_notes_gru.weight_hh_l0'], P['dec_notes_gru.bias_hh_l0']
    st = stream_ptr()
    HN = _empty(16, B, Hn, dev=dev)
    gemm(ns, P['dec_time_to_notes_hid.weight'], HN[0], bias=P['dec_time_to_notes_hid.bias'], prec=prec)
    GC = gemm(ns, w_ih_n[:, :Ht], bias=P['dec_notes_gru.bias_ih_l0'], prec=prec)
    TOK = _empty(15, B, E, dev=dev)
    PRED = _zeros(16, B, E, dev=dev)
    if inference:
        sos = _sos_grid(dev)
        sos_emb = _empty(16, 32, 1, E, dev=dev)
        call('ptv_embed_fwd', ptr(sos), ptr(w_emb), ptr(b_emb), ptr(sos_emb), None, 1, E, st)
        copy2d(TOK[0], sos_emb.view(-1, E)[0:1], lds=0)
    else:
        nt = notes.float().transpose(0, 1).contiguous()                  # [16, B, E]
        copy2d(TOK[0], nt[0])
    copy2d(PRED[0], TOK[0])
    pitch = _empty(15, B, NP, dev=dev)
    dur = _empty(15, B, 5, 2, dev=dev)
    xhat = torch.full((B, 16, 6), 2, device=dev, dtype=torch.long)
    plen = torch.zeros(B, device=dev, dtype=torch.int32)
    for n in range(15):
        gi_tok = gemm(TOK[n], w_ih_n[:, Ht:], prec=prec)
        gru_step(prec, HN[n], gi_tok, 3 * Hn, w_hh_n, b_hh_n, HN[n + 1], gi2=GC, plane=B * Hn)
        p_, d_, idx = decode_note_rows(P, HN[n + 1], prec)
        copy2d(pitch[n], p_)
        copy2d(dur[n].view(B, 10), d_.view(B, 10))
        call('ptv_note_token', ptr(p_), NP, ptr(idx), B, ptr(w_emb), ptr(b_emb), E, ptr(PRED[n + 1]), E, ptr(xhat[0, n + 1]), 16 * 6,
             ptr(plen), n + 1, int(n == 14), None, B, st)
        if n < 14:
            use_gt = (not inference) and coins[n]
            copy2d(TOK[n + 1], nt[n + 1] if use_gt else PRED[n + 1])
    return (pitch.transpose(0, 1), dur.transpose(0, 1), PRED.transpose(0, 1), plen.float())


class ChordDecoderStepFn(torch.autograd.Function):
    """(z_chd, c_sm [8,B,36] or None, coins [8] bools, force, prec, *params) -> root [8,B,12], chroma [8,B,24], bass [8,B,12]
    force (tests: replay mode): {'root' [8,B,12], 'chroma' [8,B,24], 'bass' [8,B,12]} logits of a recorded run whose
    argmax decisions build the fed-back tokens instead of this run's own (a near-tie flipped by rounding changes the trajectory)"""

    @staticmethod
    def forward(ctx, z, c_sm, coins, force, prec, *params):
        P = dict(zip(F_.CHD_PARAM_NAMES, params))
        dev = z.device
        z = z.contiguous()
        B = z.shape[0]
        T = len(coins)
        H = P['gru.weight_hh_l0'].shape[1]
        I = P['init_input'].shape[0]
        hall = _empty(T + 1, B, H, dev=dev)
        gemm(z, P['z2dec_hid.weight'], hall[0], bias=P['z2dec_hid.bias'], prec=prec)
        z_in = gemm(z, P['z2dec_in.weight'], bias=P['z2dec_in.bias'], prec=prec)
        w_ih = P['gru.weight_ih_l0']
        zg = gemm(z_in, w_ih[:, I:], bias=P['gru.bias_ih_l0'], prec=prec)
        toks = _empty(T, B, I, dev=dev)
        copy2d(toks[0], P['init_input'].view(1, -1), lds=0)
        gates = _empty(T, 4, B, H, dev=dev, dtype=F_._act_dtype(prec, H))
        root, chroma, bass = _empty(T, B, 12, dev=dev), _empty(T, B, 24, dev=dev), _empty(T, B, 12, dev=dev)
        masks = torch.zeros(2, device=dev, dtype=torch.int32)
        for t in range(T):
            gi = gemm(toks[t], w_ih[:, :I], prec=prec)
            gru_step(prec, hall[t], gi, 3 * H, P['gru.weight_hh_l0'], P['gru.bias_hh_l0'], hall[t + 1], gi2=zg, gates=gates[t],
                     plane=B * H)
            gemm(hall[t + 1], P['root_out.weight'], root[t], bias=P['root_out.bias'], prec=prec)
            gemm(hall[t + 1], P['chroma_out.weight'], chroma[t], bias=P['chroma_out.bias'], prec=prec)
            gemm(hall[t + 1], P['bass_out.weight'], bass[t], bias=P['bass_out.bias'], prec=prec)
            if t + 1 < T:
                if coins[t] and c_sm is not None:
                    copy2d(toks[t + 1], c_sm[t])
                else:
                    src = (force['root'], force['chroma'], force['bass']) if force else (root, chroma, bass)
                    call('ptv_chord_token', ptr(src[0][t]), ptr(src[1][t]), ptr(src[2][t]), ptr(masks), ptr(toks[t + 1]), B,
                         stream_ptr())
        ctx.save_for_backward(z, *params)
        ctx.st = dict(hall=hall, gates=gates, toks=toks, z_in=z_in, prec=prec, T=T, B=B, H=H, I=I)
        return root, chroma, bass

    @staticmethod
    def backward(ctx, droot, dchroma, dbass):
        # tokens are constants (ground truth or argmax one-hots): same BPTT as the teacher-forced path
        g = F_.ChordDecoderTFFn.backward(ctx, droot, dchroma, dbass)
        return (g[0], None, None, None, None) + tuple(g[3:])

"""Standalone timing of ptv_free_note_loop (the free-running decoder's step loop, freerun.hip): 32 launches (one per time step) of 15
note steps each, the configuration functional_free uses for training (replay mode: train word 2, cluster S = 4 up to 64 panels).
Prints us per dependent note step for the head-weights-resident kernel and the streamed one (train bit 21), and with phases skipped
(dbg bits: 1 gate MFMAs, 4 pitch head, 8 duration GRU -- results invalid, timing only; bit 2 = no cell would stop the cluster's state
exchange, which leaves from the cell epilogue), and the per-phase timers of wave 0.
usage: python scripts/bench_noteloop.py [B ...]"""
import sys
import torch

sys.path.insert(0, '.')
from polyphonic_chord_texture_disentanglement_amd import functional as F_            # noqa: E402
from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_      # noqa: E402
from polyphonic_chord_texture_disentanglement_amd._lib import call, stream_ptr       # noqa: E402
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE        # noqa: E402


def main():
    Bs = [int(v) for v in sys.argv[1:]] or [512, 1024]
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    m = DisentangleVAE.init_model(dev).to(dev)
    P = dict(m.decoder.named_parameters())
    pk = FF_._free_packs(P, 1024)
    w_ih_d, b_ih_d = P['dec_dur_gru.weight_ih_l0'], P['dec_dur_gru.bias_ih_l0']
    tab0 = F_.gemm(P['dur_sos_token'].view(1, -1), w_ih_d, bias=b_ih_d, prec=0)
    tab = F_.gemm(F_._onehot2x5(dev), w_ih_d, bias=b_ih_d, prec=0)
    wl = F_._parr([pk['wg_h'], pk['wg_t'], pk['wp'], pk['wd_h'], pk['wd_p'], pk['wdur'], P['dec_notes_gru.bias_hh_l0'],
                   P['pitch_out_linear.bias'], P['dur_hid_linear.bias'], P['dec_dur_gru.bias_hh_l0'], tab0, tab,
                   P['dur_out_linear.weight'], P['dur_out_linear.bias'], pk['w_embT'], P['note_embedding.bias']])
    bf = torch.bfloat16
    wE = [P['dec_notes_emb_gru.' + n] for n in FF_.EMB_GRU]
    wr = F_._parr([pk['e_ih'], pk['e_hh'], pk['e_ih_r'], pk['e_hh_r'], wE[2], wE[3], wE[6], wE[7]])
    for B in Bs:
        R, M = 32 * B, 15 * 32 * B
        panels = (B + 15) // 16
        S = 8 if panels * 8 <= 256 else (4 if panels * 4 <= 256 else (2 if panels * 2 <= 256 else 1))
        sbits = 0x400000 if S == 8 else ((S if S > 1 else 0) << 18)
        g = torch.Generator(device=dev).manual_seed(5)
        GC = torch.randn(32, B, 1536, device=dev, generator=g) * 0.6
        emb = torch.randn(16, R, 128, device=dev, generator=g) * 0.5
        HN = torch.randn(16, R, 512, device=dev, generator=g) * 0.5
        pitch = torch.zeros(M, 136, device=dev)
        dur = torch.zeros(M, 10, device=dev)
        idx = torch.zeros(5, M, device=dev, dtype=torch.int32)
        TOK = torch.randn(15, R, 128, device=dev, generator=g) * 0.5
        PRED = torch.zeros(16, R, 128, device=dev)
        xhat = torch.zeros(B, 32, 16, 6, device=dev, dtype=torch.long)
        plen = torch.zeros(R, device=dev, dtype=torch.int32)
        xch = torch.zeros(panels * 2 * 16 * 512 * 2, device=dev, dtype=bf)
        cnt = torch.zeros(panels + 1, device=dev, dtype=torch.int32)
        io = F_._parr([GC[0], emb, HN, None, pitch, None, None, dur, idx, TOK, PRED, xhat, plen, None, None, None, None, None, None,
                       xch if S > 1 else None, cnt if S > 1 else None])
        ios = [F_._parr([GC[t], emb, HN, None, pitch, None, None, dur, idx, TOK, PRED, xhat, plen, None, None, None, None, None, None,
                         xch if S > 1 else None, cnt if S > 1 else None]) for t in range(32)]
        del io
        print(f'B = {B}: {panels} panels x S = {S}')
        for name, extra in (('resident heads', 0), ('streamed heads (bit 21)', 0x200000),
                            ('resident, no gate MFMAs', 1 << 8), ('resident, no pitch head', 4 << 8), ('resident, no duration GRU', 8 << 8),
                            ('resident heads', 0), ('streamed heads (bit 21)', 0x200000), ('8-wave kernel, no cluster', -1), ('4-wave resident, no cluster', -2), ('four members per panel', -3), ('resident, no watch phase (dbg 16)', 16 << 8)):
            flags = 2 | 0x10000 | sbits | extra
            if extra == -1:
                flags = 2 | 0x20000                      # the 8-wave producer / head kernel, one workgroup per panel
            if extra == -3:
                flags = 2 | 0x10000 | (4 << 18)          # four members per panel (the default before the eight-member kernel)
                if panels * 4 > 256:
                    continue
            if extra == -2:
                flags = 2 | 0x10000                      # the 4-wave kernel, one workgroup per panel
            best = 1e9
            for rep in range(4):
                cnt.zero_(); plen.zero_(); xch.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(32):
                    call('ptv_free_note_loop', wl, ios[t], 136, B, t, 0x15a5 if t % 3 == 0 else 0, flags, stream_ptr())
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            err = int(cnt[-1]) if S > 1 else 0
            print(f'  {name:34s} {best:7.3f} ms per forward   {best * 1e3 / (32 * 15):6.2f} us per note step   err={err}')
        # the re-summarisation of the predicted notes (ptv_free_resummarize): one launch per time step, 16 dependent bi-GRU steps each
        toks = torch.zeros(33, B, 256, device=dev)
        plen.fill_(9)
        ior = [F_._parr([PRED, plen, None, None, None, None, toks[t + 1]]) for t in range(32)]
        for name, fl in (('weights resident', 0), ('weights streamed (train bit 1)', 2)):
            best = 1e9
            toks.zero_()
            for rep in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(32):
                    call('ptv_free_resummarize', wr, ior[t], B, t, fl, stream_ptr())
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            print(f'  re-summarisation, {name:30s} {best:7.3f} ms per forward   {best * 1e3 / 32:6.2f} us per launch   {best * 1e3 / (32 * 16):6.2f} us per GRU step   checksum {float(toks.double().abs().sum()):.9f}')
        # per-phase time of wave 0 (100-MHz ticks summed over a launch's 15 note steps), one launch at t = 5, member 0 of panel 0 .. 3 and a foreign member
        grid = (panels + 7) // 8 * 8 * S if S > 1 else panels
        dbg = torch.zeros(3 * grid + 8 * grid, device=dev, dtype=torch.long)
        for name, extra in (('resident', 0), ('streamed', 0x200000)):
            iod = F_._parr([GC[5], emb, HN, None, pitch, None, None, dur, idx, TOK, PRED, xhat, plen, None, None, None, None, dbg, None,
                            xch if S > 1 else None, cnt if S > 1 else None])
            cnt.zero_(); xch.zero_()
            for t in range(6):
                call('ptv_free_note_loop', wl, iod if t == 5 else ios[t], 136, B, t, 0, 2 | 0x10000 | sbits | extra | ((64 << 8) if t == 5 else 0), stream_ptr())
            torch.cuda.synchronize()
            ph = dbg[3 * grid:].view(grid, 8)[:, :6].float().cpu() * 10.0 / 15.0 / 1e3          # us per note step
            live = ph.sum(1) > 0
            print(f'  phases ({name}), us per note step, mean over {int(live.sum())} workgroups [cell, barrier+exchange, pitch head, argmax+dur_hid, dur GRU, embed]:',
                  [round(float(v), 2) for v in ph[live].mean(0)], ' max-wg:', [round(float(v), 2) for v in ph[live].max(0).values])


if __name__ == '__main__':
    main()

import sys, torch
sys.path.insert(0, '.')
from polyphonic_chord_texture_disentanglement_amd import functional as F_
from polyphonic_chord_texture_disentanglement_amd import functional_free as FF_
from polyphonic_chord_texture_disentanglement_amd._lib import call, stream_ptr
from polyphonic_chord_texture_disentanglement_amd.model import DisentangleVAE
dev = torch.device('cuda:0'); torch.manual_seed(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = DisentangleVAE.init_model(dev).to(dev); P = dict(m.decoder.named_parameters())
R, M = 32 * B, 15 * 32 * B
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
for old in (0, 1):
    HN = torch.zeros(16, R, 512, device=dev); HN[0] = HN0
    gates_n = torch.zeros(15, 4, R, 512, device=dev, dtype=bf)
    pitch = torch.zeros(M, 136, device=dev); HD = torch.zeros(6, M, 64, device=dev)
    gates_d = torch.zeros(5, 4, M, 64, device=dev, dtype=bf); dur = torch.zeros(M, 10, device=dev)
    idx = torch.zeros(5, M, device=dev, dtype=torch.int32)
    TOK = torch.zeros(15, R, 128, device=dev); TOK[0] = TOK0
    PRED = torch.zeros(16, R, 128, device=dev)
    xhat = torch.zeros(B, 32, 16, 6, device=dev, dtype=torch.long); plen = torch.zeros(R, device=dev, dtype=torch.int32)
    io = F_._parr([GC[0], emb, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen] + [None] * 8)
    call('ptv_free_note_loop', wl, io, 136, B, 0, 0, 1 | 0x10000 | (0x200000 if old else 0), stream_ptr())
    torch.cuda.synchronize()
    res[old] = dict(pitch=pitch.view(15, R, 136)[:, :B].clone(), dur=dur.view(15, R, 10)[:, :B].clone(), idx=idx.view(5, 15, R)[:, :, :B].clone(),
                    HN=HN[:, :B].clone(), HD=HD.view(6, 15, R, 64)[:, :, :B].clone(), PRED=PRED[:, :B].clone(), xhat=xhat[:, 0].clone())
for k in res[0]:
    a, b = res[0][k].float(), res[1][k].float()
    d = (a - b).abs()
    print(k, float(d.max()))
    if d.max() > 0:
        nz = d.nonzero()
        print('   first mismatches', nz[:6].tolist(), 'count', len(nz))
p0, p1 = res[0]['pitch'], res[1]['pitch']
d = (p0 - p1).abs()
for n in range(15):
    print(n, float(d[n].max()), 'cols', sorted(set(d[n].nonzero()[:, 1].tolist()))[:12], 'HN', float((res[0]['HN'][n + 1] - res[1]['HN'][n + 1]).abs().max()),
          'dur', float((res[0]['dur'][n] - res[1]['dur'][n]).abs().max()), 'HD0', float((res[0]['HD'][0, n] - res[1]['HD'][0, n]).abs().max()))
    if d[n].max() > 0: break

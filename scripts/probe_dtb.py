import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from helpers import full_params
from polyphonic_chord_texture_disentanglement_amd import functional as F_, model as M
from polyphonic_chord_texture_disentanglement_amd.synthetic import synth_batch
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
orig = F_._decoder_bwd_composite
def wrapped(P, st, z, tok_op, dP, ddur, top_h, side, G):
    Bq, R, E, He, Ht, Hn, Hd, NP, prec = (st[k] for k in ('B', 'R', 'E', 'He', 'Ht', 'Hn', 'Hd', 'NP', 'prec'))
    HN16, HD16, NS16, gates_n, gates_t = st.get('HN16'), st.get('HD16'), st.get('NS16'), st['gates_n'], st['gates_t']
    conds = dict(prec=prec != 1, zs=not F_.ZERO_SKIP, fam=F_.SUMMARY_FAMILY_SLOT >= 0, cap=torch.cuda.is_current_stream_capturing(),
                 dur16=not st.get('dur16_only'), gd=st['gates_d'] is not None, tabs=st.get('dur_tabs') is None, hd16=HD16 is None,
                 rowk=not st.get('gates_n_rowk'), gn=gates_n.dtype != F_.BF16, gt=gates_t.dtype != F_.BF16, hw=not F_.HEADS_WGRAD_FUSED,
                 hok=not F_.heads_ok(prec, Hn, NP, Hd, HN16, HD16), npok=not F_.notes_persist_ok(prec, Hn, E, F_.BF16),
                 dps=dP.stride(0) != F_._pad8(NP), dpa=dP.data_ptr() % 16, dd=not ddur.is_contiguous(), tok=not tok_op.is_contiguous(),
                 z=not z.is_contiguous(), ps=not F_.persist_supported(1, Bq, Ht, 32), side=side.s == side.main)
    print('declined by:', [k for k, v in conds.items() if v], flush=True)
    r = orig(P, st, z, tok_op, dP, ddur, top_h, side, G)
    print('ran:', r is not None, flush=True)
    return r
F_._decoder_bwd_composite = wrapped
m = M.DisentangleVAE.init_model(torch.device(DEV)); m.load_state_dict(full_params()); m.to(DEV).set_precision('bf16'); m.use_philox(11, 0)
x, c, pr = (torch.from_numpy(a).to(DEV) for a in synth_batch(B, 77))
m.zero_grad(); losses = m.loss(x, c, pr, 1., 1., 1., 0.1, [1, 0.5]); losses[0].backward(); torch.cuda.synchronize()
print('ok')
